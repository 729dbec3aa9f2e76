// renderer.cpp — see renderer.h.
#include "renderer.h"

#include <chrono>
#include <cstdio>
#include <stdexcept>

#include <hip/hip_runtime_api.h>

namespace qaray_hip {

static std::chrono::time_point<std::chrono::system_clock> g_start;

Renderer::Renderer(RendererParam &p, int dev, size_t rank, size_t size) : param(p), mpiSize(size), mpiRank(rank), device(dev)
{
  tasking::signal_start();  // src/renderers/renderer.cpp:67-70
}
Renderer::~Renderer() { if (ctx) qa_ctx_destroy(ctx); }

void Renderer::Init()
{
  if (qa_ctx_create(device, &ctx) != QA_OK) throw std::runtime_error(std::string("qa_ctx_create: ") + qa_last_error());
}

// src/renderers/renderer.cpp:71-113: canvas + framebuffer here, camera frame inside the flattener
void Renderer::ComputeScene(FrameBuffer &fb, Scene &sc)
{
  image = &fb;
  scene = &sc;
  pixelW = static_cast<size_t>(sc.camera.imgWidth);
  pixelH = static_cast<size_t>(sc.camera.imgHeight);
  image->Init(static_cast<unsigned>(pixelW), static_cast<unsigned>(pixelH));
  const std::vector<unsigned char> blob = FlattenScene(sc);
  if (qa_scene_upload(ctx, blob.data(), blob.size()) != QA_OK)
    throw std::runtime_error(std::string("qa_scene_upload: ") + qa_last_error());
  // src/renderers/renderer.cpp:114-291: both maps, when asked for and non-empty
  if (param.usePhotonMap && param.photonMapSize > 0 && param.causticsMapSize > 0) {
    const auto t0 = std::chrono::system_clock::now();
    qa_photon_params pp;
    pp.photon.size = (uint32_t) param.photonMapSize;
    pp.photon.bounce = (uint32_t) param.photonMapBounce;
    pp.photon.radius = param.photonMapRadius;
    pp.caustics.size = (uint32_t) param.causticsMapSize;
    pp.caustics.bounce = (uint32_t) param.causticsMapBounce;
    pp.caustics.radius = param.causticsMapRadius;
    if (qa_photon_maps_build(ctx, &pp, param.seed) != QA_OK)
      throw std::runtime_error(std::string("qa_photon_maps_build: ") + qa_last_error());
    uint64_t emitted[2], emissions[2];
    qa_photon_maps_info(ctx, emitted, emissions);
    const std::chrono::duration<double> dt = std::chrono::system_clock::now() - t0;
    printf("\nPhoton Map (%zu photons, %llu emitted rays) and Caustics Map (%zu, %llu) Take %f s to Build\n", param.photonMapSize,
           (unsigned long long) emitted[0], param.causticsMapSize, (unsigned long long) emitted[1], dt.count());
  }
}

// src/renderers/renderer.cpp:42-63
void Renderer::StartTimer() { g_start = std::chrono::system_clock::now(); }
void Renderer::StopTimer()
{
  const std::chrono::duration<double> el = std::chrono::system_clock::now() - g_start;
  lastSeconds = el.count();
  printf("\nElapsed Time is %f s\n", lastSeconds);
  if (++numFrames > 0) avgSeconds += (lastSeconds - avgSeconds) / numFrames;
}
void Renderer::KillTimer() { printf("\nProgram Ends, Average Frame Time %f s\n\n", avgSeconds); }

// src/renderers/renderer.cpp:370-423
void Renderer::ThreadRender()
{
  StartTimer();
  if (mpiRank == 0) printf("\nRunning on HIP device %d, rank %zu of %zu\n", device, mpiRank, mpiSize);
  const int W = (int) pixelW, H = (int) pixelH;
  if (tasking::has_stop_signal()) qa_request_stop(ctx);
  else qa_clear_stop(ctx);
  qa_reset_counters(ctx);
  if (mpiSize == 1) {
    std::vector<float> rgb((size_t) 3 * W * H), depth((size_t) W * H);
    std::vector<uint32_t> ns((size_t) W * H);
    if (qa_render_region(ctx, 0, 0, W, H, (int) param.sppMin, (int) param.sppMax, Material::maxBounce, param.seed, 0,
                         rgb.data(), depth.data(), ns.data()) != QA_OK)
      throw std::runtime_error(std::string("qa_render_region: ") + qa_last_error());
    image->Deposit(0, 0, W, H, rgb.data(), depth.data(), ns.data(), (int) param.sppMax, param.useSRGB);
  } else {
    // rank-strided strips, like ThreadRender(tileStart = rank, step = size); this rank deposits
    // only the rows it owns (mask = 1 there), the caller gathers (Renderer_MPI::Render's PlaceImage)
    const int strips = qa_strip_count(0, H, (int) mpiRank, (int) mpiSize);
    const size_t n = (size_t) strips * QA_STRIP_ROWS * W;
    float *dRgb = nullptr, *dDepth = nullptr;
    uint32_t *dNs = nullptr;
    if (n) {
      if (hipMalloc((void **) &dRgb, n * 12) != hipSuccess || hipMalloc((void **) &dDepth, n * 4) != hipSuccess ||
          hipMalloc((void **) &dNs, n * 4) != hipSuccess)
        throw std::runtime_error("hipMalloc failed");
      if (qa_render_strips_device(ctx, 0, 0, W, H, (int) mpiRank, (int) mpiSize, (int) param.sppMin, (int) param.sppMax,
                                  Material::maxBounce, param.seed, 0, dRgb, dDepth, dNs, nullptr) != QA_OK)
        throw std::runtime_error(std::string("qa_render_strips_device: ") + qa_last_error());
      qa_synchronize(ctx);
      std::vector<float> rgb(3 * n), depth(n);
      std::vector<uint32_t> ns(n);
      (void) hipMemcpy(rgb.data(), dRgb, n * 12, hipMemcpyDeviceToHost);
      (void) hipMemcpy(depth.data(), dDepth, n * 4, hipMemcpyDeviceToHost);
      (void) hipMemcpy(ns.data(), dNs, n * 4, hipMemcpyDeviceToHost);
      for (int k = 0; k < strips; ++k) {
        const int y0 = ((int) mpiRank + k * (int) mpiSize) * QA_STRIP_ROWS;
        const int y1 = y0 + QA_STRIP_ROWS < H ? y0 + QA_STRIP_ROWS : H;
        const size_t off = (size_t) k * QA_STRIP_ROWS * W;
        image->Deposit(0, y0, W, y1, rgb.data() + 3 * off, depth.data() + off, ns.data() + off, (int) param.sppMax, param.useSRGB);
      }
      (void) hipFree(dRgb); (void) hipFree(dDepth); (void) hipFree(dNs);
    }
  }
  qa_get_counters(ctx, &counters);
  StopTimer();
}

// Renderer_MPI::Render (src/renderers/Renderer_MPI.cpp:123-139): render, then dump the three images
void Renderer::Render()
{
  ThreadRender();
  image->ComputeZBufferImage();
  image->ComputeSampleCountImage();
  image->SaveImage((outputPrefix + "colorBuffer.png").c_str());
  image->SaveZImage((outputPrefix + "depthBuffer.png").c_str());
  image->SaveSampleCountImage((outputPrefix + "sampleBuffer.png").c_str());
}

void Renderer::Terminate()
{
  if (ctx) { qa_ctx_destroy(ctx); ctx = nullptr; }
}

}  // namespace qaray_hip
