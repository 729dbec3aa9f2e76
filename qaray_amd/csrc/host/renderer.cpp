// renderer.cpp — see renderer.h.
#include "renderer.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>

#include <hip/hip_runtime_api.h>

namespace qaray_hip {

static std::chrono::time_point<std::chrono::system_clock> g_start;

Renderer::Renderer(RendererParam &p, int dev, size_t rank, size_t size) : param(p), mpiSize(size), mpiRank(rank), device(dev)
{
  tasking::signal_start();  // src/renderers/renderer.cpp:67-70
}
Renderer::~Renderer()
{
  for (qa_ctx *c : ctxs) if (c) qa_ctx_destroy(c);
  if (ctx) qa_ctx_destroy(ctx);
}

void Renderer::Init()
{
  if (multi.empty()) {
    if (qa_ctx_create(device, &ctx) != QA_OK) throw std::runtime_error(std::string("qa_ctx_create: ") + qa_last_error());
    return;
  }
  ctxs.assign(multi.size(), nullptr);
  for (size_t i = 0; i < multi.size(); ++i)
    if (qa_ctx_create(multi[i], &ctxs[i]) != QA_OK) throw std::runtime_error(std::string("qa_ctx_create: ") + qa_last_error());
}

// Run f(i) on one host thread per device and rethrow the first failure
template <class F>
static void PerDevice(size_t n, F f)
{
  std::vector<std::thread> th;
  std::vector<std::string> err(n);
  for (size_t i = 0; i < n; ++i)
    th.emplace_back([&, i]() {
      try { f(i); } catch (const std::exception &e) { err[i] = e.what(); }
    });
  for (std::thread &t : th) t.join();
  for (const std::string &e : err) if (!e.empty()) throw std::runtime_error(e);
}
#define HIP_OR_THROW(expr)                                                                               \
  do {                                                                                                   \
    const hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

// src/renderers/renderer.cpp:71-113: canvas + framebuffer here, camera frame inside the flattener
void Renderer::ComputeScene(FrameBuffer &fb, Scene &sc)
{
  image = &fb;
  scene = &sc;
  pixelW = static_cast<size_t>(sc.camera.imgWidth);
  pixelH = static_cast<size_t>(sc.camera.imgHeight);
  image->Init(static_cast<unsigned>(pixelW), static_cast<unsigned>(pixelH));
  const std::vector<unsigned char> blob = FlattenScene(sc);
  if (!multi.empty()) {
    // the flattened scene goes to the first device from the host and to the others device-to-device (the reference re-parses
    // the XML on every rank, Renderer_MPI.cpp:54; across processes bench.py broadcasts the same blob over RCCL)
    unsigned char *d0 = nullptr;
    HIP_OR_THROW(hipSetDevice(multi[0]));
    HIP_OR_THROW(hipMalloc((void **) &d0, blob.size()));
    HIP_OR_THROW(hipMemcpy(d0, blob.data(), blob.size(), hipMemcpyHostToDevice));
    PerDevice(multi.size(), [&](size_t i) {
      HIP_OR_THROW(hipSetDevice(multi[i]));
      unsigned char *di = d0;
      if (i > 0) {
        HIP_OR_THROW(hipMalloc((void **) &di, blob.size()));
        HIP_OR_THROW(hipMemcpyPeer(di, multi[i], d0, multi[0], blob.size()));
      }
      const int rc = qa_scene_upload_device(ctxs[i], di, blob.size());
      if (i > 0) (void) hipFree(di);
      if (rc != QA_OK) throw std::runtime_error(std::string("qa_scene_upload_device: ") + qa_last_error());
    });
    HIP_OR_THROW(hipSetDevice(multi[0]));
    (void) hipFree(d0);
    if (param.usePhotonMap) throw std::runtime_error("-use-photon-map with -devices is not wired up (every device would build the same maps)");
    return;
  }
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

// Renderer_MPI::Render's render + gather (Renderer_MPI.cpp:123-207) across the GPUs of this process
void Renderer::ThreadRenderMulti()
{
  StartTimer();
  const int W = (int) pixelW, H = (int) pixelH, N = (int) multi.size();
  printf("\nRunning on %d HIP device(s), 8-row strips round-robin\n", N);
  const int maxStrips = (((H + QA_STRIP_ROWS - 1) / QA_STRIP_ROWS) + N - 1) / N;
  const size_t n = (size_t) maxStrips * QA_STRIP_ROWS * W;          // pixels of a rank's packed buffer (equal for all ranks)
  const size_t words = 5 * n;                                         // rgb | depth | sample counts
  // the gather target: one packed buffer per rank, on the first device
  float *gathered = nullptr;
  HIP_OR_THROW(hipSetDevice(multi[0]));
  HIP_OR_THROW(hipMalloc((void **) &gathered, (size_t) N * words * sizeof(float)));
  memset(&counters, 0, sizeof(counters));
  std::vector<qa_counters> cnt(N);
  PerDevice((size_t) N, [&](size_t i) {
    HIP_OR_THROW(hipSetDevice(multi[i]));
    qa_ctx *c = ctxs[i];
    if (tasking::has_stop_signal()) qa_request_stop(c);
    else qa_clear_stop(c);
    qa_reset_counters(c);
    float *buf = nullptr;
    HIP_OR_THROW(hipMalloc((void **) &buf, words * sizeof(float)));
    HIP_OR_THROW(hipMemset(buf, 0, words * sizeof(float)));
    hipStream_t s = nullptr;
    HIP_OR_THROW(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (qa_strip_count(0, H, (int) i, N) > 0 &&
        qa_render_strips_device(c, 0, 0, W, H, (int) i, N, (int) param.sppMin, (int) param.sppMax, Material::maxBounce, param.seed, 0, buf, buf + 3 * n,
                                reinterpret_cast<uint32_t *>(buf + 4 * n), s) != QA_OK)
      throw std::runtime_error(std::string("qa_render_strips_device: ") + qa_last_error());
    // the strips travel to the first device right behind the kernel, on the same stream (xGMI peer copy)
    HIP_OR_THROW(hipMemcpyPeerAsync(gathered + i * words, multi[0], buf, multi[i], words * sizeof(float), s));
    HIP_OR_THROW(hipStreamSynchronize(s));
    qa_get_counters(c, &cnt[i]);
    (void) hipStreamDestroy(s);
    (void) hipFree(buf);
  });
  std::vector<float> host((size_t) N * words);
  HIP_OR_THROW(hipSetDevice(multi[0]));
  HIP_OR_THROW(hipMemcpy(host.data(), gathered, host.size() * sizeof(float), hipMemcpyDeviceToHost));
  (void) hipFree(gathered);
  for (int i = 0; i < N; ++i) {
    const float *b = host.data() + (size_t) i * words;
    PlaceStrips(*image, W, H, N, i, b, b + 3 * n, reinterpret_cast<const uint32_t *>(b + 4 * n), (int) param.sppMax, param.useSRGB);
    counters.samples += cnt[i].samples; counters.casts_normal += cnt[i].casts_normal; counters.casts_shadow += cnt[i].casts_shadow;
    counters.bvh_nodes += cnt[i].bvh_nodes; counters.tri_tests += cnt[i].tri_tests; counters.pixels += cnt[i].pixels;
  }
  StopTimer();
}

// src/renderers/renderer.cpp:370-423
void Renderer::ThreadRender()
{
  if (!multi.empty()) { ThreadRenderMulti(); return; }
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
  for (qa_ctx *&c : ctxs) if (c) { qa_ctx_destroy(c); c = nullptr; }
  if (ctx) { qa_ctx_destroy(ctx); ctx = nullptr; }
}

}  // namespace qaray_hip
