// renderer.h — host-side Renderer: the reference's batch renderer with its integrator loop
// replaced by calls into the HIP layer (include/qaray_hip.h).
//
// Mirrors src/renderers/renderer.h:47-111 (RendererParam, Renderer::Init / ComputeScene / Render /
// ThreadRender / Start-Stop-KillTimer / Terminate) and the batch flow of Renderer_MPI::Render
// (src/renderers/Renderer_MPI.cpp:123-215: render, save colorBuffer / depthBuffer / sampleBuffer
// PNGs).  ThreadRender() keeps its name and its place in the call stack; what used to be two nested
// tasking::parallel_for loops over tiles and pixels (renderer.cpp:383-420) is one
// qa_render_region / qa_render_strips call.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "framebuffer.h"
#include "qaray_hip.h"
#include "scene.h"

namespace qaray_hip {

struct RendererParam {  // src/renderers/renderer.h:47-68
  bool useSRGB = true;
  size_t sppMax = 8;
  size_t sppMin = 4;
  uint32_t seed = 0x51A7A7;
  bool usePhotonMap = false;
  size_t photonMapSize = 10000;
  size_t photonMapBounce = 20;
  float photonMapRadius = 0.2f;
  size_t causticsMapSize = 1000;
  size_t causticsMapBounce = 20;
  float causticsMapRadius = 1.0f;
  void SetSPPMax(int spp) { sppMax = static_cast<size_t>(spp); }
  void SetSPPMin(int spp) { sppMin = static_cast<size_t>(spp); }
  void SetSRGBFlag(bool flag) { useSRGB = flag; }
  void SetPhotonMappingFlag(bool flag) { usePhotonMap = flag; }
  void SetPhotonMapBounce(size_t b) { photonMapBounce = b; }
  void SetPhotonMapSize(size_t sz) { photonMapSize = sz; }
  void SetPhotonMapRadius(float r) { photonMapRadius = r; }
  void SetCausticsMapBounce(size_t b) { causticsMapBounce = b; }
  void SetCausticsMapSize(size_t sz) { causticsMapSize = sz; }
  void SetCausticsMapRadius(float r) { causticsMapRadius = r; }
};

class Renderer {
 public:
  explicit Renderer(RendererParam &param, int device = 0, size_t rank = 0, size_t size = 1);
  // One process, several GPUs (qaray_hip -devices N): the role Renderer_MPI plays across MPI ranks
  // (src/renderers/Renderer_MPI.cpp:123-215), inside one address space.  One host thread and one qa_ctx per device; the
  // flattened scene is uploaded to the first device and copied to the others device-to-device (xGMI); device i renders the
  // 8-row strips i, i + N, ...; the packed colour | depth | sample-count strips are copied to the first device
  // (hipMemcpyPeerAsync) and placed from there (PlaceStrips = PlaceImage<T>).  devices.size() == 1 takes the same path.
  void UseDevices(const std::vector<int> &devices) { multi = devices; }
  virtual ~Renderer();
  virtual void Init();                                  // creates the HIP context
  void ComputeScene(FrameBuffer &renderImage, Scene &scene);  // camera frame, fb, scene upload, photon maps
  virtual void Render();                                // ThreadRender + image dumps (batch mode)
  void ThreadRender();                                  // the hot path: one HIP call
  virtual void StartTimer();
  virtual void StopTimer();
  virtual void KillTimer();
  virtual void Terminate();
  double LastSeconds() const { return lastSeconds; }
  const qa_counters &Counters() const { return counters; }
  std::string outputPrefix;                             // like Renderer_MPI's mpiPrefix

 protected:
  RendererParam &param;
  Scene *scene = nullptr;
  FrameBuffer *image = nullptr;
  size_t pixelW = 0, pixelH = 0;
  size_t mpiSize = 1, mpiRank = 0;                      // strip partition: rank, rank+size, ...
  int device = 0;
  qa_ctx *ctx = nullptr;
  std::vector<int> multi;                               // UseDevices: empty = the single-context path
  std::vector<qa_ctx *> ctxs;                           // one per entry of `multi`
  void ThreadRenderMulti();
  double lastSeconds = 0, avgSeconds = 0;
  int numFrames = -1;
  qa_counters counters{};
};

}  // namespace qaray_hip
