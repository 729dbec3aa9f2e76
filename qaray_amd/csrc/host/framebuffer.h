// framebuffer.h — the "renderImage" output sink (reference: src/fb/framebuffer.h:35-95).
//
// Same products as the reference's FrameBuffer: 8-bit RGB image, float z-buffer, per-pixel
// sample-count byte, rendered-pixel mask and counter, plus the z / sample-count visualisations
// and PNG dumps.  The GPU path delivers LINEAR FLOAT radiance per region; Deposit() applies the
// post-process the reference does at the end of PixelRender (src/renderers/renderer.cpp:347-365).
#pragma once
#include <cstddef>
#include <functional>
#include <map>
#include <memory>
#include <atomic>
#include <cstdint>
#include <vector>

namespace qaray_hip {

float LinearToSRGB(float c);  // src/renderers/renderer.cpp:34-39

class FrameBuffer {
 public:
  void Init(unsigned w, unsigned h);
  int GetWidth() const { return (int) width; }
  int GetHeight() const { return (int) height; }
  uint8_t *GetPixels() { return img.data(); }
  const uint8_t *GetPixels() const { return img.data(); }
  uint8_t *GetMasks() { return mask.data(); }
  const uint8_t *GetMasks() const { return mask.data(); }
  float *GetZBuffer() { return zbuffer.data(); }
  const float *GetZBuffer() const { return zbuffer.data(); }
  uint8_t *GetSampleCount() { return sampleCount.data(); }
  const uint8_t *GetSampleCount() const { return sampleCount.data(); }
  const uint8_t *GetZBufferImage() const { return zbufferImg.data(); }
  const uint8_t *GetSampleCountImage() const { return sampleCountImg.data(); }
  void ResetNumRenderedPixels();
  int GetNumRenderedPixels() const { return numRenderedPixels; }
  void IncrementNumRenderPixel(int n) { numRenderedPixels += n; }
  bool IsRenderDone() const { return numRenderedPixels >= (int) (width * height); }
  void ComputeZBufferImage();
  int ComputeSampleCountImage();
  bool SaveImage(const char *filename) const;
  bool SaveZImage(const char *filename) const;
  bool SaveSampleCountImage(const char *filename) const;

  // Region [x0,x1)x[y0,y1) of float results -> 8-bit products (renderer.cpp:347-365).
  void Deposit(int x0, int y0, int x1, int y1, const float *rgb, const float *depth,
               const uint32_t *nsamples, int sppMax, bool useSRGB);

 private:
  unsigned width = 0, height = 0;
  std::vector<uint8_t> img, mask, sampleCount, zbufferImg, sampleCountImg;
  std::vector<float> zbuffer;
  std::atomic<int> numRenderedPixels{0};
};

extern FrameBuffer renderImage;  // src/scene/scene.cpp:78

// Image-space partition between GPUs (include/qaray_hip.h qa_render_strips_device): rank r of n owns the 8-row strips
// r, r + n, ...; its outputs are packed strip after strip.  PlaceStrips is the analogue of PlaceImage<T>
// (src/renderers/Renderer_MPI.cpp:103-122): rank `rank`'s packed float results go through Deposit into the rows they
// belong to.  Returns the number of strips placed.  StripRowRange: image rows [y0, y1) of packed strip k (false: no such strip).
bool StripRowRange(int height, int world, int rank, int k, int &y0, int &y1);
int PlaceStrips(FrameBuffer &fb, int width, int height, int world, int rank, const float *rgb, const float *depth, const uint32_t *nsamples,
                int sppMax, bool useSRGB);

// src/tasking/parallel_for.h:59-95, same names and signatures.  The pixel work itself runs on the GPU (one
// qa_render_* call replaces ThreadRender's two nested parallel_for loops); what stays on the host - one worker per
// GPU of a node, strip assembly, file output - can still be spread with parallel_for, and every work item polls
// the same stop flag (the HIP layer sees it through qa_request_stop).
namespace tasking {
size_t get_num_of_threads();
void set_num_of_threads(size_t num_of_threads);
void init();                       // thread count <- hardware concurrency unless set_num_of_threads was called
void signal_start();
void signal_stop();
bool has_stop_signal();
// calls T(i) for i = start, start + step, ... < end on get_num_of_threads() host threads (work items are handed out
// one at a time, like TBB's simple_partitioner in the reference); returns when all of them have returned
void parallel_for(size_t start, size_t end, size_t step, std::function<void(size_t)> T);

// one instance of T per worker thread, created from a prototype on first use (the reference keeps one sampler per
// thread this way, src/core/sampler.h); local() is valid on the calling thread and inside parallel_for bodies
template <typename T>
struct ThreadLocalStorage {
  const T data;
  explicit ThreadLocalStorage(const T &t) : data(t) {}
  T &local()
  {
    thread_local std::map<const void *, std::unique_ptr<T>> mine;
    std::unique_ptr<T> &p = mine[this];
    if (!p) p.reset(new T(data));
    return *p;
  }
};
}  // namespace tasking

}  // namespace qaray_hip
