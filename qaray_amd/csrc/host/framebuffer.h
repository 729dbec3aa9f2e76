// framebuffer.h — the "renderImage" output sink (reference: src/fb/framebuffer.h:35-95).
//
// Same products as the reference's FrameBuffer: 8-bit RGB image, float z-buffer, per-pixel
// sample-count byte, rendered-pixel mask and counter, plus the z / sample-count visualisations
// and PNG dumps.  The GPU path delivers LINEAR FLOAT radiance per region; Deposit() applies the
// post-process the reference does at the end of PixelRender (src/renderers/renderer.cpp:347-365).
#pragma once
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

namespace tasking {  // src/tasking/parallel_for.h:59-68: the stop flag every work item polls
void signal_start();
void signal_stop();
bool has_stop_signal();
}  // namespace tasking

}  // namespace qaray_hip
