// framebuffer.cpp — see framebuffer.h.
#include "framebuffer.h"

#include <algorithm>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

#include <cmath>

#include "image.h"

namespace qaray_hip {

FrameBuffer renderImage;

// src/renderers/renderer.cpp:34-39
float LinearToSRGB(const float c)
{
  const float a = 0.055f;
  if (c < 0.0031308f) return 12.92f * c;
  return (1.f + a) * std::pow(c, 1.f / 2.4f) - a;
}

// src/fb/framebuffer.cpp:32-52
void FrameBuffer::Init(unsigned w, unsigned h)
{
  width = w;
  height = h;
  const size_t n = (size_t) w * h;
  mask.assign(n, 0);
  img.assign(3 * n, 0);
  zbuffer.assign(n, 0.f);
  sampleCount.assign(n, 0);
  zbufferImg.clear();
  sampleCountImg.clear();
  numRenderedPixels = 0;
}

void FrameBuffer::ResetNumRenderedPixels()
{
  mask.assign((size_t) width * height, 0);
  numRenderedPixels = 0;
}

// src/fb/framebuffer.cpp:62-84
void FrameBuffer::ComputeZBufferImage()
{
  const size_t size = (size_t) width * height;
  zbufferImg.assign(size, 0);
  float zmin = 1.0e30f, zmax = 0;
  for (size_t i = 0; i < size; i++) {
    if (zbuffer[i] == 1.0e30f) continue;
    if (zmin > zbuffer[i]) zmin = zbuffer[i];
    if (zmax < zbuffer[i]) zmax = zbuffer[i];
  }
  for (size_t i = 0; i < size; i++) {
    if (zbuffer[i] == 1.0e30f) zbufferImg[i] = 0;
    else {
      const float f = (zmax - zbuffer[i]) / (zmax - zmin);
      zbufferImg[i] = (uint8_t) (f * 255);
    }
  }
}

// src/fb/framebuffer.cpp:86-107
int FrameBuffer::ComputeSampleCountImage()
{
  const size_t size = (size_t) width * height;
  sampleCountImg.assign(size, 0);
  uint8_t smin = 255, smax = 0;
  for (size_t i = 0; i < size; i++) {
    if (smin > sampleCount[i]) smin = sampleCount[i];
    if (smax < sampleCount[i]) smax = sampleCount[i];
  }
  if (smax != smin)
    for (size_t i = 0; i < size; i++) sampleCountImg[i] = (uint8_t) ((255 * (sampleCount[i] - smin)) / (smax - smin));
  return smax;
}

bool FrameBuffer::SaveImage(const char *fn) const { return SavePNG(fn, img.data(), (int) width, (int) height, 3); }
bool FrameBuffer::SaveZImage(const char *fn) const
{
  return !zbufferImg.empty() && SavePNG(fn, zbufferImg.data(), (int) width, (int) height, 1);
}
bool FrameBuffer::SaveSampleCountImage(const char *fn) const
{
  return !sampleCountImg.empty() && SavePNG(fn, sampleCountImg.data(), (int) width, (int) height, 1);
}

void FrameBuffer::Deposit(int x0, int y0, int x1, int y1, const float *rgb, const float *depth,
                          const uint32_t *nsamples, int sppMax, bool useSRGB)
{
  const int cw = x1 - x0;
  for (int j = y0; j < y1; ++j)
    for (int i = x0; i < x1; ++i) {
      const size_t q = (size_t) (j - y0) * cw + (i - x0);
      const size_t idx = (size_t) j * width + i;
      // nsamples = 0: the pixel was skipped (tasking::signal_stop); the reference leaves such pixels untouched,
      // mask 0 (src/renderers/renderer.cpp:365,402)
      if (nsamples[q] == 0) continue;
      float c[3] = {rgb[3 * q], rgb[3 * q + 1], rgb[3 * q + 2]};
      for (int k = 0; k < 3; ++k) {
        if (useSRGB) c[k] = LinearToSRGB(c[k]);
        const float lo = (1.f < c[k]) ? 1.f : c[k];   // MIN(1, c)
        c[k] = (0.f > lo) ? 0.f : lo;                  // MAX(0, ..)
        img[3 * idx + k] = static_cast<uint8_t>(roundf(c[k] * 255.f));
      }
      zbuffer[idx] = depth[q];
      sampleCount[idx] = static_cast<uint8_t>(255.f * nsamples[q] / static_cast<float>(sppMax));
      mask[idx] = 1;
    }
  // the reference counts every pixel of a tile it has been through, rendered or skipped by a stop request
  // (src/renderers/renderer.cpp:399-404), so IsRenderDone() turns true after a stopped frame as well
  IncrementNumRenderPixel((x1 > x0 && y1 > y0) ? cw * (y1 - y0) : 0);
}

bool StripRowRange(int height, int world, int rank, int k, int &y0, int &y1)
{
  if (height <= 0 || world < 1 || rank < 0 || rank >= world || k < 0) return false;
  const int strip = rank + k * world;
  y0 = strip * 8;
  if (y0 >= height) return false;
  y1 = y0 + 8 < height ? y0 + 8 : height;
  return true;
}

int PlaceStrips(FrameBuffer &fb, int width, int height, int world, int rank, const float *rgb, const float *depth, const uint32_t *nsamples, int sppMax,
                bool useSRGB)
{
  int k = 0, y0, y1;
  for (; StripRowRange(height, world, rank, k, y0, y1); ++k) {
    const size_t off = (size_t) k * 8 * (size_t) width;
    fb.Deposit(0, y0, width, y1, rgb + 3 * off, depth + off, nsamples + off, sppMax, useSRGB);
  }
  return k;
}

namespace tasking {
static std::atomic<bool> threadStop{false};
static std::atomic<size_t> numThreads{0};   // 0 = not initialised
void signal_start() { threadStop = false; }
void signal_stop() { threadStop = true; }
bool has_stop_signal() { return threadStop; }
size_t get_num_of_threads()
{
  if (numThreads == 0) init();
  return numThreads;
}
void set_num_of_threads(size_t n) { numThreads = n ? n : 1; }
void init()
{
  if (numThreads == 0) {
    const unsigned hw = std::thread::hardware_concurrency();
    numThreads = hw ? hw : 1;
  }
}
void parallel_for(size_t start, size_t end, size_t step, std::function<void(size_t)> T)
{
  if (step == 0 || start >= end) return;
  const size_t items = (end - start + step - 1) / step;
  const size_t workers = std::min(get_num_of_threads(), items);
  if (workers <= 1) {
    for (size_t i = start; i < end; i += step) T(i);
    return;
  }
  std::atomic<size_t> next{0};
  std::exception_ptr failure;
  std::mutex failureLock;
  auto body = [&]() {
    for (;;) {
      const size_t k = next.fetch_add(1);
      if (k >= items) return;
      try { T(start + k * step); }
      catch (...) {
        std::lock_guard<std::mutex> g(failureLock);
        if (!failure) failure = std::current_exception();
        next = items;   // no further items are handed out
        return;
      }
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(workers - 1);
  for (size_t w = 1; w < workers; ++w) pool.emplace_back(body);
  body();
  for (std::thread &t : pool) t.join();
  if (failure) std::rethrow_exception(failure);
}
}  // namespace tasking

}  // namespace qaray_hip
