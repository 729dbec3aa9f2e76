// host_capi.cpp — extern "C" surface of libqaray_host.so (include/qaray_host.h).
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "framebuffer.h"
#include "qaray_host.h"
#include "scene.h"

using namespace qaray_hip;

struct qa_host_scene { Scene scene; };
struct qa_fb { FrameBuffer fb; };

static thread_local std::string g_err;
static int Fail(int code, const std::string &msg) { g_err = msg; return code; }

extern "C" {

const char *qa_host_last_error(void) { return g_err.c_str(); }

int qa_host_scene_load(const char *xml_path, const char *asset_root, qa_host_scene **out)
{
  if (!xml_path || !out) return Fail(QA_EINVAL, "null argument");
  try {
    qa_host_scene *s = new qa_host_scene;
    s->scene.assetRoot = asset_root ? asset_root : "";
    if (!s->scene.assetRoot.empty() && s->scene.assetRoot.back() != '/') s->scene.assetRoot += '/';
    if (!LoadScene(xml_path, s->scene)) {
      delete s;
      return Fail(QA_EIO, std::string("cannot load scene ") + xml_path);
    }
    *out = s;
    return QA_OK;
  } catch (const std::bad_alloc &) {
    return Fail(QA_ENOMEM, "out of memory");
  } catch (const std::exception &e) {
    return Fail(QA_EIO, e.what());
  }
}

void qa_host_scene_destroy(qa_host_scene *scene) { delete scene; }

int qa_host_scene_set_size(qa_host_scene *scene, int width, int height)
{
  if (!scene || width <= 0 || height <= 0) return Fail(QA_EINVAL, "bad size");
  scene->scene.camera.imgWidth = width;
  scene->scene.camera.imgHeight = height;
  return QA_OK;
}

int qa_host_scene_get_size(const qa_host_scene *scene, int *width, int *height)
{
  if (!scene || !width || !height) return Fail(QA_EINVAL, "null argument");
  *width = scene->scene.camera.imgWidth;
  *height = scene->scene.camera.imgHeight;
  return QA_OK;
}

int qa_host_scene_flatten(const qa_host_scene *scene, unsigned char **blob, uint64_t *nbytes)
{
  if (!scene || !blob || !nbytes) return Fail(QA_EINVAL, "null argument");
  try {
    const std::vector<unsigned char> b = FlattenScene(scene->scene);
    unsigned char *p = static_cast<unsigned char *>(malloc(b.size()));
    if (!p) return Fail(QA_ENOMEM, "out of memory");
    memcpy(p, b.data(), b.size());
    *blob = p;
    *nbytes = b.size();
    return QA_OK;
  } catch (const std::bad_alloc &) {
    return Fail(QA_ENOMEM, "out of memory");
  }
}

void qa_host_free(void *p) { free(p); }

int qa_fb_create(int width, int height, qa_fb **out)
{
  if (!out || width <= 0 || height <= 0) return Fail(QA_EINVAL, "bad size");
  try {
    qa_fb *f = new qa_fb;
    f->fb.Init((unsigned) width, (unsigned) height);
    *out = f;
    return QA_OK;
  } catch (const std::bad_alloc &) {
    return Fail(QA_ENOMEM, "out of memory");
  }
}
void qa_fb_destroy(qa_fb *fb) { delete fb; }

int qa_fb_deposit(qa_fb *fb, int x0, int y0, int x1, int y1, const float *rgb, const float *depth,
                  const uint32_t *nsamples, int spp_max, int use_srgb)
{
  if (!fb || !rgb || !depth || !nsamples) return Fail(QA_EINVAL, "null argument");
  if (x0 < 0 || y0 < 0 || x1 > fb->fb.GetWidth() || y1 > fb->fb.GetHeight() || x1 < x0 || y1 < y0 || spp_max <= 0)
    return Fail(QA_EINVAL, "region outside the framebuffer");
  fb->fb.Deposit(x0, y0, x1, y1, rgb, depth, nsamples, spp_max, use_srgb != 0);
  return QA_OK;
}
const uint8_t *qa_fb_pixels(const qa_fb *fb) { return fb ? fb->fb.GetPixels() : nullptr; }
const float *qa_fb_zbuffer(const qa_fb *fb) { return fb ? fb->fb.GetZBuffer() : nullptr; }
const uint8_t *qa_fb_sample_count(const qa_fb *fb) { return fb ? fb->fb.GetSampleCount() : nullptr; }
const uint8_t *qa_fb_mask(const qa_fb *fb) { return fb ? fb->fb.GetMasks() : nullptr; }
int qa_fb_num_rendered_pixels(const qa_fb *fb) { return fb ? fb->fb.GetNumRenderedPixels() : 0; }
int qa_fb_place_strips(qa_fb *fb, int world, int rank, const float *rgb, const float *depth, const uint32_t *nsamples, int spp_max, int use_srgb)
{
  if (!fb || !rgb || !depth || !nsamples || world < 1 || rank < 0 || rank >= world) return QA_EINVAL;
  return PlaceStrips(fb->fb, fb->fb.GetWidth(), fb->fb.GetHeight(), world, rank, rgb, depth, nsamples, spp_max, use_srgb != 0);
}
int qa_strip_row_range(int height, int world, int rank, int k, int *y0, int *y1)
{
  int a = 0, b = 0;
  const bool ok = StripRowRange(height, world, rank, k, a, b);
  if (y0) *y0 = a;
  if (y1) *y1 = b;
  return ok ? 1 : 0;
}
int qa_fb_save_image(const qa_fb *fb, const char *p) { return (fb && p && fb->fb.SaveImage(p)) ? QA_OK : Fail(QA_EIO, "cannot write image"); }
int qa_fb_save_z_image(qa_fb *fb, const char *p)
{
  if (!fb || !p) return Fail(QA_EINVAL, "null argument");
  fb->fb.ComputeZBufferImage();
  return fb->fb.SaveZImage(p) ? QA_OK : Fail(QA_EIO, "cannot write image");
}
int qa_fb_save_sample_count_image(qa_fb *fb, const char *p)
{
  if (!fb || !p) return Fail(QA_EINVAL, "null argument");
  fb->fb.ComputeSampleCountImage();
  return fb->fb.SaveSampleCountImage(p) ? QA_OK : Fail(QA_EIO, "cannot write image");
}

const uint8_t *qa_fb_z_image(qa_fb *fb)
{
  if (!fb) return nullptr;
  fb->fb.ComputeZBufferImage();
  return fb->fb.GetZBufferImage();
}
const uint8_t *qa_fb_sample_count_image(qa_fb *fb)
{
  if (!fb) return nullptr;
  fb->fb.ComputeSampleCountImage();
  return fb->fb.GetSampleCountImage();
}

void qa_tasking_init(void) { tasking::init(); }
uint64_t qa_tasking_get_num_of_threads(void) { return (uint64_t) tasking::get_num_of_threads(); }
void qa_tasking_set_num_of_threads(uint64_t n) { tasking::set_num_of_threads((size_t) n); }
int qa_tasking_parallel_for(uint64_t start, uint64_t end, uint64_t step, void (*fn)(uint64_t, void *), void *user)
{
  if (!fn || step == 0) return QA_EINVAL;
  try { tasking::parallel_for((size_t) start, (size_t) end, (size_t) step, [&](size_t i) { fn((uint64_t) i, user); }); }
  catch (...) { return QA_EINVAL; }
  return QA_OK;
}
void qa_tasking_signal_start(void) { tasking::signal_start(); }
void qa_tasking_signal_stop(void) { tasking::signal_stop(); }
int qa_tasking_has_stop_signal(void) { return tasking::has_stop_signal() ? 1 : 0; }

}  // extern "C"
