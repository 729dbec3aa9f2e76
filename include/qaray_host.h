/* qaray_host.h — C ABI of the host-side library (libqaray_host.so).
 *
 * The host side is C++ and mirrors qaray's own host layer: XML scene loader, Object / Material /
 * Light / Texture plugin surface, FrameBuffer ("renderImage") and the tasking entry point.  This
 * header is the thin C surface a foreign-language harness (ctypes, cgo, JNI ...) binds; C++
 * callers use the headers under qaray_amd/csrc/host directly.  Every function returns 0 on success or a
 * negative QA_E* code and never throws.
 *
 * Reference interfaces replaced (paths in the reference repo):
 *   qa_host_scene_load      LoadScene(const char*)                    src/parser/xmlload.h:16-18
 *   qa_host_scene_set_size  (no CLI flag exists; the oracle harness pokes scene.camera.imgWidth)
 *   qa_host_scene_flatten   Renderer::ComputeScene + scene graph      src/renderers/renderer.cpp:71-113
 *   qa_fb_*                 FrameBuffer                               src/fb/framebuffer.h:35-95
 *   qa_tasking_*            tasking::signal_start/stop, thread count  src/tasking/parallel_for.h:59-68
 *   qa_host_render          Renderer::ThreadRender (+ Renderer_MPI::Render's image saves)
 *                                                                     src/renderers/renderer.cpp:370-423
 */
#ifndef QARAY_HOST_H
#define QARAY_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QA_OK            0
#define QA_EINVAL       -1   /* bad argument */
#define QA_EIO          -2   /* file missing / unreadable / malformed */
#define QA_ENOMEM       -3
#define QA_EHIP         -4   /* HIP runtime error (qa_last_error() has the text) */
#define QA_ENOSCENE     -5   /* render requested before a scene was uploaded */
#define QA_EUNSUPPORTED -6   /* scene uses a feature the HIP path does not implement yet */
#define QA_ESTOPPED     -7   /* stop was requested (tasking::signal_stop) */

typedef struct qa_host_scene qa_host_scene;

/* Load an XML scene.  asset_root (may be NULL or "") is prepended to every relative OBJ /
 * texture path, replacing the reference's "run from the asset directory" convention. */
int qa_host_scene_load(const char *xml_path, const char *asset_root, qa_host_scene **out);
void qa_host_scene_destroy(qa_host_scene *scene);
/* Override <camera><width>/<height> (BASELINE configs render the reference scenes at other sizes). */
int qa_host_scene_set_size(qa_host_scene *scene, int width, int height);
int qa_host_scene_get_size(const qa_host_scene *scene, int *width, int *height);
/* Flatten into one relocatable blob (include/qa_flat_scene.h); release with qa_host_free. */
int qa_host_scene_flatten(const qa_host_scene *scene, unsigned char **blob, uint64_t *nbytes);
void qa_host_free(void *p);
const char *qa_host_last_error(void);

/* ---- FrameBuffer ("renderImage"): float radiance in, the reference's 8-bit products out ---- */
typedef struct qa_fb qa_fb;
int qa_fb_create(int width, int height, qa_fb **out);
void qa_fb_destroy(qa_fb *fb);
/* Deposit a rendered region (linear float RGB, depth, samples taken); applies the reference's
 * post-process: optional sRGB, clamp, round to 8 bit, sample-count byte, mask = 1
 * (src/renderers/renderer.cpp:347-365). */
int qa_fb_deposit(qa_fb *fb, int x0, int y0, int x1, int y1, const float *rgb, const float *depth,
                  const uint32_t *nsamples, int spp_max, int use_srgb);
const uint8_t *qa_fb_pixels(const qa_fb *fb);        /* RGB8 */
const float *qa_fb_zbuffer(const qa_fb *fb);
const uint8_t *qa_fb_sample_count(const qa_fb *fb);
const uint8_t *qa_fb_mask(const qa_fb *fb);
/* FrameBuffer::ComputeZBufferImage / ComputeSampleCountImage (src/fb/framebuffer.cpp:62-107), then the 8-bit image */
const uint8_t *qa_fb_z_image(qa_fb *fb);
const uint8_t *qa_fb_sample_count_image(qa_fb *fb);
int qa_fb_num_rendered_pixels(const qa_fb *fb);
/* Multi-GPU placement (the analogue of PlaceImage<T>, src/renderers/Renderer_MPI.cpp:103-122): rank `rank` of `world` owns
 * the 8-row strips rank, rank + world, ...; its PACKED float results (include/qaray_hip.h qa_render_strips_device) are
 * deposited into the rows they belong to.  Returns the number of strips placed (negative QA_E* on bad arguments).
 * qa_strip_row_range: image rows [y0, y1) of packed strip k of that rank; 0 when the rank has no such strip. */
int qa_fb_place_strips(qa_fb *fb, int world, int rank, const float *rgb, const float *depth, const uint32_t *nsamples,
                       int spp_max, int use_srgb);
int qa_strip_row_range(int height, int world, int rank, int k, int *y0, int *y1);
int qa_fb_save_image(const qa_fb *fb, const char *png_path);
int qa_fb_save_z_image(qa_fb *fb, const char *png_path);
int qa_fb_save_sample_count_image(qa_fb *fb, const char *png_path);

/* ---- tasking (src/tasking/parallel_for.h:59-68); C++ callers use qaray_hip::tasking of csrc/host/framebuffer.h,
 * which keeps the reference's names and signatures including parallel_for(start, end, step, std::function) and
 * ThreadLocalStorage<T> ---- */
void qa_tasking_init(void);
uint64_t qa_tasking_get_num_of_threads(void);
void qa_tasking_set_num_of_threads(uint64_t n);
void qa_tasking_signal_start(void);
void qa_tasking_signal_stop(void);
int qa_tasking_has_stop_signal(void);
/* parallel_for with a C callback: fn(i, user) for i = start, start + step, ... < end on the tasking threads */
int qa_tasking_parallel_for(uint64_t start, uint64_t end, uint64_t step, void (*fn)(uint64_t, void *), void *user);

#ifdef __cplusplus
}
#endif
#endif /* QARAY_HOST_H */
