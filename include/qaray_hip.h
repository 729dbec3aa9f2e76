/* qaray_hip.h — C ABI of the MI355X hot path (libqaray_hip.so, gfx950 only).
 *
 * The reference has no FFI: its hot path is reached through in-process C++ virtual calls
 * (SURVEY.md §8b).  This header is the boundary a maintainer of the reference would bind from
 * Renderer::ThreadRender; each entry point names the reference interface it stands in for
 * (paths in the reference repo).  Plain pointers and sizes only; one host thread per context;
 * every call returns QA_OK (0) or a negative QA_E* code (include/qaray_host.h) and never throws.
 * The library owns all device memory behind the opaque handle; callers own every buffer they
 * pass in.  There is NO CPU fallback: without a gfx950 device qa_ctx_create fails.
 *
 *   qa_ctx_create / destroy        Renderer::Renderer / Terminate      src/renderers/renderer.cpp:67-70,293-298
 *   qa_scene_upload                Renderer::ComputeScene (scene side)  src/renderers/renderer.cpp:71-113
 *   qa_scene_upload_device         same, blob already in HBM (after an RCCL broadcast); the
 *                                  reference instead re-parses the XML on every rank
 *                                                                       src/renderers/Renderer_MPI.cpp:54
 *   qa_render_region[_device]      Renderer::ThreadRender -> PixelRender over a pixel region:
 *                                  camera ray, Scene::TraceNodeNormal, Material::Shade,
 *                                  Light::Illuminate/GenLight::Shadow, SuperSamplerHalton
 *                                                                       src/renderers/renderer.cpp:302-423
 *   qa_render_strips_device        Renderer_MPI's rank-strided ThreadRender (every rank renders
 *                                  tiles rank, rank+size, ...)          src/renderers/renderer.cpp:383-387
 *   qa_request_stop / qa_clear_stop   tasking::signal_stop / signal_start  src/tasking/parallel_for.cpp:70-73
 *   qa_photon_maps_build / clear   the photon-map block of Renderer::ComputeScene with
 *                                  RendererParam::usePhotonMap (-use-photon-map): photon tracing,
 *                                  power scaling, kd-tree; afterwards qa_render_* shade with
 *                                  Scene::usePhotonMap = true        src/renderers/renderer.cpp:114-291,
 *                                                                       src/materials/MtlBlinn_PhotonMap.cpp:349-458
 *   qa_get_counters                (no counterpart: the reference only prints wall-clock)
 *   qa_get_kernel_time             Renderer::StartTimer/StopTimer       src/renderers/renderer.cpp:42-63
 */
#ifndef QARAY_HIP_H
#define QARAY_HIP_H

#include <stdint.h>

#include "qa_photon.h"
#include "qaray_host.h" /* QA_OK / QA_E* */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qa_ctx qa_ctx;

typedef struct qa_counters {
  uint64_t samples;        /* camera paths started (1 sample = 1 iteration of PixelRender's loop) */
  uint64_t casts_normal;   /* closest-hit casts: camera + secondary rays */
  uint64_t casts_shadow;   /* any-hit casts */
  uint64_t bvh_nodes;      /* BVH nodes popped (only counted by stats launches, else 0) */
  uint64_t tri_tests;      /* triangle tests   (only counted by stats launches, else 0) */
  uint64_t pixels;         /* pixels completed */
} qa_counters;

/* flags for qa_render_region* */
#define QA_RENDER_STATS 1u  /* also count BVH nodes / triangle tests (slower kernel variant) */

int qa_ctx_create(int device_id, qa_ctx **out);
int qa_ctx_destroy(qa_ctx *ctx);

/* Upload a flattened scene (include/qa_flat_scene.h) from host memory / adopt a copy of one that
 * already sits in device memory.  Replaces any previous scene of the context. */
int qa_scene_upload(qa_ctx *ctx, const void *host_blob, uint64_t nbytes);
int qa_scene_upload_device(qa_ctx *ctx, const void *device_blob, uint64_t nbytes);

/* Render pixels [x0,x1) x [y0,y1) of the scene's image.  Outputs are region-local, row-major:
 * rgb (y1-y0)*(x1-x0)*3 floats of LINEAR mean radiance (sRGB/quantisation stay in FrameBuffer),
 * depth: hit distance of sample 0 (1e30 on a miss), nsamples: samples taken (0 = pixel skipped
 * because a stop was requested).  Pixel (i,j) always uses RNG stream qa_pixel_seed(seed,
 * j*width+i) (include/qa_seed.h), so any partition of the image gives identical pixels.
 * 1 <= spp_min <= spp_max: spp_min samples always, up to spp_max while the running variance
 * exceeds the reference's thresholds (SuperSamplerHalton::Loop, src/scene/scene.cpp:92-97).
 * The host variant synchronises and copies back; the device variant writes device buffers and
 * only enqueues work on `hip_stream` (a hipStream_t; NULL = the context's own non-blocking stream -
 * note that NULL is also the handle of the legacy default stream, which therefore cannot be
 * selected: consumers on other streams must wait for qa_synchronize or pass their own stream). */
int qa_render_region(qa_ctx *ctx, int x0, int y0, int x1, int y1, int spp_min, int spp_max,
                     int max_bounce, uint32_t seed, uint32_t flags, float *rgb, float *depth,
                     uint32_t *nsamples);
int qa_render_region_device(qa_ctx *ctx, int x0, int y0, int x1, int y1, int spp_min, int spp_max,
                            int max_bounce, uint32_t seed, uint32_t flags, float *d_rgb,
                            float *d_depth, uint32_t *d_nsamples, void *hip_stream);
/* Image-space partition between GPUs.  The region is cut into horizontal strips of QA_STRIP_ROWS
 * (8) pixel rows; this call renders strips first_strip, first_strip + strip_step, ... (rank r of
 * n: first_strip = r, strip_step = n), i.e. the reference's round-robin tile ownership
 *   tasking::parallel_for(tileStart = mpiRank, tileStop, tileStep = mpiSize)   src/renderers/renderer.cpp:383-387
 * Outputs are PACKED strip after strip: row (k*8 + i) of the buffers is row
 * y0 + (first_strip + k*strip_step)*8 + i of the image; buffers hold qa_strip_count()*8 rows of
 * (x1-x0) pixels (rows of a ragged last strip that fall below y1 keep nsamples = 0). */
#define QA_STRIP_ROWS 8
int qa_render_strips_device(qa_ctx *ctx, int x0, int y0, int x1, int y1, int first_strip, int strip_step,
                            int spp_min, int spp_max, int max_bounce, uint32_t seed, uint32_t flags,
                            float *d_rgb, float *d_depth, uint32_t *d_nsamples, void *hip_stream);
int qa_strip_count(int y0, int y1, int first_strip, int strip_step);
/* Wait for everything enqueued by this context. */
int qa_synchronize(qa_ctx *ctx);

/* Photon / caustics maps (the reference's -use-photon-map mode).  qa_photon_maps_build traces
 * photons from the scene's point lights on the GPU - one RNG stream per emission, see
 * include/qa_photon.h - keeps the first params->*.size of them in the reference's order, scales
 * their powers by 1 / emitted rays, balances each map into cyPhotonMap's kd-tree and leaves both
 * resident in HBM; from then on qa_render_* shade with Scene::usePhotonMap = true (a DIFFUSE
 * selection gathers 100 nearest photons from the caustics map, and from the photon map instead of
 * bouncing after a diffuse bounce).  The maps live until qa_photon_maps_clear or the next scene
 * upload.  Deterministic in (scene, params, seed): every rank of a multi-GPU job builds the same maps.
 * Errors: QA_EUNSUPPORTED when the scene has no point light or a map cannot be filled within
 * QA_PHOTON_MAX_EMISSIONS (the reference divides by zero / loops forever in these cases).
 * qa_photon_maps_info: numOfEmittedRays and loop iterations per map ([0] photon, [1] caustics);
 * qa_photon_maps_download: the balanced records as they sit in HBM, size + 1 entries, [0] unused
 * ([1..size] is byte-compatible with the reference's photonmap.dat / caustics.dat dumps). */
int qa_photon_maps_build(qa_ctx *ctx, const qa_photon_params *params, uint32_t seed);
int qa_photon_maps_clear(qa_ctx *ctx);
int qa_photon_maps_info(qa_ctx *ctx, uint64_t emitted[2], uint64_t emissions[2]);
int qa_photon_maps_download(qa_ctx *ctx, int which, qa_photon *out, uint64_t capacity);

int qa_request_stop(qa_ctx *ctx);
int qa_clear_stop(qa_ctx *ctx);

/* Counters accumulated since the last reset (synchronises the context first). */
int qa_get_counters(qa_ctx *ctx, qa_counters *out);
int qa_reset_counters(qa_ctx *ctx);
/* Sum of the integrator kernel's durations (HIP events recorded around each launch, on the
 * stream it was launched on) and the number of launches since the last reset; synchronises. */
int qa_get_kernel_time(qa_ctx *ctx, double *total_ms, uint64_t *launches);
int qa_reset_kernel_time(qa_ctx *ctx);

/* Which integrator the uploaded scene runs on, e.g. "qa_integrate<RES=1,LIGHTS=0,TEX=0,AREA=0>" (one persistent
 * megakernel, LDS-resident scene), "qa_integrate_cs<LIGHTS=1,TEX=1>" (megakernel with cooperative mesh walks) or
 * "staged: wf_logic + wf_cull + wf_trace + wf_redo (n tile groups)".  Before the first frame after an upload (or after
 * qa_set_pipeline / qa_set_option) this is the plan; afterwards it names what the LAST qa_render_* call launched,
 * including "+ photon-map gathers" / "counting variant" for those frames.  Valid until the next call on the context. */
const char *qa_get_kernel_name(qa_ctx *ctx);
/* Diagnostics of the staged integrator since the last qa_reset_counters (synchronises):
 * [0] passes, [1] closest-hit rays, [2] shadow rays, [3] BVH jobs queued, [4] rays repeated exactly,
 * [5] jobs finished, [6] node steps, [7] leaf steps, [8] triangle tests, [9] hits that failed the order check,
 * [10] jobs suspended by the step budget, [11] lane slots used in traversal rounds, [12] traversal rounds (waves).
 * Lane utilisation of the traversal = [11] / (64 * [12]); geometry bytes = [6] * 64 + [8] * 48. */
#define QA_STAGED_STATS 13
int qa_get_staged_stats(qa_ctx *ctx, uint64_t out[QA_STAGED_STATS]);

/* Scenes whose geometry does not fit LDS can run on two integrators that return the same bits: the persistent
 * megakernel (for scenes without area lights with cooperative mesh walks: the whole wave walks its mesh queries from a pool
 * of (ray, node) items in LDS, qa_kernel_cs.h) and the staged pipeline (logic / cull / trace / redo stages exchanging rays
 * through queues in HBM, qa_wf.h).  QA_PIPE_AUTO (the default) and QA_PIPE_MEGA run the megakernel - the faster integrator
 * on every scene measured; QA_PIPE_STAGED runs the staged pipeline, which is kept as an independent bitwise cross-check.
 * Scenes the staged pipeline cannot take (LDS-resident scenes, area lights, photon maps, QA_RENDER_STATS frames, > 4
 * non-ambient lights, > 31 nodes) always run on the megakernel, whatever the mode. */
#define QA_PIPE_MEGA 0
#define QA_PIPE_STAGED 1
#define QA_PIPE_AUTO 2
int qa_set_pipeline(qa_ctx *ctx, int mode);

/* Options an embedding application or a test may set (the product library reads no environment variable of its own;
 * the developer knobs of the A/B scripts exist only in builds made with -DQA_DEV_KNOBS):
 *   "coop"           1 (default) / 0: cooperative mesh walks where the scene allows them; 0 = every lane walks its own ray
 *   "cs_cull"        1 (default) / 0: the cooperative kernel skips scene-graph nodes whose bounds a wave's rays all miss; 0 = every
 *                    node is visited as the reference does (same bits either way: A/B tests)
 *   "cs_force_exact" tests: bit 0 / bit 1 send every closest-hit / shadow query of the cooperative kernel to its exact sequential
 *                    walks (the path a tie, a failed order check or a full pool takes); same bits, much slower
 *   "walk_zero_terms" tests: 1 = the shadow ray of a light whose term is zero in every component whatever the ray finds (the surface
 *                    faces away from the light) is walked all the same, as the reference does; 0 (default) = counted, not walked
 *                    (same bits, same counters)
 *   "cs_pool_limit"  n > 0: upper bound for the pool of the cooperative walks (tests: forces the overflow path); 0 = none
 *   "sync_samples"   -1 (default: per scene) / 0 / 1: a wave starts the next samples of its 64 pixels together; n >= 2 (cooperative
 *                    kernel; elsewhere like 1): finished paths wait until n of the wave's have gathered
 *   "chunk_spp"      -1 (default: per frame) / 0 / n: the per-lane kernels hand a tile's samples out in chunks - n samples first, then
 *   "chunk_tail"     chunks of this many (0: an eighth of the frame's spp) - so that a frame of few tiles per wave ends on small work
 *                    items; a pixel's state waits in device memory between chunks (same samples in the same order: same bits)
 *   "tile_order"     1 (default) / 0: tiles handed out centre-first
 *   "staged_groups"  1 (default) .. 8 tile groups of the staged pipeline, each on its own stream; more than one only pays
 *                    when the process started the HIP runtime with GPU_MAX_HW_QUEUES >= 8
 *   "verbose"        1: tree statistics and launch shapes on stderr at upload
 * Unknown names return QA_EINVAL. */
int qa_set_option(qa_ctx *ctx, const char *name, long long value);

/* Test support: fills the private (scratch) segment of every wave slot of the device with `pattern`, on every stream this context
 * launches on, and waits.  A frame rendered afterwards must not depend on the pattern: one that does reads scratch memory it never
 * wrote (the compiler hazard of DESIGN.md 5b; tests/test_gpu_parity.py renders with several patterns).  No reference counterpart. */
int qa_debug_scrub_scratch(qa_ctx *ctx, uint32_t pattern);

/* Launch geometry (0 = library default). blocks_per_cu * CUs persistent workgroups of `threads`. */
int qa_set_launch_config(qa_ctx *ctx, int blocks_per_cu, int threads_per_block);

const char *qa_last_error(void);

/* Self-test hooks for the device math library (qaray_amd/csrc/hip/qa_device_math.h): the device
 * sinf/cosf evaluated on the GPU, and the same source compiled for the host (fn: 0 sinf, 1 cosf,
 * 2 powf(x,y), 3 expf). */
int qa_test_sincosf_device(const float *x, int n, float *s, float *c);
int qa_test_math_device(int fn, const float *x, const float *y, int n, float *out);
int qa_test_math_host(int fn, const float *x, const float *y, int n, float *out);

#ifdef __cplusplus
}
#endif
#endif /* QARAY_HIP_H */
