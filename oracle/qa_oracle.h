/* oracle/qa_oracle.h — TEST INFRASTRUCTURE ONLY (checker, never the product path).
 *
 * CPU restatement, in plain C, of the reference's per-pixel Monte-Carlo integrator.  It reads
 * the flattened scene blob (include/qa_flat_scene.h) and follows the reference function by
 * function and operation by operation (each function in qa_oracle.c cites the reference
 * file:line it restates), calling the same glibc libm entry points the reference build calls,
 * so that on the same host its output is expected to equal the real reference's
 * (oracle/_ref/ref_harness) BIT FOR BIT.  tests/test_oracle_vs_reference.py checks exactly
 * that against committed golden vectors produced by the reference.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 */
#ifndef QA_ORACLE_H
#define QA_ORACLE_H

#include <stdint.h>

#include "qa_photon.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qa_oracle_counters {
  uint64_t samples;
  uint64_t casts_normal;   /* closest-hit casts (camera + secondary rays) */
  uint64_t casts_shadow;   /* any-hit casts */
  uint64_t bvh_nodes;      /* BVH nodes popped from the traversal stack */
  uint64_t tri_tests;      /* IntersectTriangle calls */
} qa_oracle_counters;

/* Renders pixels [x0,x1) x [y0,y1) of the blob's image.  rgb: (y1-y0)*(x1-x0)*3 floats, linear
 * mean radiance; depth: first-sample hit distance (1e30 on miss); ns: samples taken per pixel.
 * threads <= 0: all OpenMP threads.  Returns 0, or a negative value on a malformed blob. */
int qa_oracle_render(const void *blob, int x0, int y0, int x1, int y1, int spp_min, int spp_max,
                     int max_bounce, uint32_t seed, float *rgb, float *depth, uint32_t *ns,
                     int threads, qa_oracle_counters *counters);

/* Photon / caustics maps of Renderer::ComputeScene (-use-photon-map), one RNG stream per emission
 * (include/qa_photon.h).  photon / caustics: pp->*.size + 1 records, [0] unused, [1..size] the
 * balanced kd-tree in heap order (byte-compatible with the reference's photonmap.dat / caustics.dat
 * dumps).  emitted: numOfEmittedRays per map; emissions: loop iterations per map. */
int qa_oracle_photon_build(const void *blob, const qa_photon_params *pp, uint32_t seed, qa_photon *photon,
                           qa_photon *caustics, uint64_t emitted[2], uint64_t emissions[2]);

/* qa_oracle_render with Scene::usePhotonMap = true when pp != NULL (maps from qa_oracle_photon_build). */
int qa_oracle_render_pm(const void *blob, int x0, int y0, int x1, int y1, int spp_min, int spp_max,
                        int max_bounce, uint32_t seed, float *rgb, float *depth, uint32_t *ns,
                        int threads, qa_oracle_counters *counters, const qa_photon_params *pp,
                        const qa_photon *photon, const qa_photon *caustics);

/* Small pieces exposed for unit tests. */
float qa_oracle_halton(int index, int base);
void  qa_oracle_rng_stream(uint32_t seed, uint32_t pixel, int n, float *out);

#ifdef __cplusplus
}
#endif
#endif
