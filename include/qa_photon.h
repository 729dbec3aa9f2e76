/* qa_photon.h — photon / caustics map records and parameters shared by the HIP path, the CPU
 * oracle and the reference harness (plain C).
 *
 * Reference: Renderer::ComputeScene builds two cyPhotonMap instances when started with
 * -use-photon-map (src/renderers/renderer.cpp:114-291), MtlBlinn_PhotonMap::Shade gathers from
 * them (src/materials/MtlBlinn_PhotonMap.cpp:349-458).  qa_photon is byte-compatible with
 * cy::PhotonMap::Photon (src/ext/cyPhotonMap.h:83-103; 24 bytes), so the harness can dump the
 * reference's balanced arrays and the tests can compare them record by record.
 *
 * RNG contract.  The reference traces photons in one serial loop on the main thread's stream;
 * like the per-pixel streams of qa_seed.h, every EMISSION (one iteration of the while(true) loop,
 * renderer.cpp:146/217) gets its own xorshift32 stream here:
 *     seed[0] = qa_pixel_seed(seed ^ QA_STREAM_PHOTON | QA_STREAM_CAUSTICS, emission_index)
 * so that emissions can be traced in parallel while the stored photons keep the serial loop's
 * order (emission-major, bounce-minor) and its stopping rule.  oracle/ref_harness.cpp drives the
 * reference's own Light::RandomPhoton / Scene::TraceNodeNormal / Material::RandomPhotonBounce /
 * cyPhotonMap code on these streams.
 */
#ifndef QA_PHOTON_H
#define QA_PHOTON_H

#include <stdint.h>

#include "qa_seed.h"

#define QA_STREAM_PHOTON 0x50484F54u   /* "PHOT" */
#define QA_STREAM_CAUSTICS 0x43415553u /* "CAUS" */
#define QA_PHOTON_GATHER 100           /* EstimateIrradiance<100>, MtlBlinn_PhotonMap.cpp:429,447 */

/* The reference's emission loop only ends when the map is full, i.e. never in a scene whose
 * surfaces cannot store a photon of that kind (e.g. a caustics map without specular objects).
 * Every implementation here gives up after this many emissions and reports an error instead. */
#define QA_PHOTON_MAX_EMISSIONS(size) (1024ull * (uint64_t) (size) + 65536ull)

typedef struct qa_photon {
  float pos[3];
  float power;           /* largest channel of the power */
  uint8_t rgb[3];        /* 255 * power colour / largest channel */
  uint8_t plane_dirz;    /* bits 0-1: kd-tree split axis; bit 3: direction z <= 0 */
  int16_t dirx, diry;    /* direction x, y * 0x7FFF */
} qa_photon;

typedef struct qa_photon_map_params {
  uint32_t size;    /* photons to store   (RendererParam::photonMapSize 10000 / causticsMapSize 1000) */
  uint32_t bounce;  /* hits per emission  (photonMapBounce / causticsMapBounce, 20) */
  float radius;     /* gather radius      (photonMapRadius 0.2 / causticsMapRadius 1.0) */
} qa_photon_map_params;

typedef struct qa_photon_params {   /* src/renderers/renderer.h:51-57 */
  qa_photon_map_params photon, caustics;
} qa_photon_params;

static inline QA_HD uint32_t qa_photon_seed(uint32_t seed, uint32_t stream, uint32_t emission)
{
  return qa_pixel_seed(seed ^ stream, emission);
}

#endif /* QA_PHOTON_H */
