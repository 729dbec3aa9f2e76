/* qa_seed.h — per-pixel RNG stream seeding shared by the HIP path, the CPU oracle and the
 * reference harness (plain C, header-only).
 *
 * The reference seeds one xorshift32 stream per worker thread from srand(time())/rand()
 * (reference src/samplers/Sampler_Marsaglia.cpp:32-42), which is neither reproducible nor
 * expressible on a GPU.  Every implementation in this repo instead gives each PIXEL its own
 * stream, created exactly the way the reference creates one: seed[0] = rand() % 999999999 + 1,
 * with rand() replaced by a deterministic function of (global seed, pixel index).  The harness
 * built against the real reference interposes rand() with qa_pixel_rand() and re-creates the
 * thread's sampler before each pixel, so its output is the reference's own arithmetic on this
 * stream.  pixel = j * image_width + i on the FULL image, so crops, regions and
 * multi-GPU partitions see identical streams.
 */
#ifndef QA_SEED_H
#define QA_SEED_H

#include <stdint.h>

#define QA_DEFAULT_SEED 0x51A7A7u

#if defined(__HIPCC__)
#define QA_HD __host__ __device__
#else
#define QA_HD
#endif

/* value the interposed rand() returns: non-negative 31-bit, like glibc's rand() */
static inline QA_HD uint32_t qa_pixel_rand(uint32_t seed, uint32_t pixel)
{
  uint32_t h = seed ^ (pixel * 0x9E3779B9u);
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h & 0x7FFFFFFFu;
}

/* initial xorshift32 state of the pixel's stream (never 0) */
static inline QA_HD uint32_t qa_pixel_seed(uint32_t seed, uint32_t pixel)
{
  return qa_pixel_rand(seed, pixel) % 999999999u + 1u;
}

#endif /* QA_SEED_H */
