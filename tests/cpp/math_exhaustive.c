/* tests/cpp/math_exhaustive.c - TEST TOOL: qa_device_math.h's expf / powf restatements (compiled for the
 * host inside libqaray_hip.so, entry point qa_test_math_host) against the host libm, bit for bit.
 *   math_exhaustive <libqaray_hip.so> 0 [stride]   expf over every stride-th float bit pattern
 *   math_exhaustive <libqaray_hip.so> 1 [stride]   powf over positive bases below 2 x 17 exponents + random pairs
 * stride 1 = the full sweep (4.3e9 / 3.9e9 evaluations, ~30 s on 8 cores); the test suite uses a larger stride. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
typedef int (*fn_t)(int, const float *, const float *, int, float *);
int main(int argc, char **argv)
{
  void *h = dlopen(argv[1], RTLD_NOW);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
  fn_t f = (fn_t) dlsym(h, "qa_test_math_host");
  const int mode = atoi(argv[2]);
  const long long stride = argc > 3 ? atoll(argv[3]) : 1;
  unsigned long long bad = 0, total = 0;
  if (mode == 0) {  // expf over all floats
#pragma omp parallel reduction(+ : bad, total)
    {
      const int CH = 1 << 16;
      float *x = malloc(CH * 4), *o = malloc(CH * 4);
#pragma omp for schedule(dynamic, 64)
      for (long long c = 0; c < (1LL << 32) / CH; c += stride) {
        for (int i = 0; i < CH; ++i) { uint32_t u = (uint32_t) (c * CH + i); memcpy(&x[i], &u, 4); }
        f(3, x, NULL, CH, o);
        for (int i = 0; i < CH; ++i) {
          float e = expf(x[i]);
          if (e != e && o[i] != o[i]) { total++; continue; }
          uint32_t a, b; memcpy(&a, &e, 4); memcpy(&b, &o[i], 4);
          total++;
          if (a != b) { if (bad < 5) fprintf(stderr, "expf(%a) libm %a mine %a\n", x[i], e, o[i]); bad++; }
        }
      }
    }
  } else {  // powf
    const float ys[] = {5.f, 2.f, 80.f, 20.f, 10.f, 50.f, 100.f, 40.f, 15.f, 25.f, 60.f, 1.f, 0.5f, 3.f, 7.5f, 1000.f, 33.3f};
    const int ny = sizeof(ys) / 4;
#pragma omp parallel reduction(+ : bad, total)
    {
      const int CH = 1 << 16;
      float *x = malloc(CH * 4), *y = malloc(CH * 4), *o = malloc(CH * 4);
#pragma omp for schedule(dynamic, 16)
      for (long long c = 0; c < 0x40000000LL / 5 / CH; c += stride) {   // positive floats below 2.0, every 5th
        for (int yi = 0; yi < ny; ++yi) {
          for (int i = 0; i < CH; ++i) { uint32_t u = (uint32_t) ((c * CH + i) * 5 + yi % 5); memcpy(&x[i], &u, 4); y[i] = ys[yi]; }
          f(2, x, y, CH, o);
          for (int i = 0; i < CH; ++i) {
            float e = powf(x[i], y[i]);
            uint32_t a, b; memcpy(&a, &e, 4); memcpy(&b, &o[i], 4);
            total++;
            if (a != b) { if (bad < 5) fprintf(stderr, "powf(%a,%a) libm %a mine %a\n", x[i], y[i], e, o[i]); bad++; }
          }
        }
      }
      // random pairs
      unsigned s = 12345u + 977u * omp_get_thread_num();
      for (int rep = 0; rep < 400; rep += (int) (stride > 400 ? 400 : stride)) {
        for (int i = 0; i < CH; ++i) {
          s = s * 1664525u + 1013904223u; uint32_t u = (s >> 2) % 0x41000000u; memcpy(&x[i], &u, 4);   // (0, 8)
          s = s * 1664525u + 1013904223u; y[i] = (float) (s >> 8) / 16777216.f * 300.f - 50.f;
        }
        f(2, x, y, CH, o);
        for (int i = 0; i < CH; ++i) {
          float e = powf(x[i], y[i]);
          uint32_t a, b; memcpy(&a, &e, 4); memcpy(&b, &o[i], 4);
          total++;
          if (a != b) { if (bad < 5) fprintf(stderr, "powf(%a,%a) libm %a mine %a\n", x[i], y[i], e, o[i]); bad++; }
        }
      }
    }
  }
  printf("mode %d: %llu evaluations, %llu mismatches\n", mode, total, bad);
  return bad != 0;
}
