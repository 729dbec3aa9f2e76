// qa_device_math.h — device-side float3 algebra and libm replacements for the integrator.
//
// Parity rule: the CPU reference evaluates everything in IEEE fp32 without FMA contraction, in
// the order GLM's scalar operators spell out (oracle/qa_oracle.c restates it).  Device code uses
// the same order and this library is built with -ffp-contract=off and correctly rounded
// divide/sqrt, so +,-,*,/ and sqrt give the reference's bits.  The only places where bits can
// differ are libm calls (sinf, cosf, powf, expf, asinf, tanf and the double asin/atan2 of the
// sphere's texture coordinates): glibc's float routines are evaluated in double and are correctly
// rounded for the vast majority of arguments, so the device versions below also evaluate in fp64
// (MI355X runs fp64 vector math at half the fp32 rate) and round once to fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qa {

struct f3 { float x, y, z; };

__host__ __device__ __forceinline__ f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__host__ __device__ __forceinline__ f3 ld3(const float *p) { return F3(p[0], p[1], p[2]); }
__host__ __device__ __forceinline__ f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ __forceinline__ f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ __forceinline__ f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
__host__ __device__ __forceinline__ f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
__host__ __device__ __forceinline__ f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
__host__ __device__ __forceinline__ f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
// glm::dot (func_geometric.inl:54-61): (x*x' + y*y') + z*z'
__host__ __device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// glm::cross (func_geometric.inl:74-85)
__host__ __device__ __forceinline__ f3 cross(f3 a, f3 b)
{
  return F3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__host__ __device__ __forceinline__ float qsqrt(float x) { return __builtin_sqrtf(x); }
__host__ __device__ __forceinline__ float length(f3 a) { return qsqrt(dot(a, a)); }
// glm::normalize: v * (1 / sqrt(dot(v, v)))
__host__ __device__ __forceinline__ f3 normalize(f3 a) { return a * (1.f / qsqrt(dot(a, a))); }
// mat3 (column-major) * vec (type_mat3x3.inl:428-434)
__host__ __device__ __forceinline__ f3 mulMV(const float *m, f3 v)
{
  return F3(m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z,
            m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
// Transformation::TransposeMult (src/core/transform.cpp:49-56)
__host__ __device__ __forceinline__ f3 mulTMV(const float *m, f3 d)
{
  return F3(dot(F3(m[0], m[1], m[2]), d), dot(F3(m[3], m[4], m[5]), d), dot(F3(m[6], m[7], m[8]), d));
}

// MIN/MAX/ABS exactly as the reference's macros (src/math/math.h:104-107), NaN behaviour included
__host__ __device__ __forceinline__ float qmin(float x, float y) { return x < y ? x : y; }
__host__ __device__ __forceinline__ float qmax(float x, float y) { return x > y ? x : y; }
__host__ __device__ __forceinline__ float qabs(float x) { return x > 0 ? x : -x; }

#define QA_PI 3.14159274101257324219f /* (float) M_PI */

// Rec.709 luma (src/math/math.h:128-131)
__host__ __device__ __forceinline__ float luma(f3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }

// ---------------------------------------------------------------------------------------------
// sinf / cosf: the algorithm glibc >= 2.28 uses (Arm Optimized Routines' sincosf: reduce by pi/2
// with a 2^24-prescaled 2/pi, then degree-7 / degree-8 minimax kernels evaluated in fp64, one
// rounding to fp32).  Restated here with the published coefficients so that the device returns
// glibc's bits: tests/test_device_math.py compares it with the host libm over every float in
// [0, 2*pi] (the only range the integrator uses: phi = 2*pi*r with r in [0,1]).
// Arguments outside [2^-12, 120) fall back to the correctly rounded fp64 evaluation below.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ float sincos_kernel(double x, double x2, bool cosine, bool negate)
{
  // coefficients of the quadrant-0 table; quadrants 2,3 use the negated cosine set
  const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
               c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  if (!cosine) {
    const double x3 = x * x2;
    const double t1 = s2 + x2 * s3;
    const double x5 = x3 * x2;
    const double s = x + x3 * s1;
    return (float) (s + x5 * t1);
  }
  const double sg = negate ? -1.0 : 1.0;
  const double x4 = x2 * x2;
  const double t2 = sg * c3 + x2 * (sg * c4);
  const double t1 = sg * c0 + x2 * (sg * c1);
  const double x6 = x4 * x2;
  const double c = t1 + x4 * (sg * c2);
  return (float) (c + x6 * t2);
}

__host__ __device__ __forceinline__ void sincos_f64(double x, double *s, double *c);

// want_cos = false: sinf(y); true: cosf(y)
__host__ __device__ __forceinline__ float sincosf_one(float y, bool want_cos)
{
  const float ay = y > 0 ? y : -y;
  double x = (double) y;
  if (ay < 0x1.92p-1f) {  // top-12-bit compare against pi/4, as the original does
    if (ay < 0x1p-12f) return want_cos ? 1.0f : y;
    return sincos_kernel(x, x * x, want_cos, false);
  }
  if (ay < 120.0f) {
    const double r = x * 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
    const int n = ((int32_t) r + 0x800000) >> 24;
    x = x - (double) n * 0x1.921FB54442D18p0;
    const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    const int m = want_cos ? (n ^ 1) : n;
    return sincos_kernel(x * sign, x * x, (m & 1) != 0, (n & 2) != 0);
  }
  double ds, dc;
  sincos_f64(x, &ds, &dc);
  return want_cos ? (float) dc : (float) ds;
}
__host__ __device__ __forceinline__ float qsinf(float x) { return sincosf_one(x, false); }
__host__ __device__ __forceinline__ float qcosf(float x) { return sincosf_one(x, true); }

// Correctly rounded fallback: Cody-Waite reduction by pi/2 (three-term constant) + Taylor
// kernels on |r| <= pi/4 in fp64 (truncation error < 3e-14).  Valid for |x| < ~1e5.
__host__ __device__ __forceinline__ void sincos_f64(double x, double *s, double *c)
{
  const double two_over_pi = 0.63661977236758134308;
  const double pio2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
  const double pio2_2 = 6.07710050630396597660e-11;   // next 33 bits
  const double pio2_3 = 2.02226624879595063154e-21;   // remainder
  const double fn = __builtin_rint(x * two_over_pi);
  double r = x - fn * pio2_1;
  r = r - fn * pio2_2;
  r = r - fn * pio2_3;
  const double r2 = r * r;
  double ps = 1.0 - r2 * (1.0 / 156.0);
  ps = 1.0 - r2 * (1.0 / 110.0) * ps;
  ps = 1.0 - r2 * (1.0 / 72.0) * ps;
  ps = 1.0 - r2 * (1.0 / 42.0) * ps;
  ps = 1.0 - r2 * (1.0 / 20.0) * ps;
  ps = 1.0 - r2 * (1.0 / 6.0) * ps;
  const double sr = r * ps;
  double pc = 1.0 - r2 * (1.0 / 182.0);
  pc = 1.0 - r2 * (1.0 / 132.0) * pc;
  pc = 1.0 - r2 * (1.0 / 90.0) * pc;
  pc = 1.0 - r2 * (1.0 / 56.0) * pc;
  pc = 1.0 - r2 * (1.0 / 30.0) * pc;
  pc = 1.0 - r2 * (1.0 / 12.0) * pc;
  const double cr = 1.0 - r2 * 0.5 * pc;
  const int q = ((int) fn) & 3;
  const double ss = (q & 1) ? cr : sr;
  const double cc = (q & 1) ? sr : cr;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}

// powf for the integrator's domain (base >= 0): (float) 2^(y * log2(x)) in fp64.
__host__ __device__ __forceinline__ float qpowf(float x, float y)
{
  if (y == 0.f) return 1.f;
  if (x == 1.f) return 1.f;
  if (x == 0.f) return y > 0.f ? 0.f : __builtin_inff();
  if (y == 1.f) return x;
  return (float) exp2((double) y * log2((double) x));
}
__host__ __device__ __forceinline__ float qexpf(float x) { return (float) exp((double) x); }
__host__ __device__ __forceinline__ float qasinf(float x) { return (float) asin((double) x); }
__host__ __device__ __forceinline__ float qtanf(float x) { return (float) tan((double) x); }

}  // namespace qa
