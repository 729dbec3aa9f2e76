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
#ifdef QA_ABL_FAST_SINCOS   /* ablation builds only (timing what the exact routines cost): never parity */
__host__ __device__ __forceinline__ float qsinf(float x) { return __builtin_sinf(x); }
__host__ __device__ __forceinline__ float qcosf(float x) { return __builtin_cosf(x); }
#else
__host__ __device__ __forceinline__ float qsinf(float x) { return sincosf_one(x, false); }
__host__ __device__ __forceinline__ float qcosf(float x) { return sincosf_one(x, true); }
#endif

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

// ---------------------------------------------------------------------------------------------
// expf / powf: the algorithms glibc >= 2.28 uses (Arm Optimized Routines' expf and powf: table-
// driven 2^(k/32) with a cubic in fp64; powf takes log2(x) from a 16-entry table plus a degree-5
// polynomial), restated with the published tables.  Every a*b+c below is a fused multiply-add
// where glibc's x86-64 build for FMA-capable CPUs (the ifunc variant __expf_fma / __powf_fma that
// the host runs, here and on the GPU box) fuses it, and a separate multiply and add where it does
// not, so that the device returns the bits the reference's libm calls return.
// tests/test_device_math.py compares both with the host libm (expf over every float; powf over
// random pairs and over the integrator's exponents).
// ---------------------------------------------------------------------------------------------
namespace mathtab {
// 2^(i/32) as a double bit pattern minus (i << 47): adding (k << 47) yields 2^(k/32) for any int k
constexpr uint64_t kExp2[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
    0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
    0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
// log2 table: for the 16 sub-intervals of [0x1.66p-1, 0x1.66p0): 1/c and log2(c)
constexpr double kLog2[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
}  // namespace mathtab

__host__ __device__ __forceinline__ double qa_asdouble(uint64_t u)
{
  double d;
  __builtin_memcpy(&d, &u, 8);
  return d;
}
__host__ __device__ __forceinline__ uint64_t qa_asuint64(double d)
{
  uint64_t u;
  __builtin_memcpy(&u, &d, 8);
  return u;
}
__host__ __device__ __forceinline__ uint32_t qa_asuint(float f)
{
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}

__host__ __device__ __forceinline__ float qexpf(float x)
{
  const uint32_t abstop = (qa_asuint(x) >> 20) & 0x7ff;
  if (abstop >= 0x42b) {   // |x| >= 88 or NaN: the special cases of the original
    if (qa_asuint(x) == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8) return x + x;
    if (x > 0x1.62e42ep6f) return __builtin_inff();   // overflow
    if (x < -0x1.9fe368p6f) return 0.0f;              // underflow
  }
  const double xd = (double) x;
  const double InvLn2N = 0x1.71547652b82fep+5, SHIFT = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
  // x * N/ln2 = k + r, r in [-1/2, 1/2]
  double kd = __builtin_fma(InvLn2N, xd, SHIFT);
  const uint64_t ki = qa_asuint64(kd);
  kd -= SHIFT;
  const double r = __builtin_fma(InvLn2N, xd, -kd);
  const double s = qa_asdouble(mathtab::kExp2[ki % 32] + (ki << 47));
  const double z = __builtin_fma(C0, r, C1);
  const double r2 = r * r;
  double y = __builtin_fma(C2, r, 1.0);
  y = __builtin_fma(z, r2, y);
  return (float) (y * s);
}

// powf for a non-negative base (all the integrator needs: cosines, 1 - |cos|, cone ratios)
__host__ __device__ __forceinline__ float qpowf(float x, float y)
{
  if (y == 0.f) return 1.f;
  if (x == 1.f) return 1.f;
  if (x == 0.f) return y > 0.f ? 0.f : __builtin_inff();
  if (!(x > 0.f) || !(x < __builtin_inff()) || !(y == y) || y == __builtin_inff() || y == -__builtin_inff())
    return (float) exp2((double) y * log2((double) x));  // outside the restated domain: not reached by the integrator
  uint32_t ix = qa_asuint(x);
  if (ix < 0x00800000u) {  // subnormal base: normalise
    ix = qa_asuint(x * 0x1p23f);
    ix &= 0x7fffffffu;
    ix -= 23u << 23;
  }
  // log2(x) = log1p(z/c - 1)/ln2 + log2(c) + k
  const uint32_t tmp = ix - 0x3f330000u;
  const uint32_t i = (tmp >> 19) % 16;
  const uint32_t top = tmp & 0xff800000u;
  const uint32_t iz = ix - top;
  const int k = (int32_t) top >> 23;
  const double invc = mathtab::kLog2[i][0], logc = mathtab::kLog2[i][1];
  float zf;
  __builtin_memcpy(&zf, &iz, 4);
  const double z = (double) zf;
  const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
               A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp+0;
  const double r = __builtin_fma(z, invc, -1.0);
  const double y0 = logc + (double) k;
  const double r2 = r * r;
  double yy = __builtin_fma(A0, r, A1);
  const double p = __builtin_fma(A2, r, A3);
  const double r4 = r2 * r2;
  double q = __builtin_fma(A4, r, y0);
  q = __builtin_fma(p, r2, q);
  yy = __builtin_fma(yy, r4, q);
  const double ylogx = (double) y * yy;
  if (((qa_asuint64(ylogx) >> 47) & 0xffff) >= (qa_asuint64(126.0) >> 47)) {
    if (ylogx > 0x1.fffffffd1d571p+6) return __builtin_inff();  // |y*log(x)| >= 126: overflow ...
    if (ylogx <= -150.0) return 0.0f;                            // ... or underflow
  }
  // 2^ylogx: N*x = k + r
  const double SHIFT = 0x1.8p+47;
  const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
  double kd = ylogx + SHIFT;
  const uint64_t ki = qa_asuint64(kd);
  kd -= SHIFT;
  const double rr = ylogx - kd;
  const double s = qa_asdouble(mathtab::kExp2[ki % 32] + (ki << 47));
  const double zz = __builtin_fma(C0, rr, C1);
  const double rr2 = rr * rr;
  double e = __builtin_fma(C2, rr, 1.0);
  e = __builtin_fma(zz, rr2, e);
  return (float) (e * s);
}
__host__ __device__ __forceinline__ float qasinf(float x) { return (float) asin((double) x); }
__host__ __device__ __forceinline__ float qtanf(float x) { return (float) tan((double) x); }

}  // namespace qa
