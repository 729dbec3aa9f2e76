// qa_kernel.h — the integrator: one persistent HIP kernel that runs qaray's whole per-pixel
// Monte-Carlo loop on gfx950.
//
// What it replaces (reference file:line): Renderer::ThreadRender/PixelRender
// (src/renderers/renderer.cpp:302-423), SuperSamplerHalton (src/scene/scene.cpp:83-123),
// Scene::TraceNodeNormal/TraceNodeShadow (src/scene/scene.cpp:35-74), Sphere/Plane/TriObj
// intersectors and the BVH walk (src/objects/objects.cpp:55-420), MtlBlinn_PhotonMap::Shade with
// MultiMtl dispatch (src/materials/MtlBlinn_PhotonMap.cpp:65-500, materials.h:70-76), the lights
// (src/lights/lights.cpp:39-144) and the xorshift32 sampler (src/samplers/Sampler_Marsaglia.cpp).
//
// Execution model (MI355X-first, not a translation of the CPU recursion):
//  * Persistent threads.  The grid is sized to what is resident on the chip; every LANE owns one
//    pixel at a time and runs a small state machine: [fetch pixel] -> [start sample: camera ray]
//    -> trace -> shade -> (next bounce | sample finished).  A finished path is replaced in the same
//    loop iteration by the pixel's next sample, a finished pixel by a new pixel pulled from a
//    global work counter with ONE atomic per wavefront (ballot + mbcnt prefix), so all 64 lanes of
//    a wave trace a ray in every iteration and no ray/hit queue ever goes through HBM: path state
//    lives in VGPRs, the BVH stack and (for scenes that fit) the whole scene image in LDS.
//  * The reference's recursion (Shade -> TraceNodeNormal -> Shade ...) is a chain, because exactly
//    one of reflect / transmit / diffuse is followed per hit; it is unrolled into an iteration
//    that carries a throughput.  Random numbers are drawn in the reference's order (select, lobe
//    sample, then the deeper hits); lights that draw random numbers (area lights) are evaluated by a
//    post-order replay of the path's hit log when the path ends (AREA kernel variants).
//  * BVH traversal is "while-while": a lane descends inner nodes until it holds a leaf, then the
//    wave intersects leaves together; the per-lane visiting order is exactly the reference's
//    (near child first, far child pushed), which decides ties between equal hit distances.
//  * One xorshift32 stream per pixel (include/qa_seed.h); a pixel's samples are sequential by
//    construction, pixels are the parallel dimension.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "qa_device_math.h"
#include "qa_scene_dev.h"
#include "qa_seed.h"
#include "qa_texture_dev.h"

namespace qa {

#define QA_BLOCK 256
#ifndef QA_MIN_WAVES
#define QA_MIN_WAVES 4          /* waves per SIMD the register allocator must leave room for */
#endif
#define QA_BIAS 0.005f         /* src/objects/objects.cpp:19 */
#define QA_DX 0.01f            /* DiffRay::dx = dy, src/core/ray.cpp:31-32 */
#define QA_DONE 0xFFFFFFFFu    /* traversal sentinel (has the leaf bit set, never a real node word) */
// qa_integrate's DCounters are per LANE, reduced by shuffles at the end.  (-DQA_WAVE_TALLIES: the lanes that reach a tally add
// their number to a wave-uniform count in scalar registers instead - eight vector registers less on paper; measured on the
// Cornell-box kernel: 101 instead of 87 spilled registers and 12.4 instead of 13.1 Gsamples/s, profiles/round03/experiments.txt.
// qa_integrate_cs does tally per wave.)
#ifdef QA_WAVE_TALLIES
#define QA_TALLY(x) (x) += (unsigned long long) __popcll(__ballot(1))
#else
#define QA_LANE_TALLIES 1
#define QA_TALLY(x) (x)++
#endif

#ifdef QA_STAMPS
#define QA_T(var) const unsigned long long var = __builtin_readcyclecounter();
#define QA_TACC(dst, since)                                                                                  \
  {                                                                                                          \
    const unsigned long long d_ = __builtin_readcyclecounter() - (since);                                    \
    if ((int) __lane_id() == __ffsll((long long) __ballot(1)) - 1) dst += d_;                                \
  }
#else
#define QA_T(var)
#define QA_TACC(dst, since)
#endif

struct Ray { f3 p, d; };
struct RayDiff { f3 dx, dy; };  // directions of the x / y differential rays (they share the origin)

struct Hit {
  float z;      // world-parametric distance (rays are not renormalised in node space)
  f3 p, N;      // node-local until traceClosest() maps them to world space
  int node;     // instance index, -1 = none
  int mtlID;
  bool front;
};

// Where the traversal reads the scene from: the LDS-resident image or global memory.
template <bool RES>
struct SceneMem {
  const uint4 *img;  // LDS image (RES) - unused otherwise
};

__device__ __forceinline__ float asF(uint32_t u) { return __uint_as_float(u); }

}  // namespace qa
#include "qa_photon_dev.h"   // needs QA_BLOCK and asF
namespace qa {

// Scene-graph nodes and mesh descriptors: resident scenes carry them BY VALUE in the kernel
// arguments (constant address space: wave-uniform indices become scalar loads into SGPRs, no
// vector-memory round trip on the critical path of every cast); larger scenes read the tables
// from global memory.
//
// The tables in global memory are read through the CONSTANT address space: no kernel writes them, and only then may
// the compiler turn a wave-uniform index into scalar loads (s_load into SGPRs: lower latency, no VGPRs for the
// matrices) - through a plain pointer out of DScene it cannot rule out that the frame's stores alias the table and
// issues one vector (flat) load per lane and field instead.  The record is returned by value; fields that are not
// used are never loaded.
template <class T>
__device__ __forceinline__ T ldTable(const T *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
  // dword by dword: a whole-struct copy becomes one wide load, which the compiler keeps in the vector path when its
  // users are vector ALU operations
  static_assert(sizeof(T) % 4 == 0 && alignof(T) >= 4, "table records are made of dwords");
  const __attribute__((address_space(4))) uint32_t *w = (const __attribute__((address_space(4))) uint32_t *) p;
  T v;
  uint32_t *d = reinterpret_cast<uint32_t *>(&v);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) d[i] = w[i];
  return v;
#else
  return *p;
#endif
}
// A pointer that comes out of such a record (DMesh::wnodes, wtris, ...) has no address space the compiler can see, and
// a load through it is a FLAT load: LDS or global decided per access at run time, and counted on both wait counters,
// so that it also waits for the LDS traffic of the traversal stack.  Where the pointer can only be global memory the
// load says so.
__device__ __forceinline__ uint4 ldGlobal(const uint4 *p);
// GMEM: the pointer is known to be global memory (non-resident scenes); otherwise a plain load (LDS image or unknown)
template <bool GMEM>
__device__ __forceinline__ uint4 ld16(const uint4 *p)
{
  if constexpr (GMEM) return ldGlobal(p);
  else return *p;
}
__device__ __forceinline__ uint4 ldGlobal(const uint4 *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
  // (through the native vector type: uint4 is a class, and its copy would go back through a generic reference)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = *(const __attribute__((address_space(1))) u32x4 *) p;
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
template <bool RES>
__device__ __forceinline__ std::conditional_t<RES, const qa_instance &, qa_instance> instAt(const DScene &sc, int k)
{
  if constexpr (RES) return sc.instv[k];
  else return ldTable(sc.inst + k);
}
template <bool RES>
__device__ __forceinline__ std::conditional_t<RES, const DMesh &, DMesh> meshAt(const DScene &sc, int k)
{
  if constexpr (RES) return sc.meshv[k];
  else return ldTable(sc.mesh + k);
}

// ---------------------------------------------------------------------------------------------
// RNG: Sampler_Marsaglia::xorshift32 (src/samplers/Sampler_Marsaglia.cpp:43-53)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float rng1(uint32_t &state)
{
  uint32_t x = state;
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  state = x;
  return (float) x / 4294967296.0f;
}

// Sampler::UniformBall (src/core/sampler.cpp:42-53); z uses r2 like the reference
__device__ __forceinline__ f3 uniformBall(uint32_t &rng, float radius)
{
  f3 p;
  do {
    const float r1 = rng1(rng), r2 = rng1(rng);
    (void) rng1(rng);
    p.x = (2.f * r1 - 1.f) * radius;
    p.y = (2.f * r2 - 1.f) * radius;
    p.z = (2.f * r2 - 1.f) * radius;
  } while (length(p) > radius);
  return p;
}

// TransformToLocalFrame (src/math/math.cpp:37-46)
__device__ __forceinline__ f3 toLocalFrame(f3 N, f3 sample)
{
  const f3 Z = N;
  const f3 Y = (qabs(Z.x) > qabs(Z.y)) ? normalize(F3(Z.z, 0, -Z.x)) : normalize(F3(0, -Z.z, Z.y));
  const f3 X = normalize(cross(Y, Z));
  const f3 unit = normalize(sample);
  return (X * unit.x + Y * unit.y) + Z * unit.z;
}

// ---------------------------------------------------------------------------------------------
// Node transforms (src/core/node.cpp:112-139, src/core/transform.h:47-61)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ Ray toNode(const qa_instance &in, const Ray &r)
{
  const f3 pos = ld3(in.pos);
  Ray o;
  o.p = mulMV(in.itm, r.p - pos);
  o.d = mulMV(in.itm, (r.p + r.d) - pos) - o.p;
  return o;
}

// Node::ToNodeCoords(DiffRay) (src/core/node.cpp:119-126) for the x / y rays: same origin as the
// central ray, so only the direction needs its own transform.  `before` is the central ray before
// this level, `afterP` its origin after it.
__device__ __forceinline__ f3 toNodeDir(const qa_instance &in, f3 beforeP, f3 afterP, f3 dir)
{
  return mulMV(in.itm, (beforeP + dir) - ld3(in.pos)) - afterP;
}
template <bool RES>
__device__ __forceinline__ void localRayDiff(const DScene &sc, int k, const Ray &world, const RayDiff &wd, Ray &r, RayDiff &rd)
{
  // root level
  Ray cur;
  RayDiff cd;
  if (sc.rootIdentity) {
    cur.p = world.p;
    cur.d = (world.p + world.d) - world.p;
    cd.dx = (world.p + wd.dx) - world.p;
    cd.dy = (world.p + wd.dy) - world.p;
  } else {
    cur = toNode(instAt<RES>(sc, 0), world);
    cd.dx = toNodeDir(instAt<RES>(sc, 0), world.p, cur.p, wd.dx);
    cd.dy = toNodeDir(instAt<RES>(sc, 0), world.p, cur.p, wd.dy);
  }
  int chain[QA_MAX_NODE_DEPTH];
  int n = 0;
  for (int a = k; a > 0 && n < QA_MAX_NODE_DEPTH; a = instAt<RES>(sc, a).parent) chain[n++] = a;
  for (int q = n - 1; q >= 0; --q) {
    const qa_instance &in = instAt<RES>(sc, chain[q]);
    const Ray nx = toNode(in, cur);
    cd.dx = toNodeDir(in, cur.p, nx.p, cd.dx);
    cd.dy = toNodeDir(in, cur.p, nx.p, cd.dy);
    cur = nx;
  }
  r = cur;
  rd = cd;
}

// The root node of a qaray scene never carries a transform (Node::Init, src/core/node.cpp:41-48):
// tm = itm = I, pos = 0.  Multiplying by the identity and subtracting zero return their operand
// (up to the sign of a zero), so Node::ToNodeCoords at the root reduces to dir' = (p + dir) - p,
// whose two roundings are what the reference performs and must be kept.
template <bool RES>
__device__ __forceinline__ Ray rootRay(const DScene &sc, const Ray &world)
{
  if (!sc.rootIdentity) return toNode(instAt<RES>(sc, 0), world);
  Ray o;
  o.p = world.p;
  o.d = (world.p + world.d) - world.p;
  return o;
}

// Local ray of instance k: the root's transform has already been applied (r0); walk the rest of
// the ancestor chain top-down.  All lanes work on the same k, so the chain is wave-uniform.
template <bool RES>
__device__ __forceinline__ Ray localRay(const DScene &sc, int k, const Ray &r0)
{
  const int depth = instAt<RES>(sc, k).depth;
  if (depth == 1) return toNode(instAt<RES>(sc, k), r0);
  int chain[QA_MAX_NODE_DEPTH];
  int n = 0;
  for (int a = k; a > 0 && n < QA_MAX_NODE_DEPTH; a = instAt<RES>(sc, a).parent) chain[n++] = a;
  Ray r = r0;
  for (int q = n - 1; q >= 0; --q) r = toNode(instAt<RES>(sc, chain[q]), r);
  return r;
}

// The same for the instance loops of global-memory scenes, which visit the nodes in pre-order: the children of a group
// follow each other, so the group's own local ray is kept from one sibling to the next instead of being rebuilt from
// the root for every child (five walls in one group: five times).  The chain is evaluated top-down either way - the
// same operations on the same operands.  (Resident kernels keep localRay: their scenes are flat, and the six registers
// of the kept ray are what the Cornell-box kernel does not have.)
struct GroupRay { int node; Ray ray; };
template <bool RES>
__device__ __forceinline__ Ray localRayInGroup(const DScene &sc, int k, const Ray &r0, GroupRay &g)
{
  if constexpr (RES) return localRay<RES>(sc, k, r0);
  else {
    const qa_instance &in = instAt<RES>(sc, k);
    if (in.depth == 1) return toNode(in, r0);
    if (in.parent != g.node) {
      g.ray = localRay<RES>(sc, in.parent, r0);
      g.node = in.parent;
    }
    return toNode(in, g.ray);
  }
}

// ---------------------------------------------------------------------------------------------
// Intersectors.  `closest` = false is a shadow query (diffray == NULL in the reference): only
// h.z is maintained and the caller stops at the first hit.
// ---------------------------------------------------------------------------------------------
// Sphere::IntersectRay (src/objects/objects.cpp:55-141)
__device__ __forceinline__ bool hitSphere(const Ray &ray, Hit &h, int k, bool closest)
{
  const float a = dot(ray.d, ray.d);
  const float b = 2.f * dot(ray.p, ray.d);
  const float c = dot(ray.p, ray.p) - 1;
  const float rcp2a = 1.f / (2.f * a);
  const float delta = b * b - 4 * a * c;
  float t = QA_BIGFLOAT;
  if (delta < 0) return false;
  if (delta == 0) {
    const float t0 = -b * rcp2a;
    if (t0 <= QA_BIAS) return false;
    t = t0;
  } else {
    const float sq = qsqrt(delta);
    const float t1 = (-b - sq) * rcp2a;
    const float t2 = (-b + sq) * rcp2a;
    if (t1 <= QA_BIAS && t2 <= QA_BIAS) return false;
    else if (t1 > QA_BIAS) t = qmin(t, t1);
    else if (t2 > QA_BIAS) t = qmin(t, t2);
  }
  if (h.z > t) {
    h.z = t;
    if (closest) {
      const f3 p = ray.p + ray.d * t;
      const f3 N = normalize(p);
      h.p = p;
      h.N = N;
      h.front = (dot(N, ray.d) <= 0);
      h.node = k;
    }
    return true;
  }
  return false;
}

// Plane::IntersectRay (src/objects/objects.cpp:149-208)
__device__ __forceinline__ bool hitPlane(const Ray &ray, Hit &h, int k, bool closest)
{
  const f3 N = F3(0, 0, 1);
  const float dz = dot(ray.d, N);
  if (qabs(dz) < 1e-7f) return false;
  const float pz = dot(ray.p, N);
  const float t = -pz / dz;
  if (t <= QA_BIAS) return false;
  if (h.z > t) {
    const f3 p = ray.p + ray.d * t;
    if (qabs(p.x) > 1.f || qabs(p.y) > 1.f) return false;
    h.z = t;
    if (closest) {
      h.p = p;
      h.N = N;
      h.front = (dot(N, ray.d) <= 0);
      h.node = k;
    }
    return true;
  }
  return false;
}

// One axis of the slab test (src/objects/objects.cpp:360-395, src/core/box.cpp:103-123)
__device__ __forceinline__ void slab(float d, float p0, float p1, float &t0, float &t1)
{
  if (qabs(d) < 1e-7f) { t0 = -QA_BIGFLOAT; t1 = QA_BIGFLOAT; }
  else { t0 = qmin(p0, p1); t1 = qmax(p0, p1); }
}
__device__ __forceinline__ void boxEntryExit(const Ray &ray, f3 drcp, f3 bmin, f3 bmax, float &entry, float &exit_)
{
  const f3 p0 = (-(ray.p - bmin)) * drcp;
  const f3 p1 = (-(ray.p - bmax)) * drcp;
  f3 t0, t1;
  slab(ray.d.x, p0.x, p1.x, t0.x, t1.x);
  slab(ray.d.y, p0.y, p1.y, t0.y, t1.y);
  slab(ray.d.z, p0.z, p1.z, t0.z, t1.z);
  entry = qmax(t0.x, qmax(t0.y, t0.z));
  exit_ = qmin(t1.x, qmin(t1.y, t1.z));
}

// Fast slab test for rays whose direction has no near-zero component (|d| >= 1e-7 on every axis,
// the reference's threshold): then every product below is finite and not NaN, and the reference's
// MIN/MAX macros return the same VALUES as the hardware min/max instructions (they can only differ
// in the sign of a zero, which the comparisons that consume entry/exit cannot see).
// -(p - b) and (b - p) are the same number up to the sign of zero for the same reason.
__device__ __forceinline__ void boxEntryExitFast(const Ray &ray, f3 drcp, f3 bmin, f3 bmax, float &entry, float &exit_)
{
  const f3 p0 = (bmin - ray.p) * drcp;
  const f3 p1 = (bmax - ray.p) * drcp;
  entry = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(p0.x, p1.x), __builtin_fminf(p0.y, p1.y)), __builtin_fminf(p0.z, p1.z));
  exit_ = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(p0.x, p1.x), __builtin_fmaxf(p0.y, p1.y)), __builtin_fmaxf(p0.z, p1.z));
}

// Slab tests of the library's own tree: the box is widened by `pad` on every side (folded into two
// copies of the ray origin, so the widening costs nothing per box).  Any conservative form will do
// here - these tests only decide where the own tree is searched, never what the reference accepts.
__device__ __forceinline__ void boxEntryExitPadFast(f3 pLo, f3 pHi, f3 drcp, f3 bmin, f3 bmax, float &entry, float &exit_)
{
  const f3 p0 = (bmin - pLo) * drcp;
  const f3 p1 = (bmax - pHi) * drcp;
  entry = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(p0.x, p1.x), __builtin_fminf(p0.y, p1.y)), __builtin_fminf(p0.z, p1.z));
  exit_ = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(p0.x, p1.x), __builtin_fmaxf(p0.y, p1.y)), __builtin_fmaxf(p0.z, p1.z));
}
__device__ __forceinline__ void boxEntryExitPad(f3 pLo, f3 pHi, f3 d, f3 drcp, f3 bmin, f3 bmax, float &entry, float &exit_)
{
  const f3 p0 = (bmin - pLo) * drcp;
  const f3 p1 = (bmax - pHi) * drcp;
  f3 t0, t1;
  slab(d.x, p0.x, p1.x, t0.x, t1.x);   // a near-zero direction component leaves the axis unbounded
  slab(d.y, p0.y, p1.y, t0.y, t1.y);
  slab(d.z, p0.z, p1.z, t0.z, t1.z);
  entry = qmax(t0.x, qmax(t0.y, t0.z));
  exit_ = qmin(t1.x, qmin(t1.y, t1.z));
}

// TriObj::IntersectTriangle (src/objects/objects.cpp:212-306) on a precomputed 48-byte record
// (three 16-byte words q0..q2, see DTri).  TriangleArea(axis, P, Q, R) =
// (Q.u-P.u)*(R.v-P.v) - (R.u-P.u)*(Q.v-P.v) with (u,v) the two coordinates kept after dropping
// `axis` (objects.cpp:30-41).  Written without early exits: in a 64-wide wave some lane almost
// always survives each of the reference's five rejections, so the branches only cost; every
// quantity is a pure function of the inputs, and the accept condition is the conjunction of the
// reference's tests in its own order (a rejected test's later values are simply ignored).
__device__ __forceinline__ bool hitTriangle(const uint4 q0, const uint4 q1, const uint4 q2, const Ray &ray, Hit &h,
                                            float &ba, float &bb)
{
  const f3 N = F3(asF(q0.x), asF(q0.y), asF(q0.z));
  const f3 A = F3(asF(q0.w), asF(q1.x), asF(q1.y));
  const float dz = dot(ray.d, N);
  const float pz = dot(ray.p - A, N);
  const float t = -pz / dz;
  bool ok = !(qabs(dz) < 1e-7f) && !(t <= QA_BIAS) && (h.z > t);
  const uint32_t axis = q2.w;
  const f3 p = ray.p + ray.d * t;
  const bool ax0 = (axis == 0), ax2 = (axis == 2);
  const float pu = ax0 ? p.y : p.x;
  const float pv = ax2 ? p.y : p.z;
  const float au = ax0 ? A.y : A.x;
  const float av = ax2 ? A.y : A.z;
  const float bu = asF(q1.z), bv = asF(q1.w), cu = asF(q2.x), cv = asF(q2.y), s = asF(q2.z);
  const float a = ((bu - pu) * (cv - pv) - (cu - pu) * (bv - pv)) * s;
  const float b = ((cu - pu) * (av - pv) - (au - pu) * (cv - pv)) * s;
  const float c = 1.f - a - b;
  ok = ok && !(a < 0 || b < 0 || c < 0);
  if (ok) {
    h.z = t;
    h.p = p;
    h.front = (dz <= 0);
    ba = a;
    bb = b;
  }
  return ok;
}

// The same test reduced to what the traversal itself needs: accept / reject and the distance.
// The hit point, facing and barycentrics of the finally accepted triangle are recomputed once per
// mesh by triangleDetails() with the very same expressions (t is the stored h.z), instead of being
// carried through seven conditional moves on every one of the ~20 tests of a cast.
__device__ __forceinline__ bool hitTriangleZ(const uint4 q0, const uint4 q1, const uint4 q2, const Ray &ray, float &hz)
{
  const f3 N = F3(asF(q0.x), asF(q0.y), asF(q0.z));
  const f3 A = F3(asF(q0.w), asF(q1.x), asF(q1.y));
  const float dz = dot(ray.d, N);
  const float pz = dot(ray.p - A, N);
  const float t = -pz / dz;
  bool ok = !(qabs(dz) < 1e-7f) && !(t <= QA_BIAS) && (hz > t);
  const uint32_t axis = q2.w;
  const f3 p = ray.p + ray.d * t;
  const bool ax0 = (axis == 0), ax2 = (axis == 2);
  const float pu = ax0 ? p.y : p.x;
  const float pv = ax2 ? p.y : p.z;
  const float au = ax0 ? A.y : A.x;
  const float av = ax2 ? A.y : A.z;
  const float bu = asF(q1.z), bv = asF(q1.w), cu = asF(q2.x), cv = asF(q2.y), s = asF(q2.z);
  const float a = ((bu - pu) * (cv - pv) - (cu - pu) * (bv - pv)) * s;
  const float b = ((cu - pu) * (av - pv) - (au - pu) * (cv - pv)) * s;
  const float c = 1.f - a - b;
  ok = ok && !(a < 0 || b < 0 || c < 0);
  if (ok) hz = t;
  return ok;
}
// The same test on the library's own tree: the accept rule is unchanged, and `tie` is raised when a
// triangle passes everything but arrives at exactly the distance already held - the one situation
// in which the order of the tests decides who wins (hitMesh then repeats the query in the
// reference's order).
// PACKED: the record comes from DMesh::wtris (element id above the 2-bit axis).
template <bool PACKED = false>
__device__ __forceinline__ bool hitTriangleZTie(const uint4 q0, const uint4 q1, const uint4 q2, const Ray &ray, float &hz, bool &tie)
{
  const f3 N = F3(asF(q0.x), asF(q0.y), asF(q0.z));
  const f3 A = F3(asF(q0.w), asF(q1.x), asF(q1.y));
  const float dz = dot(ray.d, N);
  const float pz = dot(ray.p - A, N);
  const float t = -pz / dz;
  const bool pre = !(qabs(dz) < 1e-7f) && !(t <= QA_BIAS);
  const uint32_t axis = PACKED ? (q2.w & 3u) : q2.w;
  const f3 p = ray.p + ray.d * t;
  const bool ax0 = (axis == 0), ax2 = (axis == 2);
  const float pu = ax0 ? p.y : p.x;
  const float pv = ax2 ? p.y : p.z;
  const float au = ax0 ? A.y : A.x;
  const float av = ax2 ? A.y : A.z;
  const float bu = asF(q1.z), bv = asF(q1.w), cu = asF(q2.x), cv = asF(q2.y), s = asF(q2.z);
  const float a = ((bu - pu) * (cv - pv) - (cu - pu) * (bv - pv)) * s;
  const float b = ((cu - pu) * (av - pv) - (au - pu) * (cv - pv)) * s;
  const float c = 1.f - a - b;
  const bool inside = pre && !(a < 0 || b < 0 || c < 0);
  const bool ok = inside && (hz > t);
  tie = tie || (inside && hz == t);
  if (ok) hz = t;
  return ok;
}
__device__ __forceinline__ void triangleDetails(const uint4 q0, const uint4 q1, const uint4 q2, const Ray &ray, Hit &h,
                                                float &ba, float &bb)
{
  const f3 N = F3(asF(q0.x), asF(q0.y), asF(q0.z));
  const f3 A = F3(asF(q0.w), asF(q1.x), asF(q1.y));
  const float dz = dot(ray.d, N);
  const float t = h.z;
  const uint32_t axis = q2.w;
  const f3 p = ray.p + ray.d * t;
  const bool ax0 = (axis == 0), ax2 = (axis == 2);
  const float pu = ax0 ? p.y : p.x;
  const float pv = ax2 ? p.y : p.z;
  const float au = ax0 ? A.y : A.x;
  const float av = ax2 ? A.y : A.z;
  const float bu = asF(q1.z), bv = asF(q1.w), cu = asF(q2.x), cv = asF(q2.y), s = asF(q2.z);
  ba = ((bu - pu) * (cv - pv) - (cu - pu) * (bv - pv)) * s;
  bb = ((cu - pu) * (av - pv) - (au - pu) * (cv - pv)) * s;
  h.p = p;
  h.front = (dz <= 0);
}

// TriObj::IntersectRay + TraceBVHNode (src/objects/objects.cpp:310-420).  The traversal stack
// holds node DATA words (leaf flag + range, or child index) instead of ids: the word arrives with
// the node's box when the parent tests its two children, so an inner visit is a single 64-byte
// read of the sibling pair.
struct TriPick { uint32_t tri; float a, b; };  // accepted triangle (element order) and its barycentrics

// One BVH walk.  FAST = false: the reference's tree and rules - near child first, far child stacked,
// strict tests (entry < t_max, entry < exit), a leaf's triangles in element order, first accepted
// wins at equal distance (objects.cpp:342-419).  FAST = true: the library's own tree (qa_fastbvh.h)
// with non-strict box tests, and `tie` is raised when a triangle passes the inside test at exactly
// the distance already held.  Returns whether a triangle was accepted; best = its element index in
// the walked tree's order.  closest = false stops at the first accepted triangle.
// STRIDE: distance between a lane's stack entries (QA_BLOCK: the per-lane columns of the LDS stacks; 1: a private array).
template <bool FAST, bool STATS, bool GMEM = false, int STRIDE = QA_BLOCK>
__device__ __forceinline__ bool walkBVH(const uint4 *nodes, const uint4 *tris, uint32_t rootData, const Ray &ray, f3 drcp,
                                        bool fastSlab, float &hz, bool closest, uint32_t *stack, DCounters &cnt,
                                        uint32_t &best, bool &tie, float pad = 0.f)
{
  const f3 pLo = ray.p + F3(pad, pad, pad), pHi = ray.p - F3(pad, pad, pad);   // FAST only
  bool hasHit = false;
  int sp = 0;
  uint32_t cur = rootData;
  while (cur != QA_DONE) {
    // ---- descend inner nodes until this lane holds a leaf (or runs out of work) --------------
    while (!(cur & QA_BVH_LEAF_BIT)) {
      if (STATS) QA_TALLY(cnt.bvh_nodes);
      const uint4 *pair = nodes + 2 * (size_t) (cur & QA_BVH_CHILD_MASK);
      const uint4 a0 = ld16<GMEM>(pair), a1 = ld16<GMEM>(pair + 1), b0 = ld16<GMEM>(pair + 2), b1 = ld16<GMEM>(pair + 3);
      float entry0, exit0, entry1, exit1;
      const f3 min0 = F3(asF(a0.x), asF(a0.y), asF(a0.z)), max0 = F3(asF(a0.w), asF(a1.x), asF(a1.y));
      const f3 min1 = F3(asF(b0.x), asF(b0.y), asF(b0.z)), max1 = F3(asF(b0.w), asF(b1.x), asF(b1.y));
      if constexpr (FAST) {
        if (fastSlab) {
          boxEntryExitPadFast(pLo, pHi, drcp, min0, max0, entry0, exit0);
          boxEntryExitPadFast(pLo, pHi, drcp, min1, max1, entry1, exit1);
        } else {
          boxEntryExitPad(pLo, pHi, ray.d, drcp, min0, max0, entry0, exit0);
          boxEntryExitPad(pLo, pHi, ray.d, drcp, min1, max1, entry1, exit1);
        }
      } else if (fastSlab) {
        boxEntryExitFast(ray, drcp, min0, max0, entry0, exit0);
        boxEntryExitFast(ray, drcp, min1, max1, entry1, exit1);
      } else {
        boxEntryExit(ray, drcp, min0, max0, entry0, exit0);
        boxEntryExit(ray, drcp, min1, max1, entry1, exit1);
      }
      const float t_max = hz;
      const bool hit0 = FAST ? (entry0 <= t_max && entry0 <= exit0) : (entry0 < t_max && entry0 < exit0);
      const bool hit1 = FAST ? (entry1 <= t_max && entry1 <= exit1) : (entry1 < t_max && entry1 < exit1);
      const uint32_t d0 = a1.z, d1 = b1.z;
      if (hit0 && hit1) {
        // the reference pushes the far child, then the near one, and pops the near one next
        const bool nearFirst = entry0 < entry1;
        stack[(sp++) * STRIDE] = nearFirst ? d1 : d0;
        cur = nearFirst ? d0 : d1;
      } else if (hit0) cur = d0;
      else if (hit1) cur = d1;
      else cur = sp ? stack[(--sp) * STRIDE] : QA_DONE;
    }
    if (cur == QA_DONE) break;
    // ---- leaf: its triangles in element order -----------------------------------------------
    if (STATS) QA_TALLY(cnt.bvh_nodes);
    const uint32_t count = ((cur >> QA_BVH_COUNT_SHIFT) & QA_BVH_COUNT_MASK) + 1;
    const uint32_t first = cur & QA_BVH_OFFSET_MASK;
    for (uint32_t i = 0; i < count; ++i) {
      if (STATS) QA_TALLY(cnt.tri_tests);
      const uint4 *t = tris + 3 * (size_t) (first + i);
      const uint4 t2 = ld16<GMEM>(t + 2);
      bool accepted;
      if constexpr (FAST) accepted = hitTriangleZTie<true>(ld16<GMEM>(t), ld16<GMEM>(t + 1), t2, ray, hz, tie);
      else accepted = hitTriangleZ(ld16<GMEM>(t), ld16<GMEM>(t + 1), t2, ray, hz);
      if (accepted) {
        hasHit = true;
        best = FAST ? (t2.w >> 2) : first + i;   // own tree: element | reference leaf << 15 (DMesh::ftris)
        if (!closest) return true;
      }
    }
    cur = sp ? stack[(--sp) * STRIDE] : QA_DONE;
  }
  return hasHit;
}

// The four children of a DWideNode (qa_scene_dev.h, its 4 x 16 bytes in q0..q3) against a ray whose origin has been
// split into pLo / pHi (origin -/+ pad: the boxes widened by `pad`): K = entry distance of child c, or +inf when the
// ray misses it, the slot is empty or the box lies beyond hz; W = its child word.  Boxes are decoded with one fma per
// plane - conservative by construction (quantised outwards, checked by the builder with the same fma).
#define QA_WIDE_BYTE(X, C) ((float) (((X) >> (8 * (C))) & 0xFFu))
#define QA_WIDE_CHILD(K, W, C)                                                                                                       \
  {                                                                                                                                  \
    const f3 lo = F3(__builtin_fmaf(QA_WIDE_BYTE(q1.z, C), asF(q0.w), asF(q0.x)), __builtin_fmaf(QA_WIDE_BYTE(q1.w, C), asF(q1.x), asF(q0.y)), \
                     __builtin_fmaf(QA_WIDE_BYTE(q2.x, C), asF(q1.y), asF(q0.z)));                                                   \
    const f3 hi = F3(__builtin_fmaf(QA_WIDE_BYTE(q2.y, C), asF(q0.w), asF(q0.x)), __builtin_fmaf(QA_WIDE_BYTE(q2.z, C), asF(q1.x), asF(q0.y)), \
                     __builtin_fmaf(QA_WIDE_BYTE(q2.w, C), asF(q1.y), asF(q0.z)));                                                   \
    const f3 p0 = (lo - pLo) * drcp, p1 = (hi - pHi) * drcp;                                                                         \
    const float en = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(p0.x, p1.x), __builtin_fminf(p0.y, p1.y)), __builtin_fminf(p0.z, p1.z)); \
    const float ex = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(p0.x, p1.x), __builtin_fmaxf(p0.y, p1.y)), __builtin_fmaxf(p0.z, p1.z)); \
    K = (W != QA_DONE && en <= hz && en <= ex) ? en : INF;                                                                           \
  }
#define QA_WIDE_CE(KA, WA, KB, WB)  \
  {                                 \
    const bool sw = KA > KB;        \
    const float tk = sw ? KB : KA;  \
    KB = sw ? KA : KB;              \
    KA = tk;                        \
    const uint32_t tw = sw ? WB : WA; \
    WB = sw ? WA : WB;              \
    WA = tw;                        \
  }
// all four children, sorted by entry distance (missed ones last, K = +inf)
#define QA_WIDE_NODE(q0, q1, q2, q3)                          \
  float k0, k1, k2, k3;                                       \
  uint32_t w0 = q3.x, w1 = q3.y, w2 = q3.z, w3 = q3.w;        \
  QA_WIDE_CHILD(k0, w0, 0)                                    \
  QA_WIDE_CHILD(k1, w1, 1)                                    \
  QA_WIDE_CHILD(k2, w2, 2)                                    \
  QA_WIDE_CHILD(k3, w3, 3)                                    \
  QA_WIDE_CE(k0, w0, k1, w1)                                  \
  QA_WIDE_CE(k2, w2, k3, w3)                                  \
  QA_WIDE_CE(k0, w0, k2, w2)                                  \
  QA_WIDE_CE(k1, w1, k3, w3)                                  \
  QA_WIDE_CE(k1, w1, k2, w2)

// One walk of the library's 4-wide tree (qa_widebvh.h).  Boxes are widened by `pad` (folded into two copies of the
// origin) and tested non-strictly, so every triangle the reference's inside test can accept at or before the distance
// held is tested; children are entered nearest first.  `tris` = DMesh::wtris (leaf order, element id in the axis
// word); `best` returns the element.  `tie` is raised when a triangle passes the inside test at exactly the
// distance already held.  closest = false stops at the first accepted triangle.  `stack`: LDS, stride QA_BLOCK,
// `cap` entries - on overflow `tie` is raised too (the caller then repeats the query on the reference tree).
__device__ __forceinline__ bool walkWide(const uint4 *wn, const uint4 *tris, uint32_t rootWord, const Ray &ray, f3 drcp, float pad,
                                         float &hz, bool closest, uint32_t *stack, uint32_t cap, uint32_t &best, bool &tie)
{
  const f3 pLo = ray.p + F3(pad, pad, pad), pHi = ray.p - F3(pad, pad, pad);
  const float INF = __builtin_inff();
  bool hasHit = false;
  uint32_t sp = 0;
  uint32_t cur = rootWord;
  while (cur != QA_DONE) {
    while (!(cur & QA_BVH_LEAF_BIT)) {
      const uint4 *nd = wn + 4 * (size_t) cur;
      const uint4 q0 = ldGlobal(nd), q1 = ldGlobal(nd + 1), q2 = ldGlobal(nd + 2), q3 = ldGlobal(nd + 3);
      QA_WIDE_NODE(q0, q1, q2, q3)
      // nearest child next, the others stacked farthest first
      if (k3 < INF) { if (sp < cap) stack[(sp++) * QA_BLOCK] = w3; else tie = true; }
      if (k2 < INF) { if (sp < cap) stack[(sp++) * QA_BLOCK] = w2; else tie = true; }
      if (k1 < INF) { if (sp < cap) stack[(sp++) * QA_BLOCK] = w1; else tie = true; }
      if (k0 < INF) cur = w0;
      else cur = sp ? stack[(--sp) * QA_BLOCK] : QA_DONE;
    }
    if (cur == QA_DONE) break;
    const uint32_t count = ((cur >> QA_BVH_COUNT_SHIFT) & QA_BVH_COUNT_MASK) + 1;
    const uint32_t first = cur & QA_BVH_OFFSET_MASK;
    for (uint32_t i = 0; i < count; ++i) {
      const uint4 *t = tris + 3 * (size_t) (first + i);
      const uint4 t2 = ldGlobal(t + 2);
      if (hitTriangleZTie<true>(ldGlobal(t), ldGlobal(t + 1), t2, ray, hz, tie)) {
        hasHit = true;
        best = t2.w >> 2;               // element (the reference's triangle order)
        if (!closest) return true;
      }
    }
    cur = sp ? stack[(--sp) * QA_BLOCK] : QA_DONE;
  }
  return hasHit;
}

// May this ray be searched with a pruned walk of mesh m?  The reference's inside test can accept, by cancellation of
// its 2-D areas, a point farther than DMesh::cancelDist (qa_widebvh.h ComputeMeshSlack) from a triangle.  A pruned
// search goes wrong only if such a "hit" lies before the triangle's leaf box on the ray, i.e. on the segment from the
// origin to the mesh bounds - which is never farther from any triangle than the origin's farthest corner of the
// bounds, or the bounds' diagonal.
template <class M>
__device__ __forceinline__ bool insideCancelReach(const M &m, f3 o)
{
  const f3 bmin = ld3(m.bmin), bmax = ld3(m.bmax);
  const f3 f = F3(qmax(qabs(o.x - bmin.x), qabs(o.x - bmax.x)), qmax(qabs(o.y - bmin.y), qabs(o.y - bmax.y)), qmax(qabs(o.z - bmin.z), qabs(o.z - bmax.z)));
  const f3 dg = bmax - bmin;
  const float reach2 = qmax(dot(f, f), dot(dg, dg));
  return reach2 * 1.0001f < m.cancelDist * m.cancelDist;
}

// Would the reference's walk have reached the leaf `leaf` of its tree?  It enters a node when the
// strict box test passes against the distance held at that moment.  Every inner box of the tree is
// the union of its children's boxes (min / max of the same floats), and the slab arithmetic is
// monotone in the box bounds, so an ancestor's [entry, exit] interval contains the leaf's: if the
// LEAF's box passes the strict test against `limit`, every node above it does.  `limit` is the final
// hit distance for a closest-hit query (the distance held earlier can only be larger: sufficient),
// the fixed t_max for an any-hit query (exact).  DTriShade::pad holds an element's leaf id.
template <bool GMEM = false>
__device__ __forceinline__ bool refReaches(const uint4 *nodes, uint32_t leaf, const Ray &ray, f3 drcp, bool fastSlab, float limit)
{
  if (leaf <= 1) return true;   // the root is entered unconditionally (the mesh bounds were tested by the caller)
  const uint4 n0 = ld16<GMEM>(nodes + 2 * (size_t) leaf), n1 = ld16<GMEM>(nodes + 2 * (size_t) leaf + 1);
  float entry, exit_;
  const f3 bmin = F3(asF(n0.x), asF(n0.y), asF(n0.z)), bmax = F3(asF(n0.w), asF(n1.x), asF(n1.y));
  if (fastSlab) boxEntryExitFast(ray, drcp, bmin, bmax, entry, exit_);
  else boxEntryExit(ray, drcp, bmin, bmax, entry, exit_);
  return entry < limit && entry < exit_;
}

// TriObj::IntersectRay + TraceBVHNode (src/objects/objects.cpp:310-420).
//
// The closest hit of a mesh does not depend on the tree it is searched with - except where the
// reference's own search is not exhaustive: it enters a child box only on strict inequalities (a
// flat box is never entered), and at equal distance it keeps the triangle it met first.  Counting
// kernels (STATS) therefore walk the reference's tree exactly as the reference does.  The others
// walk the library's SAH tree (qa_fastbvh.h: two triangles per leaf, padded boxes, non-strict
// tests, so that it reaches every triangle the reference can reach; 2.7 instead of 15 triangle
// tests per cast on the Cornell box) and then check the answer against the reference's rules:
// the found triangle must be reachable in the reference's tree (refReaches) and no tie may have
// been seen; otherwise the lane repeats the query on the reference's tree.
template <bool RES, bool STATS>
__device__ __forceinline__ bool hitMesh(const SceneMem<RES> mem, const DMesh &m, const Ray &ray, Hit &h, int k,
                                        bool closest, uint32_t *stack /* LDS, stride QA_BLOCK */, DCounters &cnt,
                                        TriPick &pick, uint32_t stackCap = 0xFFFFu)
{
  const f3 drcp = F3(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
  // wave-uniform choice: the exact MIN/MAX/threshold form only when some lane needs it
  const bool fastSlab = !__any(qabs(ray.d.x) < 1e-7f || qabs(ray.d.y) < 1e-7f || qabs(ray.d.z) < 1e-7f);
  float meshExit;
  {
    float entry;
    if (fastSlab) boxEntryExitFast(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
    else boxEntryExit(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
    if (entry > h.z || entry > meshExit) return false;  // Box::IntersectRay, src/core/box.cpp:94-128
  }
  if (m.num_faces == 0) return false;
  const uint4 *nodes = RES ? mem.img + m.resNodes : reinterpret_cast<const uint4 *>(m.nodes);
  const uint4 *tris = RES ? mem.img + m.resTris : reinterpret_cast<const uint4 *>(m.tris);
  const uint4 *shade = RES ? mem.img + m.resShade : reinterpret_cast<const uint4 *>(m.shade);
  bool hasHit = false;
  uint32_t bestTri = 0;
  bool tie = false;
  // Global-memory scenes keep the reference tree: there every inner step is a dependent memory round
  // trip, and the SAH tree's longer chains of small nodes cost more than its fewer triangle tests save
  // (measured: tower scene 182 -> 199 ms, glossy caustics 52 -> 61 ms, project7_object 99 -> 93 ms),
  // while carrying both walks in one kernel costs those scenes 5 - 10 % in registers alone.
  if constexpr (STATS) {
    hasHit = walkBVH<false, STATS>(nodes, tris, m.rootData, ray, drcp, fastSlab, h.z, closest, stack, cnt, bestTri, tie);
  } else if constexpr (!RES) {
    // Global-memory meshes: the library's 4-wide tree (qa_widebvh.h) - a third of the dependent node reads, a fifth
    // of the triangle tests - with the answer checked against the reference's rules: the found triangle's leaf must
    // pass the reference's strict box test at the found distance (refReaches: then the reference's walk, whose
    // running distance is at least that, reaches the triangle too), and no tie may have been seen.  A miss needs no
    // check: every triangle the reference can accept is tested (boxes are unions of triangle bounds, widened by the
    // fp32 slack of its inside test).  Otherwise, and for ray origins so
    // far out that the inside test's areas can cancel, the lane walks the reference tree as the reference does.
    const float oMax = qmax(qmax(qabs(ray.p.x), qabs(ray.p.y)), qabs(ray.p.z));
    bool redo = true;
    const float hz0 = h.z;
    if (m.useWide && insideCancelReach(m, ray.p)) {
      const float pad = m.nearPad + (QA_SLACK_SCALE * 1e-6f) * (oMax + m.absMax);
      hasHit = walkWide(reinterpret_cast<const uint4 *>(m.wnodes), reinterpret_cast<const uint4 *>(m.wtris), m.wrootWord, ray, drcp, pad, h.z, closest, stack, stackCap, bestTri, tie);
      redo = tie;
      if (hasHit && !redo) {
        const uint32_t leaf = ldGlobal(shade + 3 * (size_t) bestTri + 2).w;   // DTriShade::pad
        redo = !refReaches<true>(nodes, leaf, ray, drcp, fastSlab, closest ? h.z : hz0);
      }
    }
    if (redo) {
      h.z = hz0;
      tie = false;
      hasHit = walkBVH<false, false, true>(nodes, tris, m.rootData, ray, drcp, fastSlab, h.z, closest, stack, cnt, bestTri, tie);
    }
  } else if (!m.useFast) {
    hasHit = walkBVH<false, STATS>(nodes, tris, m.rootData, ray, drcp, fastSlab, h.z, closest, stack, cnt, bestTri, tie);
  } else {
    const uint4 *fnodes = RES ? mem.img + m.resFNodes : reinterpret_cast<const uint4 *>(m.fnodes);
    const uint4 *ftris = RES ? mem.img + m.resFTris : reinterpret_cast<const uint4 *>(m.ftris);
    const float hz0 = h.z;
    uint32_t bestF = 0;
    // How far from a triangle can a point be that the reference's inside test still accepts?  The test
    // evaluates 2-D signed areas of magnitude <= (2P)^2 (P: largest coordinate involved) in fp32 and
    // scales them by 1 / area: the barycentrics are off by at most ~64 eps P^2 / area, i.e. the point
    // may lie up to ~200 eps P^2 / h outside an edge (h: the smallest altitude of any triangle of the
    // mesh, DMesh::invH = 1 / h), plus the error of the point itself.  The own tree's boxes are widened
    // by that much for this ray, so it reaches every triangle the reference can accept up to the far
    // side of the mesh bounds.
    // (every point of the ray up to the far side of the bounds lies between the origin and the bounds)
    const float oMax = qmax(qmax(qabs(ray.p.x), qabs(ray.p.y)), qabs(ray.p.z));
    const float P = qmax(m.absMax, oMax);
    const float pad = ((QA_SLACK_SCALE * 1.2e-5f) * m.invH) * (P * P) + (QA_SLACK_SCALE * 1e-6f) * P;
    hasHit = walkBVH<true, false>(fnodes, ftris, m.frootData, ray, drcp, fastSlab, h.z, closest, stack, cnt, bestF, tie, pad);
    // Beyond the mesh bounds a triangle can only be "hit" by cancellation: at a distance D from the
    // triangle the inside test is off by ~64 eps D^2 / (L h) and would have to be off by D / L, i.e.
    // D >= h / (64 eps) ~ 2.6e5 h (all three areas then round to the same product: barycentrics 0, 0, 1).
    // The reference does accept such a hit when nothing nearer holds the ray.  The ray meets a triangle's
    // plane that far out only if it runs within (|o| + |mesh|) / D of parallel to it, so a miss on the
    // own tree is trusted unless the ray is that close to one of the mesh's (few) distinct face normals;
    // D is taken as 1e5 h.  With a hit in hand the question does not arise: D is far beyond the bounds.
    bool redo = tie;
    if (!hasHit && hz0 > meshExit) {
      const float theta = ((QA_SLACK_SCALE * 1.8e-5f) * m.invH) * (oMax + m.absMax) + QA_SLACK_SCALE * 2e-5f;   // + the tolerance the normal list was merged with
      const float lim = (theta * theta) * dot(ray.d, ray.d);
      bool parallel = m.numNormals == 0;   // no list (too many distinct normals): never trust a miss
      const float4 *nrm = reinterpret_cast<const float4 *>(mem.img + m.resNormals);
      for (uint32_t i = 0; i < m.numNormals; ++i) {
        const float4 n = nrm[i];
        const float dn = __builtin_fmaf(ray.d.x, n.x, __builtin_fmaf(ray.d.y, n.y, ray.d.z * n.z));   // not reference arithmetic: fused is fine
        parallel = parallel || (dn * dn <= lim);
      }
      redo = redo || parallel;
    }
    if (hasHit) {
      bestTri = bestF & 0x7FFFu;                  // element
      const uint32_t leaf = bestF >> 15;          // its leaf in the reference tree (the record's word, see DMesh::ftris)
      // closest: the distance just found; any-hit: the fixed t_max (h.z is untouched by walkBVH then... it is
      // set to the accepted distance, so take the saved one)
      redo = redo || !refReaches(nodes, leaf, ray, drcp, fastSlab, closest ? h.z : hz0);
    }
    if (redo) {
      h.z = hz0;
      hasHit = walkBVH<false, false>(nodes, tris, m.rootData, ray, drcp, fastSlab, h.z, closest, stack, cnt, bestTri, tie);
    }
  }
  if (!closest) return hasHit;
  if (hasHit) {
    float ba = 0, bb = 0;
    {
      const uint4 *t = tris + 3 * (size_t) bestTri;
      triangleDetails(ld16<!RES>(t), ld16<!RES>(t + 1), ld16<!RES>(t + 2), ray, h, ba, bb);
    }
    // shading normal: TriMesh::GetNormal (src/mesh/TriMesh.h:196-204), left un-normalised
    const uint4 *s = shade + 3 * (size_t) bestTri;
    const uint4 s0 = ld16<!RES>(s), s1 = ld16<!RES>(s + 1), s2 = ld16<!RES>(s + 2);
    const float bc = 1.f - ba - bb;
    const f3 n0 = F3(asF(s0.x), asF(s0.y), asF(s0.z)), n1 = F3(asF(s0.w), asF(s1.x), asF(s1.y)),
             n2 = F3(asF(s1.z), asF(s1.w), asF(s2.x));
    h.N = (n0 * ba + n1 * bb) + n2 * bc;
    h.mtlID = (int) s2.y;
    h.node = k;
    pick.tri = bestTri;
    pick.a = ba;
    pick.b = bb;
  }
  return hasHit;
}

// Scene::TraceNodeNormal (src/scene/scene.cpp:50-74): closest hit over every node in pre-order.
// TEX: also maintains HitInfo::uvw / duvw / hasTexture exactly as the intersectors do (fields are
// only overwritten by the object types that set them, so stale values survive like in the reference).
template <bool RES, bool TEX, bool STATS>
__device__ __forceinline__ bool traceClosest(const SceneMem<RES> mem, const DScene &sc, const Ray &world, const RayDiff &wd,
                                             Hit &h, TexHit &th, uint32_t *stack, DCounters &cnt)
{
  QA_TALLY(cnt.casts_normal);
  const Ray r0 = rootRay<RES>(sc, world);
  GroupRay grp;
  grp.node = -1;
  grp.ray = r0;
  bool any = false;
  for (int k = 1; k < sc.num_inst; ++k) {
    const int type = instAt<RES>(sc, k).obj_type;
    if (type == QA_OBJ_NONE) continue;
    Ray r;
    RayDiff rd;
    if (TEX) localRayDiff<RES>(sc, k, world, wd, r, rd);
    else r = localRayInGroup<RES>(sc, k, r0, grp);
    bool hit;
    if (type == QA_OBJ_SPHERE) {
      hit = hitSphere(r, h, k, true);
      if (TEX && hit) texSphere(r.p, rd.dx, rd.dy, h.p, h.N, th);
    } else if (type == QA_OBJ_PLANE) {
      hit = hitPlane(r, h, k, true);
      if (TEX && hit) texPlane(r.p, rd.dx, rd.dy, h.p, th);
    } else {
      const DMesh &m = meshAt<RES>(sc, instAt<RES>(sc, k).mesh);
      TriPick pick;
      QA_T(tm0)
      hit = hitMesh<RES, STATS>(mem, m, r, h, k, true, stack, cnt, pick, sc.stackDepth);
      QA_TACC(cnt.sl[3], tm0)
      if (TEX && hit && m.hasVT) {
        const uint4 *t = (RES ? mem.img + m.resTris : reinterpret_cast<const uint4 *>(m.tris)) + 3 * (size_t) pick.tri;
        const float *vt = m.vt + 6 * (size_t) pick.tri;
        texTriangle(t[0], t[1], t[2], vt, r.p, rd.dx, rd.dy, pick.a, pick.b, th);
      }
    }
    any |= hit;
  }
  if (any) {
    // Node::FromNodeCoords at every level from the hit node up to and including the root
    // (src/core/node.cpp:127-139); the reference applies them as its recursion unwinds.
    for (int a = h.node; a >= 0; a = instAt<RES>(sc, a).parent) {
      if (a == 0 && sc.rootIdentity) {
        h.N = normalize(h.N);  // identity root: p unchanged, the normal is still re-normalised
        break;
      }
      const qa_instance &in = instAt<RES>(sc, a);
      h.p = mulMV(in.tm, h.p) + ld3(in.pos);
      h.N = normalize(mulTMV(in.itm, h.N));
    }
  }
  return any;
}

// GenLight::Shadow -> Scene::TraceNodeShadow (src/lights/lights.cpp:39-48, src/scene/scene.cpp:35-46)
template <bool RES, bool STATS>
__device__ __forceinline__ float shadow(const SceneMem<RES> mem, const DScene &sc, const Ray &world, float t_max,
                                        uint32_t *stack, DCounters &cnt)
{
  QA_TALLY(cnt.casts_shadow);
  Hit h;
  h.z = t_max;
  h.node = -1;
  const Ray r0 = rootRay<RES>(sc, world);
  GroupRay grp;
  grp.node = -1;
  grp.ray = r0;
  for (int k = 1; k < sc.num_inst; ++k) {
    const int type = instAt<RES>(sc, k).obj_type;
    if (type == QA_OBJ_NONE) continue;
    const Ray r = localRayInGroup<RES>(sc, k, r0, grp);
    bool hit;
    if (type == QA_OBJ_SPHERE) hit = hitSphere(r, h, k, false);
    else if (type == QA_OBJ_PLANE) hit = hitPlane(r, h, k, false);
    else {
      TriPick pick;
      QA_T(tm0)
      hit = hitMesh<RES, STATS>(mem, meshAt<RES>(sc, instAt<RES>(sc, k).mesh), r, h, k, false, stack, cnt, pick, sc.stackDepth);
      QA_TACC(cnt.sl[6], tm0)
    }
    if (hit) return 0.0f;
  }
  return 1.0f;
}

// ---------------------------------------------------------------------------------------------
// Lights (src/lights/lights.h:35-171, src/lights/lights.cpp:23-144); area lights (size > 0.01)
// shoot 16 - 64 shadow rays per evaluation and run in the AREA kernel variants.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float inverseSquareFalloff(f3 v) { return qmin(1.f, 1.f / dot(v, v)); }

// SpotLight::GetAttenuation(Direction(p)) (src/lights/lights.cpp:128-143)
__device__ __forceinline__ float spotAttenuation(const qa_light &l, f3 p)
{
  const f3 d = normalize(p - ld3(l.position));
  const float cosTheta = dot(d, ld3(l.direction));
  if (cosTheta < 0) return 0;
  const float rr = qsqrt(1.f - cosTheta * cosTheta) / cosTheta;
  if (rr > l.outer) return 0;
  return rr < l.inner ? 1.f : qpowf((l.outer - rr) / (l.outer - l.inner), 2.f);
}

template <bool RES, bool STATS>
__device__ __forceinline__ f3 illuminate(const SceneMem<RES> mem, const DScene &sc, const qa_light &l, f3 p,
                                         uint32_t *stack, DCounters &cnt, uint32_t &rng)
{
  const f3 intensity = ld3(l.intensity);
  if (l.type != QA_LIGHT_DIRECT && l.size > 0.01f) {
    // area light: 16 shadow rays towards points of a ball around the light, 64 as soon as the
    // running estimate is a penumbra value (src/lights/lights.cpp:52-65,88-100; shadow_spp 16/64 :16-17)
    int spp = 16, n = 0;
    float inshadow = 0.0f;
    while (n < spp) {
      const f3 dir = (ld3(l.position) + uniformBall(rng, l.size)) - p;
      Ray r;
      r.p = p;
      r.d = normalize(dir);
      inshadow += (shadow<RES, STATS>(mem, sc, r, length(dir), stack, cnt) - inshadow) * inverseSquareFalloff(dir) / (float) (n + 1);
      n++;
      if (inshadow > 0.f && inshadow < 1.f) spp = 64;
    }
    f3 I = intensity * inshadow;
    if (l.type == QA_LIGHT_SPOT) I = I * spotAttenuation(l, p);
    return I;
  }
  if (l.type == QA_LIGHT_DIRECT) {
    Ray r;
    r.p = p;
    r.d = normalize(-ld3(l.direction));
    return intensity * shadow<RES, STATS>(mem, sc, r, QA_BIGFLOAT, stack, cnt);
  }
  // point / spot
  const f3 dir = ld3(l.position) - p;
  Ray r;
  r.p = p;
  r.d = normalize(dir);
  f3 I = (intensity * shadow<RES, STATS>(mem, sc, r, length(dir), stack, cnt)) * inverseSquareFalloff(dir);
  if (l.type == QA_LIGHT_SPOT) I = I * spotAttenuation(l, p);
  return I;
}

__device__ __forceinline__ f3 lightDirection(const qa_light &l, f3 p)
{
  if (l.type == QA_LIGHT_DIRECT) return ld3(l.direction);
  return normalize(p - ld3(l.position));
}

// Direct lighting of one shading point (MtlBlinn_PhotonMap.cpp:481-498): every non-ambient light,
// weight 1/#lights (ambient counted), Blinn lobe around the half vector.
template <bool RES, bool STATS>
__device__ __forceinline__ f3 directLight(const SceneMem<RES> mem, const DScene &sc, f3 p, f3 N, f3 V, f3 kd, f3 ks,
                                          float gloss, uint32_t *stack, DCounters &cnt, uint32_t &rng)
{
  f3 sum = F3(0, 0, 0);
  const float normCoefDI = 1.f / (float) sc.num_lights;
  for (int li = 0; li < sc.num_lights; ++li) {
    const qa_light l = ldTable(sc.light + li);
    if (l.type == QA_LIGHT_AMBIENT) continue;
    if (!STATS && !(l.type != QA_LIGHT_DIRECT && l.size > 0.01f)) {
      // Not an area light, nothing to count: the term first, as if unshadowed (a shadow factor of 1.0f multiplies exactly), the shadow
      // ray only when that term is not zero in every component - a surface facing away from the light (cosNL = max(0, N.L) = 0) adds
      // the same zero whether it is occluded or not; an occluded light's term is the unshadowed one times 0.0f (the reference's term
      // with the factor 0.0f: a zero of some sign - or a NaN exactly when the unshadowed term is not finite; a sum that started
      // from +0 does not see the sign of a zero).  qa_kernel_cs.h's csLightTerms / csLightSum, one lane at a time.
      f3 I;
      Ray r;
      r.p = p;
      float tmax = QA_BIGFLOAT;
      if (l.type == QA_LIGHT_DIRECT) {
        I = ld3(l.intensity) * 1.0f;
        r.d = normalize(-ld3(l.direction));
      } else {
        const f3 dir = ld3(l.position) - p;
        r.d = normalize(dir);
        tmax = length(dir);
        I = (ld3(l.intensity) * 1.0f) * inverseSquareFalloff(dir);
        if (l.type == QA_LIGHT_SPOT) I = I * spotAttenuation(l, p);
      }
      const f3 intensity = I * normCoefDI;
      const f3 Ld = normalize(-lightDirection(l, p));
      const f3 H = normalize(V + Ld);
      const float cosNL = qmax(0.f, dot(N, Ld));
      const float cosNH = qmax(0.f, dot(N, H));
      const f3 brdf = kd + ks * qpowf(cosNH, gloss);
      const f3 u = (intensity * cosNL) * brdf;
      const bool walk = sc.walkZeroTerms || !(u.x == 0.f && u.y == 0.f && u.z == 0.f);
      if (!walk) QA_TALLY(cnt.casts_shadow);   // (the reference casts it: counted, not walked)
      const bool occluded = walk && shadow<RES, STATS>(mem, sc, r, tmax, stack, cnt) == 0.0f;
      sum = sum + (occluded ? u * 0.0f : u);
      continue;
    }
    const f3 intensity = illuminate<RES, STATS>(mem, sc, l, p, stack, cnt, rng) * normCoefDI;
    const f3 Ld = normalize(-lightDirection(l, p));
    const f3 H = normalize(V + Ld);
    const float cosNL = qmax(0.f, dot(N, Ld));
    const float cosNH = qmax(0.f, dot(N, H));
    const f3 brdf = kd + ks * qpowf(cosNH, gloss);
    sum = sum + (intensity * cosNL) * brdf;
  }
  return sum;
}

// ---------------------------------------------------------------------------------------------
// MtlBlinn_PhotonMap::Shade up to (not including) the light loop
// (src/materials/MtlBlinn_PhotonMap.cpp:256-479): sampled colours, Fresnel, the lobe / Russian-
// roulette draw and the sampled secondary direction.  Shared by both kernels.
// ---------------------------------------------------------------------------------------------
struct Surface {
  f3 emission, kd, ks;   // sampled emission / diffuse / specular colours
  float gloss;
  bool spawn;            // a secondary ray follows
  f3 nextDir, bxdf;      // its (un-normalised) direction and the BxDF weight (PDF = 1)
  bool nextFromDiffuse;
  bool selDiffuse;       // RandomSelectMtl returned DIFFUSE (photon-map gathers hang off this)
};

// The textured kernels' form of shadeSurface: the same operations with every texture lookup of the hit FIRST (below).  A function of
// its own because reordering the shared text changes the register allocation of the untextured kernels too (the Cornell-box kernel:
// 87 -> 113 spilled registers, 13.16 -> 12.76 Gsamples/s).
template <bool TEX, class S = DScene>
__device__ __forceinline__ Surface shadeSurfaceTexFirst(const uint4 *mtlTable, const S &sc, const TexTables &tt, int mi, f3 N,
                                                f3 V, bool front, const TexHit &th, int bounceLeft, bool fromDiffuse,
                                                uint32_t &rng)
{
    const uint4 *mr = mtlTable + 6 * (size_t) mi;
    const uint4 m0 = mr[0], m1 = mr[1], m2 = mr[2], m5 = mr[5];
    f3 sampleDiffuse = F3(asF(m0.x), asF(m0.y), asF(m0.z));
    const float kill = asF(m0.w);
    f3 sampleSpecular = F3(asF(m1.x), asF(m1.y), asF(m1.z));
    const float glossSpec = asF(m1.w);
    f3 emission = F3(asF(m2.x), asF(m2.y), asF(m2.z));
    const uint32_t mflags = m5.w;
    f3 rK = F3(0, 0, 0), tK = F3(0, 0, 0);
    float glossRefl = 0.f, glossRefr = 0.f;
    if (mflags & QA_MTL_SPECULAR_LOBES) {
      const uint4 m3 = mr[3], m4 = mr[4];
      rK = F3(asF(m3.x), asF(m3.y), asF(m3.z));
      tK = F3(asF(m4.x), asF(m4.y), asF(m4.z));
      glossRefl = asF(m3.w);
      glossRefr = asF(m4.w);
    }
    if (TEX) {
      // Every texture lookup of the hit FIRST (the reference samples the colours where it uses them, MtlBlinn_PhotonMap.cpp:262-330;
      // the lookups are pure functions of the hit, so their place does not show): the 32-tap loops then run with nothing of the
      // Fresnel / lobe arithmetic alive beside them - they are where the textured kernels spill.
      const int *mt = sc.mtlTex + 8 * (size_t) mi;   // texmaps: diffuse, specular, emission, reflection, refraction
      const int4 mtex0 = make_int4(mt[0], mt[1], mt[2], mt[3]);
      const int mtex4 = mt[4];
      emission = mtlSample(tt, th, emission, mtex0.z);
      if (mflags & QA_MTL_SPECULAR_LOBES) {
        tK = mtlSample(tt, th, tK, mtex4);
        rK = mtlSample(tt, th, rK, mtex0.w);
      }
      sampleSpecular = mtlSample(tt, th, sampleSpecular, mtex0.y);
      sampleDiffuse = mtlSample(tt, th, sampleDiffuse, mtex0.x);
    }
    const f3 Y = dot(N, V) > 0.f ? N : -N;

    // ComputeFresnel (:65-105); skipped when neither lobe can receive energy: with
    // tK = rK = 0 both products below are exactly 0 for any finite Fresnel term.
    f3 sampleTransmission = F3(0, 0, 0), sampleReflection = F3(0, 0, 0);
    f3 tDir = F3(0, 0, 0), rDir = F3(0, 0, 0);
    if (mflags & QA_MTL_SPECULAR_LOBES) {
      const float ior = asF(m2.w);
      const f3 Z = cross(V, Y);
      const f3 X = normalize(cross(Y, Z));
      const float nIOR = front ? 1.f / ior : ior;
      const float cosI = dot(N, V);
      const float sinI = qsqrt(1 - cosI * cosI);
      const float sinO = qmax(0.f, qmin(1.f, sinI * nIOR));
      const float cosO = qsqrt(1.f - sinO * sinO);
      tDir = ((-X) * sinO) - (Y * cosO);
      rDir = ((N * 2.f) * dot(N, V)) - V;
      const bool totReflection = (nIOR * sinI) > 1.001f;
      const float C = (nIOR - 1.f) * (nIOR - 1.f) / ((nIOR + 1.f) * (nIOR + 1.f));
      const float rC = C + (1.f - C) * qpowf(1.f - qabs(cosI), 5.f);
      const float tC = 1.f - rC;
      sampleTransmission = totReflection ? F3(0, 0, 0) : tK * tC;
      sampleReflection = totReflection ? (rK + tK) : (rK + tK * rC);
    }

    // RandomSelectMtl (:107-150): one draw, luma-weighted lobes + Russian roulette
    const float lumaT = luma(sampleTransmission), lumaR = luma(sampleReflection), lumaD = luma(sampleDiffuse);
    const float rsel = rng1(rng);
    const float coefTransmit = lumaT;
    const float coefReflection = coefTransmit + lumaR;
    const float coefDiffuse = coefReflection + lumaD;
    const float coefSum = coefDiffuse + kill;
    const float sel = rsel * coefSum;
    int select;  // 0 transmit, 1 reflect, 2 diffuse, 3 absorb
    if (sel < coefTransmit && lumaT > 0.00001f) select = 0;
    else if (sel < coefReflection && lumaR > 0.00001f) select = 1;
    else if (sel < coefDiffuse && lumaD > 0.00001f) select = 2;
    else select = 3;

    // secondary ray (at most one): reflect / transmit / diffuse blocks (:374-479)
    bool spawn = false;
    f3 nextDir = F3(0, 0, 0), bxdf = F3(0, 0, 0);
    bool nextFromDiffuse = false;
    if (bounceLeft > 0) {
      if (select == 1) {
        if (glossRefl > 0.f) {
          do { nextDir = normalize(normalize(rDir) + uniformBall(rng, 2.f * glossRefl)); } while (dot(nextDir, Y) < 0);
        } else nextDir = rDir;
        bxdf = sampleReflection;
        spawn = true;
      } else if (select == 0) {
        if (glossRefr > 0.f) {
          do { nextDir = normalize(normalize(tDir) + uniformBall(rng, 2.f * glossRefr)); } while (dot(nextDir, Y) > 0);
        } else nextDir = tDir;
        bxdf = sampleTransmission;
        spawn = true;
      } else if (select == 2 && !fromDiffuse && front) {
        // SampleDiffuseBxDF (:199-224) + CosWeightedHemisphere (src/core/sampler.cpp:87-103)
        const float r1 = rng1(rng), r2 = rng1(rng);
        const float cosTheta = qsqrt(r1);
        const float sinTheta = qsqrt(1 - r1);
        const float phi = 2 * QA_PI * r2;
        const f3 smp = F3(sinTheta * qcosf(phi), sinTheta * qsinf(phi), cosTheta);
        nextDir = toLocalFrame(N, smp);
        bxdf = sampleDiffuse;
        if (TEX || (mflags & QA_MTL_HAS_SPECULAR)) {
          const f3 Ld = normalize(nextDir);
          const f3 H = normalize(V + Ld);
          const float cosNH = qmax(0.f, dot(N, H));
          bxdf = sampleDiffuse + sampleSpecular * qpowf(cosNH, glossSpec);
        }
        nextFromDiffuse = true;
        spawn = true;
      }
    }


  Surface o;
  o.emission = emission;
  o.kd = sampleDiffuse;
  o.ks = sampleSpecular;
  o.gloss = glossSpec;
  o.spawn = spawn;
  o.nextDir = nextDir;
  o.bxdf = bxdf;
  o.nextFromDiffuse = nextFromDiffuse;
  o.selDiffuse = (select == 2);
  return o;
}

// (S: anything with the material -> texmap table `mtlTex`: DScene, or the few words an out-of-line caller hands over)
template <bool TEX, class S = DScene>
__device__ __forceinline__ Surface shadeSurface(const uint4 *mtlTable, const S &sc, const TexTables &tt, int mi, f3 N,
                                                f3 V, bool front, const TexHit &th, int bounceLeft, bool fromDiffuse,
                                                uint32_t &rng)
{
    if constexpr (TEX) return shadeSurfaceTexFirst<TEX, S>(mtlTable, sc, tt, mi, N, V, front, th, bounceLeft, fromDiffuse, rng);
    const uint4 *mr = mtlTable + 6 * (size_t) mi;
    const uint4 m0 = mr[0], m1 = mr[1], m2 = mr[2], m5 = mr[5];
    f3 sampleDiffuse = F3(asF(m0.x), asF(m0.y), asF(m0.z));
    const float kill = asF(m0.w);
    f3 sampleSpecular = F3(asF(m1.x), asF(m1.y), asF(m1.z));
    const float glossSpec = asF(m1.w);
    f3 emission = F3(asF(m2.x), asF(m2.y), asF(m2.z));
    int4 mtex0 = make_int4(-1, -1, -1, -1);  // texmaps: diffuse, specular, emission, reflection
    int mtex4 = -1;                          //          refraction
    if (TEX) {
      const int *mt = sc.mtlTex + 8 * (size_t) mi;
      mtex0 = make_int4(mt[0], mt[1], mt[2], mt[3]);
      mtex4 = mt[4];
      emission = mtlSample(tt, th, emission, mtex0.z);
    }
    const uint32_t mflags = m5.w;
        const f3 Y = dot(N, V) > 0.f ? N : -N;
    
    // ComputeFresnel (:65-105); skipped when neither lobe can receive energy: with
    // tK = rK = 0 both products below are exactly 0 for any finite Fresnel term.
    f3 sampleTransmission = F3(0, 0, 0), sampleReflection = F3(0, 0, 0);
    f3 tDir = F3(0, 0, 0), rDir = F3(0, 0, 0);
    float glossRefl = 0.f, glossRefr = 0.f;
    if (mflags & QA_MTL_SPECULAR_LOBES) {
      const uint4 m3 = mr[3], m4 = mr[4];
      f3 rK = F3(asF(m3.x), asF(m3.y), asF(m3.z)), tK = F3(asF(m4.x), asF(m4.y), asF(m4.z));
      if (TEX) {
        tK = mtlSample(tt, th, tK, mtex4);
        rK = mtlSample(tt, th, rK, mtex0.w);
      }
      glossRefl = asF(m3.w);
      glossRefr = asF(m4.w);
      const float ior = asF(m2.w);
      const f3 Z = cross(V, Y);
      const f3 X = normalize(cross(Y, Z));
      const float nIOR = front ? 1.f / ior : ior;
      const float cosI = dot(N, V);
      const float sinI = qsqrt(1 - cosI * cosI);
      const float sinO = qmax(0.f, qmin(1.f, sinI * nIOR));
      const float cosO = qsqrt(1.f - sinO * sinO);
      tDir = ((-X) * sinO) - (Y * cosO);
      rDir = ((N * 2.f) * dot(N, V)) - V;
      const bool totReflection = (nIOR * sinI) > 1.001f;
      const float C = (nIOR - 1.f) * (nIOR - 1.f) / ((nIOR + 1.f) * (nIOR + 1.f));
      const float rC = C + (1.f - C) * qpowf(1.f - qabs(cosI), 5.f);
      const float tC = 1.f - rC;
      sampleTransmission = totReflection ? F3(0, 0, 0) : tK * tC;
      sampleReflection = totReflection ? (rK + tK) : (rK + tK * rC);
    }

    if (TEX) {
      sampleSpecular = mtlSample(tt, th, sampleSpecular, mtex0.y);
      sampleDiffuse = mtlSample(tt, th, sampleDiffuse, mtex0.x);
    }
    // RandomSelectMtl (:107-150): one draw, luma-weighted lobes + Russian roulette
    const float lumaT = luma(sampleTransmission), lumaR = luma(sampleReflection), lumaD = luma(sampleDiffuse);
    const float rsel = rng1(rng);
    const float coefTransmit = lumaT;
    const float coefReflection = coefTransmit + lumaR;
    const float coefDiffuse = coefReflection + lumaD;
    const float coefSum = coefDiffuse + kill;
    const float sel = rsel * coefSum;
    int select;  // 0 transmit, 1 reflect, 2 diffuse, 3 absorb
    if (sel < coefTransmit && lumaT > 0.00001f) select = 0;
    else if (sel < coefReflection && lumaR > 0.00001f) select = 1;
    else if (sel < coefDiffuse && lumaD > 0.00001f) select = 2;
    else select = 3;

    // secondary ray (at most one): reflect / transmit / diffuse blocks (:374-479)
    bool spawn = false;
    f3 nextDir = F3(0, 0, 0), bxdf = F3(0, 0, 0);
    bool nextFromDiffuse = false;
    if (bounceLeft > 0) {
      if (select == 1) {
        if (glossRefl > 0.f) {
          do { nextDir = normalize(normalize(rDir) + uniformBall(rng, 2.f * glossRefl)); } while (dot(nextDir, Y) < 0);
        } else nextDir = rDir;
        bxdf = sampleReflection;
        spawn = true;
      } else if (select == 0) {
        if (glossRefr > 0.f) {
          do { nextDir = normalize(normalize(tDir) + uniformBall(rng, 2.f * glossRefr)); } while (dot(nextDir, Y) > 0);
        } else nextDir = tDir;
        bxdf = sampleTransmission;
        spawn = true;
      } else if (select == 2 && !fromDiffuse && front) {
        // SampleDiffuseBxDF (:199-224) + CosWeightedHemisphere (src/core/sampler.cpp:87-103)
        const float r1 = rng1(rng), r2 = rng1(rng);
        const float cosTheta = qsqrt(r1);
        const float sinTheta = qsqrt(1 - r1);
        const float phi = 2 * QA_PI * r2;
        const f3 smp = F3(sinTheta * qcosf(phi), sinTheta * qsinf(phi), cosTheta);
        nextDir = toLocalFrame(N, smp);
        bxdf = sampleDiffuse;
        if (TEX || (mflags & QA_MTL_HAS_SPECULAR)) {
          const f3 Ld = normalize(nextDir);
          const f3 H = normalize(V + Ld);
          const float cosNH = qmax(0.f, dot(N, H));
          bxdf = sampleDiffuse + sampleSpecular * qpowf(cosNH, glossSpec);
        }
        nextFromDiffuse = true;
        spawn = true;
      }
    }


  Surface o;
  o.emission = emission;
  o.kd = sampleDiffuse;
  o.ks = sampleSpecular;
  o.gloss = glossSpec;
  o.spawn = spawn;
  o.nextDir = nextDir;
  o.bxdf = bxdf;
  o.nextFromDiffuse = nextFromDiffuse;
  o.selDiffuse = (select == 2);
  return o;
}

// ---------------------------------------------------------------------------------------------
// Per-lane path state
// ---------------------------------------------------------------------------------------------
struct Path {
  Ray ray;          // next ray to trace (world space)
  f3 T;             // throughput
  f3 L;             // radiance gathered by this sample so far
  int absorbMtl;    // material the current ray was spawned from: its absorption applies on a
                    // back-face exit (Beer's law, ComputeSecondaryRay :244-248); -1 for camera rays
  int bounce;       // bounceCount the next hit is shaded with
  bool fromDiffuse; // hInfo.c.hasDiffuseHit of the next hit
  bool primary;     // camera ray
};

// ---------------------------------------------------------------------------------------------
// The kernel.  Dynamic LDS: [resident scene image (RES) | traversal stacks (stackDepth x 256)]
// ---------------------------------------------------------------------------------------------
// Area lights draw random numbers, and the reference evaluates a hit's lights only AFTER the whole
// recursive subtree below it (MtlBlinn_PhotonMap.cpp:374-479 precede :484-498).  AREA variants
// therefore log one record per hit (19 floats, SoA in a global scratch slab) and replay the lights
// deepest hit first when the path ends, continuing the same xorshift32 stream.
#define QA_MAX_PATH 8           /* hits per path an AREA variant can log (maxBounce <= 7) */
#define QA_REC_FLOATS 19

// PHOTON variants (Scene::usePhotonMap, -use-photon-map): a DIFFUSE selection gathers from the
// caustics map, and from the photon map instead of bouncing when the ray already comes from a
// diffuse bounce (MtlBlinn_PhotonMap.cpp:349-359,426-458).  The gathers draw no random numbers, so
// they are evaluated at the hit (also by AREA variants).
// Waves per SIMD the register allocator must leave room for.  The LDS-resident kernel without lights
// (the Cornell box class: VALU-issue bound, smallest register footprint) gains from a fifth wave (+5 %);
// every other variant loses more to the extra spills than the added latency hiding returns
// (resident glass-sphere room -36 %, project7_object -8 %, glossy caustics -10 %); 6 loses everywhere.
#ifndef QA_C2_EXTRA_WAVE
#define QA_C2_EXTRA_WAVE 1   /* 0: four waves per SIMD for that variant too - spill-free and 6 % slower (profiles/round02/c2_register_variants.txt) */
#endif
#define QA_WAVES_FOR(RES, LIGHTS) (((RES) && !(LIGHTS)) ? QA_MIN_WAVES + QA_C2_EXTRA_WAVE : QA_MIN_WAVES)

template <bool RES, bool LIGHTS, bool TEX, bool AREA, bool STATS, bool PHOTON = false>
__global__ __launch_bounds__(QA_BLOCK, QA_WAVES_FOR(RES, LIGHTS)) void qa_integrate(const DScene sc, const RenderParams rp)
{
  extern __shared__ uint4 s_dyn[];
  SceneMem<RES> mem;
  mem.img = s_dyn;
  if (RES) {
    for (uint32_t i = threadIdx.x; i < sc.residentVec4; i += QA_BLOCK) s_dyn[i] = sc.resident[i];
    __syncthreads();
  }
  uint32_t *stack = reinterpret_cast<uint32_t *>(s_dyn + (RES ? sc.residentVec4 : 0)) + threadIdx.x;
  // per-lane sample accumulators (running mean + variance of SuperSamplerHalton) live in LDS: they
  // are touched once per sample, registers are better spent on the traversal
  float *acc = reinterpret_cast<float *>(stack + (size_t) sc.stackDepth * QA_BLOCK - threadIdx.x) + threadIdx.x;
  const uint4 *mtlTable = RES ? s_dyn + sc.resMaterials : reinterpret_cast<const uint4 *>(sc.mtl);
  // LDS-resident scenes have shallow stacks: their workgroups also keep the path's throughput and radiance, the pixel, its output
  // index and the sample index in LDS columns (QA_LANE_SLOTS_RES) - state touched at a handful of points of an iteration
  constexpr bool LCOLS = RES && !PHOTON;   // (a photon gather's stack is deep: those variants keep the registers)
  if (LCOLS)
    for (int i = QA_LANE_SLOTS; i < QA_LANE_SLOTS_RES; ++i) acc[i * QA_BLOCK] = 0.f;
#define QA_GET_T() (LCOLS ? F3(acc[6 * QA_BLOCK], acc[7 * QA_BLOCK], acc[8 * QA_BLOCK]) : path.T)
#define QA_GET_L() (LCOLS ? F3(acc[9 * QA_BLOCK], acc[10 * QA_BLOCK], acc[11 * QA_BLOCK]) : path.L)
#define QA_PUT_T(...) { const f3 v_ = (__VA_ARGS__); if (LCOLS) { acc[6 * QA_BLOCK] = v_.x; acc[7 * QA_BLOCK] = v_.y; acc[8 * QA_BLOCK] = v_.z; } else path.T = v_; }
#define QA_PUT_L(...) { const f3 v_ = (__VA_ARGS__); if (LCOLS) { acc[9 * QA_BLOCK] = v_.x; acc[10 * QA_BLOCK] = v_.y; acc[11 * QA_BLOCK] = v_.z; } else path.L = v_; }
#define QA_GET_Q() (LCOLS ? __float_as_uint(acc[13 * QA_BLOCK]) : q)
#define QA_GET_SIDX() (LCOLS ? __float_as_int(acc[14 * QA_BLOCK]) : sidx)

  // work items walk 8x8 pixel tiles (a wave starts on a compact screen patch); ragged right /
  // bottom tiles contain padding slots that are simply skipped.
  // A launch may own only every tile_row_step-th 8-row strip of the region (round-robin image
  // partition between GPUs, the reference's ThreadRender(tileStart=rank, step=size),
  // src/renderers/renderer.cpp:383-387); its outputs are packed strip after strip.
  const int rw = rp.x1 - rp.x0, rh = rp.y1 - rp.y0;
  const unsigned tilesX = (unsigned) (rw + 7) / 8;
  // Tiles in sample chunks (RenderParams::chunk_spp): a frame of few tiles per wave ends with waves idle while the last tiles finish
  // their hundreds of samples (1080p at 512 spp on 5 120 waves: 6.3 tiles of ~12 ms per wave, 11 % of the wave slots empty on
  // average).  A pixel's samples cannot be shared out - one random-number stream, one running variance - but they can be HANDED ON:
  // a work item is (chunk, tile), all tiles' chunk 0 first; at the end of a chunk every lane stores its pixel's state, the wave
  // publishes the tile's progress (agent-scope release), and whoever fetches (chunk + 1, tile) - a whole pass of the frame later -
  // reads the state back behind an acquire.  Same samples in the same order for every pixel: same bits.
  const unsigned numTiles = tilesX * (unsigned) rp.own_tile_rows;
  const unsigned total = numTiles * (rp.chunk_spp ? rp.num_chunks : 1u) * 64u;
  const unsigned lane = __lane_id();
  unsigned curTile = 0xFFFFFFFFu, curChunk = 0;   // (wave-uniform) the work item in hand
  int chunkEnd = 0x7FFFFFFF;                      // samples a pixel has when its chunk is complete

  DCounters cnt = {};
#ifdef QA_STAMPS
  __shared__ unsigned long long s_stamps[QA_BLOCK / 64][QA_NSTAMPS];
  cnt.sl = s_stamps[threadIdx.x / 64];
  if (__lane_id() < QA_NSTAMPS) cnt.sl[__lane_id()] = 0;
#endif
  QA_T(tKernel)
  TexTables tt;
  tt.blob = sc.blob;
  tt.texels = sc.texels;
  tt.texOff = sc.texOff;
  tt.texmap = sc.texmap;
  tt.tex = sc.tex;
  tt.filter = sc.texFilter;
  int nrec = 0;               // AREA: hits logged for the current path
  float *rec = sc.areaScratch + (size_t) blockIdx.x * QA_BLOCK + threadIdx.x;  // + (lvl*19+f) * recStride
  const size_t recStride = (size_t) gridDim.x * QA_BLOCK;
  RayDiff pathDiff;           // TEX: differential directions of the current ray
  pathDiff.dx = pathDiff.dy = F3(0, 0, 1);

  // pixel state
  int px = 0, py = 0;
  unsigned q = 0;           // output index of the pixel
  uint32_t rng = 1;
  int sidx = 0;
  Path path;
  path.primary = true;
  path.ray.p = F3(0, 0, 0);
  path.ray.d = F3(0, 0, 1);
  path.T = F3(0, 0, 0);
  path.L = F3(0, 0, 0);
  path.absorbMtl = -1;
  path.bounce = 0;
  path.fromDiffuse = false;
  f3 texpos = F3(0, 0, 0);

  bool alive = true, needPixel = true, needSample = false;

  for (;;) {
    QA_T(tA)
    // ---- A. tile fetch: a wave owns one 8x8 pixel tile at a time (lane = pixel).  Rays of one
    // tile are coherent and cost about the same, so background tiles (every ray misses the scene
    // bounds) never share a wave with expensive ones.  One atomic per wave and tile; lanes that
    // finish their pixel early wait for the rest of the tile (the spread is a few percent).
    const unsigned long long aliveMask = __ballot(alive);
    const unsigned long long want = __ballot(alive && needPixel);
    if (want && want == aliveMask) {
      if (rp.chunk_spp && curTile != 0xFFFFFFFFu) {
        // the chunk in hand is complete: every lane has stored its pixel's state (section E); publish it
        // (the state words are agent-scope atomic stores - written through, no line of them stays in this XCD's L2 - and this wave
        // has waited for all of them: no release fence, whose write-back of the L2's dirty lines - the spilled registers of every
        // wave of the XCD - cost 170 us per hand-over; MI355X_MICROARCH.md, inter-workgroup visibility)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(rp.tile_progress + curTile, curChunk + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        curTile = 0xFFFFFFFFu;
      }
      unsigned base = 0;
      const int leader = __ffsll((long long) want) - 1;
      if ((int) lane == leader) base = (*rp.stop_flag) ? total : atomicAdd(rp.work_counter, 64u);
      base = __shfl(base, leader);
      unsigned item = base / 64;   // (wave-uniform) tile, or chunk * numTiles + tile
      if (rp.chunk_spp && base < total) {
        curChunk = item / numTiles;
        item -= curChunk * numTiles;
        curTile = item;
        chunkEnd = (int) (rp.chunk_spp + curChunk * rp.chunk_tail);
        if (curChunk > 0) {
          // the tile's previous chunk was handed out a whole pass of the frame ago: this wait ends at once, except on frames of
          // fewer tiles than waves.  Its holder is a resident wave that waits for nothing this wave holds; the bound is a guard
          // against a lost update, not a path that is taken (a frame that hit it would fail every parity test).
          for (int spins = 0; spins < (1 << 22); ++spins) {
            if (__hip_atomic_load(rp.tile_progress + curTile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= curChunk) break;
            __builtin_amdgcn_s_sleep(16);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
      }
      if (alive) {
        const unsigned w = base + lane;
        if (base >= total) {
          alive = false;
        } else {
          const unsigned in = w % 64;
          const unsigned tile = rp.tile_order ? rp.tile_order[item] : item;
          const unsigned otr = tile / tilesX;  // index among the strips this launch owns
          const unsigned tx = (tile % tilesX) * 8 + (in % 8);
          const unsigned ty = ((unsigned) rp.tile_row0 + otr * (unsigned) rp.tile_row_step) * 8 + (in / 8);
          if (tx < (unsigned) rw && ty < (unsigned) rh) {
            px = rp.x0 + (int) tx;
            py = rp.y0 + (int) ty;
            q = (otr * 8 + (in / 8)) * (unsigned) rw + tx;
            rng = qa_pixel_seed(rp.seed, (uint32_t) py * (uint32_t) sc.cam.width + (uint32_t) px);
            sidx = 0;
            for (int i = 0; i < 6; ++i) acc[i * QA_BLOCK] = 0.f;
            needSample = true;
            needPixel = false;
            if (rp.chunk_spp && curChunk > 0) {
              // the pixel as the previous chunk left it
              const unsigned long long *st = reinterpret_cast<const unsigned long long *>(rp.pix_state) + 4 * (size_t) q;
              const unsigned long long s0 = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), s1 = __hip_atomic_load(st + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                       s2 = __hip_atomic_load(st + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), s3 = __hip_atomic_load(st + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const uint4 a = make_uint4((uint32_t) s0, (uint32_t) (s0 >> 32), (uint32_t) s1, (uint32_t) (s1 >> 32));
              const uint4 b = make_uint4((uint32_t) s2, (uint32_t) (s2 >> 32), (uint32_t) s3, (uint32_t) (s3 >> 32));
              if (a.y & 0x80000000u) {   // finished in an earlier chunk: the lane sits this one out
                needSample = false;
                needPixel = true;
              } else {
                rng = a.x;
                sidx = (int) a.y;
                acc[0] = __uint_as_float(a.z); acc[QA_BLOCK] = __uint_as_float(a.w); acc[2 * QA_BLOCK] = __uint_as_float(b.x);
                acc[3 * QA_BLOCK] = __uint_as_float(b.y); acc[4 * QA_BLOCK] = __uint_as_float(b.z); acc[5 * QA_BLOCK] = __uint_as_float(b.w);
              }
            }
            if (LCOLS) {
              acc[12 * QA_BLOCK] = __uint_as_float((unsigned) px | ((unsigned) py << 16));
              acc[13 * QA_BLOCK] = __uint_as_float(q);
              acc[14 * QA_BLOCK] = __int_as_float(sidx);
            }
          }
          // else: padding slot of a ragged tile - this lane sits the tile out
        }
      }
    }
    if (!__any(alive)) break;

    // ---- B. start a sample: camera ray (src/renderers/renderer.cpp:312-328) ------------------
    // sync_samples: lanes wait until the whole wave is between samples, so that the coherent
    // camera rays of a tile are traced together instead of next to incoherent secondary rays
    const bool goSample = !rp.sync_samples || (__ballot(needSample) == __ballot(alive && !needPixel));
    if (alive && needSample && goSample) {
      const int si = QA_GET_SIDX();
      const float hx = sc.halton[2 * si], hy = sc.halton[2 * si + 1];
      if (LCOLS) {
        const unsigned pxy = __float_as_uint(acc[12 * QA_BLOCK]);
        texpos = F3(hx, hy, 0.f) + F3((float) (int) (pxy & 0xFFFFu), (float) (int) (pxy >> 16), 0.f);
      } else {
        texpos = F3(hx, hy, 0.f) + F3((float) px, (float) py, 0.f);
      }
      const f3 A = ld3(sc.cam.screenA), U = ld3(sc.cam.screenU), V = ld3(sc.cam.screenV);
      const f3 cpt = (A + U * texpos.x) + V * texpos.y;
      f3 campos = ld3(sc.cam.pos);
      if (sc.cam.dof > 0.1f) {
        // SuperSamplerHalton::NewDofSample (src/scene/scene.cpp:104-111)
        const float r1 = rng1(rng), r2 = rng1(rng);
        const float r = sc.cam.dof * qsqrt(r1);
        const float t = r2 * 2.f * QA_PI;
        campos = campos + (ld3(sc.cam.screenX) * (r * qcosf(t)) + ld3(sc.cam.screenY) * (r * qsinf(t)));
      }
      path.ray.p = campos;
      path.ray.d = normalize(cpt - campos);
      if (TEX) {
        // DiffRay x / y: the same pixel sample shifted by DiffRay::dx / dy (renderer.cpp:314-317)
        const f3 xpt = (A + U * (texpos.x + QA_DX)) + V * texpos.y;
        const f3 ypt = (A + U * texpos.x) + V * (texpos.y + QA_DX);
        pathDiff.dx = normalize(xpt - campos);
        pathDiff.dy = normalize(ypt - campos);
      }
      QA_PUT_T(F3(1, 1, 1))
      QA_PUT_L(F3(0, 0, 0))
      path.absorbMtl = -1;
      path.bounce = rp.max_bounce;
      path.fromDiffuse = false;
      path.primary = true;
      needSample = false;
      nrec = 0;
      QA_TALLY(cnt.samples);
    }

    QA_TACC(cnt.sl[1], tA)
    // ---- C. trace ----------------------------------------------------------------------------
    bool done = false;  // path finished in this iteration
    if (alive && !needPixel && !needSample) {
      Hit h;
      h.z = QA_BIGFLOAT;
      h.node = -1;
      h.mtlID = 0;
      h.front = true;
      h.p = F3(0, 0, 0);
      h.N = F3(0, 0, 0);
      TexHit th;
      th.uvw = F3(0.5f, 0.5f, 0.5f);   // HitInfo::Init (src/core/hitinfo.cpp:31-42)
      th.duvw0 = th.duvw1 = F3(0, 0, 0);
      th.hasTexture = false;
      QA_T(tC)
      const bool found = traceClosest<RES, TEX, STATS>(mem, sc, path.ray, pathDiff, h, th, stack, cnt);
      QA_TACC(cnt.sl[2], tC)
      if (path.primary && QA_GET_SIDX() == 0) rp.depth[QA_GET_Q()] = found ? h.z : QA_BIGFLOAT;

      QA_T(tM)
      if (!found) {
        // background for camera rays (renderer.cpp:337-341), environment otherwise
        // (MtlBlinn_PhotonMap.cpp:249-251); textured versions: TEX kernel variants
        f3 c = path.primary ? ld3(sc.background) : ld3(sc.environment);
        if (TEX) {
          if (path.primary)
            c = texColorSample(tt, c, sc.bgTexmap, F3(texpos.x / (float) sc.cam.width, texpos.y / (float) sc.cam.height, 0.f));
          else
            c = sampleEnvironment(tt, c, sc.envTexmap, path.ray.d);
        }
        QA_PUT_L(QA_GET_L() + QA_GET_T() * c)
        done = true;
        QA_TACC(cnt.sl[10], tM)
      } else {
        // ---- D. shade: MtlBlinn_PhotonMap::Shade (MtlBlinn_PhotonMap.cpp:256-500) -----------
        // Beer-Lambert attenuation of everything this hit returns, when the ray arrives from
        // inside (ComputeSecondaryRay :244-248)
        if (!path.primary && !h.front && path.absorbMtl >= 0) {
          const uint4 ab = mtlTable[6 * (size_t) path.absorbMtl + 5];
          const f3 att = F3(qexpf(-asF(ab.x) * h.z), qexpf(-asF(ab.y) * h.z), qexpf(-asF(ab.z) * h.z));
          QA_PUT_T(QA_GET_T() * att)
        }
        const qa_instance &in = instAt<RES>(sc, h.node);
        int mi = -1;
        bool white = false;
        if (in.mtlset >= 0) {
          const qa_mtlset ms = sc.mtlset[in.mtlset];
          if (ms.multi) {
            if (h.mtlID >= 0 && h.mtlID < ms.count) mi = ms.first + h.mtlID;
            else white = true;  // MultiMtl::Shade returns (1,1,1) (materials.h:70-76)
          } else mi = ms.first;
        }
        if (mi < 0) {
          if (white) QA_PUT_L(QA_GET_L() + QA_GET_T())
          done = true;
        } else {
          const f3 V = -path.ray.d;
          const f3 N = h.N;
          const f3 p = h.p;
          QA_TACC(cnt.sl[11], tM)
          QA_T(tD)
          const Surface sf = shadeSurface<TEX>(mtlTable, sc, tt, mi, N, V, h.front, th, path.bounce, path.fromDiffuse, rng);
          QA_TACC(cnt.sl[4], tD)
          QA_PUT_L(QA_GET_L() + QA_GET_T() * sf.emission)
          const f3 sampleDiffuse = sf.kd, sampleSpecular = sf.ks;
          const float glossSpec = sf.gloss;
          const bool spawn = sf.spawn;
          const f3 nextDir = sf.nextDir, bxdf = sf.bxdf;
          const bool nextFromDiffuse = sf.nextFromDiffuse;

          if (PHOTON && sf.selDiffuse) {
            // this lane's heap: QA_PHOTON_GATHER + 1 consecutive elements of the scratch slab (the top levels of
            // a heap share a cache line that way: 10 - 12 % faster than slot-major columns)
            uint2 *heap = rp.heap + ((size_t) blockIdx.x * QA_BLOCK + threadIdx.x) * (QA_PHOTON_GATHER + 1);
            if (path.fromDiffuse)
              QA_PUT_L(QA_GET_L() + QA_GET_T() * photonGather(rp.pm[0], p, N, V, sampleDiffuse, sampleSpecular, glossSpec, stack, heap))
            QA_PUT_L(QA_GET_L() + QA_GET_T() * photonGather(rp.pm[1], p, N, V, sampleDiffuse, sampleSpecular, glossSpec, stack, heap))
          }

          // direct lighting (:481-498)
          if (LIGHTS && !AREA) {
            QA_T(tL)
            QA_PUT_L(QA_GET_L() + QA_GET_T() * directLight<RES, STATS>(mem, sc, p, N, V, sampleDiffuse, sampleSpecular, glossSpec, stack, cnt, rng))
            QA_TACC(cnt.sl[5], tL)
          }
          if (AREA && nrec < QA_MAX_PATH) {
            const f3 pT = QA_GET_T();
            const float v[QA_REC_FLOATS] = {p.x, p.y, p.z, N.x, N.y, N.z, V.x, V.y, V.z, pT.x, pT.y, pT.z,
                                            sampleDiffuse.x, sampleDiffuse.y, sampleDiffuse.z,
                                            sampleSpecular.x, sampleSpecular.y, sampleSpecular.z, glossSpec};
            for (int f = 0; f < QA_REC_FLOATS; ++f) rec[(size_t) (nrec * QA_REC_FLOATS + f) * recStride] = v[f];
            ++nrec;
          }

          QA_T(tS)
          if (spawn) {
            // ComputeSecondaryRay (:226-254): DiffRay(pos, dir).Normalize()
            path.ray.p = p;
            path.ray.d = normalize(nextDir);
            if (TEX) pathDiff.dx = pathDiff.dy = path.ray.d;  // DiffRay(pos, dir): x = y = c (ray.h:57-63)
            QA_PUT_T(QA_GET_T() * bxdf)
            path.absorbMtl = mi;
            path.bounce -= 1;
            path.fromDiffuse = nextFromDiffuse;
            path.primary = false;
          } else {
            done = true;
          }
          QA_TACC(cnt.sl[12], tS)
        }
      }
    }

    // ---- E. sample finished: SuperSamplerHalton::Accumulate / Loop (scene.cpp:92-121) ---------
    QA_T(tE)
#ifdef QA_STAMPS
    if (lane == 0) cnt.sl[8] += 1;
#endif
    if (alive && done) {
      if (AREA) {
        for (int lvl = nrec - 1; lvl >= 0; --lvl) {
          float v[QA_REC_FLOATS];
          for (int f = 0; f < QA_REC_FLOATS; ++f) v[f] = rec[(size_t) (lvl * QA_REC_FLOATS + f) * recStride];
          const f3 d = directLight<RES, STATS>(mem, sc, F3(v[0], v[1], v[2]), F3(v[3], v[4], v[5]), F3(v[6], v[7], v[8]),
                                               F3(v[12], v[13], v[14]), F3(v[15], v[16], v[17]), v[18], stack, cnt, rng);
          QA_PUT_L(QA_GET_L() + F3(v[9], v[10], v[11]) * d)
        }
        nrec = 0;
      }
      if (LCOLS) sidx = QA_GET_SIDX();
      const unsigned qo = QA_GET_Q();
      const f3 pL = QA_GET_L();
      const float inv = (float) (sidx + 1);
      f3 mean = F3(acc[0], acc[QA_BLOCK], acc[2 * QA_BLOCK]);
      f3 cstd = F3(0, 0, 0);
      const f3 dc = (pL - mean) / inv;
      mean = mean + dc;
      acc[0] = mean.x; acc[QA_BLOCK] = mean.y; acc[2 * QA_BLOCK] = mean.z;
      // The running variance decides one thing - whether a pixel past sppMin takes another sample (below) - and is no output: with
      // sppMin == sppMax (every BASELINE config) nothing reads it, and its three correctly rounded divisions per sample are not made
      // (Cornell box + 1.5 %).  In the variants without lights only: the lit LDS-resident kernel lost 8 % to the changed register
      // allocation around this branch (project3_sphere 18 500 -> 16 900), profiles/round03/experiments.txt 32.
      if (LIGHTS || rp.spp_min < rp.spp_max) {
        cstd = F3(acc[3 * QA_BLOCK], acc[4 * QA_BLOCK], acc[5 * QA_BLOCK]);
        if (sidx > 0) cstd = cstd + ((dc * dc) * inv - cstd / (float) sidx);
        acc[3 * QA_BLOCK] = cstd.x; acc[4 * QA_BLOCK] = cstd.y; acc[5 * QA_BLOCK] = cstd.z;
      }
      ++sidx;
      if (LCOLS) acc[14 * QA_BLOCK] = __int_as_float(sidx);
      const bool more = sidx < rp.spp_min ||
                        (sidx < rp.spp_max && (cstd.x > 0.005f || cstd.y > 0.001f || cstd.z > 0.005f));
      if (more) {
        if (rp.chunk_spp && sidx >= chunkEnd) {
          // the chunk's last sample of this pixel: its state waits for whoever takes the tile's next chunk
          unsigned long long *st = reinterpret_cast<unsigned long long *>(rp.pix_state) + 4 * (size_t) qo;
#define QA_PAIR(lo, hi) ((unsigned long long) (lo) | ((unsigned long long) (hi) << 32))
          __hip_atomic_store(st, QA_PAIR(rng, (uint32_t) sidx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(st + 1, QA_PAIR(__float_as_uint(mean.x), __float_as_uint(mean.y)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(st + 2, QA_PAIR(__float_as_uint(mean.z), __float_as_uint(cstd.x)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(st + 3, QA_PAIR(__float_as_uint(cstd.y), __float_as_uint(cstd.z)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#undef QA_PAIR
          needPixel = true;
        } else {
          needSample = true;
        }
      } else {
        rp.rgb[3 * qo + 0] = mean.x;
        rp.rgb[3 * qo + 1] = mean.y;
        rp.rgb[3 * qo + 2] = mean.z;
        rp.ns[qo] = (uint32_t) sidx;
        if (rp.chunk_spp)   // (later chunks of the tile skip this pixel)
          __hip_atomic_store(reinterpret_cast<unsigned long long *>(rp.pix_state) + 4 * (size_t) qo, 0x80000000ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        QA_TALLY(cnt.pixels);
        needPixel = true;
      }
    }
    QA_TACC(cnt.sl[7], tE)
  }

  // ---- counters: wave reduction, one atomic per wave and counter -----------------------------
  unsigned long long v[6] = {cnt.samples, cnt.casts_normal, cnt.casts_shadow, cnt.bvh_nodes, cnt.tri_tests, cnt.pixels};
  unsigned long long *dst = reinterpret_cast<unsigned long long *>(rp.counters);
  for (int i = 0; i < 6; ++i) {
    unsigned long long x = v[i];
#ifdef QA_LANE_TALLIES
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
#endif
    if (lane == 0 && x) atomicAdd(&dst[i], x);
  }
#ifdef QA_STAMPS
  if (lane == 0) {
    cnt.sl[0] = __builtin_readcyclecounter() - tKernel;
    cnt.sl[9] = 1;
    for (int i = 0; i < QA_NSTAMPS; ++i) atomicAdd(&dst[6 + i], cnt.sl[i]);
  }
#endif
}

}  // namespace qa
