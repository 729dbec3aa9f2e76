// qa_photon.hip — photon / caustics maps of the reference's -use-photon-map mode on the GPU
// (second translation unit of libqaray_hip.so).
//
// Reference: Renderer::ComputeScene's two emission loops (src/renderers/renderer.cpp:114-291),
// PointLight::RandomPhoton (src/lights/lights.cpp:76-80), MtlBlinn_PhotonMap::RandomPhotonBounce
// (src/materials/MtlBlinn_PhotonMap.cpp:503-578), cy::PhotonMap (src/ext/cyPhotonMap.h).
//
// The reference traces photons one after the other on the main thread.  Here every emission has
// its own RNG stream (include/qa_photon.h), so a batch of emissions is traced by one kernel, one
// lane per emission, each writing the photons it would store to its own slots; a second kernel
// scans the per-emission counts and packs the photons in emission order, which reproduces the
// serial loop's map, its stopping rule ("the first size candidates") and its numOfEmittedRays.
// The kd-tree balancing is the reference's serial quick-select, restated on the host: it runs once
// per frame, and its swap sequence decides how ties are split, so it is kept serial.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "qa_kernel.h"
#include "qa_ctx.h"

namespace qa {

struct EmitParams {
  uint32_t first, count;     // emission indices [first, first + count)
  uint32_t max_bounce;
  int32_t caustics;
  uint32_t seed;
  int32_t num_sources;
  const int32_t *sources;    // light indices of the photon sources (Light::IsPhotonSource)
  uint32_t max_rec;          // record slots per emission
  uint32_t *counts;          // [count]
  uint32_t *cand;            // [count][max_rec] records of 6 dwords (qa_photon)
};

// MtlBlinn_PhotonMap::RandomPhotonBounce (MtlBlinn_PhotonMap.cpp:503-578).  On success the ray
// continues from the hit in the sampled direction and the power is multiplied by
// BxDF / (PDF * selection probability), and by Beer's law on a back-face hit.
template <bool TEX>
__device__ __forceinline__ bool photonBounce(const uint4 *mtlTable, const DScene &sc, const TexTables &tt, int mi, Ray &ray,
                                             f3 &power, const Hit &h, const TexHit &th, uint32_t &rng)
{
  const uint4 *mr = mtlTable + 6 * (size_t) mi;
  const uint4 m0 = mr[0], m1 = mr[1], m2 = mr[2], m5 = mr[5];
  f3 sampleDiffuse = F3(asF(m0.x), asF(m0.y), asF(m0.z));
  const float kill = asF(m0.w);
  f3 sampleSpecular = F3(asF(m1.x), asF(m1.y), asF(m1.z));
  const float glossSpec = asF(m1.w);
  int4 mtex0 = make_int4(-1, -1, -1, -1);
  int mtex4 = -1;
  if (TEX) {
    const int *mt = sc.mtlTex + 8 * (size_t) mi;
    mtex0 = make_int4(mt[0], mt[1], mt[2], mt[3]);
    mtex4 = mt[4];
  }
  const uint32_t mflags = m5.w;
  const f3 V = -ray.d;
  const f3 N = h.N;
  const f3 Y = dot(N, V) > 0.f ? N : -N;
  // ComputeFresnel (:65-105), skipped like in shadeSurface when both specular colours are black
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
    const float nIOR = h.front ? 1.f / ior : ior;
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
    sampleDiffuse = mtlSample(tt, th, sampleDiffuse, mtex0.x);
    sampleSpecular = mtlSample(tt, th, sampleSpecular, mtex0.y);
  }
  // RandomSelectMtl with its scale output (:107-150)
  const float lumaT = luma(sampleTransmission), lumaR = luma(sampleReflection), lumaD = luma(sampleDiffuse);
  const float rsel = rng1(rng);
  const float coefTransmit = lumaT;
  const float coefReflection = coefTransmit + lumaR;
  const float coefDiffuse = coefReflection + lumaD;
  const float coefAbsorb = coefDiffuse + kill;
  const float rcpCoefSum = 1.f / coefAbsorb;
  const float sel = rsel * coefAbsorb;
  f3 sampleDir = F3(0, 0, 0), bxdf = F3(0, 0, 0);
  float PDF = 1.f, scale = 1.f;
  bool go = false;
  if (sel < coefTransmit && lumaT > 0.00001f) {
    scale = lumaT * rcpCoefSum;
    if (glossRefr > 0.f) {
      do { sampleDir = normalize(normalize(tDir) + uniformBall(rng, 2.f * glossRefr)); } while (dot(sampleDir, Y) > 0);
    } else sampleDir = tDir;
    bxdf = sampleTransmission;
    go = true;
  } else if (sel < coefReflection && lumaR > 0.00001f) {
    scale = lumaR * rcpCoefSum;
    if (glossRefl > 0.f) {
      do { sampleDir = normalize(normalize(rDir) + uniformBall(rng, 2.f * glossRefl)); } while (dot(sampleDir, Y) < 0);
    } else sampleDir = rDir;
    bxdf = sampleReflection;
    go = true;
  } else if (sel < coefDiffuse && lumaD > 0.00001f) {
    scale = lumaD * rcpCoefSum;
    if (h.front) {
      // SampleDiffuseBxDF(..., photonMap = true) (:203-224): Sampler::UniformHemisphere, PDF 1/2
      const float r1 = rng1(rng), r2 = rng1(rng);
      const float cosTheta = r1;
      const float sinTheta = qsqrt(1 - r1 * r1);
      const float phi = 2 * QA_PI * r2;
      sampleDir = toLocalFrame(N, F3(sinTheta * qcosf(phi), sinTheta * qsinf(phi), cosTheta));
      const f3 Ld = normalize(sampleDir);
      const f3 H = normalize(V + Ld);
      const float cosNH = qmax(0.f, dot(N, H));
      bxdf = sampleDiffuse + sampleSpecular * qpowf(cosNH, glossSpec);
      PDF = 0.5f;
      go = true;
    }
  }
  if (!go) return false;
  ray.p = h.p;
  ray.d = normalize(normalize(sampleDir));   // Normalize() in RandomPhotonBounce and again in the loop (renderer.cpp:183,253)
  power = (power * bxdf) / (PDF * scale);
  if (!h.front) power = power * F3(qexpf(-asF(m5.x) * h.z), qexpf(-asF(m5.y) * h.z), qexpf(-asF(m5.z) * h.z));
  return true;
}

// One lane = one iteration of the emission loop (renderer.cpp:146-197 photon map, :217-271
// caustics map); the photons it would store go to the lane's own slots, in order.
template <bool RES, bool TEX>
__global__ __launch_bounds__(QA_BLOCK) void qa_photon_emit(const DScene sc, const EmitParams ep)
{
  extern __shared__ uint4 s_dyn[];
  SceneMem<RES> mem;
  mem.img = s_dyn;
  if (RES) {
    for (uint32_t i = threadIdx.x; i < sc.residentVec4; i += QA_BLOCK) s_dyn[i] = sc.resident[i];
    __syncthreads();
  }
  uint32_t *stack = reinterpret_cast<uint32_t *>(s_dyn + (RES ? sc.residentVec4 : 0)) + threadIdx.x;
  const uint4 *mtlTable = RES ? s_dyn + sc.resMaterials : reinterpret_cast<const uint4 *>(sc.mtl);
  const uint32_t t = blockIdx.x * QA_BLOCK + threadIdx.x;
  if (t >= ep.count) return;
  TexTables tt;
  tt.blob = sc.blob;
  tt.texels = sc.texels;
  tt.texOff = sc.texOff;
  tt.texmap = sc.texmap;
  tt.tex = sc.tex;
  tt.filter = sc.texFilter;
  DCounters cnt = {0, 0, 0, 0, 0, 0};

  uint32_t rng = qa_photon_seed(ep.seed, ep.caustics ? QA_STREAM_CAUSTICS : QA_STREAM_PHOTON, ep.first + t);
  const float lightScale = 1.f / (float) ep.num_sources;
  int li = ep.sources[0];
  if (ep.num_sources > 1) {
    const float r = rng1(rng);
    const float n = (float) ep.num_sources;
    uint32_t id;
    if (!ep.caustics) {
      const float fl = floorf(r * n), lim = (float) (ep.num_sources - 1);
      id = (uint32_t) (fl < lim ? fl : lim);
    } else {  // the caustics loop rounds up (renderer.cpp:224)
      const uint32_t ce = (uint32_t) ceilf(r * n);
      id = ce < (uint32_t) (ep.num_sources - 1) ? ce : (uint32_t) (ep.num_sources - 1);
    }
    li = ep.sources[id];
  }
  const qa_light &l = sc.light[li];
  // PointLight::RandomPhoton: Sampler::UniformSphere (src/core/sampler.cpp:55-70)
  Ray ray;
  {
    float r1 = rng1(rng);
    const float r2 = rng1(rng);
    r1 = r1 * 2.f - 1.f;
    const float sinTheta = qsqrt(1 - r1 * r1);
    const float phi = 2 * QA_PI * r2;
    ray.p = ld3(l.position);
    ray.d = normalize(F3(sinTheta * qcosf(phi), sinTheta * qsinf(phi), r1));
  }
  f3 power = ld3(l.intensity) * lightScale;
  uint32_t *out = ep.cand + (size_t) t * ep.max_rec * 6;
  uint32_t stored = 0;
  bool fromDiffuse = false;
  for (uint32_t bounce = 0; bounce < ep.max_bounce;) {
    Hit h;
    h.z = QA_BIGFLOAT;
    h.node = -1;
    h.mtlID = 0;
    h.front = true;
    h.p = F3(0, 0, 0);
    h.N = F3(0, 0, 0);
    TexHit th;
    th.uvw = F3(0.5f, 0.5f, 0.5f);
    th.duvw0 = th.duvw1 = F3(0, 0, 0);
    th.hasTexture = false;
    RayDiff rd;
    rd.dx = rd.dy = ray.d;   // DiffRay(p, dir): x = y = c
    if (!traceClosest<RES, TEX, false>(mem, sc, ray, rd, h, th, stack, cnt)) break;
    const qa_instance &in = instAt<RES>(sc, h.node);
    if (in.mtlset < 0) break;
    const qa_mtlset ms = sc.mtlset[in.mtlset];
    // MultiMtl keeps Material's defaults: always a photon surface, never bounces (src/core/material.h:53-63)
    const bool blinn = !ms.multi;
    bool surface = true;
    if (blinn) {
      const uint4 m0 = mtlTable[6 * (size_t) ms.first];
      surface = luma(F3(asF(m0.x), asF(m0.y), asF(m0.z))) > 0;   // IsPhotonSurface: the plain diffuse colour
    }
    if (surface && bounce != 0 && !(ep.caustics && fromDiffuse) && stored < ep.max_rec) {
      uint32_t *rec = out + 6 * (size_t) stored++;
      float maxPower;
      uint32_t w4, w5;
      photonPack(power, ray.d, maxPower, w4, w5);
      rec[0] = __float_as_uint(h.p.x);
      rec[1] = __float_as_uint(h.p.y);
      rec[2] = __float_as_uint(h.p.z);
      rec[3] = __float_as_uint(maxPower);
      rec[4] = w4;
      rec[5] = w5;
    }
    if (!blinn || !photonBounce<TEX>(mtlTable, sc, tt, ms.first, ray, power, h, th, rng)) break;
    ++bounce;
    if (ep.caustics) fromDiffuse = fromDiffuse || surface;
  }
  ep.counts[t] = stored;
}

// Packs a batch: photons of emission e go to map[1 + offset(e) ...] while offset < size, where
// offset = candidates of all earlier emissions.  state[0] candidates so far, state[1] emitted rays
// (emissions that stored at least one photon), state[2] 1 + index of the emission that holds
// candidate number `size` (the one in which the reference's loop notices that the map is full).
__global__ __launch_bounds__(1024) void qa_photon_pack(const uint32_t *counts, const uint32_t *cand, uint32_t n, uint32_t max_rec,
                                                       uint32_t first, uint32_t *map, uint32_t size,
                                                       unsigned long long *state)
{
  __shared__ unsigned long long s_sum[1024];
  __shared__ unsigned int s_emitted;
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t b = threadIdx.x * per, e = (b + per < n) ? b + per : n;
  unsigned long long local = 0;
  for (uint32_t i = b; i < e; ++i) local += counts[i];
  s_sum[threadIdx.x] = local;
  if (threadIdx.x == 0) s_emitted = 0;
  __syncthreads();
  // exclusive scan of 1024 partial sums (Hillis-Steele in LDS)
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const unsigned long long v = threadIdx.x >= off ? s_sum[threadIdx.x - off] : 0;
    __syncthreads();
    s_sum[threadIdx.x] += v;
    __syncthreads();
  }
  const unsigned long long base = state[0];
  unsigned long long off = base + s_sum[threadIdx.x] - local;
  unsigned int emitted = 0;
  for (uint32_t i = b; i < e; ++i) {
    const uint32_t c = counts[i];
    if (c && off < size) {
      ++emitted;
      for (uint32_t j = 0; j < c && off + j < size; ++j) {
        const uint32_t *src = cand + ((size_t) i * max_rec + j) * 6;
        uint32_t *dst = map + (size_t) (1 + off + j) * 6;
        for (int k = 0; k < 6; ++k) dst[k] = src[k];
      }
    }
    if (off <= size && size < off + c) state[2] = (unsigned long long) first + i + 1;
    off += c;
  }
  if (emitted) atomicAdd(&s_emitted, emitted);
  __syncthreads();
  if (threadIdx.x == 1023) {
    state[0] = base + s_sum[1023];
    state[1] += s_emitted;
  }
}

template <bool RES, bool STATS>
static KernelFn PickPm(bool tex, bool area)
{
  if (area) return tex ? (KernelFn) qa_integrate<RES, true, true, true, STATS, true> : (KernelFn) qa_integrate<RES, true, false, true, STATS, true>;
  return tex ? (KernelFn) qa_integrate<RES, true, true, false, STATS, true> : (KernelFn) qa_integrate<RES, true, false, false, STATS, true>;
}
static KernelFn PickPmKernel(bool resident, bool tex, bool area, bool stats)
{
  if (resident) return stats ? PickPm<true, true>(tex, area) : PickPm<true, false>(tex, area);
  return stats ? PickPm<false, true>(tex, area) : PickPm<false, false>(tex, area);
}

}  // namespace qa

// -------------------------------------------------------------------------------------------------
// kd-tree balancing on the host: PhotonMap::PrepareForIrradianceEstimation / BalanceSegment
// (cyPhotonMap.h:272-372).  Left-balanced tree in heap order; the median of a segment is found with
// the reference's quick-select, whose swaps decide on which side equal keys end up.
// -------------------------------------------------------------------------------------------------
namespace {

struct Box3 { float lo[3], hi[3]; };

void BalanceSegment(std::vector<qa_photon> &work, std::vector<qa_photon> &tree, Box3 box, uint32_t index, uint32_t start, uint32_t end)
{
  const uint32_t n = end - start + 1;
  uint32_t median = 1;
  while (4 * median <= n) median += median;
  if (3 * median <= n) median = 2 * median + start - 1;
  else median = end - median + 1;

  int axis = 2;
  const float dx = box.hi[0] - box.lo[0], dy = box.hi[1] - box.lo[1], dz = box.hi[2] - box.lo[2];
  if (dx > dy) { if (dx > dz) axis = 0; }
  else if (dy > dz) axis = 1;

  uint32_t left = start, right = end;
  while (right > left) {
    const float pivot = work[right].pos[axis];
    uint32_t i = left - 1, j = right;
    for (;;) {
      while (work[++i].pos[axis] < pivot) {}
      while (work[--j].pos[axis] > pivot && j > left) {}
      if (i >= j) break;
      std::swap(work[i], work[j]);
    }
    std::swap(work[i], work[right]);
    if (i >= median) right = i - 1;
    if (i <= median) left = i + 1;
  }
  tree[index] = work[median];
  tree[index].plane_dirz = (uint8_t) ((tree[index].plane_dirz & 0x8) | axis);
  const float split = tree[index].pos[axis];
  if (median > start) {
    if (start < median - 1) {
      Box3 b = box;
      b.hi[axis] = split;
      BalanceSegment(work, tree, b, 2 * index, start, median - 1);
    } else tree[2 * index] = work[start];
  }
  if (median < end) {
    if (median + 1 < end) {
      Box3 b = box;
      b.lo[axis] = split;
      BalanceSegment(work, tree, b, 2 * index + 1, median + 1, end);
    } else tree[2 * index + 1] = work[end];
  }
}

// work: count + 1 records, [0] all zero like the reference's value-initialised dummy (it takes part in the box)
void Balance(std::vector<qa_photon> &work)
{
  const uint32_t count = (uint32_t) work.size() - 1;
  Box3 box;
  for (int a = 0; a < 3; ++a) box.lo[a] = box.hi[a] = work[0].pos[a];
  for (uint32_t i = 1; i <= count; ++i)
    for (int a = 0; a < 3; ++a) {
      if (box.lo[a] > work[i].pos[a]) box.lo[a] = work[i].pos[a];
      if (box.hi[a] < work[i].pos[a]) box.hi[a] = work[i].pos[a];
    }
  std::vector<qa_photon> tree(work.size());
  memset(tree.data(), 0, tree.size() * sizeof(qa_photon));
  BalanceSegment(work, tree, box, 1, 1, count);
  work.swap(tree);
}

// Photon::GetDirection (cyPhotonMap.h:233-254): x, y from the two shorts; z from x ALONE - the
// reference computes "dirX*dirX + dirY - dirY" - through a digit-by-digit integer square root.
void PhotonDirection(const qa_photon &ph, float out[3])
{
  out[0] = (float) ph.dirx / (float) 0x7FFF;
  out[1] = (float) ph.diry / (float) 0x7FFF;
  int xy2 = ph.dirx * ph.dirx + ph.diry - ph.diry;
  if (xy2 > 0x3FFF0001) xy2 = 0x3FFF0001;
  int root = 0, bit = 0x40000000, rem = 0x3FFF0001 - xy2;
  while (bit > rem) bit >>= 2;
  while (bit) {
    if (rem >= root + bit) {
      rem = rem - root - bit;
      root = root + (bit << 1);
    }
    root >>= 1;
    bit >>= 2;
  }
  out[2] = (float) root / (float) 0x7FFF;
  if (ph.plane_dirz & 0x8) out[2] = -out[2];
}

struct Scratch {   // device buffers of one build, released on every exit path
  std::vector<void *> ptrs;
  ~Scratch() { for (void *p : ptrs) (void) hipFree(p); }
  template <class T> hipError_t alloc(T **p, size_t bytes)
  {
    *p = nullptr;
    const hipError_t e = hipMalloc((void **) p, bytes);
    if (e == hipSuccess) ptrs.push_back(*p);
    return e;
  }
};

}  // namespace

void FreePhotonMaps(qa_ctx *c)
{
  for (int k = 0; k < 2; ++k) {
    if (c->dPhotons[k]) (void) hipFree(c->dPhotons[k]);
    c->dPhotons[k] = nullptr;
    for (int t = 0; t < 3; ++t) {
      if (c->dPmTables[k][t]) (void) hipFree(c->dPmTables[k][t]);
      c->dPmTables[k][t] = nullptr;
    }
    c->hostPhotons[k].clear();
    c->photonEmitted[k] = c->photonEmissions[k] = 0;
  }
  if (c->dHeap) (void) hipFree(c->dHeap);
  c->dHeap = nullptr;
  c->photonReady = false;
}

static int BuildOne(qa_ctx *c, int which, const qa_photon_map_params &mp, uint32_t seed, const int32_t *dSources, int numSources)
{
  typedef void (*EmitFn)(const DScene, const EmitParams);
  const EmitFn emit = c->resident ? (c->textured ? (EmitFn) qa_photon_emit<true, true> : (EmitFn) qa_photon_emit<true, false>)
                                  : (c->textured ? (EmitFn) qa_photon_emit<false, true> : (EmitFn) qa_photon_emit<false, false>);
  const uint32_t maxRec = mp.bounce > 1 ? mp.bounce - 1 : 1;   // a photon is stored at hits 1 .. bounce - 1
  const uint32_t batch = 32768;
  Scratch sc;
  uint32_t *dCounts, *dCand, *dMap;
  unsigned long long *dState;
  HIP_TRY(sc.alloc(&dCounts, batch * sizeof(uint32_t)));
  HIP_TRY(sc.alloc(&dCand, (size_t) batch * maxRec * sizeof(qa_photon)));
  HIP_TRY(sc.alloc(&dMap, ((size_t) mp.size + 1) * sizeof(qa_photon)));
  HIP_TRY(sc.alloc(&dState, 3 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(dMap, 0, ((size_t) mp.size + 1) * sizeof(qa_photon), c->stream));
  HIP_TRY(hipMemsetAsync(dState, 0, 3 * sizeof(unsigned long long), c->stream));
  const uint64_t cap = QA_PHOTON_MAX_EMISSIONS(mp.size);
  unsigned long long state[3] = {0, 0, 0};
  uint64_t first = 0;
  while (state[0] <= mp.size) {   // the loop ends with the first candidate that no longer fits
    if (first >= cap)
      return Fail(QA_EUNSUPPORTED, std::string(which ? "caustics" : "photon") + " map not full after " + std::to_string(first) +
                                       " emissions: no surface of this scene can store such a photon (the reference would loop forever)");
    EmitParams ep;
    ep.first = (uint32_t) first;
    ep.count = batch;
    ep.max_bounce = mp.bounce;
    ep.caustics = which;
    ep.seed = seed;
    ep.num_sources = numSources;
    ep.sources = dSources;
    ep.max_rec = maxRec;
    ep.counts = dCounts;
    ep.cand = dCand;
    hipLaunchKernelGGL(emit, dim3(batch / QA_BLOCK), dim3(QA_BLOCK), (unsigned) c->ldsBytes, c->stream, c->ds, ep);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(qa_photon_pack, dim3(1), dim3(1024), 0, c->stream, dCounts, dCand, batch, maxRec, (uint32_t) first, dMap,
                       mp.size, dState);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(state, dState, sizeof(state), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    first += batch;
  }
  std::vector<qa_photon> &host = c->hostPhotons[which];
  try { host.resize((size_t) mp.size + 1); } catch (const std::bad_alloc &) { return Fail(QA_ENOMEM, "out of memory"); }
  HIP_TRY(hipMemcpy(host.data(), dMap, host.size() * sizeof(qa_photon), hipMemcpyDeviceToHost));
  // ScalePhotonPowers(1.f / numOfEmittedRays) (cyPhotonMap.h:128-132), then the kd-tree
  const float scale = 1.f / (float) (uint32_t) state[1];
  for (uint32_t i = 1; i <= mp.size; ++i) host[i].power *= scale;
  try { Balance(host); } catch (const std::bad_alloc &) { return Fail(QA_ENOMEM, "out of memory"); }
  HIP_TRY(hipMalloc(&c->dPhotons[which], host.size() * sizeof(qa_photon)));
  HIP_TRY(hipMemcpy(c->dPhotons[which], host.data(), host.size() * sizeof(qa_photon), hipMemcpyHostToDevice));
  // the gather's tables (DPhotonMap): what Photon::GetPlane / GetDirection / GetMaxPower / GetPower return
  {
    std::vector<float> tab;
    try { tab.resize(host.size() * 4); } catch (const std::bad_alloc &) { return Fail(QA_ENOMEM, "out of memory"); }
    for (int t = 0; t < 3; ++t) {
      for (size_t i = 0; i < host.size(); ++i) {
        const qa_photon &ph = host[i];
        float *o = &tab[4 * i];
        if (t == 0) {
          const uint32_t axis = ph.plane_dirz & 0x3u;
          o[0] = ph.pos[0]; o[1] = ph.pos[1]; o[2] = ph.pos[2];
          memcpy(&o[3], &axis, 4);
        } else if (t == 1) {
          PhotonDirection(ph, o);
          o[3] = ph.power;
        } else {
          o[0] = (ph.rgb[0] / 255.0f) * ph.power;   // ToColor(color) * power, src/math/math.h:119-123
          o[1] = (ph.rgb[1] / 255.0f) * ph.power;
          o[2] = (ph.rgb[2] / 255.0f) * ph.power;
          o[3] = 0.f;
        }
      }
      HIP_TRY(hipMalloc(&c->dPmTables[which][t], tab.size() * sizeof(float)));
      HIP_TRY(hipMemcpy(c->dPmTables[which][t], tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  c->photonEmitted[which] = state[1];
  c->photonEmissions[which] = state[2];
  return QA_OK;
}

extern "C" {

int qa_photon_maps_clear(qa_ctx *c)
{
  if (!c) return Fail(QA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  FreePhotonMaps(c);
  return QA_OK;
}

int qa_photon_maps_build(qa_ctx *c, const qa_photon_params *pp, uint32_t seed)
{
  if (!c || !pp) return Fail(QA_EINVAL, "null argument");
  if (!c->haveScene) return Fail(QA_ENOSCENE, "no scene uploaded");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  FreePhotonMaps(c);
  const qa_photon_map_params *mps[2] = {&pp->photon, &pp->caustics};
  for (const qa_photon_map_params *mp : mps) {
    if (mp->size < 1 || mp->size > (1u << 28) || mp->bounce < 1 || mp->bounce > 4096 || !(mp->radius > 0))
      return Fail(QA_EINVAL, "photon map: size in [1, 2^28], bounce in [1, 4096] and radius > 0 required");
  }
  // photon sources: PointLight only (Light::IsPhotonSource, src/lights/lights.h:114,156)
  const qa_flat_header *h = reinterpret_cast<const qa_flat_header *>(c->hostBlob.data());
  const qa_light *light = QA_BLOB_PTR(qa_light, c->hostBlob.data(), h->off_lights);
  std::vector<int32_t> sources;
  for (uint32_t i = 0; i < h->num_lights; ++i)
    if (light[i].type == QA_LIGHT_POINT) sources.push_back((int32_t) i);
  if (sources.empty())
    return Fail(QA_EUNSUPPORTED, "photon map: the scene has no point light (the reference divides by zero photon sources)");
  // the gather walks the kd-tree on the lane's LDS stack: two words per tree level + 1
  uint32_t levels = 1;
  for (uint32_t n = std::max(pp->photon.size, pp->caustics.size); n > 1; n >>= 1) ++levels;
  const uint32_t needDepth = std::max(c->stackDepth, 2 * (levels + 2));
  const size_t stackBytes = ((size_t) needDepth + QA_LANE_SLOTS) * QA_BLOCK * sizeof(uint32_t);
  const size_t imageBytes = c->resident ? (size_t) c->ds.residentVec4 * sizeof(uint4) : 0;
  if (imageBytes + stackBytes > 64 * 1024) return Fail(QA_EUNSUPPORTED, "photon map too deep for the LDS stack");

  Scratch tmp;
  int32_t *dSources;
  HIP_TRY(tmp.alloc(&dSources, sources.size() * sizeof(int32_t)));
  HIP_TRY(hipMemcpy(dSources, sources.data(), sources.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  for (int which = 0; which < 2; ++which) {
    const int rc = BuildOne(c, which, *mps[which], seed, dSources, (int) sources.size());
    if (rc != QA_OK) { FreePhotonMaps(c); return rc; }
  }
  c->photonParams = *pp;
  c->stackDepthPm = needDepth;
  c->ldsBytesPm = imageBytes + stackBytes;
  c->kernelPm = PickPmKernel(c->resident, c->textured, c->area, false);
  c->kernelPmStats = PickPmKernel(c->resident, c->textured, c->area, true);
  int resident = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, (const void *) c->kernelPm, QA_BLOCK, c->ldsBytesPm) != hipSuccess || resident < 1)
    resident = 1;
  c->blocksPerCUPm = resident > 8 ? 8 : resident;
  // nearest-photon heaps: QA_PHOTON_GATHER + 1 slots per thread of the largest grid
  const size_t threads = (size_t) c->numCUs * 8 * QA_BLOCK;
  const hipError_t e = hipMalloc(&c->dHeap, threads * (QA_PHOTON_GATHER + 1) * sizeof(uint2));
  if (e != hipSuccess) {
    FreePhotonMaps(c);
    return Fail(QA_EHIP, std::string("photon heaps: ") + hipGetErrorString(e));
  }
  c->photonReady = true;
  return QA_OK;
}

int qa_photon_maps_info(qa_ctx *c, uint64_t emitted[2], uint64_t emissions[2])
{
  if (!c || !emitted || !emissions) return Fail(QA_EINVAL, "null argument");
  if (!c->photonReady) return Fail(QA_ENOSCENE, "no photon maps built");
  for (int k = 0; k < 2; ++k) { emitted[k] = c->photonEmitted[k]; emissions[k] = c->photonEmissions[k]; }
  return QA_OK;
}

int qa_photon_maps_download(qa_ctx *c, int which, qa_photon *out, uint64_t capacity)
{
  if (!c || !out || which < 0 || which > 1) return Fail(QA_EINVAL, "bad argument");
  if (!c->photonReady) return Fail(QA_ENOSCENE, "no photon maps built");
  const std::vector<qa_photon> &host = c->hostPhotons[which];
  if (capacity < host.size()) return Fail(QA_EINVAL, "output smaller than size + 1 records");
  // read back from the device: what the render kernels actually gather from
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy(out, c->dPhotons[which], host.size() * sizeof(qa_photon), hipMemcpyDeviceToHost));
  return QA_OK;
}

}  // extern "C"
