// qa_wf.h — the STAGED integrator for scenes whose geometry does not fit LDS (BASELINE C3 / C4 / C5).
//
// The megakernel of qa_kernel.h keeps a whole path in one lane; on big meshes a wave then enters a
// BVH walk with 13 - 34 % of its lanes (most rays miss a given mesh's bounds) and waits for the
// slowest of them.  Here the same per-pixel Monte-Carlo loop (reference: Renderer::PixelRender
// src/renderers/renderer.cpp:302-366, MtlBlinn_PhotonMap::Shade src/materials/MtlBlinn_PhotonMap.cpp:
// 256-500, Scene::TraceNodeNormal / TraceNodeShadow src/scene/scene.cpp:35-74, TriObj::TraceBVHNode
// src/objects/objects.cpp:324-420) is cut into stages that exchange work through queues in HBM:
//
//   wf_logic   one lane per pixel slot.  Finishes the direct light of the previous hit (shadow results),
//              shades the closest hit that came back (BSDF lobe selection / sampling, Russian roulette),
//              accumulates finished samples (SuperSamplerHalton, src/scene/scene.cpp:83-123), starts camera
//              rays, and appends every new ray - the next segment and one shadow ray per light - to the ray
//              queues.  All random numbers of a pixel are drawn here, in the reference's order.
//   wf_cull    one lane per queued ray: the scene-graph loop of TraceNodeNormal / TraceNodeShadow.  Spheres
//              and planes are intersected on the spot; for every mesh whose bounds the ray enters a JOB
//              (node-local ray, distance limit, ray id) goes to the job queue.
//   wf_trace   persistent waves pull jobs (one BVH walk each, all of the same shape) and refill finished
//              lanes from the queue with one atomic per wave (ballot + mbcnt), so a wave walks with all of its
//              lanes.  Results meet in a 64-bit atomicMin per ray (distance, instance, triangle).
//   wf_redo    rays whose answer could depend on the ORDER in which the reference visits instances and
//              triangles (ties, hits in front of their own leaf box) are repeated exactly as the reference
//              walks them, one lane per ray.  Rare.
//
// One live path per pixel (its xorshift32 stream is sequential), every pixel of the region in flight at once:
// a stage works on 10^5..10^7 items.  Per-slot state is SoA in HBM (16-byte columns, coalesced).
#pragma once
#include "qa_kernel.h"
#include "qa_wf_types.h"

namespace qa {

__device__ __forceinline__ uint32_t wfInfo(uint32_t sidx, uint32_t bounce, bool fromDiff, bool primary, uint32_t phase, uint32_t pend)
{
  return (sidx & 0xFFFFu) | ((bounce & 0xFu) << 16) | ((fromDiff ? 1u : 0u) << 20) | ((primary ? 1u : 0u) << 21) | (phase << 22) | (pend << 24);
}

// Decode a slot: tile-major (a wave of wf_logic = one 8x8 pixel tile, like the megakernel's work items).
struct WfPixel { int px, py; unsigned q; bool valid; };
__device__ __forceinline__ WfPixel wfPixel(const RenderParams &rp, const WfBuf &b, unsigned slot)
{
  const int rw = rp.x1 - rp.x0, rh = rp.y1 - rp.y0;
  const unsigned tilesX = (unsigned) (rw + 7) / 8;
  const unsigned in = slot % 64, tile = (slot / 64) * b.groupCount + b.groupIndex;   // the group's tiles are interleaved with the others'
  const unsigned otr = tile / tilesX;
  const unsigned tx = (tile % tilesX) * 8 + (in % 8);
  const unsigned ty = ((unsigned) rp.tile_row0 + otr * (unsigned) rp.tile_row_step) * 8 + (in / 8);
  WfPixel o;
  o.valid = tx < (unsigned) rw && ty < (unsigned) rh && otr < (unsigned) rp.own_tile_rows;
  o.px = rp.x0 + (int) tx;
  o.py = rp.y0 + (int) ty;
  o.q = (otr * 8 + (in / 8)) * (unsigned) rw + tx;
  return o;
}

__device__ __forceinline__ unsigned long long wfKey(float z, int k, uint32_t tri)
{
  return ((unsigned long long) __float_as_uint(z) << 32) | ((unsigned long long) (uint32_t) k << 24) | (unsigned long long) (tri & 0xFFFFFFu);
}

// ---------------------------------------------------------------------------------------------
// wf_init: pixel streams (include/qa_seed.h), empty accumulators
// ---------------------------------------------------------------------------------------------
__global__ void wf_init(const DScene sc, const RenderParams rp, WfBuf b)
{
  const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= b.n) return;
  const WfPixel px = wfPixel(rp, b, slot);
  uint32_t rng = 1, info = wfInfo(0, 0, false, true, WF_PH_DONE, 0);
  if (px.valid) {
    rng = qa_pixel_seed(rp.seed, (uint32_t) px.py * (uint32_t) sc.cam.width + (uint32_t) px.px);
    info = wfInfo(0, 0, false, true, WF_PH_SAMPLE, 0);
  }
  b.P[slot] = make_float4(0, 0, 0, __uint_as_float(rng));
  b.D[slot] = make_float4(0, 0, 1, __uint_as_float(info));
  b.T[slot] = make_float4(0, 0, 0, __uint_as_float(0xFFFFFFFFu));
  b.L[slot] = make_float4(0, 0, 0, 0);
  b.mean[slot] = make_float4(0, 0, 0, 0);
  b.cstd[slot] = make_float4(0, 0, 0, 0);
  b.out[slot] = 0;
  b.redoFlag[slot] = 0;
  if (slot < 2) b.contCount[slot] = 0;
}

// TriObj::IntersectRay's gate: the ray against the mesh bounds (Box::IntersectRay, src/core/box.cpp:94-128)
__device__ __forceinline__ void wfMeshGate(const DMesh &m, const Ray &r, float &entry, float &exit_)
{
  const f3 drcp = F3(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
  // per-lane choice: both forms return the same values where the fast one applies (boxEntryExitFast)
  if (qabs(r.d.x) < 1e-7f || qabs(r.d.y) < 1e-7f || qabs(r.d.z) < 1e-7f) boxEntryExit(r, drcp, ld3(m.bmin), ld3(m.bmax), entry, exit_);
  else boxEntryExitFast(r, drcp, ld3(m.bmin), ld3(m.bmax), entry, exit_);
}

// ---------------------------------------------------------------------------------------------
// Details of a known closest hit (instance k, triangle tri, distance z): what the intersector that won
// wrote into HitInfo, recomputed with the very expressions of hitSphere / hitPlane / hitMesh, then
// Node::FromNodeCoords up to the root (the tail of traceClosest).
// ---------------------------------------------------------------------------------------------
//
// Returns the ORDER CHECK of a mesh hit (see wf_cull): would the reference's sequential walk, whose running
// distance is never larger than a job's, have reached this triangle too?  Yes if the triangle's leaf box passes
// the reference's strict test at the found distance (refReaches: every box above it then passes as well, and so
// does the mesh gate, whose box is the root's).  Otherwise the ray has to be repeated exactly (wf_redo).
template <bool TEX>
__device__ __forceinline__ bool wfHitDetails(const DScene &sc, int k, uint32_t tri, float z, const Ray &world, const RayDiff &wd,
                                             Hit &h, TexHit &th)
{
  bool orderOk = true;
  Ray r;
  RayDiff rd;
  rd.dx = rd.dy = F3(0, 0, 1);
  if (TEX) localRayDiff<false>(sc, k, world, wd, r, rd);
  else r = localRay<false>(sc, k, rootRay<false>(sc, world));
  const qa_instance &in = sc.inst[k];
  h.z = QA_BIGFLOAT;
  h.node = -1;
  h.mtlID = 0;
  h.front = true;
  h.p = F3(0, 0, 0);
  h.N = F3(0, 0, 0);
  if (in.obj_type == QA_OBJ_SPHERE) {
    hitSphere(r, h, k, true);
    if (TEX) texSphere(r.p, rd.dx, rd.dy, h.p, h.N, th);
  } else if (in.obj_type == QA_OBJ_PLANE) {
    hitPlane(r, h, k, true);
    if (TEX) texPlane(r.p, rd.dx, rd.dy, h.p, th);
  } else {
    const DMesh &m = sc.mesh[in.mesh];
    const uint4 *t = reinterpret_cast<const uint4 *>(m.tris) + 3 * (size_t) tri;
    const uint4 q0 = ldGlobal(t), q1 = ldGlobal(t + 1), q2 = ldGlobal(t + 2);
    float ba = 0, bb = 0;
    h.z = z;
    triangleDetails(q0, q1, q2, r, h, ba, bb);
    const uint4 *s = reinterpret_cast<const uint4 *>(m.shade) + 3 * (size_t) tri;
    const uint4 s0 = ldGlobal(s), s1 = ldGlobal(s + 1), s2 = ldGlobal(s + 2);
    const float bc = 1.f - ba - bb;
    const f3 n0 = F3(asF(s0.x), asF(s0.y), asF(s0.z)), n1 = F3(asF(s0.w), asF(s1.x), asF(s1.y)), n2 = F3(asF(s1.z), asF(s1.w), asF(s2.x));
    h.N = (n0 * ba + n1 * bb) + n2 * bc;
    h.mtlID = (int) s2.y;
    h.node = k;
    if (TEX && m.hasVT) texTriangle(q0, q1, q2, m.vt + 6 * (size_t) tri, r.p, rd.dx, rd.dy, ba, bb, th);
    {
      const uint32_t leaf = s2.w;   // DTriShade::pad: the reference-tree leaf that holds this element
      const f3 drcp = F3(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
      const bool nearZero = qabs(r.d.x) < 1e-7f || qabs(r.d.y) < 1e-7f || qabs(r.d.z) < 1e-7f;
      orderOk = refReaches<true>(reinterpret_cast<const uint4 *>(m.nodes), leaf, r, drcp, !nearZero, z);
      if (leaf <= 1 || !m.gateIsRoot) {
        float gEntry, gExit;
        wfMeshGate(m, r, gEntry, gExit);
        orderOk = orderOk && !(gEntry > z);
      }
    }
  }
  h.z = z;
  h.node = k;
  for (int a = k; a >= 0; a = sc.inst[a].parent) {
    if (a == 0 && sc.rootIdentity) {
      h.N = normalize(h.N);
      break;
    }
    const qa_instance &ia = sc.inst[a];
    h.p = mulMV(ia.tm, h.p) + ld3(ia.pos);
    h.N = normalize(mulTMV(ia.itm, h.N));
  }
  return orderOk;
}

// Light j seen from p, everything but the shadow test (illuminate + directLight of qa_kernel.h with the
// shadow factor 1, which multiplies exactly): the shadow ray and the contribution it gates.
__device__ __forceinline__ void wfLight(const DScene &sc, const qa_light &l, f3 p, f3 N, f3 V, f3 kd, f3 ks, float gloss,
                                        float4 &sh, f3 &contrib)
{
  const float normCoefDI = 1.f / (float) sc.num_lights;
  f3 I;
  f3 dirN;
  float tmax;
  if (l.type == QA_LIGHT_DIRECT) {
    dirN = normalize(-ld3(l.direction));
    tmax = QA_BIGFLOAT;
    I = ld3(l.intensity) * 1.0f;
  } else {
    const f3 dir = ld3(l.position) - p;
    dirN = normalize(dir);
    tmax = length(dir);
    I = (ld3(l.intensity) * 1.0f) * inverseSquareFalloff(dir);
    if (l.type == QA_LIGHT_SPOT) I = I * spotAttenuation(l, p);
  }
  sh = make_float4(dirN.x, dirN.y, dirN.z, tmax);
  const f3 intensity = I * normCoefDI;
  const f3 Ld = normalize(-lightDirection(l, p));
  const f3 H = normalize(V + Ld);
  const float cosNL = qmax(0.f, dot(N, Ld));
  const float cosNH = qmax(0.f, dot(N, H));
  const f3 brdf = kd + ks * qpowf(cosNH, gloss);
  contrib = (intensity * cosNL) * brdf;
}

// ---------------------------------------------------------------------------------------------
// Per-wave append buffers in LDS: a wave collects its queue entries (prefix by ballot + mbcnt, no atomics)
// and reserves space in the global queue with ONE atomic per flush.  One returning atomic per wave and
// tile on a single counter is what bounded the first version of these stages (~90 per microsecond).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wfWaveSync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CAP>
struct WaveQueue32 {
  uint32_t *buf;        // LDS, CAP entries, this wave's
  uint32_t fill;        // wave-uniform
  __device__ __forceinline__ void flush(uint32_t *gq, uint32_t *gcount)
  {
    if (!fill) return;
    wfWaveSync();
    unsigned base = 0;
    if (__lane_id() == 0) base = atomicAdd(gcount, fill);
    base = __shfl(base, 0);
    for (uint32_t i = __lane_id(); i < fill; i += 64) gq[base + i] = buf[i];
    wfWaveSync();
    fill = 0;
  }
  __device__ __forceinline__ void push(bool pred, uint32_t value, uint32_t *gq, uint32_t *gcount)
  {
    const unsigned long long m = __ballot(pred);
    if (!m) return;
    const uint32_t n = (uint32_t) __popcll(m);
    if (fill + n > (uint32_t) CAP) flush(gq, gcount);
    if (pred) buf[fill + __popcll(m & ((1ull << __lane_id()) - 1ull))] = value;
    fill += n;
  }
};

// ---------------------------------------------------------------------------------------------
// wf_logic
// ---------------------------------------------------------------------------------------------
#define QA_WF_RAYQ_CAP 512
struct WfEmit { bool closest; uint32_t shadow; uint32_t redo; bool newSample, pixelDone, live; };

template <bool TEX>
__device__ __forceinline__ WfEmit wfLogicSlot(const DScene &sc, const RenderParams &rp, const WfBuf &b, unsigned slot, bool inRange)
{
  float4 P4 = make_float4(0, 0, 0, 0), D4 = P4;
  if (inRange) D4 = b.D[slot];
  uint32_t info = inRange ? __float_as_uint(D4.w) : wfInfo(0, 0, false, true, WF_PH_DONE, 0);
  uint32_t phase = WF_INFO_PHASE(info);
  const bool live = phase != WF_PH_DONE;
  // a slot whose rays are still being walked (suspended jobs) sits this pass out; one whose answer needs the
  // exact repeat queues it (wf_redo runs at the end of the pass) and waits one pass more
  bool ready = live, exactKey = false;
  uint32_t redoBits = 0;
  if (live) {
    if (b.out[slot] != 0) ready = false;
    else {
      const uint32_t flags = b.redoFlag[slot];
      redoBits = flags & 0x1Fu;
      if (redoBits) { ready = false; b.redoFlag[slot] = 0; }
      else if (flags & QA_WF_EXACT) { exactKey = true; b.redoFlag[slot] = 0; }   // wf_redo's answer: no order check
    }
  }
  bool emitClosest = false;
  uint32_t emitShadow = 0;       // bit j: shadow ray towards light j
  bool newSample = false, pixelDone = false;
  if (ready) {
    P4 = b.P[slot];
    float4 T4 = b.T[slot], L4 = b.L[slot];
    uint32_t rng = __float_as_uint(P4.w);
    uint32_t sidx = WF_INFO_SIDX(info), bounce = WF_INFO_BOUNCE(info), pend = WF_INFO_PEND(info);
    bool fromDiffuse = WF_INFO_FROMDIFF(info), primary = WF_INFO_PRIMARY(info);
    int absorbMtl = (int) __float_as_uint(T4.w);
    f3 T = F3(T4.x, T4.y, T4.z), L = F3(L4.x, L4.y, L4.z);
    Ray ray;
    ray.p = F3(P4.x, P4.y, P4.z);
    ray.d = F3(D4.x, D4.y, D4.z);
    const WfPixel px = wfPixel(rp, b, slot);
    TexTables tt;
    tt.blob = sc.blob;
    tt.texels = sc.texels;
    tt.texOff = sc.texOff;
    tt.texmap = sc.texmap;
    tt.tex = sc.tex;
    tt.filter = sc.texFilter;
    const uint4 *mtlTable = reinterpret_cast<const uint4 *>(sc.mtl);

    // ---- 1. direct light of the previous hit: the shadow rays have come back (directLight's sum, lights in order)
    if (pend) {
      const uint32_t vis = b.vis[slot];
      const float4 Tp4 = b.Tp[slot];
      f3 sum = F3(0, 0, 0);
      for (uint32_t j = 0; j < b.numLights; ++j)
        if ((pend >> j) & (vis >> j) & 1u) {
          const float4 c = b.C[(size_t) j * b.n + slot];
          sum = sum + F3(c.x, c.y, c.z);
        }
      L = L + F3(Tp4.x, Tp4.y, Tp4.z) * sum;
      pend = 0;
    }

    bool bail = false;
    bool done = (phase == WF_PH_LIGHTS);
    if (phase == WF_PH_TRACED) {
      // ---- 2. the closest-hit query has come back
      const unsigned long long key = b.key[slot];
      const bool found = key != ~0ull;
      const float z = __uint_as_float((uint32_t) (key >> 32));
      if (primary && sidx == 0) rp.depth[px.q] = found ? z : QA_BIGFLOAT;
      RayDiff pathDiff;
      pathDiff.dx = pathDiff.dy = ray.d;
      f3 texpos = F3(0, 0, 0);
      if (TEX && primary) {
        const float hx = sc.halton[2 * sidx], hy = sc.halton[2 * sidx + 1];
        texpos = F3(hx, hy, 0.f) + F3((float) px.px, (float) px.py, 0.f);
        const f3 A = ld3(sc.cam.screenA), U = ld3(sc.cam.screenU), V = ld3(sc.cam.screenV);
        const f3 xpt = (A + U * (texpos.x + QA_DX)) + V * texpos.y;
        const f3 ypt = (A + U * texpos.x) + V * (texpos.y + QA_DX);
        pathDiff.dx = normalize(xpt - ray.p);
        pathDiff.dy = normalize(ypt - ray.p);
      }
      if (!found) {
        f3 c = primary ? ld3(sc.background) : ld3(sc.environment);
        if (TEX) {
          if (primary) c = texColorSample(tt, c, sc.bgTexmap, F3(texpos.x / (float) sc.cam.width, texpos.y / (float) sc.cam.height, 0.f));
          else c = sampleEnvironment(tt, c, sc.envTexmap, ray.d);
        }
        L = L + T * c;
        done = true;
      } else {
        const int hk = (int) ((key >> 24) & 0xFFu);
        const uint32_t htri = (uint32_t) (key & 0xFFFFFFu);
        Hit h;
        TexHit th;
        th.uvw = F3(0.5f, 0.5f, 0.5f);
        th.duvw0 = th.duvw1 = F3(0, 0, 0);
        th.hasTexture = false;
        // a hit that fails the order check changes nothing: the slot keeps its state until wf_redo has answered
        bail = !wfHitDetails<TEX>(sc, hk, htri, z, ray, pathDiff, h, th) && !exactKey;
        if (!bail) {
        if (!primary && !h.front && absorbMtl >= 0) {
          const uint4 ab = mtlTable[6 * (size_t) absorbMtl + 5];
          const f3 att = F3(qexpf(-asF(ab.x) * h.z), qexpf(-asF(ab.y) * h.z), qexpf(-asF(ab.z) * h.z));
          T = T * att;
        }
        const qa_instance &in = sc.inst[h.node];
        int mi = -1;
        bool white = false;
        if (in.mtlset >= 0) {
          const qa_mtlset ms = sc.mtlset[in.mtlset];
          if (ms.multi) {
            if (h.mtlID >= 0 && h.mtlID < ms.count) mi = ms.first + h.mtlID;
            else white = true;
          } else mi = ms.first;
        }
        if (mi < 0) {
          if (white) L = L + T;
          done = true;
        } else {
          const f3 V = -ray.d;
          const f3 N = h.N;
          const f3 p = h.p;
          const Surface sf = shadeSurface<TEX>(mtlTable, sc, tt, mi, N, V, h.front, th, (int) bounce, fromDiffuse, rng);
          L = L + T * sf.emission;
          // direct lighting (:481-498): the shadow rays go to the queue, the sum is taken when they are back
          if (b.numLights) {
            for (uint32_t j = 0; j < b.numLights; ++j) {
              float4 sh;
              f3 contrib;
              wfLight(sc, ldTable(sc.light + b.lightIdx[j]), p, N, V, sf.kd, sf.ks, sf.gloss, sh, contrib);
              b.SH[(size_t) j * b.n + slot] = sh;
              b.C[(size_t) j * b.n + slot] = make_float4(contrib.x, contrib.y, contrib.z, 0.f);
            }
            pend = (1u << b.numLights) - 1u;
            emitShadow = pend;
            b.vis[slot] = pend;
            b.Tp[slot] = make_float4(T.x, T.y, T.z, 0.f);
          }
          ray.p = p;   // origin of the shadow rays and of the secondary ray alike
          if (sf.spawn) {
            ray.d = normalize(sf.nextDir);
            T = T * sf.bxdf;
            absorbMtl = mi;
            bounce -= 1;
            fromDiffuse = sf.nextFromDiffuse;
            primary = false;
            emitClosest = true;
          } else if (pend) {
            phase = WF_PH_LIGHTS;
          } else {
            done = true;
          }
        }
        }
      }
    }
    if (bail) redoBits = 1u;
    else {

    // ---- 3. sample finished: SuperSamplerHalton::Accumulate / Loop (scene.cpp:92-121)
    if (done) {
      const float4 m4 = b.mean[slot], c4 = b.cstd[slot];
      const float inv = (float) (sidx + 1);
      f3 mean = F3(m4.x, m4.y, m4.z), cstd = F3(c4.x, c4.y, c4.z);
      const f3 dc = (L - mean) / inv;
      mean = mean + dc;
      if (sidx > 0) cstd = cstd + ((dc * dc) * inv - cstd / (float) sidx);
      b.mean[slot] = make_float4(mean.x, mean.y, mean.z, 0.f);
      b.cstd[slot] = make_float4(cstd.x, cstd.y, cstd.z, 0.f);
      ++sidx;
      const bool more = (int) sidx < rp.spp_min || ((int) sidx < rp.spp_max && (cstd.x > 0.005f || cstd.y > 0.001f || cstd.z > 0.005f));
      if (more) phase = WF_PH_SAMPLE;
      else {
        rp.rgb[3 * px.q + 0] = mean.x;
        rp.rgb[3 * px.q + 1] = mean.y;
        rp.rgb[3 * px.q + 2] = mean.z;
        rp.ns[px.q] = sidx;
        phase = WF_PH_DONE;
        pixelDone = true;
      }
    }

    // ---- 4. camera ray (src/renderers/renderer.cpp:312-328).  New samples start only in passes whose gate is open
    // (every few passes, qa_wf.hip): the camera rays of neighbouring pixels then travel and are walked together
    // instead of trickling into every pass next to incoherent bounce rays, once the pixels' paths have drifted apart.
    if (phase == WF_PH_SAMPLE && b.gateOpen) {
      const float hx = sc.halton[2 * sidx], hy = sc.halton[2 * sidx + 1];
      const f3 texpos = F3(hx, hy, 0.f) + F3((float) px.px, (float) px.py, 0.f);
      const f3 A = ld3(sc.cam.screenA), U = ld3(sc.cam.screenU), V = ld3(sc.cam.screenV);
      const f3 cpt = (A + U * texpos.x) + V * texpos.y;
      f3 campos = ld3(sc.cam.pos);
      if (sc.cam.dof > 0.1f) {
        const float r1 = rng1(rng), r2 = rng1(rng);
        const float r = sc.cam.dof * qsqrt(r1);
        const float t = r2 * 2.f * QA_PI;
        campos = campos + (ld3(sc.cam.screenX) * (r * qcosf(t)) + ld3(sc.cam.screenY) * (r * qsinf(t)));
      }
      ray.p = campos;
      ray.d = normalize(cpt - campos);
      T = F3(1, 1, 1);
      L = F3(0, 0, 0);
      absorbMtl = -1;
      bounce = (uint32_t) rp.max_bounce;
      fromDiffuse = false;
      primary = true;
      phase = WF_PH_TRACED;
      emitClosest = true;
      newSample = true;
    } else if (emitClosest) {
      phase = WF_PH_TRACED;
    }

    info = wfInfo(sidx, bounce, fromDiffuse, primary, phase, pend);
    b.P[slot] = make_float4(ray.p.x, ray.p.y, ray.p.z, __uint_as_float(rng));
    b.D[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(info));
    b.T[slot] = make_float4(T.x, T.y, T.z, __uint_as_float((uint32_t) absorbMtl));
    b.L[slot] = make_float4(L.x, L.y, L.z, 0.f);
    if (emitClosest) b.key[slot] = ~0ull;
    }
  }

  WfEmit e;
  e.closest = emitClosest;
  e.shadow = emitShadow;
  e.redo = redoBits;
  e.newSample = newSample;
  e.pixelDone = pixelDone;
  e.live = live;
  return e;
}

// waves per SIMD the allocator must leave room for: four without textures (+5 - 8 % on the tower / glossy scenes against
// two); the textured variant spills 167 registers at that budget and is 2 % faster at two
#ifndef QA_WF_LOGIC_WAVES
#define QA_WF_LOGIC_WAVES 4
#endif
template <bool TEX>
__global__ __launch_bounds__(QA_BLOCK, TEX ? 2 : QA_WF_LOGIC_WAVES) void wf_logic(const DScene sc, const RenderParams rp, WfBuf b, WfCounters *ctr, DCounters *frame,
                                                     uint32_t parity)
{
  __shared__ uint32_t s_q[QA_BLOCK / 64][2][QA_WF_RAYQ_CAP];
  const unsigned lane = __lane_id(), wave = threadIdx.x / 64;
  if (blockIdx.x == 0 && threadIdx.x == 0) b.contCount[(parity & 1u) ^ 1u] = 0;   // wf_trace of this pass fills it
  WaveQueue32<QA_WF_RAYQ_CAP> qC, qS;
  qC.buf = s_q[wave][0];
  qC.fill = 0;
  qS.buf = s_q[wave][1];
  qS.fill = 0;
  unsigned nLive = 0, nNew = 0, nClosest = 0, nPix = 0, nShadow = 0;   // wave-uniform tallies
  // Only some of a tile's pixels have work in a given pass (their rays are back, their gate is open) and what they do
  // differs by an order of magnitude (shading a hit against adding a light term or starting a camera ray): the wave sorts
  // the slots of its tiles into two LDS lists, "shade" and "other", and runs the slot logic on 64 of one kind at a time
  // (QA_WF_LOGIC_COMPACT=0: one tile at a time, lane = slot).  The slot logic is the same either way, and a slot's
  // result does not depend on its neighbours: same bits.
#ifndef QA_WF_LOGIC_COMPACT
#define QA_WF_LOGIC_COMPACT 1
#endif
  const unsigned tiles = b.n / 64, wavesTotal = gridDim.x * (QA_BLOCK / 64);
#if QA_WF_LOGIC_COMPACT
  __shared__ uint32_t s_list[QA_BLOCK / 64][2][128];
  unsigned nList[2] = {0, 0};      // wave-uniform
#endif
  for (unsigned tile = blockIdx.x * (QA_BLOCK / 64) + wave;; tile += wavesTotal) {
    const bool last = tile >= tiles;
#if QA_WF_LOGIC_COMPACT
    if (!last) {
      const unsigned slot = tile * 64 + lane;
      const uint32_t info = __float_as_uint(b.D[slot].w);
      const uint32_t phase = WF_INFO_PHASE(info);
      const bool live = phase != WF_PH_DONE;
      int cat = -1;                // -1: nothing to do in this pass, 0: a hit to shade, 1: anything else
      if (live && b.out[slot] == 0 && !(phase == WF_PH_SAMPLE && !b.gateOpen)) {
        const uint32_t flags = b.redoFlag[slot];
        cat = (phase == WF_PH_TRACED && !(flags & 0x1Fu) && b.key[slot] != ~0ull) ? 0 : 1;
      }
      nLive += (unsigned) __popcll(__ballot(live && cat < 0));
      for (int c = 0; c < 2; ++c) {
        const unsigned long long m = __ballot(cat == c);
        if (cat == c) s_list[wave][c][nList[c] + __popcll(m & ((1ull << lane) - 1ull))] = slot;
        nList[c] += (unsigned) __popcll(m);
      }
      wfWaveSync();
    }
    while (nList[0] >= 64 || nList[1] >= 64 || (last && (nList[0] | nList[1]))) {
      const int c = nList[0] >= 64 ? 0 : (nList[1] >= 64 ? 1 : (nList[0] ? 0 : 1));
      const unsigned n = min(64u, nList[c]), base = nList[c] - n;
      const bool have = lane < n;
      const unsigned slot = have ? s_list[wave][c][base + lane] : 0u;
      nList[c] = base;
      wfWaveSync();
#else
    if (last) break;
    {
      const bool have = true;
      const unsigned slot = tile * 64 + lane;
#endif
      const WfEmit e = wfLogicSlot<TEX>(sc, rp, b, slot, have);
      qC.push(e.closest, slot, b.rayq, &ctr->nClosest);
      for (uint32_t j = 0; j < b.numLights; ++j) qS.push((e.shadow >> j) & 1u, slot | (j << 24), b.rayq + b.n, &ctr->nShadow);
      for (uint32_t t = 0; t <= b.numLights; ++t) {
        // rare: straight to the global queue
        const bool r = (e.redo >> t) & 1u;
        const unsigned long long mR = __ballot(r);
        if (!mR) continue;
        unsigned base2 = 0;
        const int leader = __ffsll((long long) mR) - 1;
        if ((int) lane == leader) base2 = atomicAdd(&ctr->nRedo, (unsigned) __popcll(mR));
        base2 = __shfl(base2, leader);
        if (r) b.redoq[base2 + __popcll(mR & ((1ull << lane) - 1ull))] = slot | (t << 24);
      }
      nLive += (unsigned) __popcll(__ballot(e.live && !e.pixelDone));
      nNew += (unsigned) __popcll(__ballot(e.newSample));
      nClosest += (unsigned) __popcll(__ballot(e.closest));
      nPix += (unsigned) __popcll(__ballot(e.pixelDone));
      unsigned nsh = __popc(e.shadow);
      for (int off = 32; off > 0; off >>= 1) nsh += __shfl_down(nsh, off);
      nShadow += __shfl(nsh, 0);
    }
#if QA_WF_LOGIC_COMPACT
    if (last) break;
#endif
  }
  qC.flush(b.rayq, &ctr->nClosest);
  qS.flush(b.rayq + b.n, &ctr->nShadow);
  if (lane == 0) {
    if (nLive) atomicAdd(&ctr->active, nLive);
    if (nNew) atomicAdd(&frame->samples, (unsigned long long) nNew);
    if (nClosest) atomicAdd(&frame->casts_normal, (unsigned long long) nClosest);
    if (nShadow) atomicAdd(&frame->casts_shadow, (unsigned long long) nShadow);
    if (nPix) atomicAdd(&frame->pixels, (unsigned long long) nPix);
  }
}

// ---------------------------------------------------------------------------------------------
// Exact search of ONE ray through the whole scene graph, as the reference walks it (and as the
// megakernel's traceClosest / shadow do): instances in pre-order, the running distance carried from one
// to the next.  Used by the first-generation trace stage (one lane per ray) and by wf_redo.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool wfMeshSearch(const DMesh &m, const Ray &ray, float &hz, bool closest, uint32_t *stack, uint32_t &bestTri,
                                             DCounters &cnt)
{
  const f3 drcp = F3(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
  const bool fastSlab = !__any(qabs(ray.d.x) < 1e-7f || qabs(ray.d.y) < 1e-7f || qabs(ray.d.z) < 1e-7f);
  float meshExit, entry;
  if (fastSlab) boxEntryExitFast(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
  else boxEntryExit(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
  if (entry > hz || entry > meshExit) return false;
  if (m.num_faces == 0) return false;
  bool tie = false;
  return walkBVH<false, false, true>(reinterpret_cast<const uint4 *>(m.nodes), reinterpret_cast<const uint4 *>(m.tris), m.rootData, ray, drcp,
                               fastSlab, hz, closest, stack, cnt, bestTri, tie);
}

__device__ __forceinline__ unsigned long long wfExactClosest(const DScene &sc, const Ray &world, uint32_t *stack)
{
  DCounters cnt = {0, 0, 0, 0, 0, 0};
  Hit h;
  h.z = QA_BIGFLOAT;
  h.node = -1;
  int bestK = -1;
  uint32_t bestTri = 0;
  const Ray r0 = rootRay<false>(sc, world);
  for (int k = 1; k < sc.num_inst; ++k) {
    const int type = instAt<false>(sc, k).obj_type;
    if (type == QA_OBJ_NONE) continue;
    const Ray r = localRay<false>(sc, k, r0);
    bool hit;
    uint32_t tri = 0;
    if (type == QA_OBJ_SPHERE) hit = hitSphere(r, h, k, false);
    else if (type == QA_OBJ_PLANE) hit = hitPlane(r, h, k, false);
    else hit = wfMeshSearch(meshAt<false>(sc, instAt<false>(sc, k).mesh), r, h.z, true, stack, tri, cnt);
    if (hit) { bestK = k; bestTri = tri; }
  }
  return bestK < 0 ? ~0ull : wfKey(h.z, bestK, bestTri);
}

__device__ __forceinline__ bool wfExactOccluded(const DScene &sc, const Ray &world, float t_max, uint32_t *stack)
{
  DCounters cnt = {0, 0, 0, 0, 0, 0};
  Hit h;
  h.z = t_max;
  h.node = -1;
  const Ray r0 = rootRay<false>(sc, world);
  for (int k = 1; k < sc.num_inst; ++k) {
    const int type = instAt<false>(sc, k).obj_type;
    if (type == QA_OBJ_NONE) continue;
    const Ray r = localRay<false>(sc, k, r0);
    bool hit;
    uint32_t tri = 0;
    if (type == QA_OBJ_SPHERE) hit = hitSphere(r, h, k, false);
    else if (type == QA_OBJ_PLANE) hit = hitPlane(r, h, k, false);
    else hit = wfMeshSearch(meshAt<false>(sc, instAt<false>(sc, k).mesh), r, h.z, false, stack, tri, cnt);
    if (hit) return true;
  }
  return false;
}

// ---------------------------------------------------------------------------------------------
// wf_cull: the scene-graph loop of Scene::TraceNodeNormal / TraceNodeShadow (src/scene/scene.cpp:35-74) for
// one queued ray per lane.  Spheres and planes are intersected here (exact, and order-independent: the closest
// of them wins, the first in node order at equal distance - the key's instance bits break ties that way).
// Every mesh whose bounds the ray enters within the distance found so far becomes a job for wf_trace.
//
// Why jobs may be walked independently although the reference carries one running distance through all
// nodes: the answer of the sequential walk is the minimum over the instances' own answers (ties: first
// instance) PROVIDED the winning triangle would also have been reached by the sequential walk, whose
// running distance can only be smaller than a job's.  wf_trace checks exactly that for every hit it
// commits (strict test of the hit's leaf box and the mesh gate against the found distance); a ray with a
// hit that fails the check is repeated by wf_redo as the reference walks it.  Shadow queries never depend
// on the order (any hit within the fixed t_max, src/lights/lights.cpp:39-48).
// ---------------------------------------------------------------------------------------------
#define QA_WF_JOBQ_CAP 128
__global__ __launch_bounds__(QA_BLOCK) void wf_cull(const DScene sc, WfBuf b, WfCounters *ctr)
{
  __shared__ float4 s_jobs[QA_BLOCK / 64][2][QA_WF_JOBQ_CAP];
  const unsigned lane = __lane_id(), wave = threadIdx.x / 64;
  float4 *jA = s_jobs[wave][0], *jB = s_jobs[wave][1];
  uint32_t fill = 0;   // wave-uniform
  auto flush = [&]() {
    if (!fill) return;
    wfWaveSync();
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&ctr->nJobs, fill);
    base = __shfl(base, 0);
    for (uint32_t i = lane; i < fill; i += 64)
      if (base + i < b.jobCap) { b.jobA[base + i] = jA[i]; b.jobB[base + i] = jB[i]; }
    // entries beyond the queue's capacity: their rays go to the exact repeat
    for (uint32_t i = lane; i < fill; i += 64)
      if (base + i >= b.jobCap) {
        const uint32_t bits = __float_as_uint(jB[i].w);
        atomicOr(&b.redoFlag[bits & QA_WF_SLOT_MASK], 1u << ((bits >> 24) & 7u));
        atomicSub(&b.out[bits & QA_WF_SLOT_MASK], 1u);
      }
    wfWaveSync();
    fill = 0;
  };
  const unsigned nC = ctr->nClosest, nS = ctr->nShadow;
  const unsigned nCpad = (nC + 63u) & ~63u;          // closest and shadow queries never share a wave
  const unsigned total = nCpad + nS;
  const unsigned wavesTotal = gridDim.x * (QA_BLOCK / 64);
  for (unsigned base = (blockIdx.x * (QA_BLOCK / 64) + wave) * 64u; base < total; base += wavesTotal * 64u) {
    const unsigned i = base + lane;
    const bool closest = base < nCpad;
    const bool valid = closest ? (i < nC) : (i < total);
    unsigned slot = 0, j = 0;
    Ray w;
    w.p = F3(0, 0, 0);
    w.d = F3(0, 0, 1);
    Hit h;
    h.z = QA_BIGFLOAT;
    h.node = -1;
    if (valid) {
      const unsigned e = closest ? b.rayq[i] : b.rayq[b.n + (i - nCpad)];
      slot = e & QA_WF_SLOT_MASK;
      j = e >> 24;
      const float4 P4 = b.P[slot];
      w.p = F3(P4.x, P4.y, P4.z);
      if (closest) {
        const float4 D4 = b.D[slot];
        w.d = F3(D4.x, D4.y, D4.z);
      } else {
        const float4 S4 = b.SH[(size_t) j * b.n + slot];
        w.d = F3(S4.x, S4.y, S4.z);
        h.z = S4.w;
      }
    }
    const Ray r0 = rootRay<false>(sc, w);
    GroupRay grp;   // the local ray of the group whose children are being visited (localRayInGroup, qa_kernel.h)
    grp.node = -1;
    grp.ray = r0;
    // ---- spheres and planes
    int bestK = -1;
    bool occluded = false;
    for (int k = 1; k < sc.num_inst; ++k) {
      const int type = instAt<false>(sc, k).obj_type;
      if (type != QA_OBJ_SPHERE && type != QA_OBJ_PLANE) continue;
      const Ray r = localRayInGroup<false>(sc, k, r0, grp);
      const bool hit = (type == QA_OBJ_SPHERE) ? hitSphere(r, h, k, false) : hitPlane(r, h, k, false);
      // hitSphere / hitPlane update only on a strictly smaller distance: bestK ends as the first instance at h.z
      if (hit) { bestK = k; occluded = !closest; }
    }
    // ---- meshes: one job per entered bound.  A ray whose origin lies so far from a mesh that the reference's inside test
    // can accept by cancellation (qa_widebvh.h ComputeMeshSlack) cannot be searched with a pruned walk: it goes to wf_redo.
    unsigned njobs = 0;
    bool exact = false;
    for (int k = 1; k < sc.num_inst; ++k) {
      const qa_instance ik = instAt<false>(sc, k);
      if (ik.obj_type != QA_OBJ_MESH) continue;
      const DMesh m = meshAt<false>(sc, ik.mesh);
      if (m.num_faces == 0) continue;
      const Ray r = localRayInGroup<false>(sc, k, r0, grp);
      float entry, exit_;
      wfMeshGate(m, r, entry, exit_);
      bool go = valid && !occluded && !exact && !(entry > h.z || entry > exit_);
      if (go && !insideCancelReach(m, r.p)) { exact = true; go = false; }
      const unsigned long long mask = __ballot(go);
      if (!mask) continue;
      const uint32_t n = (uint32_t) __popcll(mask);
      if (fill + n > QA_WF_JOBQ_CAP) flush();
      if (go) {
        const unsigned at = fill + __popcll(mask & ((1ull << lane) - 1ull));
        const uint32_t bits = slot | ((closest ? 0u : j + 1u) << 24) | ((uint32_t) k << 27);
        jA[at] = make_float4(r.p.x, r.p.y, r.p.z, h.z);
        jB[at] = make_float4(r.d.x, r.d.y, r.d.z, __uint_as_float(bits));
        ++njobs;
      }
      fill += n;
    }
    if (valid) {
      if (closest) b.key[slot] = bestK < 0 ? ~0ull : wfKey(h.z, bestK, 0);
      else if (occluded) atomicAnd(&b.vis[slot], ~(1u << j));
      if (njobs) atomicAdd(&b.out[slot], njobs);
      if (exact) atomicOr(&b.redoFlag[slot], 1u << (closest ? 0u : j + 1u));
    }
  }
  flush();
}

// ---------------------------------------------------------------------------------------------
// wf_trace: persistent waves, one BVH walk (job) per lane, finished lanes refilled from the queue.
//
// The walk searches the library's 4-wide tree (qa_widebvh.h, walkWide of qa_kernel.h cut into rounds): boxes widened
// by the fp32 slack of the reference's inside test, non-strict tests, nearest child first; leaves hold up to three
// triangles (DMesh::wtris, leaf order), tested with the reference's arithmetic.
// A lane is walking (at an inner node or at a leaf), finished, over its step budget, or idle.  Each round the
// wave runs the body more of its lanes wait for (node step: one 64-byte node, four slab tests; leaf step: the
// leaf's triangles), so both bodies execute with most lanes enabled.  Once a quarter of the wave is not
// walking, the finished lanes commit (closest: one 64-bit atomicMin of (distance, instance, triangle); shadow:
// one atomicAnd), the over-budget lanes write their job back with its stack (it continues in the next pass: a
// pass never waits for the ray that grazes a gridded wall for thousands of steps), and all of them take new
// jobs from a range of the queue the wave has reserved with one atomic per 128 jobs (WfBuf::reserve).
// What makes the answers the reference's: wf_logic's order check of the winning hit (wfHitDetails); here, a tie
// (a second triangle passing the inside test at exactly the held distance), a full stack, and a shadow hit
// whose reference leaf fails the strict test against the ray's t_max all send the ray to wf_redo.
// Dynamic LDS: the lanes' traversal stacks.  (Serving the top 85 - 340 nodes of every tree from LDS was 10 % slower and is gone:
// profiles/round02/staged_experiments.txt.)
// ---------------------------------------------------------------------------------------------
#ifndef QA_WF_TRACE_WAVES
#define QA_WF_TRACE_WAVES 5
#endif
#define QA_WF_NOBEST 0xFFFFFFFFu

__global__ __launch_bounds__(QA_BLOCK, QA_WF_TRACE_WAVES) void wf_trace(const DScene sc, WfBuf b, WfCounters *ctr, uint32_t parity,
                                                                          uint32_t budget)
{
  extern __shared__ uint4 s_dyn[];
  __shared__ unsigned long long s_wnodes[32], s_tris[32], s_nodes[32], s_shade[32];
  __shared__ uint32_t s_root[32];
  __shared__ float s_pad[32], s_absMax[32];
  if (threadIdx.x < 32) {
    const int k = (int) threadIdx.x;
    unsigned long long wn = 0, tr = 0, nd = 0, sh = 0;
    uint32_t root = QA_DONE;
    float pad = 0.f, am = 0.f;
    if (k < sc.num_inst && sc.inst[k].obj_type == QA_OBJ_MESH) {
      const DMesh &m = sc.mesh[sc.inst[k].mesh];
      wn = (unsigned long long) m.wnodes;
      tr = (unsigned long long) m.wtris;
      nd = (unsigned long long) m.nodes;
      sh = (unsigned long long) m.shade;
      root = m.wrootWord;
      pad = m.nearPad;
      am = m.absMax;
    }
    s_wnodes[k] = wn;
    s_tris[k] = tr;
    s_nodes[k] = nd;
    s_shade[k] = sh;
    s_root[k] = root;
    s_pad[k] = pad;
    s_absMax[k] = am;
  }
  uint32_t *stack = reinterpret_cast<uint32_t *>(s_dyn) + threadIdx.x;   // entry s at stack[s * QA_BLOCK]
  const uint32_t cap = b.traceStack;
  __syncthreads();
  const unsigned lane = __lane_id();
  const unsigned nCont = min(b.contCount[parity], b.contCap), nNew = min(ctr->nJobs, b.jobCap);
  const unsigned total = nCont + nNew;
  const unsigned reserve = b.reserve;   // jobs a wave reserves with one atomic
  const float INF = __builtin_inff();

  bool have = false, over = false, exhausted = false, tie = false;
  f3 lo = F3(0, 0, 0), ld = F3(0, 0, 1), drcp = F3(0, 0, 1);
  float hz = 0.f, hz0 = 0.f, pad = 0.f;
  uint32_t bits = 0, best = QA_WF_NOBEST, cur = QA_DONE, sp = 0, steps = 0;
  const uint4 *wn = nullptr, *tris = nullptr;
  uint32_t rNext = 0, rEnd = 0;                       // the wave's reserved range of job indices (wave-uniform)
  uint32_t nNode = 0, nLeaf = 0, nTri = 0, nJobsDone = 0, nSusp = 0, nFlag = 0;
  unsigned long long nSlots = 0, nRounds = 0;

  for (;;) {
    const unsigned long long mWalk = __ballot(have && cur != QA_DONE && !over);
    const unsigned long long mFin = __ballot(have && (cur == QA_DONE || over));
    const int nWalk = __popcll(mWalk);
    const bool canRefill = !exhausted || rNext < rEnd;
    if ((64 - nWalk >= (int) b.refillAt && (mFin || canRefill)) || (nWalk == 0)) {
      // ---- finished jobs commit
      if (have && cur == QA_DONE) {
        const unsigned slot = bits & QA_WF_SLOT_MASK, type = (bits >> 24) & 7u, k = bits >> 27;
        bool flag = tie;
        if (best != QA_WF_NOBEST && !flag) {
          if (type == 0) atomicMin(&b.key[slot], wfKey(hz, (int) k, best));
          else {
            // the reference reports "occluded" iff it reaches an accepted triangle: this one's leaf must pass its strict
            // test against the ray's fixed t_max (refReaches); if it does not, only the exact walk can tell
            Ray ray;
            ray.p = lo;
            ray.d = ld;
            const uint32_t leaf = ldGlobal(reinterpret_cast<const uint4 *>(s_shade[k]) + 3 * (size_t) best + 2).w;
            const bool nearZero = qabs(ld.x) < 1e-7f || qabs(ld.y) < 1e-7f || qabs(ld.z) < 1e-7f;
            if (refReaches<true>(reinterpret_cast<const uint4 *>(s_nodes[k]), leaf, ray, drcp, !nearZero, hz0)) atomicAnd(&b.vis[slot], ~(1u << (type - 1u)));
            else flag = true;
          }
        }
        if (flag) { atomicOr(&b.redoFlag[slot], 1u << type); ++nFlag; }
        atomicSub(&b.out[slot], 1u);   // read by the next pass's wf_logic: ordered by the kernel boundary
        have = false;
        ++nJobsDone;
      }
      // ---- jobs over their step budget continue in the next pass
      const unsigned long long mOver = __ballot(have && over);
      if (mOver) {
        unsigned cb = 0;
        const int leader = __ffsll((long long) mOver) - 1;
        if ((int) lane == leader) cb = atomicAdd(&b.contCount[parity ^ 1u], (unsigned) __popcll(mOver));
        cb = __shfl(cb, leader);
        if (have && over) {
          const unsigned at = cb + __popcll(mOver & ((1ull << lane) - 1ull));
          if (at < b.contCap) {
            b.contA[parity ^ 1u][at] = make_float4(lo.x, lo.y, lo.z, hz);
            b.contB[parity ^ 1u][at] = make_float4(ld.x, ld.y, ld.z, __uint_as_float(bits));
            b.contC[parity ^ 1u][at] = make_uint4(best, cur, sp | (tie ? 0x80000000u : 0u), __float_as_uint(hz0));
            uint32_t *sv = b.contStack[parity ^ 1u] + (size_t) at * b.traceStack;
            for (uint32_t q = 0; q < sp; ++q) sv[q] = stack[q * QA_BLOCK];
            have = false;
            ++nSusp;
          } else steps = 0;   // continuation queue full: keep walking
          over = false;
        }
      }
      // ---- refill
      const unsigned long long idle = __ballot(!have);
      if (idle) {
        if (rNext >= rEnd && !exhausted) {
          unsigned rb = 0;
          if (lane == 0) rb = atomicAdd(&ctr->jobHead, reserve);
          rb = __shfl(rb, 0);
          rNext = rb;
          rEnd = min(rb + reserve, total);
          if (rb + reserve >= total) exhausted = true;
          if (rb >= total) { rNext = rEnd = 0; }
        }
        const uint32_t avail = rEnd - rNext;
        const uint32_t rank = (uint32_t) __popcll(idle & ((1ull << lane) - 1ull));
        if (!have && rank < avail) {
          const unsigned my = rNext + rank;
          float4 A, B;
          if (my < nCont) {
            A = b.contA[parity][my];
            B = b.contB[parity][my];
            const uint4 C = b.contC[parity][my];
            best = C.x;
            cur = C.y;
            sp = C.z & 0x7FFFFFFFu;
            tie = (C.z >> 31) != 0;
            hz0 = __uint_as_float(C.w);
            const uint32_t *sv = b.contStack[parity] + (size_t) my * b.traceStack;
            for (uint32_t q = 0; q < sp; ++q) stack[q * QA_BLOCK] = sv[q];
          } else {
            A = b.jobA[my - nCont];
            B = b.jobB[my - nCont];
            best = QA_WF_NOBEST;
            sp = 0;
            tie = false;
            hz0 = A.w;
          }
          lo = F3(A.x, A.y, A.z);
          hz = A.w;
          ld = F3(B.x, B.y, B.z);
          bits = __float_as_uint(B.w);
          drcp = F3(1.f / ld.x, 1.f / ld.y, 1.f / ld.z);
          const uint32_t k = bits >> 27;
          wn = reinterpret_cast<const uint4 *>(s_wnodes[k]);
          tris = reinterpret_cast<const uint4 *>(s_tris[k]);
          pad = s_pad[k] + (QA_SLACK_SCALE * 1e-6f) * (qmax(qmax(qabs(lo.x), qabs(lo.y)), qabs(lo.z)) + s_absMax[k]);
          if (my >= nCont) cur = s_root[k];
          steps = 0;
          over = false;
          have = true;
        }
        rNext += min(avail, (uint32_t) __popcll(idle));
      }
      if (!__any(have)) break;
    }

    // ---- one round: the body more lanes wait for
    // (running both bodies every round, and picking out only the nearest child instead of sorting all four, were
    // measured: 4 % slower / neutral, profiles/round02/staged_experiments.txt)
    const bool atInner = have && !over && !(cur & QA_BVH_LEAF_BIT);
    const bool atLeaf = have && !over && (cur & QA_BVH_LEAF_BIT) && cur != QA_DONE;
    const int nI = __popcll(__ballot(atInner)), nL = __popcll(__ballot(atLeaf));
    if (nI >= nL && nI > 0) {
      if (atInner) {
        const f3 pLo = lo + F3(pad, pad, pad), pHi = lo - F3(pad, pad, pad);
        const uint4 *nd = wn + 4 * (size_t) cur;
        const uint4 q0 = ldGlobal(nd), q1 = ldGlobal(nd + 1), q2 = ldGlobal(nd + 2), q3 = ldGlobal(nd + 3);
        QA_WIDE_NODE(q0, q1, q2, q3)
        if (k3 < INF) { if (sp < cap) stack[(sp++) * QA_BLOCK] = w3; else tie = true; }
        if (k2 < INF) { if (sp < cap) stack[(sp++) * QA_BLOCK] = w2; else tie = true; }
        if (k1 < INF) { if (sp < cap) stack[(sp++) * QA_BLOCK] = w1; else tie = true; }
        if (k0 < INF) cur = w0;
        else cur = sp ? stack[(--sp) * QA_BLOCK] : QA_DONE;
        ++steps;
        ++nNode;
      }
      nSlots += (unsigned) nI;
      ++nRounds;
    }
    else if (nL > 0) {
      if (atLeaf) {
        Ray ray;
        ray.p = lo;
        ray.d = ld;
        const uint32_t count = ((cur >> QA_BVH_COUNT_SHIFT) & QA_BVH_COUNT_MASK) + 1;
        const uint32_t first = cur & QA_BVH_OFFSET_MASK;
        const bool anyHit = ((bits >> 24) & 7u) != 0;
        bool stop = false;
        for (uint32_t i = 0; i < count && !stop; ++i) {
          const uint4 *t = tris + 3 * (size_t) (first + i);
          ++nTri;
          const uint4 t2 = ldGlobal(t + 2);
          if (hitTriangleZTie<true>(ldGlobal(t), ldGlobal(t + 1), t2, ray, hz, tie)) {
            best = t2.w >> 2;               // element (the reference's triangle order)
            stop = anyHit;                  // TraceNodeShadow: the first accepted triangle ends the query
          }
        }
        cur = stop ? QA_DONE : (sp ? stack[(--sp) * QA_BLOCK] : QA_DONE);
        ++steps;
        ++nLeaf;
      }
      nSlots += (unsigned) nL;
      ++nRounds;
    }
    over = have && cur != QA_DONE && steps >= budget;
  }

  // ---- statistics: one atomic per wave and counter
  unsigned long long v[8] = {nJobsDone, nNode, nLeaf, nTri, nFlag, nSusp, nSlots, nRounds};
  unsigned long long *dst = reinterpret_cast<unsigned long long *>(b.stats);
  for (int i = 0; i < 8; ++i) {
    unsigned long long x = v[i];
    if (i < 6) for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    if (lane == 0 && x) atomicAdd(&dst[i], x);
  }
}

// ---------------------------------------------------------------------------------------------
// wf_redo: the exact repeat, one lane per flagged ray.  Dynamic LDS: traversal stacks.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(QA_BLOCK, 4) void wf_redo(const DScene sc, WfBuf b, const WfCounters *ctr)
{
  extern __shared__ uint4 s_dyn[];
  uint32_t *stack = reinterpret_cast<uint32_t *>(s_dyn) + threadIdx.x;
  const unsigned n = ctr->nRedo;
  for (unsigned base = blockIdx.x * QA_BLOCK; base < n; base += gridDim.x * QA_BLOCK) {
    const unsigned i = base + threadIdx.x;
    if (i >= n) continue;
    const unsigned e = b.redoq[i];
    const unsigned slot = e & QA_WF_SLOT_MASK, type = e >> 24;
    const float4 P4 = b.P[slot];
    Ray w;
    w.p = F3(P4.x, P4.y, P4.z);
    if (type == 0) {
      const float4 D4 = b.D[slot];
      w.d = F3(D4.x, D4.y, D4.z);
      b.key[slot] = wfExactClosest(sc, w, stack);
      b.redoFlag[slot] = QA_WF_EXACT;   // wf_logic takes this answer without the order check
    } else {
      const float4 S4 = b.SH[(size_t) (type - 1) * b.n + slot];
      w.d = F3(S4.x, S4.y, S4.z);
      // start from "visible": a job's unchecked hit may have cleared the bit
      if (wfExactOccluded(sc, w, S4.w, stack)) atomicAnd(&b.vis[slot], ~(1u << (type - 1)));
      else atomicOr(&b.vis[slot], 1u << (type - 1));
    }
  }
}

}  // namespace qa
