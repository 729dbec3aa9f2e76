// qa_kernel_cs.h — the megakernel for scenes whose meshes live in global memory, with COOPERATIVE mesh walks.
//
// Why.  Shadow queries are half of such a frame (cycle stamps, profiles/round02/megakernel_section_stamps.txt: 55 % of a
// wave's time on project7_object, 33 % on the tower scene, are the mesh walks of shadow rays), and qa_integrate walks them
// like everything else: per instance, with the 22 (9) of 64 lanes whose ray enters the bounds, for as long as the slowest
// of them takes.  But an any-hit query has no order to keep: "occluded" means that SOME triangle the reference accepts is
// reachable within t_max.  So here the wave walks the shadow rays of one (light, mesh instance) pair TOGETHER: the entered
// lanes' roots go into a pool in LDS (the wave's share of the traversal stacks, idle at that moment); every round all 64
// lanes pop one (owner, node) item each - a lane tests ANOTHER lane's ray, fetched from the owner's registers with
// ds_bpermute - and push the children the ray enters (ballot + prefix, no atomics) or test the leaf's triangles; items of a
// ray that is already settled are dropped.  Lane occupancy no longer depends on how many rays enter a mesh or on how
// unequal their walks are.
// What makes the answer the reference's is unchanged (hitMesh, qa_kernel.h): the accepted triangle's leaf in the reference
// tree must pass the reference's strict box test against t_max (refReaches), a triangle at exactly t_max, a full pool, a
// mesh without the 4-wide tree or an origin beyond the pruned search's reach repeat the query with the sequential walk
// of the reference tree.  Closest-hit queries, shading and every random draw are qa_integrate's: frames are bit-identical.
//
// Replaces (reference file:line): GenLight::Shadow -> Scene::TraceNodeShadow (src/lights/lights.cpp:39-48,
// src/scene/scene.cpp:35-46) with TriObj::IntersectRay / TraceBVHNode as the any-hit query (src/objects/objects.cpp:310-420);
// everything else as qa_kernel.h.
#pragma once
#include "qa_kernel.h"

namespace qa {

#ifdef QA_STAMPS
#define QA_FILL(cnt) (cnt).sl
#else
#define QA_FILL(cnt) nullptr
#endif
#define QA_CS_OWNER_SHIFT 22            /* pool item = child word | owner lane << 22 (inner: node index, leaf: flag, count, offset) */
#define QA_CS_INDEX_MASK 0x3FFFFFu      /* node indices / triangle offsets of a mesh must fit 22 bits (host check) */

__device__ __forceinline__ void csWaveSync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The wave's pool: entry i of wave w sits at pool[(i / 64) * QA_BLOCK + i % 64] with pool = LDS stacks + 64 w (the wave's
// own columns of the per-lane stacks).  cap entries, then 64 result words (accepted element + 1) and 64 flag words.
__device__ __forceinline__ uint32_t &csSlot(uint32_t *pool, uint32_t i) { return pool[(i >> 6) * QA_BLOCK + (i & 63u)]; }

// Any-hit walk of mesh m's 4-wide tree for the lanes with `own` (node-local ray r, padded by `pad`, limit tmax), by the
// whole wave.  found = an accepted element of the lane's own ray (or ~0u); tie = the query has to be repeated exactly.
__device__ __forceinline__ void csWalkAny(const DMesh &m, bool own, const Ray &r, f3 drcp, float pad, float tmax, uint32_t *pool, uint32_t cap,
                                          uint32_t &found, bool &tie, unsigned long long *fill = nullptr)
{
  const unsigned lane = __lane_id();
  const float INF = __builtin_inff();
  const uint4 *wn = reinterpret_cast<const uint4 *>(m.wnodes), *tris = reinterpret_cast<const uint4 *>(m.wtris);
  csSlot(pool, cap + lane) = 0;
  csSlot(pool, cap + 64 + lane) = 0;
  const unsigned long long mk = __ballot(own);
  uint32_t n = (uint32_t) __popcll(mk);
  if (own) csSlot(pool, (uint32_t) __popcll(mk & ((1ull << lane) - 1ull))) = m.wrootWord | (lane << QA_CS_OWNER_SHIFT);
  csWaveSync();
  while (n) {
    // With at most 16 items in the pool four lanes share an item: each tests ONE child of the node (or one triangle of the
    // leaf), so an underfilled round costs a quarter of the arithmetic - and all items are taken every round.
    const bool quad = n <= 16u;
    const uint32_t take = quad ? n : (n < 64u ? n : 64u);
    const uint32_t idx = quad ? (lane >> 2) : lane, sub = lane & 3u;
#ifdef QA_STAMPS
    if (fill && lane == 0) { fill[10] += quad ? 4u * take : take; fill[11] += 1; fill[12] += (take <= 16u) ? 1 : 0; }   /* lanes at work / rounds / rounds with <= 16 items */
#endif
    const bool work = idx < take;
    const uint32_t item = work ? csSlot(pool, n - take + idx) : 0u;
    n -= take;
    const uint32_t owner = (item >> QA_CS_OWNER_SHIFT) & 63u;
    // the owner's ray, out of its registers
    const f3 op = F3(__shfl(r.p.x, (int) owner), __shfl(r.p.y, (int) owner), __shfl(r.p.z, (int) owner));
    const f3 od = F3(__shfl(r.d.x, (int) owner), __shfl(r.d.y, (int) owner), __shfl(r.d.z, (int) owner));
    const f3 orc = F3(__shfl(drcp.x, (int) owner), __shfl(drcp.y, (int) owner), __shfl(drcp.z, (int) owner));
    const float opad = __shfl(pad, (int) owner), hz = __shfl(tmax, (int) owner);
    const bool live = work && csSlot(pool, cap + owner) == 0 && csSlot(pool, cap + 64 + owner) == 0;   // dropped once the ray is settled
    const bool isLeaf = (item & QA_BVH_LEAF_BIT) != 0;
    float k0 = INF, k1 = INF, k2 = INF, k3 = INF;
    uint32_t w0 = QA_DONE, w1 = QA_DONE, w2 = QA_DONE, w3 = QA_DONE;
    if (live && !isLeaf) {
      const uint4 *nd = wn + 4 * (size_t) (item & QA_CS_INDEX_MASK);
      const uint4 q0 = ldGlobal(nd), q1 = ldGlobal(nd + 1), q2 = ldGlobal(nd + 2), q3 = ldGlobal(nd + 3);
      const f3 pLo = op + F3(opad, opad, opad), pHi = op - F3(opad, opad, opad);
      const f3 drcp = orc;   // (the name QA_WIDE_CHILD uses)
      if (quad) {
        w0 = sub == 0 ? q3.x : sub == 1 ? q3.y : sub == 2 ? q3.z : q3.w;
        QA_WIDE_CHILD(k0, w0, sub)
      } else {
        w0 = q3.x; w1 = q3.y; w2 = q3.z; w3 = q3.w;
        QA_WIDE_CHILD(k0, w0, 0)
        QA_WIDE_CHILD(k1, w1, 1)
        QA_WIDE_CHILD(k2, w2, 2)
        QA_WIDE_CHILD(k3, w3, 3)
      }
    }
    // children the ray enters go back into the pool (any order will do for an any-hit query; looking at the nearest child
    // first was measured: C3 749 -> 699, C5 1436 -> 1413 Msamples/s - the sort costs more than it finds)
#define QA_CS_PUSH(K, W)                                                                                     \
    {                                                                                                        \
      const bool p = K < INF;                                                                                \
      const unsigned long long pm = __ballot(p);                                                             \
      if (p) {                                                                                               \
        const uint32_t at = n + (uint32_t) __popcll(pm & ((1ull << lane) - 1ull));                           \
        if (at < cap) csSlot(pool, at) = W | (owner << QA_CS_OWNER_SHIFT);                                   \
        else atomicOr(&csSlot(pool, cap + 64 + owner), 1u);   /* pool full: this ray is repeated exactly */  \
      }                                                                                                      \
      n += (uint32_t) __popcll(pm);                                                                          \
      n = n < cap ? n : cap;                                                                                 \
    }
    QA_CS_PUSH(k0, w0)
    if (!quad) {
      QA_CS_PUSH(k1, w1)
      QA_CS_PUSH(k2, w2)
      QA_CS_PUSH(k3, w3)
    }
#undef QA_CS_PUSH
    if (live && isLeaf) {
      Ray oray;
      oray.p = op;
      oray.d = od;
      const uint32_t count = ((item >> QA_BVH_COUNT_SHIFT) & QA_BVH_COUNT_MASK) + 1;
      const uint32_t first = item & QA_CS_INDEX_MASK;
      const uint32_t i0 = quad ? sub : 0u, di = quad ? 4u : 1u;   // quad: this lane's triangle(s) sub, sub + 4, ...
      float hzl = hz;
      bool tl = false, stop = false;
      for (uint32_t i = i0; i < count && !stop; i += di) {
        const uint4 *t = tris + 3 * (size_t) (first + i);
        const uint4 t2 = ldGlobal(t + 2);
        if (hitTriangleZTie<true>(ldGlobal(t), ldGlobal(t + 1), t2, oray, hzl, tl)) {
          csSlot(pool, cap + owner) = (t2.w >> 2) + 1u;   // several lanes may store for one owner: any accepted element will do
          stop = true;
        }
      }
      if (tl) atomicOr(&csSlot(pool, cap + 64 + owner), 1u);
    }
    csWaveSync();
  }
  found = own ? csSlot(pool, cap + lane) - 1u : ~0u;
  tie = own && csSlot(pool, cap + 64 + lane) != 0;
  csWaveSync();
}

// TriObj::IntersectRay as an any-hit query against t_max, for the lanes with `go` (hitMesh<RES = false> of qa_kernel.h
// with the 4-wide walk done by the whole wave).  Every lane of the wave calls this.
__device__ __forceinline__ bool csAnyHitMesh(const DMesh &m, bool go, const Ray &ray, float tmax, uint32_t *pool, uint32_t cap,
                                             uint32_t *stack, DCounters &cnt)
{
  const f3 drcp = F3(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
  const bool nearZero = qabs(ray.d.x) < 1e-7f || qabs(ray.d.y) < 1e-7f || qabs(ray.d.z) < 1e-7f;
  if (go) {
    float entry, meshExit;
    if (nearZero) boxEntryExit(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
    else boxEntryExitFast(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
    if (entry > tmax || entry > meshExit) go = false;   // Box::IntersectRay, src/core/box.cpp:94-128
  }
  if (m.num_faces == 0) go = false;
  const bool coop = go && m.useWide && insideCancelReach(m, ray.p);
  bool redo = go && !coop, hasHit = false;
  if (__any(coop)) {
    const float oMax = qmax(qmax(qabs(ray.p.x), qabs(ray.p.y)), qabs(ray.p.z));
    const float pad = m.nearPad + (QA_SLACK_SCALE * 1e-6f) * (oMax + m.absMax);
    uint32_t found;
    bool tie;
    QA_T(tW)
    csWalkAny(m, coop, ray, drcp, pad, tmax, pool, cap, found, tie, QA_FILL(cnt));
    QA_TACC(cnt.sl[6], tW)
    if (coop) {
      redo = tie;
      if (found != ~0u && !tie) {
        // the reference reports "occluded" iff it reaches an accepted triangle: this one's leaf must pass its strict test
        // against the ray's fixed t_max; if it does not, only the sequential walk can tell
        const uint32_t leaf = ldGlobal(reinterpret_cast<const uint4 *>(m.shade) + 3 * (size_t) found + 2).w;   // DTriShade::pad
        if (refReaches<true>(reinterpret_cast<const uint4 *>(m.nodes), leaf, ray, drcp, !nearZero, tmax)) hasHit = true;
        else redo = true;
      }
    }
  }
  if (redo) {
    float hz = tmax;
    uint32_t bestTri = 0;
    bool tie = false;
    const bool fastSlab = !__any(nearZero);
    hasHit = walkBVH<false, false, true>(reinterpret_cast<const uint4 *>(m.nodes), reinterpret_cast<const uint4 *>(m.tris), m.rootData, ray, drcp,
                                         fastSlab, hz, false, stack, cnt, bestTri, tie);
  }
  return hasHit;
}

// ---------------------------------------------------------------------------------------------
// Closest-hit walks by the whole wave.  (A twin of csWalkAny on purpose: folded into one template the textured kernel runs
// 4 % slower - C3 748 -> 719 Msamples/s - for the same instructions in another order; session3_experiments.txt, item 18.)  Same pool; behind it 64 result keys (distance bits << 32 | element: one
// ds_min_u64 per accepted triangle, and the distance half is what every lane prunes and accepts against, so a hit found by
// one lane shortens the work of all lanes on that ray at once) and 64 flag words.  A triangle that passes the inside test at
// exactly the distance held - whether that distance was there before the test or arrived from another lane at the same
// moment (the atomic's return value) - raises the flag: the one situation in which the ORDER of the tests decides
// (hitMesh then repeats the query in the reference's order, as after a tie of the sequential 4-wide walk).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long *csKey(uint32_t *pool, uint32_t cap, uint32_t owner)
{
  // 64 keys = two pool rows of 64 words; key o sits in row o / 32
  return reinterpret_cast<unsigned long long *>(&csSlot(pool, cap + (owner >> 5) * 64u)) + (owner & 31u);
}

__device__ __forceinline__ void csWalkClosest(const DMesh &m, bool own, const Ray &r, f3 drcp, float pad, float limit, uint32_t *pool, uint32_t cap,
                                              float &hz, uint32_t &found, bool &tie, unsigned long long *fill = nullptr)
{
  const unsigned lane = __lane_id();
  const float INF = __builtin_inff();
  const uint4 *wn = reinterpret_cast<const uint4 *>(m.wnodes), *tris = reinterpret_cast<const uint4 *>(m.wtris);
  *csKey(pool, cap, lane) = ((unsigned long long) __float_as_uint(limit) << 32) | 0xFFFFFFFFull;
  csSlot(pool, cap + 128 + lane) = 0;
  const unsigned long long mk = __ballot(own);
  uint32_t n = (uint32_t) __popcll(mk);
  if (own) csSlot(pool, (uint32_t) __popcll(mk & ((1ull << lane) - 1ull))) = m.wrootWord | (lane << QA_CS_OWNER_SHIFT);
  csWaveSync();
  while (n) {
    const bool quad = n <= 16u;   // four lanes per item, one child / triangle each (csWalkAny)
    const uint32_t take = quad ? n : (n < 64u ? n : 64u);
    const uint32_t idx = quad ? (lane >> 2) : lane, sub = lane & 3u;
#ifdef QA_STAMPS
    if (fill && lane == 0) { fill[10] += quad ? 4u * take : take; fill[11] += 1; fill[12] += (take <= 16u) ? 1 : 0; }
#endif
    const bool work = idx < take;
    const uint32_t item = work ? csSlot(pool, n - take + idx) : 0u;
    n -= take;
    const uint32_t owner = (item >> QA_CS_OWNER_SHIFT) & 63u;
    const f3 op = F3(__shfl(r.p.x, (int) owner), __shfl(r.p.y, (int) owner), __shfl(r.p.z, (int) owner));
    const f3 od = F3(__shfl(r.d.x, (int) owner), __shfl(r.d.y, (int) owner), __shfl(r.d.z, (int) owner));
    const f3 orc = F3(__shfl(drcp.x, (int) owner), __shfl(drcp.y, (int) owner), __shfl(drcp.z, (int) owner));
    const float opad = __shfl(pad, (int) owner);
    unsigned long long *key = csKey(pool, cap, owner);
    const bool live = work && csSlot(pool, cap + 128 + owner) == 0;
    const float hzNow = __uint_as_float((uint32_t) (*key >> 32));   // the distance the owner's ray holds right now
    const bool isLeaf = (item & QA_BVH_LEAF_BIT) != 0;
    float k0 = INF, k1 = INF, k2 = INF, k3 = INF;
    uint32_t w0 = QA_DONE, w1 = QA_DONE, w2 = QA_DONE, w3 = QA_DONE;
    if (live && !isLeaf) {
      const uint4 *nd = wn + 4 * (size_t) (item & QA_CS_INDEX_MASK);
      const uint4 q0 = ldGlobal(nd), q1 = ldGlobal(nd + 1), q2 = ldGlobal(nd + 2), q3 = ldGlobal(nd + 3);
      const f3 pLo = op + F3(opad, opad, opad), pHi = op - F3(opad, opad, opad);
      const f3 drcp = orc;
      const float hz = hzNow;   // (the names QA_WIDE_CHILD uses)
      if (quad) {
        w3 = sub == 0 ? q3.x : sub == 1 ? q3.y : sub == 2 ? q3.z : q3.w;   // (every item is taken every round while quad: no order to keep)
        QA_WIDE_CHILD(k3, w3, sub)
      } else {
        w0 = q3.x; w1 = q3.y; w2 = q3.z; w3 = q3.w;
        QA_WIDE_CHILD(k0, w0, 0)
        QA_WIDE_CHILD(k1, w1, 1)
        QA_WIDE_CHILD(k2, w2, 2)
        QA_WIDE_CHILD(k3, w3, 3)
        // farthest first: the pool is popped from its top, so the nearest child of a node is looked at first
        QA_WIDE_CE(k0, w0, k1, w1)
        QA_WIDE_CE(k2, w2, k3, w3)
        QA_WIDE_CE(k0, w0, k2, w2)
        QA_WIDE_CE(k1, w1, k3, w3)
        QA_WIDE_CE(k1, w1, k2, w2)
      }
    }
#define QA_CS_PUSH(K, W)                                                                                     \
    {                                                                                                        \
      const bool p = K < INF;                                                                                \
      const unsigned long long pm = __ballot(p);                                                             \
      if (p) {                                                                                               \
        const uint32_t at = n + (uint32_t) __popcll(pm & ((1ull << lane) - 1ull));                           \
        if (at < cap) csSlot(pool, at) = W | (owner << QA_CS_OWNER_SHIFT);                                   \
        else atomicOr(&csSlot(pool, cap + 128 + owner), 1u);                                                 \
      }                                                                                                      \
      n += (uint32_t) __popcll(pm);                                                                          \
      n = n < cap ? n : cap;                                                                                 \
    }
    QA_CS_PUSH(k3, w3)
    if (!quad) {
      QA_CS_PUSH(k2, w2)
      QA_CS_PUSH(k1, w1)
      QA_CS_PUSH(k0, w0)
    }
#undef QA_CS_PUSH
    if (live && isLeaf) {
      Ray oray;
      oray.p = op;
      oray.d = od;
      const uint32_t count = ((item >> QA_BVH_COUNT_SHIFT) & QA_BVH_COUNT_MASK) + 1;
      const uint32_t first = item & QA_CS_INDEX_MASK;
      const uint32_t i0 = quad ? sub : 0u, di = quad ? 4u : 1u;
      float hzl = hzNow;
      bool tl = false;
      for (uint32_t i = i0; i < count; i += di) {
        const uint4 *t = tris + 3 * (size_t) (first + i);
        const uint4 t2 = ldGlobal(t + 2);
        if (hitTriangleZTie<true>(ldGlobal(t), ldGlobal(t + 1), t2, oray, hzl, tl)) {
          const unsigned long long mine = ((unsigned long long) __float_as_uint(hzl) << 32) | (unsigned long long) (t2.w >> 2);
          const unsigned long long old = atomicMin(key, mine);
          if ((uint32_t) (old >> 32) == __float_as_uint(hzl) && old != mine) tl = true;   // another lane accepted this very distance meanwhile
        }
      }
      if (tl) atomicOr(&csSlot(pool, cap + 128 + owner), 1u);
    }
    csWaveSync();
  }
  const unsigned long long k = *csKey(pool, cap, lane);
  found = own ? (uint32_t) k : ~0u;            // 0xFFFFFFFF: nothing accepted
  hz = __uint_as_float((uint32_t) (k >> 32));
  tie = own && csSlot(pool, cap + 128 + lane) != 0;
  csWaveSync();
}

// hitMesh<RES = false>(closest = true) of qa_kernel.h with the 4-wide walk done by the whole wave; `go`: this lane has a ray.
__device__ __forceinline__ bool csHitMeshClosest(const DMesh &m, bool go, const Ray &ray, Hit &h, int k, uint32_t *pool, uint32_t cap,
                                                 uint32_t *stack, DCounters &cnt, TriPick &pick)
{
  const f3 drcp = F3(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
  const bool nearZero = qabs(ray.d.x) < 1e-7f || qabs(ray.d.y) < 1e-7f || qabs(ray.d.z) < 1e-7f;
  if (go) {
    float entry, meshExit;
    if (nearZero) boxEntryExit(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
    else boxEntryExitFast(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
    if (entry > h.z || entry > meshExit) go = false;
  }
  if (m.num_faces == 0) go = false;
  const uint4 *nodes = reinterpret_cast<const uint4 *>(m.nodes), *tris = reinterpret_cast<const uint4 *>(m.tris), *shade = reinterpret_cast<const uint4 *>(m.shade);
  const float hz0 = h.z;
  const bool coop = go && m.useWide && insideCancelReach(m, ray.p);
  bool redo = go && !coop, hasHit = false;
  uint32_t bestTri = 0;
  if (__any(coop)) {
    const float oMax = qmax(qmax(qabs(ray.p.x), qabs(ray.p.y)), qabs(ray.p.z));
    const float pad = m.nearPad + (QA_SLACK_SCALE * 1e-6f) * (oMax + m.absMax);
    float hz;
    uint32_t found;
    bool tie;
    QA_T(tW)
    csWalkClosest(m, coop, ray, drcp, pad, hz0, pool, cap, hz, found, tie, QA_FILL(cnt));
    QA_TACC(cnt.sl[3], tW)
    if (coop) {
      redo = tie;
      if (found != 0xFFFFFFFFu && !tie) {
        const uint32_t leaf = ldGlobal(shade + 3 * (size_t) found + 2).w;   // DTriShade::pad
        if (refReaches<true>(nodes, leaf, ray, drcp, !nearZero, hz)) {
          hasHit = true;
          bestTri = found;
          h.z = hz;
        } else redo = true;
      }
    }
  }
  if (redo) {
    h.z = hz0;
    bool tie = false;
    const bool fastSlab = !__any(nearZero);
    hasHit = walkBVH<false, false, true>(nodes, tris, m.rootData, ray, drcp, fastSlab, h.z, true, stack, cnt, bestTri, tie);
  }
  if (hasHit) {
    float ba = 0, bb = 0;
    {
      const uint4 *t = tris + 3 * (size_t) bestTri;
      triangleDetails(ldGlobal(t), ldGlobal(t + 1), ldGlobal(t + 2), ray, h, ba, bb);
    }
    // shading normal: TriMesh::GetNormal (src/mesh/TriMesh.h:196-204), left un-normalised
    const uint4 *s = shade + 3 * (size_t) bestTri;
    const uint4 s0 = ldGlobal(s), s1 = ldGlobal(s + 1), s2 = ldGlobal(s + 2);
    const float bc = 1.f - ba - bb;
    const f3 n0 = F3(asF(s0.x), asF(s0.y), asF(s0.z)), n1 = F3(asF(s0.w), asF(s1.x), asF(s1.y)), n2 = F3(asF(s1.z), asF(s1.w), asF(s2.x));
    h.N = (n0 * ba + n1 * bb) + n2 * bc;
    h.mtlID = (int) s2.y;
    h.node = k;
    pick.tri = bestTri;
    pick.a = ba;
    pick.b = bb;
  }
  return hasHit;
}

// Scene::TraceNodeNormal (traceClosest of qa_kernel.h) for the lanes with `act`; every lane of the wave calls this.
template <bool TEX>
__device__ __forceinline__ bool csTraceClosest(const DScene &sc, bool act, const Ray &world, const RayDiff &wd, Hit &h, TexHit &th, uint32_t *pool,
                                               uint32_t cap, uint32_t *stack, DCounters &cnt)
{
  if (!__any(act)) return false;
  if (act) cnt.casts_normal++;
  const Ray r0 = rootRay<false>(sc, world);
  GroupRay grp;
  grp.node = -1;
  grp.ray = r0;
  bool any = false;
  for (int k = 1; k < sc.num_inst; ++k) {
    const qa_instance in = instAt<false>(sc, k);
    const int type = in.obj_type;
    if (type == QA_OBJ_NONE) continue;
    // (the node-local differential directions are only needed by an object that is hit: they are built then, by the
    // same chain of operations localRayDiff performs for the central ray, instead of for every node of every query)
    const Ray r = localRayInGroup<false>(sc, k, r0, grp);
    bool hit = false;
    if (type == QA_OBJ_SPHERE) {
      if (act) hit = hitSphere(r, h, k, true);
      if (TEX && hit) {
        Ray r2;
        RayDiff rd;
        localRayDiff<false>(sc, k, world, wd, r2, rd);
        texSphere(r.p, rd.dx, rd.dy, h.p, h.N, th);
      }
    } else if (type == QA_OBJ_PLANE) {
      if (act) hit = hitPlane(r, h, k, true);
      if (TEX && hit) {
        Ray r2;
        RayDiff rd;
        localRayDiff<false>(sc, k, world, wd, r2, rd);
        texPlane(r.p, rd.dx, rd.dy, h.p, th);
      }
    } else {
      const DMesh m = meshAt<false>(sc, in.mesh);
      TriPick pick;
      hit = csHitMeshClosest(m, act, r, h, k, pool, cap, stack, cnt, pick);
      if (TEX && hit && m.hasVT) {
        Ray r2;
        RayDiff rd;
        localRayDiff<false>(sc, k, world, wd, r2, rd);
        const uint4 *t = reinterpret_cast<const uint4 *>(m.tris) + 3 * (size_t) pick.tri;
        texTriangle(ldGlobal(t), ldGlobal(t + 1), ldGlobal(t + 2), m.vt + 6 * (size_t) pick.tri, r.p, rd.dx, rd.dy, pick.a, pick.b, th);
      }
    }
    any |= hit;
  }
  if (any) {
    // Node::FromNodeCoords at every level from the hit node up to and including the root (src/core/node.cpp:127-139)
    for (int a = h.node; a >= 0; a = instAt<false>(sc, a).parent) {
      if (a == 0 && sc.rootIdentity) {
        h.N = normalize(h.N);
        break;
      }
      const qa_instance ia = instAt<false>(sc, a);
      h.p = mulMV(ia.tm, h.p) + ld3(ia.pos);
      h.N = normalize(mulTMV(ia.itm, h.N));
    }
  }
  return any;
}

// The shadow ray illuminate() (qa_kernel.h) shoots from p towards light l
__device__ __forceinline__ void csShadowRay(const qa_light &l, f3 p, Ray &w, float &tmax)
{
  w.p = p;
  if (l.type == QA_LIGHT_DIRECT) {
    w.d = normalize(-ld3(l.direction));
    tmax = QA_BIGFLOAT;
  } else {
    const f3 dir = ld3(l.position) - p;
    w.d = normalize(dir);
    tmax = length(dir);
  }
}

// GenLight::Shadow -> Scene::TraceNodeShadow for every non-ambient light of the lanes with `lit`: bit j of the result =
// light slot j occluded.  The reference stops at the first node that occludes; which node that is does not matter.
__device__ __forceinline__ uint32_t csShadows(const DScene &sc, bool lit, f3 p, uint32_t *pool, uint32_t cap, uint32_t *stack, DCounters &cnt)
{
  uint32_t occl = 0, j = 0;
  for (int li = 0; li < sc.num_lights; ++li) {
    const qa_light l = ldTable(sc.light + li);
    if (l.type == QA_LIGHT_AMBIENT) continue;
    Ray w;
    float tmax;
    csShadowRay(l, p, w, tmax);
    if (lit) cnt.casts_shadow++;
    const Ray r0 = rootRay<false>(sc, w);
    GroupRay grp;
    grp.node = -1;
    grp.ray = r0;
    bool open = lit;   // this lane's query is still undecided
    for (int k = 1; k < sc.num_inst; ++k) {
      const qa_instance in = instAt<false>(sc, k);
      if (in.obj_type == QA_OBJ_NONE) continue;
      if (!__any(open)) break;
      const Ray r = localRayInGroup<false>(sc, k, r0, grp);
      bool hit = false;
      if (in.obj_type == QA_OBJ_SPHERE || in.obj_type == QA_OBJ_PLANE) {
        Hit h;
        h.z = tmax;
        h.node = -1;
        if (open) hit = (in.obj_type == QA_OBJ_SPHERE) ? hitSphere(r, h, k, false) : hitPlane(r, h, k, false);
      } else {
        hit = csAnyHitMesh(meshAt<false>(sc, in.mesh), open, r, tmax, pool, cap, stack, cnt);
      }
      if (open && hit) {
        occl |= 1u << j;
        open = false;
      }
    }
    ++j;
  }
  return occl;
}

// directLight() of qa_kernel.h with the shadow factors known (1.0f multiplies exactly, 0.0f gives the same signed zeros)
__device__ __forceinline__ f3 csDirectLight(const DScene &sc, f3 p, f3 N, f3 V, f3 kd, f3 ks, float gloss, uint32_t occl)
{
  f3 sum = F3(0, 0, 0);
  const float normCoefDI = 1.f / (float) sc.num_lights;
  uint32_t j = 0;
  for (int li = 0; li < sc.num_lights; ++li) {
    const qa_light l = ldTable(sc.light + li);
    if (l.type == QA_LIGHT_AMBIENT) continue;
    const float vis = ((occl >> j) & 1u) ? 0.0f : 1.0f;
    ++j;
    f3 I;
    if (l.type == QA_LIGHT_DIRECT) I = ld3(l.intensity) * vis;
    else {
      const f3 dir = ld3(l.position) - p;
      I = (ld3(l.intensity) * vis) * inverseSquareFalloff(dir);
      if (l.type == QA_LIGHT_SPOT) I = I * spotAttenuation(l, p);
    }
    const f3 intensity = I * normCoefDI;
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
// The kernel: qa_integrate<RES = false, LIGHTS, TEX, AREA = false> with section D cut where the wave meets for its
// shadow walks.  Dynamic LDS as qa_integrate's: [traversal stacks | sample accumulators].
// Sections A, B and E are qa_integrate's text, repeated here on purpose: moving them into functions shared by both kernels
// changes the register allocation of qa_integrate's LDS-resident variants - the Cornell-box kernel lost 8 % (13.0 -> 12.0
// Gsamples/s) with only the tile fetch factored out (profiles/round02/session3_experiments.txt, item 14).
// ---------------------------------------------------------------------------------------------
// Waves per SIMD the register allocator must leave room for: three for the untextured variants (the walks' bookkeeping
// and the owner rays spill at four: C5 1262 -> 1439, C4 4007 -> 4622 Msamples/s), four for the textured ones (C3 731 vs 711).
#ifndef QA_CS_WAVES_NOTEX
#define QA_CS_WAVES_NOTEX 3
#endif
#ifndef QA_CS_WAVES_TEX
#define QA_CS_WAVES_TEX 4
#endif
template <bool LIGHTS, bool TEX>
__global__ __launch_bounds__(QA_BLOCK, TEX ? QA_CS_WAVES_TEX : QA_CS_WAVES_NOTEX) void qa_integrate_cs(const DScene sc, const RenderParams rp)
{
  extern __shared__ uint4 s_dyn[];
  SceneMem<false> mem;
  mem.img = s_dyn;
  uint32_t *stack = reinterpret_cast<uint32_t *>(s_dyn) + threadIdx.x;
  float *acc = reinterpret_cast<float *>(stack + (size_t) sc.stackDepth * QA_BLOCK - threadIdx.x) + threadIdx.x;
  const uint4 *mtlTable = reinterpret_cast<const uint4 *>(sc.mtl);
  // the wave's pool for cooperative walks: its own columns of the per-lane stacks
  uint32_t *pool = reinterpret_cast<uint32_t *>(s_dyn) + (threadIdx.x / 64) * 64;
  // behind the pool: result words / keys and flags (csWalkAny, csWalkClosest).  (A limit below the LDS there is: tests of the overflow path.)
  const uint32_t poolRoom = sc.stackDepth * 64u - 192u;
  const uint32_t poolCap = (sc.csPoolLimit && sc.csPoolLimit < poolRoom) ? (sc.csPoolLimit & ~63u) : poolRoom;

  const int rw = rp.x1 - rp.x0, rh = rp.y1 - rp.y0;
  const unsigned tilesX = (unsigned) (rw + 7) / 8;
  const unsigned total = tilesX * (unsigned) rp.own_tile_rows * 64u;
  const unsigned lane = __lane_id();

  DCounters cnt = {};
#ifdef QA_STAMPS
  __shared__ unsigned long long s_stamps[QA_BLOCK / 64][13];
  cnt.sl = s_stamps[threadIdx.x / 64];
  if (__lane_id() < 13) cnt.sl[__lane_id()] = 0;
#endif
  QA_T(tKernel)
  TexTables tt;
  tt.blob = sc.blob;
  tt.texmap = sc.texmap;
  tt.tex = sc.tex;
  tt.filter = sc.texFilter;
  RayDiff pathDiff;
  pathDiff.dx = pathDiff.dy = F3(0, 0, 1);

  int px = 0, py = 0;
  unsigned q = 0;
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
    // ---- A. tile fetch (qa_integrate, section A)
    const unsigned long long aliveMask = __ballot(alive);
    const unsigned long long want = __ballot(alive && needPixel);
    if (want && want == aliveMask) {
      unsigned base = 0;
      const int leader = __ffsll((long long) want) - 1;
      if ((int) lane == leader) base = (*rp.stop_flag) ? total : atomicAdd(rp.work_counter, 64u);
      base = __shfl(base, leader);
      if (alive) {
        const unsigned w = base + lane;
        if (base >= total) {
          alive = false;
        } else {
          const unsigned in = w % 64;
          const unsigned tile = rp.tile_order ? rp.tile_order[w / 64] : w / 64;
          const unsigned otr = tile / tilesX;
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
          }
        }
      }
    }
    if (!__any(alive)) break;

    // ---- B. start a sample (qa_integrate, section B; src/renderers/renderer.cpp:312-328)
    const bool goSample = !rp.sync_samples || (__ballot(needSample) == __ballot(alive && !needPixel));
    if (alive && needSample && goSample) {
      const float hx = sc.halton[2 * sidx], hy = sc.halton[2 * sidx + 1];
      texpos = F3(hx, hy, 0.f) + F3((float) px, (float) py, 0.f);
      const f3 A = ld3(sc.cam.screenA), U = ld3(sc.cam.screenU), V = ld3(sc.cam.screenV);
      const f3 cpt = (A + U * texpos.x) + V * texpos.y;
      f3 campos = ld3(sc.cam.pos);
      if (sc.cam.dof > 0.1f) {
        const float r1 = rng1(rng), r2 = rng1(rng);
        const float r = sc.cam.dof * qsqrt(r1);
        const float t = r2 * 2.f * QA_PI;
        campos = campos + (ld3(sc.cam.screenX) * (r * qcosf(t)) + ld3(sc.cam.screenY) * (r * qsinf(t)));
      }
      path.ray.p = campos;
      path.ray.d = normalize(cpt - campos);
      if (TEX) {
        const f3 xpt = (A + U * (texpos.x + QA_DX)) + V * texpos.y;
        const f3 ypt = (A + U * texpos.x) + V * (texpos.y + QA_DX);
        pathDiff.dx = normalize(xpt - campos);
        pathDiff.dy = normalize(ypt - campos);
      }
      path.T = F3(1, 1, 1);
      path.L = F3(0, 0, 0);
      path.absorbMtl = -1;
      path.bounce = rp.max_bounce;
      path.fromDiffuse = false;
      path.primary = true;
      needSample = false;
      cnt.samples++;
    }
    QA_TACC(cnt.sl[1], tA)
    // ---- C. trace (qa_integrate, section C)
    const bool act = alive && !needPixel && !needSample;
    bool done = false;
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
    QA_T(tC)
    const bool found = csTraceClosest<TEX>(sc, act, path.ray, pathDiff, h, th, pool, poolCap, stack, cnt);
    QA_TACC(cnt.sl[2], tC)
    QA_T(tD)

    // ---- D. shade up to the lights (qa_integrate, section D)
    bool lit = false;
    int mi = -1;
    f3 V = F3(0, 0, 1), N = F3(0, 0, 1), p = F3(0, 0, 0);
    Surface sf;
    sf.emission = sf.kd = sf.ks = sf.nextDir = sf.bxdf = F3(0, 0, 0);
    sf.gloss = 0.f;
    sf.spawn = sf.nextFromDiffuse = sf.selDiffuse = false;
    if (act) {
      if (path.primary && sidx == 0) rp.depth[q] = found ? h.z : QA_BIGFLOAT;
      if (!found) {
        f3 c = path.primary ? ld3(sc.background) : ld3(sc.environment);
        if (TEX) {
          if (path.primary)
            c = texColorSample(tt, c, sc.bgTexmap, F3(texpos.x / (float) sc.cam.width, texpos.y / (float) sc.cam.height, 0.f));
          else
            c = sampleEnvironment(tt, c, sc.envTexmap, path.ray.d);
        }
        path.L = path.L + path.T * c;
        done = true;
      } else {
        if (!path.primary && !h.front && path.absorbMtl >= 0) {
          const uint4 ab = mtlTable[6 * (size_t) path.absorbMtl + 5];
          const f3 att = F3(qexpf(-asF(ab.x) * h.z), qexpf(-asF(ab.y) * h.z), qexpf(-asF(ab.z) * h.z));
          path.T = path.T * att;
        }
        const qa_instance &in = sc.inst[h.node];
        bool white = false;
        if (in.mtlset >= 0) {
          const qa_mtlset ms = sc.mtlset[in.mtlset];
          if (ms.multi) {
            if (h.mtlID >= 0 && h.mtlID < ms.count) mi = ms.first + h.mtlID;
            else white = true;
          } else mi = ms.first;
        }
        if (mi < 0) {
          if (white) path.L = path.L + path.T;
          done = true;
        } else {
          V = -path.ray.d;
          N = h.N;
          p = h.p;
          sf = shadeSurface<TEX>(mtlTable, sc, tt, mi, N, V, h.front, th, path.bounce, path.fromDiffuse, rng);
          path.L = path.L + path.T * sf.emission;
          lit = true;
        }
      }
    }
    QA_TACC(cnt.sl[4], tD)
    // ---- direct lighting: the whole wave walks the shadow rays of its lit lanes
    if (LIGHTS) {
      if (__any(lit)) {
        QA_T(tL)
        const uint32_t occl = csShadows(sc, lit, p, pool, poolCap, stack, cnt);
        if (lit) path.L = path.L + path.T * csDirectLight(sc, p, N, V, sf.kd, sf.ks, sf.gloss, occl);
        QA_TACC(cnt.sl[5], tL)
      }
    }
    QA_T(tE)
    if (lit) {
      if (sf.spawn) {
        path.ray.p = p;
        path.ray.d = normalize(sf.nextDir);
        if (TEX) pathDiff.dx = pathDiff.dy = path.ray.d;
        path.T = path.T * sf.bxdf;
        path.absorbMtl = mi;
        path.bounce -= 1;
        path.fromDiffuse = sf.nextFromDiffuse;
        path.primary = false;
      } else {
        done = true;
      }
    }

    // ---- E. sample finished (qa_integrate, section E; scene.cpp:92-121)
    if (alive && done) {
      const float inv = (float) (sidx + 1);
      f3 mean = F3(acc[0], acc[QA_BLOCK], acc[2 * QA_BLOCK]);
      f3 cstd = F3(acc[3 * QA_BLOCK], acc[4 * QA_BLOCK], acc[5 * QA_BLOCK]);
      const f3 dc = (path.L - mean) / inv;
      mean = mean + dc;
      if (sidx > 0) cstd = cstd + ((dc * dc) * inv - cstd / (float) sidx);
      acc[0] = mean.x; acc[QA_BLOCK] = mean.y; acc[2 * QA_BLOCK] = mean.z;
      acc[3 * QA_BLOCK] = cstd.x; acc[4 * QA_BLOCK] = cstd.y; acc[5 * QA_BLOCK] = cstd.z;
      ++sidx;
      const bool more = sidx < rp.spp_min || (sidx < rp.spp_max && (cstd.x > 0.005f || cstd.y > 0.001f || cstd.z > 0.005f));
      if (more) {
        needSample = true;
      } else {
        rp.rgb[3 * q + 0] = mean.x;
        rp.rgb[3 * q + 1] = mean.y;
        rp.rgb[3 * q + 2] = mean.z;
        rp.ns[q] = (uint32_t) sidx;
        cnt.pixels++;
        needPixel = true;
      }
    }
    QA_TACC(cnt.sl[7], tE)
#ifdef QA_STAMPS
    if (lane == 0) cnt.sl[8] += 1;
#endif
  }

  unsigned long long v[6] = {cnt.samples, cnt.casts_normal, cnt.casts_shadow, cnt.bvh_nodes, cnt.tri_tests, cnt.pixels};
  unsigned long long *dst = reinterpret_cast<unsigned long long *>(rp.counters);
  for (int i = 0; i < 6; ++i) {
    unsigned long long x = v[i];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    if (lane == 0 && x) atomicAdd(&dst[i], x);
  }
#ifdef QA_STAMPS
  if (lane == 0) {
    cnt.sl[0] = __builtin_readcyclecounter() - tKernel;
    cnt.sl[9] = 1;
    for (int i = 0; i < 13; ++i) atomicAdd(&dst[6 + i], cnt.sl[i]);
  }
#endif
}

}  // namespace qa
