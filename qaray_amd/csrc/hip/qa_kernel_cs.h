// qa_kernel_cs.h — the megakernel for scenes whose meshes live in global memory, with COOPERATIVE mesh walks from ONE pool
// per query phase.
//
// Why.  On such scenes three quarters of qa_integrate's wave time are mesh walks entered with a third (or a seventh) of the
// lanes and lasting as long as the slowest of them (profiles/round02/megakernel_section_stamps.txt).  Round 2 let the whole
// wave walk ONE (ray type, light, mesh instance) at a time from a pool of (ray, node) items; what that left: a walk per
// instance and light - 18 of them per hit on project7_object, each with its own start-up, its own underfilled first and last
// rounds (35 % of all rounds held <= 16 items) - and 234 spilled registers for the rays and bookkeeping the lanes had to keep
// for each other.  Here a query PHASE (the closest-hit query of the wave's 64 paths, or the shadow queries of a batch of
// lights) fills one pool:
//   * Sweep.  Every instance is visited once (wave-uniform loop, the reference's pre-order).  Spheres and planes are
//     intersected on the spot.  A lane whose ray passes a mesh's bounding-box gate writes its NODE-LOCAL ray into a ray slot
//     in LDS (32 bytes: origin, limit, direction, owner lane | instance or light | box padding) and pushes the mesh's root.
//   * Rounds.  All 64 lanes pop one item each (node | ray slot), read the ray from its slot, and either test the node's four
//     children and push the entered ones (ballot + prefix, no atomics) or test the leaf's triangles with the reference's
//     arithmetic.  Items of every instance - and of every light - share the rounds; with <= 16 items four lanes share an item.
//     The 4-wide trees of all meshes live in ONE node array and ONE triangle array (DScene::csNodes / csTris, child words
//     rebased at upload), so an item needs no mesh descriptor.
//   * Closest hit: one 64-bit key per lane, distance bits << 32 | instance << 20 | element, updated with ds_min_u64 by whoever
//     accepts a triangle; every lane prunes and accepts against the distance half, so a hit in one instance shortens the
//     walks of all others at once (distances are world-parametric: rays are not renormalised in node space).  The minimum
//     over instances with the lower instance winning at equal distance IS the reference's sequential answer, provided the
//     winner would also have been reached by the reference's walk (qa_wf.h's argument): checked below.
//   * Any hit: the first accepted triangle of a (lane, light) pair is recorded and its remaining items die when popped.
// What makes the answer the reference's is hitMesh's argument unchanged (qa_kernel.h): the accepted triangle's leaf in the
// reference tree must pass the reference's strict box test at the found distance (t_max for a shadow ray); a triangle at
// exactly the distance held, a full pool, a mesh without the 4-wide tree or an origin beyond the pruned search's reach send
// the lane's query to the exact sequential walk of the whole scene graph on the reference's trees (csExactClosest /
// csExactShadow: out of line, private stacks - a handful of rays per million).  Shading and every random draw are
// qa_integrate's: frames are bit-identical.
// Kernel variants (template parameters; why they are variants and not branches: DESIGN.md 4d / 5 round 3 - every live value more
// in the sweeps is paid in spilled registers): CULL - the sweeps test a node's root-space bounds first and skip nodes no lane of
// the wave can meet; MANY - more than four shadow-casting lights, in batches of four with the surface parked in a global slab;
// AREA - area lights: hit log, all lights evaluated by the whole wave when its paths have ended, sample rays four per lane at a time.
//
// Replaces (reference file:line): Scene::TraceNodeNormal / TraceNodeShadow (src/scene/scene.cpp:35-74), GenLight::Shadow
// (src/lights/lights.cpp:39-48), TriObj::IntersectRay / TraceBVHNode (src/objects/objects.cpp:310-420); everything else as qa_kernel.h.
#pragma once
#include "qa_kernel.h"

namespace qa {

#ifdef QA_STAMPS
#define QA_FILL(cnt) (cnt).sl
#else
#define QA_FILL(cnt) nullptr
#endif
#define QA_CS_SLOT_SHIFT 20             /* pool item = child word | ray slot << 20 (inner: node index; leaf: flag, count - 1, triangle offset) */
#define QA_CS_INDEX_MASK 0xFFFFFu       /* scene-wide node indices / triangle offsets must fit 20 bits (host check), elements of a mesh too */
#define QA_CS_SLOT_MASK 0xFFu           /* <= 256 ray slots */
#define QA_CS_LIGHT_BATCH 4             /* shadow queries are pooled for up to four lights at a time */
#define QA_CS_EXACT_STACK 64            /* private stack entries of the exact walks (reference trees deeper than this keep qa_integrate) */
#ifndef QA_CS_LEAF_ROUND
#define QA_CS_LEAF_ROUND 48u           /* leaf items that make a leaf round go first */
#endif
#define QA_CS_RES_WORDS 256             /* per wave: 64 closest-hit keys (2 words each) or QA_CS_LIGHT_BATCH x 64 any-hit results */

// lanes below this one that are set in m (v_mbcnt: two instructions)
__device__ __forceinline__ uint32_t csLanePrefix(unsigned long long m)
{
  return __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
}
__device__ __forceinline__ void csWaveSync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The wave's share of the dynamic LDS (host: CsLdsWords in qa_capi.hip): [items | ray slots | results | flags | accumulators]
struct CsLds {
  uint32_t *items;               // [capItems]
  uint4 *rays;                   // [2 * slots]: (origin.xyz, limit), (direction.xyz, meta)
  uint32_t *res;                 // [QA_CS_RES_WORDS]
  uint32_t *flags;               // [64] closest: != 0 = repeat this lane's query exactly; any-hit: bit j = light j of the batch
  uint32_t capItems, slots;
};
__device__ __forceinline__ unsigned long long *csKeys(const CsLds &L) { return reinterpret_cast<unsigned long long *>(L.res); }

// meta word of a ray slot: owner lane | x << 6 (closest: instance, any-hit: light of the batch) | padding << 14.  The padding
// of the boxes rides as the float's upper 12 bits rounded UP (a larger padding is always safe).
__device__ __forceinline__ uint32_t csMeta(uint32_t owner, uint32_t x, float pad)
{
  const uint32_t enc = (__float_as_uint(pad) + 0x7FFFFu) >> 19;
  return owner | (x << 6) | (enc << 14);
}
__device__ __forceinline__ float csMetaPad(uint32_t meta) { return __uint_as_float((meta >> 14) << 19); }

// Rounds until the pool is empty.  CLOSEST: keys / flags per owner lane; else: results per (light of the batch, owner lane).
//
// The pool is TWO stacks sharing the item array: node items grow up from its start, leaf items down from its end.  A round is
// either a node round (every lane tests the four children of one node) or a leaf round (every lane tests the triangles of one
// leaf): the wave never executes both bodies for a mixed set of items (in round 2's single-stack form every round paid for
// both: ~390 vector instructions, of which a lane used half).  Leaf rounds go first once a wave's worth of leaves has
// gathered (a hit shortens every walk of its ray), node rounds otherwise.
template <bool CLOSEST>
__device__ __forceinline__ void csRun(const DScene &sc, const CsLds &L, uint32_t &nNode, unsigned long long *fill)
{
  const unsigned lane = __lane_id();
  const float INF = __builtin_inff();
  const uint4 *wn = sc.csNodes, *tris = sc.csTris;
  const uint32_t cap = L.capItems;
  uint32_t nLeaf = 0;
  csWaveSync();   // the sweep's ray slots and root items
#ifdef QA_STAMPS
  const unsigned long long tRun = __builtin_readcyclecounter();
#endif
  while (nNode | nLeaf) {
    const bool leafRound = nLeaf >= QA_CS_LEAF_ROUND || nNode == 0u;
    const uint32_t have = leafRound ? nLeaf : nNode;
    // With at most 16 items four lanes share an item: each tests ONE child of the node (or every fourth triangle of the
    // leaf), so an underfilled round costs a quarter of the arithmetic - and all items are taken.
    const bool quad = have <= 16u;
    const uint32_t take = quad ? have : (have < 64u ? have : 64u);
    const uint32_t idx = quad ? (lane >> 2) : lane, sub = lane & 3u;
#ifdef QA_STAMPS
    if (fill && lane == 0) { fill[10] += quad ? 4u * take : take; fill[11] += 1; fill[12] += leafRound ? 1 : 0; }   /* lanes at work / rounds / leaf rounds */
#endif
    const bool work = idx < take;
    // node items are popped from the top of their stack, leaf items from the low end of theirs (the most recent ones)
    const uint32_t item = work ? L.items[leafRound ? cap - nLeaf + idx : nNode - take + idx] : 0u;
    if (leafRound) nLeaf -= take;
    else nNode -= take;
    const uint32_t slot = (item >> QA_CS_SLOT_SHIFT) & QA_CS_SLOT_MASK;
    const uint4 ra = L.rays[2 * slot], rb = L.rays[2 * slot + 1];
    const f3 op = F3(asF(ra.x), asF(ra.y), asF(ra.z)), od = F3(asF(rb.x), asF(rb.y), asF(rb.z));
    const uint32_t meta = rb.w, owner = meta & 63u, mx = (meta >> 6) & 0xFFu;
    bool live;
    float hz;
    unsigned long long *key = csKeys(L) + owner;
    if (CLOSEST) {
      live = work && L.flags[owner] == 0;
      hz = __uint_as_float((uint32_t) (*key >> 32));   // the distance the owner's path holds right now
    } else {
      live = work && L.res[mx * 64u + owner] == 0 && ((L.flags[owner] >> mx) & 1u) == 0;   // dropped once the query is settled
      hz = asF(ra.w);
    }
    if (!leafRound) {
      // ---- node round
      const float opad = csMetaPad(meta);
      // Reciprocal direction for the box tests of the library's own tree: one ulp off the division at most, which the boxes'
      // padding covers many times over (1e-6 (|origin| + |mesh|) for the slab arithmetic, qa_widebvh.h); a zero component
      // gives +-inf and a NaN slab that min / max ignore (the axis stays unbounded: conservative).
      const f3 drcp = F3(__builtin_amdgcn_rcpf(od.x), __builtin_amdgcn_rcpf(od.y), __builtin_amdgcn_rcpf(od.z));
      float k0 = INF, k1 = INF, k2 = INF, k3 = INF;
      uint32_t w0 = QA_DONE, w1 = QA_DONE, w2 = QA_DONE, w3 = QA_DONE;
      if (live && (item & QA_BVH_LEAF_BIT)) {
        // (a mesh that is one leaf: its root goes over to the leaf stack)
        if (!quad || sub == 0) { k3 = 0.f; w3 = item & ~(QA_CS_SLOT_MASK << QA_CS_SLOT_SHIFT); }
      } else if (live) {
        const uint4 *nd = wn + 4 * (size_t) (item & QA_CS_INDEX_MASK);
        const uint4 q0 = ldGlobal(nd), q1 = ldGlobal(nd + 1), q2 = ldGlobal(nd + 2), q3 = ldGlobal(nd + 3);
        const f3 pLo = op + F3(opad, opad, opad), pHi = op - F3(opad, opad, opad);
        if (quad) {
          w3 = sub == 0 ? q3.x : sub == 1 ? q3.y : sub == 2 ? q3.z : q3.w;   // (every item is taken every round while quad: no order to keep)
          QA_WIDE_CHILD(k3, w3, sub)
        } else {
          w0 = q3.x; w1 = q3.y; w2 = q3.z; w3 = q3.w;
          QA_WIDE_CHILD(k0, w0, 0)
          QA_WIDE_CHILD(k1, w1, 1)
          QA_WIDE_CHILD(k2, w2, 2)
          QA_WIDE_CHILD(k3, w3, 3)
          if (CLOSEST) {
            // farthest first: a stack is popped from its top, so the nearest child of a node is looked at first.  (An any-hit
            // query has no order to keep; sorting there was measured: it costs more than it finds.)
            QA_WIDE_CE(k0, w0, k1, w1)
            QA_WIDE_CE(k2, w2, k3, w3)
            QA_WIDE_CE(k0, w0, k2, w2)
            QA_WIDE_CE(k1, w1, k3, w3)
            QA_WIDE_CE(k1, w1, k2, w2)
          }
        }
      }
      // children the ray enters: inner ones onto the node stack, leaves onto the leaf stack (ballot + lane prefix, no atomics).
      // Whether the pool has room is decided for the whole wave: without room (rarely) none of this push's items is stored and
      // their queries are repeated exactly.
#define QA_CS_PUSH(K, W)                                                                                     \
      {                                                                                                      \
        const bool p = K < INF, lf = (W & QA_BVH_LEAF_BIT) != 0;                                             \
        const unsigned long long mn = __ballot(p && !lf), ml = __ballot(p && lf);                            \
        const uint32_t cn = (uint32_t) __popcll(mn), cl = (uint32_t) __popcll(ml);                           \
        if (__builtin_expect(cn + cl > cap - nNode - nLeaf, 0)) {                                            \
          if (p) atomicOr(&L.flags[owner], CLOSEST ? 1u : (1u << mx));                                       \
        } else {                                                                                             \
          const uint32_t at = lf ? (cap - nLeaf - 1u) - csLanePrefix(ml) : nNode + csLanePrefix(mn);         \
          if (p) L.items[at] = W | (slot << QA_CS_SLOT_SHIFT);                                               \
          nNode += cn;                                                                                       \
          nLeaf += cl;                                                                                       \
        }                                                                                                    \
      }
      QA_CS_PUSH(k3, w3)
      if (!quad) {
        QA_CS_PUSH(k2, w2)
        QA_CS_PUSH(k1, w1)
        QA_CS_PUSH(k0, w0)
      }
#undef QA_CS_PUSH
    } else if (live) {
      // ---- leaf round: the leaf's triangles with the reference's arithmetic
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
          if (CLOSEST) {
            const unsigned long long mine = ((unsigned long long) __float_as_uint(hzl) << 32) | (unsigned long long) ((mx << QA_CS_SLOT_SHIFT) | (t2.w >> 2));
            const unsigned long long old = atomicMin(key, mine);
            // another lane accepted this very distance meanwhile: the one situation in which the ORDER of the tests decides
            if ((uint32_t) (old >> 32) == __float_as_uint(hzl) && old != mine) tl = true;
          } else {
            L.res[mx * 64u + owner] = ((slot << QA_CS_SLOT_SHIFT) | (first + i)) + 1u;   // several lanes may store for one query: any accepted triangle will do
            stop = true;
          }
        }
      }
      if (tl) atomicOr(&L.flags[owner], CLOSEST ? 1u : (1u << mx));
    }
    csWaveSync();
  }
#ifdef QA_STAMPS
  if (fill && lane == 0) fill[CLOSEST ? 3 : 6] += __builtin_readcyclecounter() - tRun;   /* the rounds alone */
#endif
}

// ---------------------------------------------------------------------------------------------
// The exact walks: the reference's sequential scene-graph loop on the reference's trees, one lane per ray, out of line.
// They decide WHO wins (distance, instance, element); the hit's details are computed by the common code below.
// ---------------------------------------------------------------------------------------------
struct CsWinner { float z; int k; uint32_t tri; };

__device__ __forceinline__ Ray csRootRay(const qa_instance *inst, uint32_t rootIdentity, const Ray &world)
{
  if (!rootIdentity) return toNode(ldTable(inst), world);
  Ray o;
  o.p = world.p;
  o.d = (world.p + world.d) - world.p;
  return o;
}
__device__ __forceinline__ Ray csLocalRay(const qa_instance *inst, int k, const qa_instance &in, const Ray &r0)
{
  if (in.depth == 1) return toNode(in, r0);
  int chain[QA_MAX_NODE_DEPTH];
  int n = 0;
  for (int a = k; a > 0 && n < QA_MAX_NODE_DEPTH; a = ldTable(inst + a).parent) chain[n++] = a;
  Ray r = r0;
  for (int q = n - 1; q >= 0; --q) r = toNode(ldTable(inst + chain[q]), r);
  return r;
}
// TriObj::IntersectRay on the reference tree (hitMesh's STATS branch without the counters)
__device__ __forceinline__ bool csExactMesh(const DMesh &m, const Ray &ray, float &hz, bool closest, uint32_t *stk, uint32_t &best)
{
  const f3 drcp = F3(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
  const bool fastSlab = !__any(qabs(ray.d.x) < 1e-7f || qabs(ray.d.y) < 1e-7f || qabs(ray.d.z) < 1e-7f);
  float entry, meshExit;
  if (fastSlab) boxEntryExitFast(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
  else boxEntryExit(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
  if (entry > hz || entry > meshExit) return false;   // Box::IntersectRay, src/core/box.cpp:94-128
  if (m.num_faces == 0) return false;
  DCounters none = {};
  bool tie = false;
  return walkBVH<false, false, true, 1>(reinterpret_cast<const uint4 *>(m.nodes), reinterpret_cast<const uint4 *>(m.tris), m.rootData, ray, drcp, fastSlab,
                                        hz, closest, stk, none, best, tie);
}

__device__ __attribute__((noinline)) CsWinner csExactClosest(const qa_instance *inst, const DMesh *mesh, int num_inst, uint32_t rootIdentity, float px, float py,
                                                             float pz, float dx, float dy, float dz)
{
  uint32_t stk[QA_CS_EXACT_STACK];
  Ray world;
  world.p = F3(px, py, pz);
  world.d = F3(dx, dy, dz);
  const Ray r0 = csRootRay(inst, rootIdentity, world);
  CsWinner w;
  w.z = QA_BIGFLOAT;
  w.k = -1;
  w.tri = 0;
  for (int k = 1; k < num_inst; ++k) {
    const qa_instance in = ldTable(inst + k);
    if (in.obj_type == QA_OBJ_NONE) continue;
    const Ray r = csLocalRay(inst, k, in, r0);
    Hit h;
    h.z = w.z;
    h.node = -1;
    if (in.obj_type == QA_OBJ_SPHERE) {
      if (hitSphere(r, h, k, false)) { w.z = h.z; w.k = k; }
    } else if (in.obj_type == QA_OBJ_PLANE) {
      if (hitPlane(r, h, k, false)) { w.z = h.z; w.k = k; }
    } else {
      const DMesh m = ldTable(mesh + in.mesh);
      uint32_t best = 0;
      if (csExactMesh(m, r, h.z, true, stk, best)) { w.z = h.z; w.k = k; w.tri = best; }
    }
  }
  return w;
}

__device__ __attribute__((noinline)) bool csExactShadow(const qa_instance *inst, const DMesh *mesh, int num_inst, uint32_t rootIdentity, float px, float py, float pz,
                                                        float dx, float dy, float dz, float tmax)
{
  uint32_t stk[QA_CS_EXACT_STACK];
  Ray world;
  world.p = F3(px, py, pz);
  world.d = F3(dx, dy, dz);
  const Ray r0 = csRootRay(inst, rootIdentity, world);
  for (int k = 1; k < num_inst; ++k) {
    const qa_instance in = ldTable(inst + k);
    if (in.obj_type == QA_OBJ_NONE) continue;
    const Ray r = csLocalRay(inst, k, in, r0);
    Hit h;
    h.z = tmax;
    h.node = -1;
    bool hit;
    if (in.obj_type == QA_OBJ_SPHERE) hit = hitSphere(r, h, k, false);
    else if (in.obj_type == QA_OBJ_PLANE) hit = hitPlane(r, h, k, false);
    else {
      const DMesh m = ldTable(mesh + in.mesh);
      uint32_t best = 0;
      hit = csExactMesh(m, r, h.z, false, stk, best);
    }
    if (hit) return true;
  }
  return false;
}

// ---------------------------------------------------------------------------------------------
// Sweep helpers
// ---------------------------------------------------------------------------------------------
// The bounding-box gate of TriObj::IntersectRay (src/objects/objects.cpp:310-322, Box::IntersectRay src/core/box.cpp:94-128)
template <class M>
__device__ __forceinline__ bool csGate(const M &m, const Ray &ray, float limit)
{
  const f3 drcp = F3(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
  const bool nearZero = qabs(ray.d.x) < 1e-7f || qabs(ray.d.y) < 1e-7f || qabs(ray.d.z) < 1e-7f;
  float entry, meshExit;
  if (nearZero) boxEntryExit(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
  else boxEntryExitFast(ray, drcp, ld3(m.bmin), ld3(m.bmax), entry, meshExit);
  return !(entry > limit || entry > meshExit) && m.num_faces != 0;
}

// The any-hit results of one run of the pool: a recorded triangle counts when the reference's walk reaches it - its leaf in
// the reference tree passes the strict box test against the ray's fixed t_max (exact for an any-hit query: the reference
// reports "occluded" iff it reaches an accepted triangle); if it does not, only the sequential walk can tell.
__device__ __forceinline__ void csSettleShadows(const DScene &sc, const CsLds &L, uint32_t nb, uint32_t &occl, uint32_t &redo)
{
  const unsigned lane = __lane_id();
  redo |= L.flags[lane];
  for (uint32_t jj = 0; jj < nb; ++jj) {
    const uint32_t f = L.res[jj * 64u + lane];
    if (!__any(f != 0)) continue;
    if (f != 0 && !((occl | redo) >> jj & 1u)) {
      const uint32_t slot = ((f - 1u) >> QA_CS_SLOT_SHIFT) & QA_CS_SLOT_MASK, gtri = (f - 1u) & QA_CS_INDEX_MASK;
      const uint4 ra = L.rays[2 * slot], rb = L.rays[2 * slot + 1];
      Ray r;
      r.p = F3(asF(ra.x), asF(ra.y), asF(ra.z));
      r.d = F3(asF(rb.x), asF(rb.y), asF(rb.z));
      const uint4 b0 = ldGlobal(sc.csLeafBox + 2 * (size_t) gtri), b1 = ldGlobal(sc.csLeafBox + 2 * (size_t) gtri + 1);
      bool reached = asF(b1.z) != 0.f;   // the leaf is the root: entered unconditionally
      if (!reached) {
        const f3 drcp = F3(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
        const bool nearZero = qabs(r.d.x) < 1e-7f || qabs(r.d.y) < 1e-7f || qabs(r.d.z) < 1e-7f;
        const f3 bmin = F3(asF(b0.x), asF(b0.y), asF(b0.z)), bmax = F3(asF(b0.w), asF(b1.x), asF(b1.y));
        float entry, exit_;
        if (nearZero) boxEntryExit(r, drcp, bmin, bmax, entry, exit_);
        else boxEntryExitFast(r, drcp, bmin, bmax, entry, exit_);
        reached = entry < asF(ra.w) && entry < exit_;
      }
      if (reached) occl |= 1u << jj;
      else redo |= 1u << jj;
    }
  }
  csWaveSync();   // results and ray slots have been read: the next run may overwrite them
  for (uint32_t jj = 0; jj < nb; ++jj) L.res[jj * 64u + lane] = 0;
  L.flags[lane] = 0;
}

// Node::ToNodeCoords through the levels of a flat instance record (toNode / localRay of qa_kernel.h: the same operations)
__device__ __forceinline__ Ray csToNode(const float *itm, const float *pos, const Ray &r)
{
  const f3 o = ld3(pos);
  Ray out;
  out.p = mulMV(itm, r.p - o);
  out.d = mulMV(itm, (r.p + r.d) - o) - out.p;
  return out;
}
__device__ __forceinline__ Ray csLocal(const CsInst &ci, const Ray &r0)
{
  Ray r = csToNode(ci.itmA, ci.posA, r0);
  if (ci.depth == 2) r = csToNode(ci.itmB, ci.posB, r);
  return r;
}
// The same in a sweep over the nodes in pre-order: the children of a group follow each other, so the group's own ray is kept
// from one child to the next (five walls in one group: the level-A transform once instead of five times; same operations)
__device__ __forceinline__ Ray csLocalSweep(const CsInst &ci, const Ray &r0, GroupRay &g)
{
  if (ci.depth != 2) return csToNode(ci.itmA, ci.posA, r0);
  if (ci.parent != g.node) {
    g.ray = csToNode(ci.itmA, ci.posA, r0);
    g.node = ci.parent;
  }
  return csToNode(ci.itmB, ci.posB, g.ray);
}
// Node::FromNodeCoords at every level from the hit node up to and including the (identity) root (src/core/node.cpp:127-139)
__device__ __forceinline__ void csToWorld(const CsInst &ci, f3 &p, f3 &N)
{
  if (ci.depth == 2) {
    p = mulMV(ci.tmB, p) + ld3(ci.posB);
    N = normalize(mulTMV(ci.itmB, N));
  }
  p = mulMV(ci.tmA, p) + ld3(ci.posA);
  N = normalize(mulTMV(ci.itmA, N));
  N = normalize(N);   // identity root: p unchanged, the normal is still re-normalised
}

// A lane with `coop` takes ray slot nSlots + (its rank among the entering lanes) and pushes the mesh's root there.
__device__ __forceinline__ void csEnter(const CsLds &L, unsigned long long mk, bool coop, const Ray &r, float limit, uint32_t x, float pad, uint32_t rootWord, uint32_t n,
                                        uint32_t nSlots)
{
  const unsigned lane = __lane_id();
  if (coop) {
    const uint32_t at = csLanePrefix(mk);
    const uint32_t slot = nSlots + at;
    L.rays[2 * slot] = make_uint4(__float_as_uint(r.p.x), __float_as_uint(r.p.y), __float_as_uint(r.p.z), __float_as_uint(limit));
    L.rays[2 * slot + 1] = make_uint4(__float_as_uint(r.d.x), __float_as_uint(r.d.y), __float_as_uint(r.d.z), csMeta(lane, x, pad));
    L.items[n + at] = rootWord | (slot << QA_CS_SLOT_SHIFT);
  }
}
template <class M>
__device__ __forceinline__ float csPad(const M &m, f3 o)
{
  const float oMax = qmax(qmax(qabs(o.x), qabs(o.y)), qabs(o.z));
  return m.nearPad + (QA_SLACK_SCALE * 1e-6f) * (oMax + m.absMax);
}

// ---------------------------------------------------------------------------------------------
// Instance culling.  The reference visits every node for every ray (Scene::TraceNodeNormal / TraceNodeShadow,
// src/scene/scene.cpp:35-74; it computes a child bounding box at src/parser/xmlload.cpp:105 and never uses it); nothing requires
// reproducing that cost.  A node's object can only be hit inside its bounds, so a ray that misses the object's ROOT-space box
// (DScene::csCull), or enters it beyond the distance held, skips the node: no record loads, no transform, no test - when the
// whole wave skips it; a lane that misses while others do not sits the node out, which changes nothing either.
// How much the box must be widened: an accepted hit is a point o_n + d_n * t of the NODE-space ray, which the fp32 transforms
// (Node::ToNodeCoords: itm * (p - pos) and itm * ((p + d) - pos) - ...) have moved by up to ~4 eps |itm| (|o| + |pos|) in the
// origin and ~8 eps |itm| (|o| + |d| + |pos|) in the direction - the latter times t, which is at most the distance to the far
// side of the box.  Back in root space that is <= 14 eps cond(tm) (oMax + |pos| + 1) (oMax + |box|): csCullK3 holds 2e-5 cond(tm)
// (20 x that), csCullK4 the meshes' own acceptance slack (DMesh::nearPad) and the rounding of this test's slab arithmetic
// (approximate reciprocals: a few ulp).  NaNs (a zero direction component against a box face through the origin) never cull.
// ---------------------------------------------------------------------------------------------
struct CsCullRay { f3 rd; float pad; };   // (four values alive during a sweep; the origin is the ray's own)
__device__ __forceinline__ CsCullRay csCullRay(const DScene &sc, const Ray &r0)
{
  const float oMax = qmax(qmax(qabs(r0.p.x), qabs(r0.p.y)), qabs(r0.p.z));
  CsCullRay c;
  c.pad = __builtin_fmaf((oMax + sc.csCullS1) * (oMax + sc.csCullS2), sc.csCullK3, sc.csCullK4);
  c.rd = F3(__builtin_amdgcn_rcpf(r0.d.x), __builtin_amdgcn_rcpf(r0.d.y), __builtin_amdgcn_rcpf(r0.d.z));
  return c;
}
__device__ __forceinline__ bool csCullPass(const CsCull &b, const CsCullRay &c, f3 o, float limit)
{
  const f3 pd = F3(c.pad, c.pad, c.pad);
  const f3 p0 = ((ld3(b.lo) - pd) - o) * c.rd, p1 = ((ld3(b.hi) + pd) - o) * c.rd;   // (the box widened: scalar operands, the additions are scalar-vector)
  const float en = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(p0.x, p1.x), __builtin_fminf(p0.y, p1.y)), __builtin_fminf(p0.z, p1.z));
  const float ex = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(p0.x, p1.x), __builtin_fmaxf(p0.y, p1.y)), __builtin_fmaxf(p0.z, p1.z));
  return !(en > limit || en > ex || ex < 0.f);
}

// Scene::TraceNodeNormal (traceClosest of qa_kernel.h) for the lanes with `act`; every lane of the wave calls this.
// The x / y differential directions of the path's current ray (DiffRay, src/core/ray.h:55-63): camera rays - the pixel sample
// shifted by DiffRay::dx / dy (renderer.cpp:314-317; the ray's origin is the lens point) - secondary rays - both equal the
// ray's direction.  A function of what the path keeps anyway, so it is evaluated when a textured hit needs it.
__device__ __forceinline__ f3 csTexPos(const DScene &sc, uint32_t pxy, int sidx)
{
  return F3(sc.halton[2 * sidx], sc.halton[2 * sidx + 1], 0.f) + F3((float) (pxy & 0xFFFFu), (float) (pxy >> 16), 0.f);
}
__device__ __forceinline__ RayDiff csRayDiff(const DScene &sc, const Ray &world, bool primary, uint32_t pxy, int sidx)
{
  RayDiff wd;
  wd.dx = wd.dy = world.d;
  if (primary) {
    const f3 texpos = csTexPos(sc, pxy, sidx);
    const f3 A = ld3(sc.cam.screenA), U = ld3(sc.cam.screenU), V = ld3(sc.cam.screenV);
    const f3 xpt = (A + U * (texpos.x + QA_DX)) + V * texpos.y;
    const f3 ypt = (A + U * texpos.x) + V * (texpos.y + QA_DX);
    wd.dx = normalize(xpt - world.p);
    wd.dy = normalize(ypt - world.p);
  }
  return wd;
}

template <bool TEX, bool CULL>
__device__ __forceinline__ bool csTraceClosest(const DScene &sc, const CsLds &L, bool act, const Ray &world, bool primary, const float *cols, Hit &h, TexHit &th,
                                               DCounters &cnt)
{
  if (!__any(act)) return false;
  const unsigned lane = __lane_id();
  cnt.casts_normal += (unsigned long long) __popcll(__ballot(act));   // (wave-uniform tallies: no registers per lane)
  const Ray r0 = rootRay<false>(sc, world);
  // ---- ONE sweep over the instances in the reference's order: spheres and planes on the spot (distance and node only; the
  // winner's details below), meshes into the pool; the pool runs when it is full, and at the end.
  float bestZ = QA_BIGFLOAT;   // closest sphere / plane so far
  int bestK = -1;
  f3 spP = F3(0, 0, 0), spN = F3(0, 0, 0);   // !TEX: that hit in node space (TEX: its details wait for the winner, below)
  bool spFront = true;
  bool exact = false;          // this lane's query goes to the exact walk
  csKeys(L)[lane] = ((unsigned long long) __float_as_uint(QA_BIGFLOAT) << 32) | 0xFFFFFFFFull;
  L.flags[lane] = 0;
  uint32_t n = 0, nSlots = 0;
  GroupRay grp;
  grp.node = -1;
  grp.ray = r0;
  QA_T(tSweep)
  // the walks prune against what the spheres and planes have already settled
#define QA_CS_SETTLE_SOLIDS                                                                                                     \
  if (bestK >= 0) {                                                                                                             \
    unsigned long long *key = csKeys(L) + lane;                                                                                 \
    const unsigned long long mine = ((unsigned long long) __float_as_uint(bestZ) << 32) | 0xFFFFFFFFull, old = *key;            \
    if ((uint32_t) (old >> 32) == __float_as_uint(bestZ) && old != mine) L.flags[lane] = 1; /* a triangle at exactly that distance: order decides */ \
    if (mine < old) *key = mine;                                                                                                \
  }
  const CsCullRay cull = csCullRay(sc, r0);
  for (int k = 1; k < sc.num_inst; ++k) {
    bool in = act;   // this lane's ray can meet node k's object
    if (CULL && sc.csCullOn) {
      in = act && csCullPass(ldTable(sc.csCull + k), cull, r0.p, bestZ);
      if (!__any(in)) continue;
    }
    const CsInst ci = ldTable(sc.csInst + k);
    const int type = ci.type;
    if (type == QA_OBJ_NONE) continue;
    const Ray r = csLocalSweep(ci, r0, grp);
    if (type != QA_OBJ_MESH) {
      Hit hh;
      hh.z = bestZ;
      hh.node = -1;
      hh.p = hh.N = F3(0, 0, 0);
      hh.front = true;
      const bool hit = in && (type == QA_OBJ_SPHERE ? hitSphere(r, hh, k, !TEX) : hitPlane(r, hh, k, !TEX));
      if (hit) {
        bestZ = hh.z;
        bestK = k;
        if (!TEX) {
          spP = hh.p;
          spN = hh.N;
          spFront = hh.front;
        }
      }
      continue;
    }
    // (gate against the spheres and planes met so far: any limit not below the final answer is safe, and a triangle at
    // exactly the distance held raises the flag whatever the gate saw)
    const bool go = in && csGate(ci, r, bestZ);
    const bool coop = go && ci.useWide && insideCancelReach(ci, r.p);
    exact = exact || (go && !coop);
    const unsigned long long mk = __ballot(coop);
    const uint32_t c = (uint32_t) __popcll(mk);
    if (!c) continue;
    // The pool runs when it cannot take this instance's rays - rarely: the run that matters is the one after the sweep, where
    // nothing of the sweep is alive any more (the branch weight keeps the allocator's spill code out of the common path)
    if (__builtin_expect(nSlots + c > L.slots || n + c > L.capItems, 0)) {
      QA_CS_SETTLE_SOLIDS
      csRun<true>(sc, L, n, QA_FILL(cnt));
      nSlots = 0;
    }
    csEnter(L, mk, coop, r, 0.f, (uint32_t) k, csPad(ci, r.p), ci.csRootWord, n, nSlots);
    n += c;
    nSlots += c;
  }
  if (n != 0) {
    QA_CS_SETTLE_SOLIDS
    csRun<true>(sc, L, n, QA_FILL(cnt));
  }
#undef QA_CS_SETTLE_SOLIDS
  QA_TACC(cnt.sl[13], tSweep)
  QA_T(tDet)
  uint32_t elem;
  float meshZ;
  {
    const unsigned long long key = csKeys(L)[lane];
    exact = exact || L.flags[lane] != 0 || (act && (sc.csForceExact & 1u));   // (option "cs_force_exact": tests of the exact walks)
    elem = (uint32_t) key;
    meshZ = __uint_as_float((uint32_t) (key >> 32));
    // a sphere or plane met after the last run of the pool
    if (elem != 0xFFFFFFFFu && bestK >= 0) {
      if (meshZ == bestZ) exact = true;
      else if (bestZ < meshZ) elem = 0xFFFFFFFFu;
    }
    csWaveSync();   // the keys have been read: the region may be reused
  }
  // ---- who wins
  h.z = QA_BIGFLOAT;
  h.node = -1;
  h.mtlID = 0;
  h.front = true;
  h.p = h.N = F3(0, 0, 0);
  th.uvw = F3(0.5f, 0.5f, 0.5f);   // HitInfo::Init (src/core/hitinfo.cpp:31-42)
  th.duvw0 = th.duvw1 = F3(0, 0, 0);
  th.hasTexture = false;
  int winK = bestK;
  float z = bestZ;
  uint32_t tri = 0;
  bool check = false;   // a mesh hit from the pool: the order check below
  bool isMesh = false;
  if (act && !exact && elem != 0xFFFFFFFFu) {
    winK = (int) ((elem >> QA_CS_SLOT_SHIFT) & 0xFFu);
    tri = elem & QA_CS_INDEX_MASK;
    z = meshZ;
    check = true;
    isMesh = true;
  }
  bool found = false;
  if (!TEX) {
    // a sphere or plane won: its node-space hit goes to world space now - once, every lane through ITS node's record (the
    // rays of a wave hit different walls: a loop over the nodes would repeat this for each of them)
    if (act && !exact && !isMesh && bestK >= 0) {
      csToWorld(sc.csInst[bestK], spP, spN);
      h.z = bestZ;
      h.p = spP;
      h.N = spN;
      h.front = spFront;
      h.node = bestK;
      h.mtlID = 0;
      found = true;
    }
  }
  // ---- details of the other winners, per instance (wave-uniform k: the tables are scalar loads).  Two rounds at most: a lane
  // whose pool answer fails the order check repeats its query exactly and is then finished like the others.
  bool pending = act && !exact && winK >= 0 && !found;
  for (int round = 0; round < 2; ++round) {
    if (round == 1) {
      if (!__any(exact)) break;
#ifdef QA_STAMPS
      if (lane == 0) cnt.sl[15] += (unsigned long long) __popcll(__ballot(exact));
#endif
      if (exact) {
        const CsWinner w = csExactClosest(sc.inst, sc.mesh, sc.num_inst, sc.rootIdentity, world.p.x, world.p.y, world.p.z, world.d.x, world.d.y, world.d.z);
        winK = w.k;
        z = w.z;
        tri = w.tri;
        check = false;
        found = false;
        pending = act && winK >= 0;
      }
      exact = false;
    }
    if (!__any(pending)) continue;
    RayDiff wd;
    wd.dx = wd.dy = world.d;
    if (TEX) wd = csRayDiff(sc, world, primary, __float_as_uint(cols[12 * 64]), __float_as_int(cols[14 * 64]));   // (pixel and sample index: the wave's LDS columns)
    for (int k = 1; k < sc.num_inst; ++k) {
      const bool mine = pending && winK == k;
      if (!__any(mine)) continue;
      const CsInst ci = ldTable(sc.csInst + k);
      const Ray r = csLocal(ci, r0);
      bool ok = mine;
      if (ci.type == QA_OBJ_SPHERE) {
        if (mine) {
          // Sphere::IntersectRay's accepted hit (hitSphere, closest = true) at the root found above
          const f3 p = r.p + r.d * z;
          const f3 N = normalize(p);
          h.p = p;
          h.N = N;
          h.front = (dot(N, r.d) <= 0);
          h.mtlID = 0;
          if (TEX) {
            Ray r2;
            RayDiff rd;
            localRayDiff<false>(sc, k, world, wd, r2, rd);
            texSphere(r.p, rd.dx, rd.dy, h.p, h.N, th);
          }
        }
      } else if (ci.type == QA_OBJ_PLANE) {
        if (mine) {
          h.p = r.p + r.d * z;
          h.N = F3(0, 0, 1);
          h.front = (dot(h.N, r.d) <= 0);
          h.mtlID = 0;
          if (TEX) {
            Ray r2;
            RayDiff rd;
            localRayDiff<false>(sc, k, world, wd, r2, rd);
            texPlane(r.p, rd.dx, rd.dy, h.p, th);
          }
        }
      } else {
        const DMesh m = meshAt<false>(sc, (int) ci.mesh);
        const uint4 *nodes = reinterpret_cast<const uint4 *>(m.nodes), *tris = reinterpret_cast<const uint4 *>(m.tris), *shade = reinterpret_cast<const uint4 *>(m.shade);
        if (mine) {
          const uint4 *s = shade + 3 * (size_t) tri;
          const uint4 s0 = ldGlobal(s), s1 = ldGlobal(s + 1), s2 = ldGlobal(s + 2);
          if (check) {
            // would the reference's walk have reached this triangle?  (refReaches: its leaf must pass the strict box test at the
            // found distance; the reference's running distance is never smaller.)  If not, only the sequential walk can tell.
            const f3 drcp = F3(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
            const bool nearZero = qabs(r.d.x) < 1e-7f || qabs(r.d.y) < 1e-7f || qabs(r.d.z) < 1e-7f;
            if (!refReaches<true>(nodes, s2.w /* DTriShade::pad */, r, drcp, !nearZero, z)) { ok = false; exact = true; }
          }
          if (ok) {
            float ba = 0, bb = 0;
            h.z = z;
            const uint4 *t = tris + 3 * (size_t) tri;
            const uint4 t0 = ldGlobal(t), t1 = ldGlobal(t + 1), t2 = ldGlobal(t + 2);
            triangleDetails(t0, t1, t2, r, h, ba, bb);
            // shading normal: TriMesh::GetNormal (src/mesh/TriMesh.h:196-204), left un-normalised
            const float bc = 1.f - ba - bb;
            const f3 n0 = F3(asF(s0.x), asF(s0.y), asF(s0.z)), n1 = F3(asF(s0.w), asF(s1.x), asF(s1.y)), n2 = F3(asF(s1.z), asF(s1.w), asF(s2.x));
            h.N = (n0 * ba + n1 * bb) + n2 * bc;
            h.mtlID = (int) s2.y;
            if (TEX && m.hasVT) {
              Ray r2;
              RayDiff rd;
              localRayDiff<false>(sc, k, world, wd, r2, rd);
              texTriangle(t0, t1, t2, m.vt + 6 * (size_t) tri, r.p, rd.dx, rd.dy, ba, bb, th);
            }
          }
        }
      }
      if (ok) {
        h.z = z;
        h.node = k;
        found = true;
        csToWorld(ci, h.p, h.N);
      }
    }
    pending = false;
  }
  QA_TACC(cnt.sl[14], tDet)
  return found;
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

// The sweep of ONE shadow ray per lane (slot jj of the running batch: nb slots so far) over the scene graph: spheres and planes on the
// spot, meshes into the pool; the pool runs in between only when it cannot take an instance's rays.
template <bool CULL>
__device__ __forceinline__ void csShadowSweep(const DScene &sc, const CsLds &L, bool lit, const Ray &w, float tmax, uint32_t jj, uint32_t nb, uint32_t &occl,
                                              uint32_t &redo, uint32_t &n, uint32_t &nSlots, DCounters &cnt)
{
  const Ray r0 = rootRay<false>(sc, w);
  GroupRay grp;
  grp.node = -1;
  grp.ray = r0;
  const CsCullRay cull = csCullRay(sc, r0);
  for (int k = 1; k < sc.num_inst; ++k) {
    bool open = lit && !(((occl | redo) >> jj) & 1u);   // this lane's query is still undecided
    if (!__any(open)) break;                             // settled for the whole wave: next ray
    if (CULL && sc.csCullOn) {
      open = open && csCullPass(ldTable(sc.csCull + k), cull, r0.p, tmax);   // ... and its ray can meet node k's object
      if (!__any(open)) continue;
    }
    const CsInst ci = ldTable(sc.csInst + k);
    const int type = ci.type;
    if (type == QA_OBJ_NONE) continue;
    const Ray r = csLocalSweep(ci, r0, grp);
    if (type != QA_OBJ_MESH) {
      Hit hh;
      hh.z = tmax;
      hh.node = -1;
      const bool hit = open && (type == QA_OBJ_SPHERE ? hitSphere(r, hh, k, false) : hitPlane(r, hh, k, false));
      if (hit) occl |= 1u << jj;
      continue;
    }
    const bool go = open && csGate(ci, r, tmax);
    bool coop = go && ci.useWide && insideCancelReach(ci, r.p);
    if (go && !coop) redo |= 1u << jj;
    unsigned long long mk = __ballot(coop);
    uint32_t c = (uint32_t) __popcll(mk);
    if (!c) continue;
    // (rarely: see csTraceClosest)
    if (__builtin_expect(nSlots + c > L.slots || n + c > L.capItems, 0)) {
      csRun<false>(sc, L, n, QA_FILL(cnt));
      csSettleShadows(sc, L, nb, occl, redo);   // before the slots are reused
      nSlots = 0;
      coop = coop && !(((occl | redo) >> jj) & 1u);   // queries that run decided do not enter
      mk = __ballot(coop);
      c = (uint32_t) __popcll(mk);
      if (!c) continue;
    }
    csEnter(L, mk, coop, r, tmax, jj, csPad(ci, r.p), ci.csRootWord, n, nSlots);
    n += c;
    nSlots += c;
  }
}

// GenLight::Shadow -> Scene::TraceNodeShadow for the next batch of up to QA_CS_LIGHT_BATCH non-ambient lights (table index li
// onwards; on return li is where the following batch starts and nb the lights taken), for the lanes with `lit`: bit jj of the
// result = light jj of the batch occluded.  The reference stops at the first node that occludes; which one does not matter.
// The pool runs once, after the sweeps of all the batch's lights (and, rarely, in between when it cannot take an instance's rays).
// `need`: bit jj = the lane's term of light jj is not zero in every component.  A light whose unshadowed term is (+-0, +-0, +-0) - the
// surface faces away from it: cosNL = max(0, N.L) = 0 - adds the same zero whether it is occluded or not (csLightSum), so its shadow
// ray is counted (the reference casts it) and not walked.
template <bool CULL>
__device__ __forceinline__ uint32_t csShadowBatch(const DScene &sc, const CsLds &L, bool lit, uint32_t need, f3 p, int &li, uint32_t &nb, DCounters &cnt)
{
  const unsigned lane = __lane_id();
  uint32_t occl = 0, redo = 0;
  for (uint32_t jj = 0; jj < QA_CS_LIGHT_BATCH; ++jj) L.res[jj * 64u + lane] = 0;
  L.flags[lane] = 0;
  uint32_t n = 0, nSlots = 0;
  nb = 0;
  QA_T(tSweep)
  for (; li < sc.num_lights && nb < QA_CS_LIGHT_BATCH; ++li) {
    const qa_light l = ldTable(sc.light + li);
    if (l.type == QA_LIGHT_AMBIENT) continue;
    const uint32_t jj = nb++;
    Ray w;
    float tmax;
    csShadowRay(l, p, w, tmax);
    cnt.casts_shadow += (unsigned long long) __popcll(__ballot(lit));
    csShadowSweep<CULL>(sc, L, lit && ((need >> jj) & 1u), w, tmax, jj, nb, occl, redo, n, nSlots, cnt);
  }
  if (n != 0) {
    csRun<false>(sc, L, n, QA_FILL(cnt));
    csSettleShadows(sc, L, nb, occl, redo);
  }
  QA_TACC(cnt.sl[17], tSweep)
  // ---- the exact repeats
  if (sc.csForceExact & 2u) { occl = 0; redo = lit ? need & ((1u << nb) - 1u) : 0u; }   // (option "cs_force_exact": tests of the exact walks)
  redo &= ~occl;
#ifdef QA_STAMPS
  if (lane == 0) cnt.sl[16] += (unsigned long long) __popcll(__ballot(lit && redo != 0));
#endif
  if (__any(lit && redo != 0)) {
    int lj = li;   // walk the batch's lights backwards from where it ended
    for (uint32_t j = nb; j-- > 0;) {
      do { --lj; } while (ldTable(sc.light + lj).type == QA_LIGHT_AMBIENT);
      const bool mine = lit && ((redo >> j) & 1u);
      if (!__any(mine)) continue;
      const qa_light l = ldTable(sc.light + lj);
      Ray wj;
      float tj;
      csShadowRay(l, p, wj, tj);
      if (mine && csExactShadow(sc.inst, sc.mesh, sc.num_inst, sc.rootIdentity, wj.p.x, wj.p.y, wj.p.z, wj.d.x, wj.d.y, wj.d.z, tj)) occl |= 1u << j;
    }
  }
  return occl;
}

// The same for up to four GIVEN shadow rays per lane from one origin (directions normalised, slot s for the lanes with bit s of
// `mask`): the samples of an area light (AREA variants).  Bit s of the result = ray s occluded.
struct CsRays4 { f3 d0, d1, d2, d3; float t0, t1, t2, t3; };
template <bool CULL>
__device__ __forceinline__ uint32_t csShadowRays(const DScene &sc, const CsLds &L, uint32_t mask, uint32_t nrays, f3 p, const CsRays4 &q, DCounters &cnt)
{
  const unsigned lane = __lane_id();
  uint32_t occl = 0, redo = 0;
  for (uint32_t jj = 0; jj < QA_CS_LIGHT_BATCH; ++jj) L.res[jj * 64u + lane] = 0;
  L.flags[lane] = 0;
  uint32_t n = 0, nSlots = 0;
#define QA_CS_RAY(S, D, T)                                                                            \
  if (nrays > S) {                                                                                    \
    Ray w;                                                                                            \
    w.p = p;                                                                                          \
    w.d = D;                                                                                          \
    cnt.casts_shadow += (unsigned long long) __popcll(__ballot(((mask >> S) & 1u) != 0));            \
    csShadowSweep<CULL>(sc, L, ((mask >> S) & 1u) != 0, w, T, S, S + 1u, occl, redo, n, nSlots, cnt); \
  }
  QA_CS_RAY(0u, q.d0, q.t0)
  QA_CS_RAY(1u, q.d1, q.t1)
  QA_CS_RAY(2u, q.d2, q.t2)
  QA_CS_RAY(3u, q.d3, q.t3)
#undef QA_CS_RAY
  if (n != 0) {
    csRun<false>(sc, L, n, QA_FILL(cnt));
    csSettleShadows(sc, L, nrays, occl, redo);
  }
  if (sc.csForceExact & 2u) { occl = 0; redo = mask; }
  redo &= ~occl & mask;
  if (__any(redo != 0)) {
#define QA_CS_REDO(S, D, T)                                                                                                                            \
    if (((redo >> S) & 1u) && csExactShadow(sc.inst, sc.mesh, sc.num_inst, sc.rootIdentity, p.x, p.y, p.z, D.x, D.y, D.z, T)) occl |= 1u << S;
    QA_CS_REDO(0u, q.d0, q.t0)
    QA_CS_REDO(1u, q.d1, q.t1)
    QA_CS_REDO(2u, q.d2, q.t2)
    QA_CS_REDO(3u, q.d3, q.t3)
#undef QA_CS_REDO
  }
  return occl & mask;
}

// One light's term of MtlBlinn_PhotonMap::Shade's light loop (directLight of qa_kernel.h) with the shadow factor known
// (1.0f multiplies exactly, 0.0f gives the same signed zeros)
__device__ __forceinline__ f3 csLightTerm(const qa_light &l, float normCoefDI, f3 p, f3 N, f3 V, f3 kd, f3 ks, float gloss, bool occluded)
{
  const float vis = occluded ? 0.0f : 1.0f;
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
  return (intensity * cosNL) * brdf;
}

// Direct lighting in two halves around the shadow queries (MtlBlinn_PhotonMap.cpp:481-498).  csLightTerms evaluates, BEFORE the
// shadow queries, every non-ambient light's term as if unshadowed (a shadow factor of 1.0f multiplies exactly); csLightSum adds
// them in table order afterwards, an occluded light's term times 0.0f (the reference's term with the factor 0.0f is a zero of
// some sign - or a NaN exactly when the unshadowed term is not finite - and so is this product; a sum that started from +0
// does not see the sign of a zero).  That way the surface (normal, view direction, the sampled colours, the glossiness: 13
// values) is dead while the wave sweeps the scene for its shadow rays, and only three values per light are kept.  Lights are
// taken QA_CS_LIGHT_BATCH at a time (csLightTerms from table index li0 on, then csShadowBatch, then csLightSum continuing the
// running sum); scenes with more shadow-casting lights than one batch park the surface in a global slab (DScene::csSurf, one
// coalesced column per value) and read it back before each further batch - explicitly, and only there, instead of leaving 13
// values to the register allocator in every scene.
struct CsTerms { f3 c0, c1, c2, c3; };
__device__ __forceinline__ CsTerms csLightTerms(const DScene &sc, bool lit, int li0, f3 p, f3 N, f3 V, f3 kd, f3 ks, float gloss)
{
  CsTerms t;
  t.c0 = t.c1 = t.c2 = t.c3 = F3(0, 0, 0);
  const float normCoefDI = 1.f / (float) sc.num_lights;
  uint32_t j = 0;
  for (int li = li0; li < sc.num_lights && j < QA_CS_LIGHT_BATCH; ++li) {
    const qa_light l = ldTable(sc.light + li);
    if (l.type == QA_LIGHT_AMBIENT) continue;
    f3 c = F3(0, 0, 0);
    if (lit) c = csLightTerm(l, normCoefDI, p, N, V, kd, ks, gloss, false);
    if (j == 0) t.c0 = c;
    else if (j == 1) t.c1 = c;
    else if (j == 2) t.c2 = c;
    else t.c3 = c;
    ++j;
  }
  return t;
}
__device__ __forceinline__ uint32_t csTermsNeed(const CsTerms &t)
{
  auto nz = [](f3 c) { return !(c.x == 0.f && c.y == 0.f && c.z == 0.f); };   // (a NaN term keeps its ray)
  return (nz(t.c0) ? 1u : 0u) | (nz(t.c1) ? 2u : 0u) | (nz(t.c2) ? 4u : 0u) | (nz(t.c3) ? 8u : 0u);
}
__device__ __forceinline__ f3 csLightSum(f3 sum, const CsTerms &t, uint32_t nb, uint32_t occl)
{
  if (nb > 0) sum = sum + ((occl & 1u) ? t.c0 * 0.0f : t.c0);
  if (nb > 1) sum = sum + ((occl & 2u) ? t.c1 * 0.0f : t.c1);
  if (nb > 2) sum = sum + ((occl & 4u) ? t.c2 * 0.0f : t.c2);
  if (nb > 3) sum = sum + ((occl & 8u) ? t.c3 * 0.0f : t.c3);
  return sum;
}

// ---------------------------------------------------------------------------------------------
// The kernel: qa_integrate<RES = false, LIGHTS, TEX, AREA = false> with its closest-hit query and its shadow queries done
// by the whole wave.  Dynamic LDS per wave: CsLds + 6 x 64 sample accumulators (host: CsLdsWords).
// Sections A, B and E are qa_integrate's text, repeated here on purpose: moving them into functions shared by both kernels
// changes the register allocation of qa_integrate's LDS-resident variants - the Cornell-box kernel lost 8 % (13.0 -> 12.0
// Gsamples/s) with only the tile fetch factored out (profiles/round02/session3_experiments.txt, item 14).
// ---------------------------------------------------------------------------------------------
// Tiles in sample chunks (qa_integrate, section A; RenderParams::chunk_spp): the rare steps out of line - compiled into the kernel's
// text they cost the textured variant 47 spilled registers (experiments.txt 28).
// The wait for a tile's previous chunk and what the chunk is: -> samples a pixel has when this chunk is complete.
__device__ __attribute__((noinline)) int csChunkBegin(uint32_t *progress, uint32_t tile, uint32_t chunk, uint32_t first, uint32_t tail)
{
  if (chunk > 0) {
    for (int spins = 0; spins < (1 << 22); ++spins) {   // (ends at once except on frames of fewer tiles than waves)
      if (__hip_atomic_load(progress + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= chunk) break;
      __builtin_amdgcn_s_sleep(16);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  return (int) (first + chunk * tail);
}
// A pixel as the previous chunk left it: columns 0 - 5 and 14 of the lane (mean, variance, samples taken); -> RNG state, or 0 with
// *finished set when the pixel finished in an earlier chunk
__device__ __attribute__((noinline)) uint32_t csChunkRestore(const uint32_t *state, uint32_t q, float *acc, bool *finished)
{
  const unsigned long long *st = reinterpret_cast<const unsigned long long *>(state) + 4 * (size_t) q;
  const unsigned long long s0 = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), s1 = __hip_atomic_load(st + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           s2 = __hip_atomic_load(st + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), s3 = __hip_atomic_load(st + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *finished = ((uint32_t) (s0 >> 32) & 0x80000000u) != 0;
  if (*finished) return 0;
  acc[14 * 64] = __uint_as_float((uint32_t) (s0 >> 32));
  acc[0] = __uint_as_float((uint32_t) s1); acc[64] = __uint_as_float((uint32_t) (s1 >> 32)); acc[2 * 64] = __uint_as_float((uint32_t) s2);
  acc[3 * 64] = __uint_as_float((uint32_t) (s2 >> 32)); acc[4 * 64] = __uint_as_float((uint32_t) s3); acc[5 * 64] = __uint_as_float((uint32_t) (s3 >> 32));
  return (uint32_t) s0;
}
// ... and as this chunk leaves it, every lane of the wave at once when the whole tile is between chunks (section A): the output index
// column says what the lane holds - 0xFFFFFFFF nothing (no pixel, or one that finished in an earlier chunk), bit 31 a pixel that
// finished in this chunk (later chunks skip it), else a pixel that goes on: RNG state, samples taken, mean, variance.  No call in
// section E: the state is the lane's columns and its RNG register, which stay as the last sample left them.
__device__ __attribute__((noinline)) void csChunkSaveAll(uint32_t *state, const float *acc, uint32_t rng)
{
  const uint32_t q = __float_as_uint(acc[13 * 64]);
  if (q == 0xFFFFFFFFu) return;
  unsigned long long *st = reinterpret_cast<unsigned long long *>(state) + 4 * (size_t) (q & 0x7FFFFFFFu);
#define QA_PAIR(lo, hi) ((unsigned long long) (lo) | ((unsigned long long) (hi) << 32))
  if (q & 0x80000000u) {
    __hip_atomic_store(st, QA_PAIR(0u, 0x80000000u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  __hip_atomic_store(st, QA_PAIR(rng, __float_as_uint(acc[14 * 64])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(st + 1, QA_PAIR(__float_as_uint(acc[0]), __float_as_uint(acc[64])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(st + 2, QA_PAIR(__float_as_uint(acc[2 * 64]), __float_as_uint(acc[3 * 64])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(st + 3, QA_PAIR(__float_as_uint(acc[4 * 64]), __float_as_uint(acc[5 * 64])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#undef QA_PAIR
}

#ifndef QA_CS_WAVES_NOTEX
#define QA_CS_WAVES_NOTEX 4
#endif
#ifndef QA_CS_WAVES_TEX
#define QA_CS_WAVES_TEX 4
#endif
#define QA_CS_LANE_SLOTS 15   /* per-lane LDS words: running mean and variance of the pixel (6), the path's throughput and radiance (6), pixel | output index | sample index */
__host__ __device__ inline uint32_t CsLdsWords(uint32_t items, uint32_t slots) { return items + 8u * slots + QA_CS_RES_WORDS + 64u + QA_CS_LANE_SLOTS * 64u; }

// What a path keeps between its segments besides its ray, throughput and radiance, in one word: bounceCount the next hit is
// shaded with | hInfo.c.hasDiffuseHit of the next hit | camera ray | 1 + the material the ray was spawned from (its absorption
// applies on a back-face exit: Beer's law, ComputeSecondaryRay MtlBlinn_PhotonMap.cpp:244-248; 0 for camera rays).
#define QA_PST_BOUNCE(s) ((int) ((s) & 0xFFu))
#define QA_PST_FROM_DIFFUSE 0x100u
#define QA_PST_PRIMARY 0x200u
#define QA_PST_ABSORB(s) ((int) ((s) >> 16) - 1)

// CULL: the sweeps test every node's root-space bounds first (instance culling).  A variant of its own because the four values
// the test keeps alive during a sweep cost the untextured kernel more in spills than culling returns on scenes of a few nodes
// (C4, 10 nodes: 4 530 without the code, 4 010 with it; a field of 38 nodes: 1 470 -> 2 100 Msamples/s): qa_capi.hip SelectKernel.
// MANY: more shadow-casting lights than one batch (QA_CS_LIGHT_BATCH): the surface is parked in DScene::csSurf and the further
// batches follow.  A variant of its own for the same reason: with that loop compiled in, scenes of one batch lose 6 - 11 %.
// AREA: scenes with area lights (point / spot lights with a size: 16 - 64 shadow rays per evaluation, src/lights/lights.cpp:52-65).
// Their samples draw random numbers, and the reference evaluates a hit's lights only after the whole recursive subtree below it
// (qa_kernel.h, AREA variants): every lit hit is logged (19 floats per hit in DScene::areaScratch) and ALL lights are evaluated when
// the path has ended, deepest hit first - here by the whole wave at once: the lanes of a wave wait for each other between samples
// (sync_samples is forced), the log is replayed level by level, and an area light's samples are walked four shadow rays per lane at a
// time from the pool (csShadowRays) - exactly the any-hit batches the pool wants.
template <bool LIGHTS, bool TEX, bool CULL, bool MANY, bool AREA = false>
__global__ __launch_bounds__(QA_BLOCK, TEX ? QA_CS_WAVES_TEX : QA_CS_WAVES_NOTEX) void qa_integrate_cs(const DScene sc, const RenderParams rp)
{
  extern __shared__ uint4 s_dyn[];
  const unsigned lane = __lane_id();
  CsLds L;
  {
    const uint32_t poolCap = (sc.csPoolLimit && sc.csPoolLimit < sc.csItems) ? sc.csPoolLimit : sc.csItems;   // (a limit below the LDS there is: tests of the overflow path)
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x / 64));                   // (wave-uniform: the pointers stay in scalar registers)
    uint32_t *base = reinterpret_cast<uint32_t *>(s_dyn) + wave * CsLdsWords(sc.csItems, sc.csSlots);
    L.rays = reinterpret_cast<uint4 *>(base);                     // 16-byte aligned: first
    L.res = base + 8u * sc.csSlots;                               // 8-byte aligned keys
    L.flags = L.res + QA_CS_RES_WORDS;
    L.items = L.flags + 64;
    L.capItems = poolCap;
    L.slots = sc.csSlots;
  }
  float *acc = reinterpret_cast<float *>(L.items + sc.csItems) + lane;   // + i * 64
  // The path's throughput and the sample's radiance live in LDS too (columns 6 - 11): they are touched at a handful of points of an
  // iteration and would otherwise be six more registers alive through every sweep and every round (- 19 / - 34 spilled registers in the
  // untextured / textured kernels); the pool gives up 256 items and 16 ray slots for them.
#define QA_PT() F3(acc[6 * 64], acc[7 * 64], acc[8 * 64])
#define QA_PL() F3(acc[9 * 64], acc[10 * 64], acc[11 * 64])
#define QA_SET_PT(v) { const f3 t_ = (v); acc[6 * 64] = t_.x; acc[7 * 64] = t_.y; acc[8 * 64] = t_.z; }
#define QA_SET_PL(v) { const f3 t_ = (v); acc[9 * 64] = t_.x; acc[10 * 64] = t_.y; acc[11 * 64] = t_.z; }
  // ... and so do the pixel (x | y << 16), its output index and the sample index (columns 12 - 14): read at the sample's start and end
#define QA_PXY() __float_as_uint(acc[12 * 64])
#define QA_Q() __float_as_uint(acc[13 * 64])
#define QA_SIDX() __float_as_int(acc[14 * 64])
  // (lanes that hold no pixel - padding lanes of ragged tiles, lanes before their first tile - still run the wave's code: their sample
  // index is an index into the Halton table in csRayDiff / csTexPos, so every column starts from zero)
  for (int i = 0; i < QA_CS_LANE_SLOTS; ++i) acc[i * 64] = 0.f;
  const uint4 *mtlTable = reinterpret_cast<const uint4 *>(sc.mtl);

  const int rw = rp.x1 - rp.x0, rh = rp.y1 - rp.y0;
  const unsigned tilesX = (unsigned) (rw + 7) / 8;
  // tiles in sample chunks (qa_integrate, section A: RenderParams::chunk_spp)
  const unsigned numTiles = tilesX * (unsigned) rp.own_tile_rows;
  // (in the textured variants only: C3 + 2.3 %, project9 + 2.4 %; the untextured ones - C4, C5: 31 tiles per wave - lose 2 % to the code alone)
  constexpr bool CHUNK = TEX;
  const unsigned total = numTiles * ((CHUNK && rp.chunk_spp) ? rp.num_chunks : 1u) * 64u;
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

  // per-lane state: pixel (x | y << 16, output index, RNG stream, sample index) and path (ray, throughput, radiance, state word)
  uint32_t rng = 1;
  Ray ray;
  ray.p = F3(0, 0, 0);
  ray.d = F3(0, 0, 1);
  uint32_t pst = QA_PST_PRIMARY;
  bool alive = true, needPixel = true, needSample = false;
  uint32_t nrec = 0;       // AREA: hits logged for the current path
  // BATCH (the textured and the many-light variants; compiled into the others it costs C4 and C5 1 %): sync_samples >= 2, below
  constexpr bool BATCH = !AREA && (TEX || MANY);
  bool awaiting = false;   // AREA: the path has ended, its lights have not been evaluated yet; BATCH: ... it waits for others to finish

  for (;;) {
    QA_T(tA)
    // ---- A. tile fetch (qa_integrate, section A)
    const unsigned long long aliveMask = __ballot(alive);
    const unsigned long long want = __ballot(alive && needPixel);
    if (want && want == aliveMask) {
      if (CHUNK && rp.chunk_spp && curTile != 0xFFFFFFFFu) {
        // the chunk in hand is complete: every lane stores what it holds (agent-scope atomic stores), the wave waits for them, publishes
        csChunkSaveAll(rp.pix_state, acc, rng);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(rp.tile_progress + curTile, curChunk + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        curTile = 0xFFFFFFFFu;
      }
      unsigned base = 0;
      const int leader = __ffsll((long long) want) - 1;
      if ((int) lane == leader) base = (*rp.stop_flag) ? total : atomicAdd(rp.work_counter, 64u);
      base = __shfl(base, leader);
      unsigned item = base / 64;   // (wave-uniform) tile, or chunk * numTiles + tile
      if (CHUNK && rp.chunk_spp && base < total) {
        curChunk = item / numTiles;
        item -= curChunk * numTiles;
        curTile = item;
        chunkEnd = csChunkBegin(rp.tile_progress, curTile, curChunk, rp.chunk_spp, rp.chunk_tail);
      }
      if (alive) {
        const unsigned w = base + lane;
        if (base >= total) {
          alive = false;
        } else {
          const unsigned in = w % 64;
          const unsigned tile = rp.tile_order ? rp.tile_order[item] : item;
          const unsigned otr = tile / tilesX;
          const unsigned tx = (tile % tilesX) * 8 + (in % 8);
          const unsigned ty = ((unsigned) rp.tile_row0 + otr * (unsigned) rp.tile_row_step) * 8 + (in / 8);
          if (tx < (unsigned) rw && ty < (unsigned) rh) {
            const uint32_t px = (uint32_t) rp.x0 + tx, py = (uint32_t) rp.y0 + ty;
            acc[12 * 64] = __uint_as_float(px | (py << 16));
            acc[13 * 64] = __uint_as_float((otr * 8 + (in / 8)) * (unsigned) rw + tx);
            rng = qa_pixel_seed(rp.seed, py * (uint32_t) sc.cam.width + px);
            acc[14 * 64] = __int_as_float(0);
            for (int i = 0; i < 6; ++i) acc[i * 64] = 0.f;
            needSample = true;
            needPixel = false;
            if (CHUNK && rp.chunk_spp && curChunk > 0) {
              bool finished;
              const uint32_t r = csChunkRestore(rp.pix_state, QA_Q(), acc, &finished);
              if (finished) {   // (finished in an earlier chunk: the lane sits this one out)
                needSample = false;
                needPixel = true;
                acc[13 * 64] = __uint_as_float(0xFFFFFFFFu);
              } else rng = r;
            }
          } else if (CHUNK) {
            acc[13 * 64] = __uint_as_float(0xFFFFFFFFu);   // (a padding lane of a ragged tile holds nothing: csChunkSaveAll)
          }
        }
      }
    }
    if (!__any(alive)) break;

    // ---- B. start a sample (qa_integrate, section B; src/renderers/renderer.cpp:312-328)
    const bool goSample = !rp.sync_samples || (BATCH && rp.sync_samples != 1) || (__ballot(needSample) == __ballot(alive && !needPixel));
    const bool starting = alive && needSample && goSample;
    cnt.samples += (unsigned long long) __popcll(__ballot(starting));   // (wave-uniform tallies: no registers per lane)
    if (starting) {
      const f3 texpos = csTexPos(sc, QA_PXY(), QA_SIDX());
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
      QA_SET_PT(F3(1, 1, 1))
      QA_SET_PL(F3(0, 0, 0))
      pst = QA_PST_PRIMARY | (uint32_t) (rp.max_bounce & 0xFF);
      needSample = false;
    }
    QA_TACC(cnt.sl[1], tA)
    // ---- C. trace (qa_integrate, section C)
    const bool act = alive && !needPixel && !needSample && !((AREA || BATCH) && awaiting);
    bool done = false;
    Hit h;
    TexHit th;
    QA_T(tC)
    const bool found = csTraceClosest<TEX, CULL>(sc, L, act, ray, (pst & QA_PST_PRIMARY) != 0, acc, h, th, cnt);
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
      const bool primary = (pst & QA_PST_PRIMARY) != 0;
      if (primary && QA_SIDX() == 0) rp.depth[QA_Q()] = found ? h.z : QA_BIGFLOAT;
      if (!found) {
        f3 c = primary ? ld3(sc.background) : ld3(sc.environment);
        if (TEX) {
          if (primary) {
            const f3 texpos = csTexPos(sc, QA_PXY(), QA_SIDX());
            c = texColorSample(tt, c, sc.bgTexmap, F3(texpos.x / (float) sc.cam.width, texpos.y / (float) sc.cam.height, 0.f));
          } else
            c = sampleEnvironment(tt, c, sc.envTexmap, ray.d);
        }
        QA_SET_PL(QA_PL() + QA_PT() * c)
        done = true;
      } else {
        const int absorbMtl = QA_PST_ABSORB(pst);
        if (!primary && !h.front && absorbMtl >= 0) {
          const uint4 ab = mtlTable[6 * (size_t) absorbMtl + 5];
          const f3 att = F3(qexpf(-asF(ab.x) * h.z), qexpf(-asF(ab.y) * h.z), qexpf(-asF(ab.z) * h.z));
          QA_SET_PT(QA_PT() * att)
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
          if (white) QA_SET_PL(QA_PL() + QA_PT())
          done = true;
        } else {
          V = -ray.d;
          N = h.N;
          p = h.p;
          // (shadeSurface inline: as a function of its own - tried for the untextured variants - its results come back through
          // memory or a block of registers that the caller spills: C4 3 790 vs 4 510, C5 1 590 vs 1 690 Msamples/s at 16 spp)
          sf = shadeSurface<TEX>(mtlTable, sc, tt, mi, N, V, h.front, th, QA_PST_BOUNCE(pst), (pst & QA_PST_FROM_DIFFUSE) != 0, rng);
          QA_SET_PL(QA_PL() + QA_PT() * sf.emission)
          lit = true;
        }
      }
    }
    QA_TACC(cnt.sl[4], tD)
    // ---- direct lighting, first half: the lights' terms as if unshadowed; then the path moves on to its next segment (or
    // ends) BEFORE the shadow queries, so that the surface is dead while the wave sweeps the scene for them: what is kept is
    // the shading point (= the next ray's origin), the throughput the lights are weighted with, and three values per light
    QA_T(tE)
    CsTerms terms;
    terms.c0 = terms.c1 = terms.c2 = terms.c3 = F3(0, 0, 0);
    f3 litT = F3(0, 0, 0);
    if (AREA) {
      // log the hit: position, normal, view direction, throughput, sampled colours, glossiness (qa_kernel.h's record)
      if (lit && nrec < QA_MAX_PATH) {
        const size_t stride = (size_t) gridDim.x * QA_BLOCK;
        float *rec = sc.areaScratch + (size_t) blockIdx.x * QA_BLOCK + threadIdx.x + (size_t) nrec * QA_REC_FLOATS * stride;
        const f3 pT = QA_PT();
        const float v[QA_REC_FLOATS] = {p.x, p.y, p.z, N.x, N.y, N.z, V.x, V.y, V.z, pT.x, pT.y, pT.z, sf.kd.x, sf.kd.y, sf.kd.z, sf.ks.x, sf.ks.y, sf.ks.z, sf.gloss};
#pragma unroll
        for (int f = 0; f < QA_REC_FLOATS; ++f) rec[(size_t) f * stride] = v[f];
        ++nrec;
      }
    } else if (LIGHTS) {
      if (__any(lit)) {
        QA_T(tLt)
        terms = csLightTerms(sc, lit, 0, p, N, V, sf.kd, sf.ks, sf.gloss);
        if (MANY && lit) {
          // more lights than one batch: the surface waits in the slab (column f of this lane: csSurf[f * lanes + lane id])
          const size_t stride = (size_t) gridDim.x * QA_BLOCK;
          float *sv = sc.csSurf + (size_t) blockIdx.x * QA_BLOCK + threadIdx.x;
          const float v[13] = {N.x, N.y, N.z, V.x, V.y, V.z, sf.kd.x, sf.kd.y, sf.kd.z, sf.ks.x, sf.ks.y, sf.ks.z, sf.gloss};
#pragma unroll
          for (int f = 0; f < 13; ++f) sv[f * stride] = v[f];
        }
        QA_TACC(cnt.sl[18], tLt)
      }
    }
    if (lit) {
      litT = QA_PT();
      ray.p = p;
      if (sf.spawn) {
        // ComputeSecondaryRay (:226-254): DiffRay(pos, dir).Normalize()
        ray.d = normalize(sf.nextDir);
        QA_SET_PT(litT * sf.bxdf)
        pst = (uint32_t) ((QA_PST_BOUNCE(pst) - 1) & 0xFF) | (sf.nextFromDiffuse ? QA_PST_FROM_DIFFUSE : 0u) | ((uint32_t) (mi + 1) << 16);
      } else {
        done = true;
      }
    }
    // ---- second half: the whole wave walks the shadow rays of its lit lanes
    if (LIGHTS && !AREA) {
      if (__any(lit)) {
        QA_T(tL)
        int li = 0;
        uint32_t nb = 0;
        uint32_t occl = csShadowBatch<CULL>(sc, L, lit, csTermsNeed(terms) | (sc.walkZeroTerms ? 15u : 0u), ray.p, li, nb, cnt);
        f3 dl = csLightSum(F3(0, 0, 0), terms, nb, occl);
        if (MANY) {
          // the further batches of a scene with many lights: surface back from the slab, terms, shadow queries, sum - in table order
          while (li < sc.num_lights) {
            const int li0 = li;
            const size_t stride = (size_t) gridDim.x * QA_BLOCK;
            const float *sv = sc.csSurf + (size_t) blockIdx.x * QA_BLOCK + threadIdx.x;
            float v[13];
#pragma unroll
            for (int f = 0; f < 13; ++f) v[f] = lit ? sv[f * stride] : 0.f;
            terms = csLightTerms(sc, lit, li0, ray.p, F3(v[0], v[1], v[2]), F3(v[3], v[4], v[5]), F3(v[6], v[7], v[8]), F3(v[9], v[10], v[11]), v[12]);
            occl = csShadowBatch<CULL>(sc, L, lit, csTermsNeed(terms) | (sc.walkZeroTerms ? 15u : 0u), ray.p, li, nb, cnt);
            if (!nb) break;   // (only ambient lights were left)
            dl = csLightSum(dl, terms, nb, occl);
          }
        }
        if (lit) QA_SET_PL(QA_PL() + litT * dl)
        QA_TACC(cnt.sl[5], tL)
      }
    }

    // sync_samples >= 2: finished paths wait until that many of the wave's have gathered (or every lane's has): sections E and B
    // then run for a group of lanes instead of a few lanes in nearly every iteration
    if (BATCH && rp.sync_samples >= 2) {
      awaiting = awaiting || (alive && done);
      done = false;
      const unsigned long long aw = __ballot(awaiting);
      if (aw && (__popcll(aw) >= rp.sync_samples || aw == __ballot(alive && !needPixel))) {
        done = awaiting;
        awaiting = false;
      }
    }
    // ---- AREA: the lights of the paths that have ended, once the whole wave is between samples
    if (AREA) {
      awaiting = awaiting || (alive && done);
      done = false;
      if (__any(awaiting) && __ballot(awaiting) == __ballot(alive && !needPixel)) {
        QA_T(tL)
        uint32_t maxrec = 0;
        for (uint32_t b = 0; b < 4u; ++b) if (__any(awaiting && ((nrec >> b) & 1u))) maxrec |= 1u << b;   // (an upper bound of the wave's deepest log)
        const size_t stride = (size_t) gridDim.x * QA_BLOCK;
        const float *rec0 = sc.areaScratch + (size_t) blockIdx.x * QA_BLOCK + threadIdx.x;
        const float normCoefDI = 1.f / (float) sc.num_lights;
        for (uint32_t lvl = maxrec < QA_MAX_PATH ? maxrec : QA_MAX_PATH; lvl-- > 0;) {
          const bool on = awaiting && nrec > lvl;
          if (!__any(on)) continue;
          float v[QA_REC_FLOATS];
#pragma unroll
          for (int f = 0; f < QA_REC_FLOATS; ++f) v[f] = on ? rec0[((size_t) lvl * QA_REC_FLOATS + f) * stride] : 0.f;
          const f3 hp = F3(v[0], v[1], v[2]), hN = F3(v[3], v[4], v[5]), hV = F3(v[6], v[7], v[8]);
          f3 sum = F3(0, 0, 0);
          // directLight + illuminate of qa_kernel.h, the shadow queries done by the wave
          for (int li = 0; li < sc.num_lights; ++li) {
            const qa_light l = ldTable(sc.light + li);
            if (l.type == QA_LIGHT_AMBIENT) continue;
            f3 I;
            if (l.type != QA_LIGHT_DIRECT && l.size > 0.01f) {
              // area light: 16 shadow rays towards points of a ball around the light, 64 as soon as the running estimate is a
              // penumbra value (src/lights/lights.cpp:52-65,88-100) - four at a time: the first 16 always exist, and whether the
              // other 48 do is decided by then
              int spp = 16, ns = 0;
              float inshadow = 0.0f;
              for (;;) {
                const bool more = on && ns < spp;
                if (!__any(more)) break;
                CsRays4 rq;
                f3 dir0 = F3(0, 0, 1), dir1 = dir0, dir2 = dir0, dir3 = dir0;
                if (more) {
                  dir0 = (ld3(l.position) + uniformBall(rng, l.size)) - hp;
                  dir1 = (ld3(l.position) + uniformBall(rng, l.size)) - hp;
                  dir2 = (ld3(l.position) + uniformBall(rng, l.size)) - hp;
                  dir3 = (ld3(l.position) + uniformBall(rng, l.size)) - hp;
                }
                rq.d0 = normalize(dir0); rq.t0 = length(dir0);
                rq.d1 = normalize(dir1); rq.t1 = length(dir1);
                rq.d2 = normalize(dir2); rq.t2 = length(dir2);
                rq.d3 = normalize(dir3); rq.t3 = length(dir3);
                const uint32_t occl = csShadowRays<CULL>(sc, L, more ? 15u : 0u, 4u, hp, rq, cnt);
                if (more) {
#define QA_CS_FOLD(S, DIR)                                                                                                   \
                  {                                                                                                            \
                    const float shadowed = ((occl >> S) & 1u) ? 0.0f : 1.0f;                                                   \
                    inshadow += (shadowed - inshadow) * inverseSquareFalloff(DIR) / (float) (ns + 1);                          \
                    ns++;                                                                                                      \
                    if (inshadow > 0.f && inshadow < 1.f) spp = 64;                                                            \
                  }
                  QA_CS_FOLD(0, dir0)
                  QA_CS_FOLD(1, dir1)
                  QA_CS_FOLD(2, dir2)
                  QA_CS_FOLD(3, dir3)
#undef QA_CS_FOLD
                }
              }
              I = ld3(l.intensity) * inshadow;
              if (l.type == QA_LIGHT_SPOT) I = I * spotAttenuation(l, hp);
            } else {
              CsRays4 rq;
              rq.d1 = rq.d2 = rq.d3 = F3(0, 0, 1);
              rq.t1 = rq.t2 = rq.t3 = 0.f;
              f3 dir = F3(0, 0, 1);
              if (l.type == QA_LIGHT_DIRECT) {
                rq.d0 = normalize(-ld3(l.direction));
                rq.t0 = QA_BIGFLOAT;
              } else {
                dir = ld3(l.position) - hp;
                rq.d0 = normalize(dir);
                rq.t0 = length(dir);
              }
              // (a surface facing away from the light - cosNL = 0 - gets the same zero whether the light is occluded or not: its shadow
              // ray is counted and not walked, csShadowBatch)
              const bool walk = on && (sc.walkZeroTerms || qmax(0.f, dot(hN, normalize(-lightDirection(l, hp)))) != 0.f);
              cnt.casts_shadow += (unsigned long long) __popcll(__ballot(on && !walk));
              const uint32_t occl = csShadowRays<CULL>(sc, L, walk ? 1u : 0u, 1u, hp, rq, cnt);
              const float shadowed = (occl & 1u) ? 0.0f : 1.0f;
              if (l.type == QA_LIGHT_DIRECT) I = ld3(l.intensity) * shadowed;
              else {
                I = (ld3(l.intensity) * shadowed) * inverseSquareFalloff(dir);
                if (l.type == QA_LIGHT_SPOT) I = I * spotAttenuation(l, hp);
              }
            }
            const f3 intensity = I * normCoefDI;
            const f3 Ld = normalize(-lightDirection(l, hp));
            const f3 H = normalize(hV + Ld);
            const float cosNL = qmax(0.f, dot(hN, Ld));
            const float cosNH = qmax(0.f, dot(hN, H));
            const f3 brdf = F3(v[12], v[13], v[14]) + F3(v[15], v[16], v[17]) * qpowf(cosNH, v[18]);
            sum = sum + (intensity * cosNL) * brdf;
          }
          if (on) QA_SET_PL(QA_PL() + F3(v[9], v[10], v[11]) * sum)
        }
        QA_TACC(cnt.sl[5], tL)
        done = awaiting;
        awaiting = false;
        nrec = 0;
      }
    }
    // ---- E. sample finished (qa_integrate, section E; scene.cpp:92-121)
    bool pixelDone = false;
    if (alive && done) {
      int sidx = QA_SIDX();
      const float inv = (float) (sidx + 1);
      f3 mean = F3(acc[0], acc[64], acc[2 * 64]);
      f3 cstd = F3(0, 0, 0);
      const f3 dc = (QA_PL() - mean) / inv;
      mean = mean + dc;
      acc[0] = mean.x; acc[64] = mean.y; acc[2 * 64] = mean.z;
      if (rp.spp_min < rp.spp_max) {   // (the running variance is read by the "another sample?" test below alone: qa_integrate, section E)
        cstd = F3(acc[3 * 64], acc[4 * 64], acc[5 * 64]);
        if (sidx > 0) cstd = cstd + ((dc * dc) * inv - cstd / (float) sidx);
        acc[3 * 64] = cstd.x; acc[4 * 64] = cstd.y; acc[5 * 64] = cstd.z;
      }
      ++sidx;
      acc[14 * 64] = __int_as_float(sidx);
      const bool more = sidx < rp.spp_min || (sidx < rp.spp_max && (cstd.x > 0.005f || cstd.y > 0.001f || cstd.z > 0.005f));
      if (more) {
        if (CHUNK && rp.chunk_spp && sidx >= chunkEnd) {
          needPixel = true;   // the chunk's last sample of this pixel: its state goes to the next chunk's wave when the tile is handed on (section A)
        } else {
          needSample = true;
        }
      } else {
        const uint32_t q = QA_Q();
        rp.rgb[3 * q + 0] = mean.x;
        rp.rgb[3 * q + 1] = mean.y;
        rp.rgb[3 * q + 2] = mean.z;
        rp.ns[q] = (uint32_t) sidx;
        if (CHUNK) acc[13 * 64] = __uint_as_float(q | 0x80000000u);   // (finished: csChunkSaveAll tells the tile's later chunks)
        pixelDone = true;
        needPixel = true;
      }
    }
    cnt.pixels += (unsigned long long) __popcll(__ballot(pixelDone));
    QA_TACC(cnt.sl[7], tE)
#ifdef QA_STAMPS
    if (lane == 0) cnt.sl[8] += 1;
#endif
  }

  // the tallies are per wave: one lane adds them
  unsigned long long *dst = reinterpret_cast<unsigned long long *>(rp.counters);
  if (lane == 0) {
    if (cnt.samples) atomicAdd(&dst[0], cnt.samples);
    if (cnt.casts_normal) atomicAdd(&dst[1], cnt.casts_normal);
    if (cnt.casts_shadow) atomicAdd(&dst[2], cnt.casts_shadow);
    if (cnt.pixels) atomicAdd(&dst[5], cnt.pixels);
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
