// qa_kernel_sm.h — the integrator as a per-lane STATE MACHINE with ray casts that can be suspended
// between BVH steps (second kernel of the library; same results as qa_integrate in qa_kernel.h).
//
// Why a second kernel: in qa_integrate all 64 lanes of a wave start a cast together and the wave
// leaves the traversal when its slowest lane does.  With a few hundred triangles per scene that
// costs ~2x; with 10^5..10^6 triangles (deep cyBVH trees, incoherent secondary rays) traversal
// lengths are heavy-tailed and measured lane utilisation drops to 15 %.  Here every lane carries
// its whole cast state (instance cursor, node-space ray, BVH cursor + LDS stack, best hit) in
// registers, and the wave loop interleaves three blocks, each entered only when enough lanes want it:
//
//   GEN   consume a finished cast: shade (MtlBlinn_PhotonMap::Shade), light bookkeeping, sample /
//         pixel bookkeeping, camera ray - and set up the next cast
//   INST  move the cast to the next scene-graph node: Node::ToNodeCoords chain, sphere / plane
//         intersection, mesh bounds gate
//   TRAV  a bounded number of BVH steps (inner-node pair tests, leaf triangles)
//
// A lane whose cast finishes early is shaded and re-armed while its neighbours are still deep in
// the tree.  Shadow rays are casts of their own (continuation = light index), so they take part in
// the same interleaving.  Per-lane order of RNG draws, node visits and triangle tests is exactly the
// reference's; only the interleaving between lanes changes.  Area lights (AREA variants) stay on
// qa_integrate.
#pragma once
#include "qa_kernel.h"

namespace qa {

enum { SM_IDLE = 0, SM_WAITPIX = 1, SM_GEN = 2, SM_INST = 3, SM_TRAV = 4 };

// Node::FromNodeCoords up the hit node's ancestry (src/core/node.cpp:127-139)
template <bool RES>
__device__ __forceinline__ void hitToWorld(const DScene &sc, Hit &h)
{
  for (int a = h.node; a >= 0; a = instAt<RES>(sc, a).parent) {
    if (a == 0 && sc.rootIdentity) { h.N = normalize(h.N); break; }
    const qa_instance &in = instAt<RES>(sc, a);
    h.p = mulMV(in.tm, h.p) + ld3(in.pos);
    h.N = normalize(mulTMV(in.itm, h.N));
  }
}

template <bool RES, bool LIGHTS, bool TEX, bool STATS>
__global__ __launch_bounds__(QA_BLOCK, QA_MIN_WAVES) void qa_integrate_sm(const DScene sc, const RenderParams rp)
{
  extern __shared__ uint4 s_dyn[];
  SceneMem<RES> mem;
  mem.img = s_dyn;
  if (RES) {
    for (uint32_t i = threadIdx.x; i < sc.residentVec4; i += QA_BLOCK) s_dyn[i] = sc.resident[i];
    __syncthreads();
  }
  uint32_t *stack = reinterpret_cast<uint32_t *>(s_dyn + (RES ? sc.residentVec4 : 0)) + threadIdx.x;
  float *acc = reinterpret_cast<float *>(stack + (size_t) sc.stackDepth * QA_BLOCK - threadIdx.x) + threadIdx.x;
  const uint4 *mtlTable = RES ? s_dyn + sc.resMaterials : reinterpret_cast<const uint4 *>(sc.mtl);

  const int rw = rp.x1 - rp.x0, rh = rp.y1 - rp.y0;
  const unsigned tilesX = (unsigned) (rw + 7) / 8;
  const unsigned total = tilesX * (unsigned) rp.own_tile_rows * 64u;
  const unsigned lane = __lane_id();
  const int genThresh = rp.sm_gen_thresh, instThresh = rp.sm_inst_thresh, travSteps = rp.sm_trav_steps;

  DCounters cnt = {0, 0, 0, 0, 0, 0};
  TexTables tt;
  tt.blob = sc.blob;
  tt.texmap = sc.texmap;
  tt.tex = sc.tex;
  tt.filter = sc.texFilter;

  // ---- pixel / path state ----------------------------------------------------------------------
  int px = 0, py = 0;
  unsigned q = 0;
  uint32_t rng = 1;
  int sidx = 0;
  f3 texpos = F3(0, 0, 0);
  Ray ray;                    // world ray of the cast in flight (path segment or shadow ray)
  ray.p = F3(0, 0, 0);
  ray.d = F3(0, 0, 1);
  RayDiff wd;                 // TEX: its differential directions
  wd.dx = wd.dy = F3(0, 0, 1);
  f3 T = F3(0, 0, 0), L = F3(0, 0, 0);
  int absorbMtl = -1, bounce = 0;
  bool fromDiffuse = false, primary = true;

  // ---- cast state ------------------------------------------------------------------------------
  int mode = SM_WAITPIX;
  int k = 1;                  // scene-graph node the cast is at
  Ray lr;                     // node-space ray of the mesh being traversed
  lr.p = lr.d = F3(0, 0, 1);
  RayDiff lrd;
  lrd.dx = lrd.dy = F3(0, 0, 1);
  f3 drcp = F3(1, 1, 1);
  uint32_t cur = QA_DONE;
  int sp = 0;
  TriPick pick = {0, 0.f, 0.f};
  bool meshHit = false, degenerate = false;
  bool castShadow = false, castDone = false, occluded = false;
  Hit h;
  h.z = QA_BIGFLOAT; h.p = h.N = F3(0, 0, 0); h.node = -1; h.mtlID = 0; h.front = true;
  TexHit th;
  th.uvw = th.duvw0 = th.duvw1 = F3(0, 0, 0);
  th.hasTexture = false;

  // ---- continuation of a hit whose lights are being evaluated (LIGHTS) ------------------------------
  f3 hp = F3(0, 0, 0), hN = F3(0, 0, 1), hV = F3(0, 0, 1), kd = F3(0, 0, 0), ks = F3(0, 0, 0);
  float gloss = 1.f;
  int li = 0;
  bool spawn = false, nextFromDiffuse = false;
  f3 nextDir = F3(0, 0, 1), bxdf = F3(0, 0, 0);
  int nextAbsorbMtl = -1;

  bool alive = true;

  for (;;) {
    // ---- tile fetch: a wave owns one 8x8 tile at a time (see qa_integrate) --------------------------
    {
      const unsigned long long aliveMask = __ballot(alive);
      if (aliveMask == 0) break;
      const unsigned long long waiting = __ballot(alive && mode == SM_WAITPIX);
      if (waiting == aliveMask) {
        unsigned base = 0;
        const int leader = __ffsll((long long) waiting) - 1;
        if ((int) lane == leader) base = (*rp.stop_flag) ? total : atomicAdd(rp.work_counter, 64u);
        base = __shfl(base, leader);
        if (alive) {
          if (base >= total) {
            alive = false;
            mode = SM_IDLE;
          } else {
            const unsigned w = base + lane;
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
              castDone = false;
              mode = SM_GEN;   // needs its first camera ray
            }
            // else: padding slot of a ragged tile, keeps waiting
          }
        }
        continue;
      }
    }

    // ================================ GEN ==========================================================
    {
      const int nG = __popcll(__ballot(mode == SM_GEN));
      const int nBusy = __popcll(__ballot(mode == SM_INST || mode == SM_TRAV));
      if (nG > 0 && (nG >= genThresh || nBusy == 0)) {
        if (mode == SM_GEN) {
          bool done = false;      // the sample is finished
          bool start = false;     // `ray` holds a new cast
          bool newSample = !castDone;  // fresh pixel: start with a camera ray
          if (castDone) {
            castDone = false;
            bool finishHit = false;
            if (LIGHTS && castShadow) {
              // ---- a light's shadow ray came back: MtlBlinn_PhotonMap.cpp:486-497 for light li ------
              const qa_light &l = sc.light[li];
              const float sh = occluded ? 0.0f : 1.0f;
              f3 I;
              if (l.type == QA_LIGHT_DIRECT) I = ld3(l.intensity) * sh;
              else {
                I = (ld3(l.intensity) * sh) * inverseSquareFalloff(ld3(l.position) - hp);
                if (l.type == QA_LIGHT_SPOT) I = I * spotAttenuation(l, hp);
              }
              const f3 intensity = I * (1.f / (float) sc.num_lights);
              const f3 Ld = normalize(-lightDirection(l, hp));
              const f3 H = normalize(hV + Ld);
              const float cosNL = qmax(0.f, dot(hN, Ld));
              const float cosNH = qmax(0.f, dot(hN, H));
              L = L + T * ((intensity * cosNL) * (kd + ks * qpowf(cosNH, gloss)));
              ++li;
              finishHit = true;   // unless another light follows (below)
            } else {
              // ---- a path segment came back ---------------------------------------------------------
              const bool found = h.node >= 0;
              if (primary && sidx == 0) rp.depth[q] = found ? h.z : QA_BIGFLOAT;
              if (!found) {
                f3 c = primary ? ld3(sc.background) : ld3(sc.environment);
                if (TEX) {
                  if (primary) c = texColorSample(tt, c, sc.bgTexmap, F3(texpos.x / (float) sc.cam.width, texpos.y / (float) sc.cam.height, 0.f));
                  else c = sampleEnvironment(tt, c, sc.envTexmap, ray.d);
                }
                L = L + T * c;
                done = true;
              } else {
                if (!primary && !h.front && absorbMtl >= 0) {
                  const uint4 ab = mtlTable[6 * (size_t) absorbMtl + 5];
                  T = T * F3(qexpf(-asF(ab.x) * h.z), qexpf(-asF(ab.y) * h.z), qexpf(-asF(ab.z) * h.z));
                }
                const qa_instance &in = instAt<RES>(sc, h.node);
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
                  const Surface sf = shadeSurface<TEX>(mtlTable, sc, tt, mi, h.N, V, h.front, th, bounce, fromDiffuse, rng);
                  L = L + T * sf.emission;
                  hp = h.p;
                  spawn = sf.spawn;
                  nextDir = sf.nextDir;
                  bxdf = sf.bxdf;
                  nextFromDiffuse = sf.nextFromDiffuse;
                  nextAbsorbMtl = mi;
                  if (LIGHTS) {
                    hN = h.N; hV = V; kd = sf.kd; ks = sf.ks; gloss = sf.gloss;
                    li = 0;
                  }
                  finishHit = true;
                }
              }
            }
            if (finishHit) {
              bool shadowNext = false;
              if (LIGHTS) {
                while (li < sc.num_lights && sc.light[li].type == QA_LIGHT_AMBIENT) ++li;
                if (li < sc.num_lights) {
                  // Light::Illuminate's shadow ray (lights.h:66-71, lights.cpp:66-73)
                  const qa_light &l = sc.light[li];
                  ray.p = hp;
                  if (l.type == QA_LIGHT_DIRECT) {
                    ray.d = normalize(-ld3(l.direction));
                    h.z = QA_BIGFLOAT;
                  } else {
                    const f3 dir = ld3(l.position) - hp;
                    ray.d = normalize(dir);
                    h.z = length(dir);
                  }
                  castShadow = true;
                  shadowNext = true;
                  start = true;
                  cnt.casts_shadow++;
                }
              }
              if (!shadowNext) {
                if (spawn) {
                  // ComputeSecondaryRay (:226-254)
                  ray.p = hp;
                  ray.d = normalize(nextDir);
                  if (TEX) wd.dx = wd.dy = ray.d;
                  T = T * bxdf;
                  absorbMtl = nextAbsorbMtl;
                  bounce -= 1;
                  fromDiffuse = nextFromDiffuse;
                  primary = false;
                  castShadow = false;
                  h.z = QA_BIGFLOAT;
                  start = true;
                  cnt.casts_normal++;
                } else done = true;
              }
            }
          }
          if (done) {
            // SuperSamplerHalton::Accumulate / Loop (scene.cpp:92-121)
            const float inv = (float) (sidx + 1);
            f3 mean = F3(acc[0], acc[QA_BLOCK], acc[2 * QA_BLOCK]);
            f3 cstd = F3(acc[3 * QA_BLOCK], acc[4 * QA_BLOCK], acc[5 * QA_BLOCK]);
            const f3 dc = (L - mean) / inv;
            mean = mean + dc;
            if (sidx > 0) cstd = cstd + ((dc * dc) * inv - cstd / (float) sidx);
            acc[0] = mean.x; acc[QA_BLOCK] = mean.y; acc[2 * QA_BLOCK] = mean.z;
            acc[3 * QA_BLOCK] = cstd.x; acc[4 * QA_BLOCK] = cstd.y; acc[5 * QA_BLOCK] = cstd.z;
            ++sidx;
            const bool more = sidx < rp.spp_min ||
                              (sidx < rp.spp_max && (cstd.x > 0.005f || cstd.y > 0.001f || cstd.z > 0.005f));
            if (more) newSample = true;
            else {
              rp.rgb[3 * q + 0] = mean.x;
              rp.rgb[3 * q + 1] = mean.y;
              rp.rgb[3 * q + 2] = mean.z;
              rp.ns[q] = (uint32_t) sidx;
              cnt.pixels++;
              mode = SM_WAITPIX;
            }
          }
          if (newSample && mode == SM_GEN) {
            // camera ray (src/renderers/renderer.cpp:312-328)
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
            ray.p = campos;
            ray.d = normalize(cpt - campos);
            if (TEX) {
              const f3 xpt = (A + U * (texpos.x + QA_DX)) + V * texpos.y;
              const f3 ypt = (A + U * texpos.x) + V * (texpos.y + QA_DX);
              wd.dx = normalize(xpt - campos);
              wd.dy = normalize(ypt - campos);
            }
            T = F3(1, 1, 1);
            L = F3(0, 0, 0);
            absorbMtl = -1;
            bounce = rp.max_bounce;
            fromDiffuse = false;
            primary = true;
            castShadow = false;
            h.z = QA_BIGFLOAT;
            start = true;
            cnt.samples++;
            cnt.casts_normal++;
          }
          if (start) {
            // arm the cast: Scene::TraceNodeNormal / TraceNodeShadow start at the root's first child
            k = 1;
            while (k < sc.num_inst && instAt<RES>(sc, k).obj_type == QA_OBJ_NONE) ++k;
            h.node = -1;
            occluded = false;
            if (!castShadow) {
              h.mtlID = 0;
              h.front = true;
              if (TEX) {
                th.uvw = F3(0.5f, 0.5f, 0.5f);
                th.duvw0 = th.duvw1 = F3(0, 0, 0);
                th.hasTexture = false;
              }
            }
            mode = SM_INST;
          }
        }
      }
    }

    // ================================ INST =========================================================
    {
      const int nI = __popcll(__ballot(mode == SM_INST));
      const int nT = __popcll(__ballot(mode == SM_TRAV));
      if (nI > 0 && (nI >= instThresh || nT == 0)) {
        if (mode == SM_INST) {
          if (k < sc.num_inst) {
            const qa_instance &in = instAt<RES>(sc, k);
            Ray r;
            RayDiff rd;
            if (TEX && !castShadow) localRayDiff<RES>(sc, k, ray, wd, r, rd);
            else r = localRay<RES>(sc, k, rootRay<RES>(sc, ray));
            const int type = in.obj_type;
            if (type == QA_OBJ_MESH) {
              const DMesh &m = meshAt<RES>(sc, in.mesh);
              const f3 rc = F3(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
              float entry, exit_;
              boxEntryExit(r, rc, ld3(m.bmin), ld3(m.bmax), entry, exit_);
              if (entry > h.z || entry > exit_ || m.num_faces == 0) ++k;  // Box::IntersectRay gate (box.cpp:94-128)
              else {
                lr = r;
                if (TEX) lrd = rd;
                drcp = rc;
                degenerate = qabs(r.d.x) < 1e-7f || qabs(r.d.y) < 1e-7f || qabs(r.d.z) < 1e-7f;
                cur = m.rootData;
                sp = 0;
                meshHit = false;
                mode = SM_TRAV;
              }
            } else {
              bool hit;
              if (type == QA_OBJ_SPHERE) {
                hit = hitSphere(r, h, k, !castShadow);
                if (TEX && hit && !castShadow) texSphere(r.p, rd.dx, rd.dy, h.p, h.N, th);
              } else {
                hit = hitPlane(r, h, k, !castShadow);
                if (TEX && hit && !castShadow) texPlane(r.p, rd.dx, rd.dy, h.p, th);
              }
              if (hit && castShadow) { occluded = true; k = sc.num_inst; }  // any hit ends a shadow query
              else ++k;
            }
          }
          if (mode == SM_INST) {
            // object-less nodes only contribute their transforms (through their children's chains)
            while (k < sc.num_inst && instAt<RES>(sc, k).obj_type == QA_OBJ_NONE) ++k;
            if (k >= sc.num_inst) {
              if (!castShadow && h.node >= 0) hitToWorld<RES>(sc, h);
              castDone = true;
              mode = SM_GEN;
            }
          }
        }
      }
    }

    // ================================ TRAV =========================================================
    if (__ballot(mode == SM_TRAV)) {
      for (int step = 0; step < travSteps; ++step) {
        const bool inTrav = mode == SM_TRAV;
        // ---- inner nodes: up to 3 descents, stop as soon as no lane holds an inner node ----------------
        for (int d = 0; d < 3; ++d) {
          const bool inner = inTrav && mode == SM_TRAV && !(cur & QA_BVH_LEAF_BIT);
          if (!__any(inner)) break;
          const bool fastSlab = !__any(inner && degenerate);
          if (inner) {
            if (STATS) cnt.bvh_nodes++;
            const DMesh &m = meshAt<RES>(sc, instAt<RES>(sc, k).mesh);
            const uint4 *nodes = RES ? mem.img + m.resNodes : reinterpret_cast<const uint4 *>(m.nodes);
            const uint4 *pair = nodes + 2 * (size_t) (cur & QA_BVH_CHILD_MASK);
            const uint4 a0 = pair[0], a1 = pair[1], b0 = pair[2], b1 = pair[3];
            float entry0, exit0, entry1, exit1;
            const f3 min0 = F3(asF(a0.x), asF(a0.y), asF(a0.z)), max0 = F3(asF(a0.w), asF(a1.x), asF(a1.y));
            const f3 min1 = F3(asF(b0.x), asF(b0.y), asF(b0.z)), max1 = F3(asF(b0.w), asF(b1.x), asF(b1.y));
            if (fastSlab) {
              boxEntryExitFast(lr, drcp, min0, max0, entry0, exit0);
              boxEntryExitFast(lr, drcp, min1, max1, entry1, exit1);
            } else {
              boxEntryExit(lr, drcp, min0, max0, entry0, exit0);
              boxEntryExit(lr, drcp, min1, max1, entry1, exit1);
            }
            const float t_max = h.z;
            const bool hit0 = (entry0 < t_max && entry0 < exit0);
            const bool hit1 = (entry1 < t_max && entry1 < exit1);
            const uint32_t d0 = a1.z, d1 = b1.z;
            if (hit0 && hit1) {
              const bool nearFirst = entry0 < entry1;
              stack[(sp++) * QA_BLOCK] = nearFirst ? d1 : d0;
              cur = nearFirst ? d0 : d1;
            } else if (hit0) cur = d0;
            else if (hit1) cur = d1;
            else cur = sp ? stack[(--sp) * QA_BLOCK] : QA_DONE;
          }
        }
        // ---- leaves ---------------------------------------------------------------------------------------
        if (inTrav && mode == SM_TRAV && (cur & QA_BVH_LEAF_BIT) && cur != QA_DONE) {
          if (STATS) cnt.bvh_nodes++;
          const DMesh &m = meshAt<RES>(sc, instAt<RES>(sc, k).mesh);
          const uint4 *tris = RES ? mem.img + m.resTris : reinterpret_cast<const uint4 *>(m.tris);
          const uint32_t count = ((cur >> QA_BVH_COUNT_SHIFT) & QA_BVH_COUNT_MASK) + 1;
          const uint32_t first = cur & QA_BVH_OFFSET_MASK;
          bool stop = false;
          for (uint32_t i = 0; i < count && !stop; ++i) {
            if (STATS) cnt.tri_tests++;
            const uint4 *t = tris + 3 * (size_t) (first + i);
            if (hitTriangleZ(t[0], t[1], t[2], lr, h.z)) {
              meshHit = true;
              pick.tri = first + i;
              if (castShadow) stop = true;
            }
          }
          cur = (stop || !sp) ? QA_DONE : stack[(--sp) * QA_BLOCK];
        }
        // ---- mesh finished -----------------------------------------------------------------------------------
        if (inTrav && mode == SM_TRAV && cur == QA_DONE) {
          if (meshHit) {
            if (castShadow) { occluded = true; k = sc.num_inst; }
            else {
              const DMesh &m = meshAt<RES>(sc, instAt<RES>(sc, k).mesh);
              {
                const uint4 *t = (RES ? mem.img + m.resTris : reinterpret_cast<const uint4 *>(m.tris)) + 3 * (size_t) pick.tri;
                triangleDetails(t[0], t[1], t[2], lr, h, pick.a, pick.b);
              }
              const uint4 *s = (RES ? mem.img + m.resShade : reinterpret_cast<const uint4 *>(m.shade)) + 3 * (size_t) pick.tri;
              const uint4 s0 = s[0], s1 = s[1], s2 = s[2];
              const float bc = 1.f - pick.a - pick.b;
              const f3 n0 = F3(asF(s0.x), asF(s0.y), asF(s0.z)), n1 = F3(asF(s0.w), asF(s1.x), asF(s1.y)),
                       n2 = F3(asF(s1.z), asF(s1.w), asF(s2.x));
              h.N = (n0 * pick.a + n1 * pick.b) + n2 * bc;
              h.mtlID = (int) s2.y;
              h.node = k;
              if (TEX && m.hasVT) {
                const uint4 *t = (RES ? mem.img + m.resTris : reinterpret_cast<const uint4 *>(m.tris)) + 3 * (size_t) pick.tri;
                texTriangle(t[0], t[1], t[2], m.vt + 6 * (size_t) pick.tri, lr.p, lrd.dx, lrd.dy, pick.a, pick.b, th);
              }
            }
          }
          ++k;
          while (k < sc.num_inst && instAt<RES>(sc, k).obj_type == QA_OBJ_NONE) ++k;
          if (k >= sc.num_inst) {
            if (!castShadow && h.node >= 0) hitToWorld<RES>(sc, h);
            castDone = true;
            mode = SM_GEN;
          } else mode = SM_INST;
        }
        if (!__any(mode == SM_TRAV)) break;
      }
    }
  }

  // ---- counters ---------------------------------------------------------------------------------------
  unsigned long long v[6] = {cnt.samples, cnt.casts_normal, cnt.casts_shadow, cnt.bvh_nodes, cnt.tri_tests, cnt.pixels};
  unsigned long long *dst = reinterpret_cast<unsigned long long *>(rp.counters);
  for (int i = 0; i < 6; ++i) {
    unsigned long long x = v[i];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    if (lane == 0 && x) atomicAdd(&dst[i], x);
  }
}

}  // namespace qa
