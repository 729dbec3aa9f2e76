// qa_scene_dev.h — device-side view of an uploaded scene.
//
// qa_scene_upload keeps the flat blob (include/qa_flat_scene.h) in HBM verbatim and derives from
// it the arrays the traversal actually streams:
//
//   DNode   32 B   cy::BVH node padded from 28 B so a sibling pair (children are adjacent:
//                  src/ext/cyBVH.h:92-116) is one aligned 64-byte read (4 x 16 B)
//   DTri    48 B   per-triangle intersection record IN LEAF (element) ORDER, so a leaf's
//                  triangles are contiguous: precomputed face normal, projection axis and
//                  1/area - the values TriObj::IntersectTriangle recomputes for every test
//                  (src/objects/objects.cpp:212-246); precomputing them on the host with the same
//                  fp32 operations yields the same bits.  3 x 16 B.
//   DTriShade 48 B per-triangle shading inputs (vertex normals, material id), read once per
//                  accepted hit
//   halton         (Halton(s,11), Halton(s,13)) per sample index - identical for every pixel
//                  (src/scene/scene.cpp:99-102)
//
// Small scenes (every BASELINE config whose assets exist) are additionally packed into ONE
// "resident image" that each workgroup copies into LDS at kernel start: nodes, triangle records,
// shading records and materials of all meshes (layout: ResidentLayout).  The traversal then never
// leaves the CU.
#pragma once
#include <stdint.h>

#include "qa_flat_scene.h"
#include "qa_photon.h"

namespace qa {

struct alignas(32) DNode {
  float box[6];
  uint32_t data;  // qa_bvh_node::data
  uint32_t pad;
};

struct alignas(16) DTri {
  float N[3];      // normalize(cross(B-A, C-A))
  float A[3];      // (au, av) are the two components of A that survive dropping `axis`
  float bu, bv, cu, cv;  // B and C projected on the plane that drops `axis`
  float s;         // 1 / TriangleArea(axis, A, B, C)
  uint32_t axis;   // 0,1,2: dominant axis of N
};
static_assert(sizeof(DTri) == 48, "DTri must be 3 x 16 bytes");

struct alignas(16) DTriShade {
  float n0[3], n1[3], n2[3];   // vertex normals (TriMesh::GetNormal interpolates them)
  int32_t mtl;
  uint32_t face;               // original face id (texture coordinates are looked up by it)
  uint32_t pad;                // id of the reference-tree leaf that holds this element; see refReaches
};
static_assert(sizeof(DTriShade) == 48, "DTriShade must be 3 x 16 bytes");

// Material record in the order the shader reads it (no texture references: plain colours)
struct alignas(16) DMaterial {
  float diffuse[3], kill;
  float specular[3], gloss_spec;
  float emission[3], ior;
  float reflection[3], gloss_refl;
  float refraction[3], gloss_refr;
  float absorption[3];
  uint32_t flags;              // bit0: reflection or refraction colour non-zero, bit1: specular non-zero
};
static_assert(sizeof(DMaterial) == 96, "DMaterial must be 6 x 16 bytes");
#define QA_MTL_SPECULAR_LOBES 1u
#define QA_MTL_HAS_SPECULAR 2u

// Node of the library's own 4-wide search tree for global-memory meshes (qa_widebvh.h), 64 bytes = 4 x 16:
//   q0 = origin.xyz, scale.x         q1 = scale.y, scale.z, lo.x[4 children as bytes], lo.y[4]
//   q2 = lo.z[4], hi.x[4], hi.y[4], hi.z[4]      q3 = child words
// Child c's box on axis a is [origin.a + lo.a[c] * scale.a, origin.a + hi.a[c] * scale.a] (one fma each), quantised
// OUTWARDS on the node's own 8-bit grid (scale = a power of two): the search only needs boxes that CONTAIN the
// reference's leaf boxes, and the traversal is bound by the number of scattered 16-byte accesses, not by arithmetic.
// Child words: inner = node index (breadth-first numbering: the first nodes of a tree are its top levels, which the
// staged trace stage can keep in LDS); leaf = flag + range of DMesh::wtris (the format of the reference's leaf words);
// QA_DONE = empty.
struct alignas(64) DWideNode {
  float origin[3];
  float scale[3];
  uint32_t lo[3];     // byte c = child c
  uint32_t hi[3];
  uint32_t child[4];
};
static_assert(sizeof(DWideNode) == 64, "DWideNode must be 4 x 16 bytes");

struct DMesh {
  float bmin[3], bmax[3];
  const DNode *nodes;          // [num_nodes], root = 1 (global memory copy)
  const DTri *tris;            // [num_faces] in element order
  const DTriShade *shade;      // [num_faces] in element order
  uint32_t num_faces, num_nodes;
  uint32_t rootData;           // nodes[1].data
  uint32_t stackNeed;          // deepest traversal stack this BVH can require
  // offsets (in 16-byte units) of this mesh's arrays inside the resident image
  uint32_t resNodes, resTris, resShade;
  uint32_t hasVT;              // every face carries texture vertices (mixed meshes are refused)
  const float *vt;             // [6 * num_faces] texture vertices per triangle, element order
  // the library's own search tree over the same triangles (qa_fastbvh.h); non-counting kernels walk it
  const DNode *fnodes;         // same node format and numbering rules as `nodes`
  const DTri *ftris;           // the records of `tris`, in this tree's leaf order, DTri::axis = axis | element << 2 | reference leaf << 17
  const uint32_t *fmap;        // element of this tree -> element of the reference tree
  uint32_t frootData;
  uint32_t useFast;            // 0: this mesh is searched with the reference tree only
  float invH;                  // 1 / smallest triangle altitude of the mesh (own-tree box widening, hitMesh)
  float absMax;                // largest |coordinate| of the mesh bounds
  uint32_t numNormals;         // distinct face normals (up to sign) listed at resNormals (float4 each); 0 = no list
  uint32_t resNormals;
  uint32_t resFNodes, resFTris, resFMap;
  uint32_t gateIsRoot;         // the mesh bounds equal the root box of the reference tree bit for bit
  // the 4-wide tree over the triangles (qa_widebvh.h; global-memory scenes, non-counting kernels)
  const DWideNode *wnodes;
  const DTri *wtris;           // the records of `tris` in that tree's leaf order, DTri::axis = axis | element << 2
  uint32_t wrootWord;          // child word of the root
  uint32_t useWide;
  uint32_t wnodeCount;         // nodes of the wide tree
  uint32_t wideStack;          // stack entries a walk of the wide tree can need (3 per level + 2)
  float nearPad;               // fp32 slack of the reference's inside test (qa_widebvh.h ComputeMeshSlack)
  float cancelDist;            // ray origins farther out than this keep the reference tree
  uint32_t csRootWord;         // root of this mesh's 4-wide tree inside the scene-wide arrays DScene::csNodes / csTris (qa_kernel_cs.h)
  uint32_t csPad;
};

// The widening of the own search trees' boxes and the parallelism / cancellation guards (qa_kernel.h hitMesh, qa_wf.h,
// qa_widebvh.h) all scale with this factor.  It is 1 in the product; `make hip_noslack` builds a test-only library with
// 0 (lib_noslack/), with which tests/test_gpu_parity.py shows that the constants matter: without them the own trees lose
// hits the reference accepts.
#ifndef QA_SLACK_SCALE
#define QA_SLACK_SCALE 1.0f
#endif

// qa_integrate_cs's view of a scene-graph node: everything a sweep over the instances needs in ONE record (one group of
// scalar loads per instance instead of the chain instance -> parent -> ... -> mesh descriptor).  Levels: the transforms between
// the root's space and the node's own, outermost first (A, then B for a node inside a group; deeper nesting keeps qa_integrate).
struct alignas(64) CsInst {
  float itmA[9], posA[3];
  float itmB[9], posB[3];
  float tmA[9], tmB[9];        // Node::FromNodeCoords of the two levels
  int32_t type, depth;         // QA_OBJ_*, 1 or 2
  uint32_t useWide, csRootWord, num_faces, mesh;
  float bmin[3], bmax[3];      // mesh bounds (node space)
  float nearPad, absMax, cancelDist;
  float wmin[3], wmax[3];      // bounds of the object in ROOT space, padded (instance culling)
  int32_t parent;              // depth 2: the group node (consecutive children of one group share its level-A ray)
};
static_assert(sizeof(CsInst) == 256, "CsInst must be 64 dwords");

// Instance culling (qa_integrate_cs's sweeps): bounds of a scene-graph node's object in ROOT space (the eight corners of
// its node-space box through tm * p + pos of every level, evaluated in double on the host), one scalar load of 8 dwords.
// A node without an object carries an empty box (lo > hi: never entered).
struct alignas(32) CsCull {
  float lo[3], pad0;
  float hi[3], pad1;
};
static_assert(sizeof(CsCull) == 32, "CsCull must be 8 dwords");

#define QA_LANE_SLOTS 6   /* per-lane LDS floats behind the traversal stack: running mean and variance of the pixel */
#define QA_LANE_SLOTS_RES 15   /* LDS-resident scenes: + throughput, radiance, pixel, output index, sample index */
#define QA_KARG_INST 12   /* scene-graph nodes / meshes a resident scene may pass by value */
#define QA_KARG_MESH 4

struct DCamera {
  float screenA[3], screenU[3], screenV[3], screenX[3], screenY[3], pos[3];
  float dof;
  int32_t width, height;
};

struct DScene {
  const unsigned char *blob;
  const qa_instance *inst;
  const qa_mtlset *mtlset;
  const DMaterial *mtl;        // [num_materials] (global memory copy)
  const qa_light *light;
  const DMesh *mesh;
  const float *halton;         // 2 floats per sample index, [halton_count]
  const uint4 *resident;       // resident image (nodes | tris | shade | materials), or nullptr
  const qa_texmap *texmap;     // TEX variants: tables inside the blob
  const qa_texture *tex;
  const float4 *texels;        // file textures as float RGB, 16 bytes per texel (qa_texture_dev.h), texture i from texOff[i]
  const uint32_t *texOff;
  const float *texFilter;      // 31 x (x, y) elliptical filter taps (src/core/texture.cpp:39-46)
  float *areaScratch;          // AREA variants: [QA_MAX_PATH * 19][grid threads] hit log
  const int32_t *mtlTex;       // 8 ints per material: texmap of diffuse, specular, emission, reflection, refraction
  DCamera cam;
  float background[3], environment[3];
  int32_t num_inst, num_lights, halton_count, num_materials;
  uint32_t residentVec4;       // size of the resident image in 16-byte units (0 = not resident)
  uint32_t resMaterials;       // offset of the material table inside it (16-byte units)
  uint32_t stackNeed;          // max over meshes
  uint32_t rootIdentity;       // instance 0 has tm = itm = I and pos = 0 (always, for XML scenes)
  uint32_t stackDepth;         // entries per lane of the LDS traversal stack
  int32_t bgTexmap, envTexmap; // texmaps of the background / environment colours (-1 = none)
  uint32_t csPoolLimit;        // qa_integrate_cs: upper bound for the pool capacity (0 = none; option "cs_pool_limit", tests: forces the overflow path)
  // qa_integrate_cs (qa_kernel_cs.h): the 4-wide trees of ALL meshes in one node array and one triangle array (child words and
  // leaf ranges rebased), so that a pool item needs no mesh: csNodes[4 * node], csTris[3 * triangle]; csLeafBox[2 * triangle] =
  // box of the triangle's leaf in the REFERENCE tree (6 floats, then 1.0f when that leaf is the root: always reached)
  const uint4 *csNodes;
  const uint4 *csTris;
  const uint4 *csLeafBox;
  const CsInst *csInst;        // [num_inst]
  // Instance culling: csCull[k] = root-space bounds of node k's object; a ray whose origin has the largest coordinate oMax
  // tests them widened by (oMax + csCullS1) * (oMax + csCullS2) * csCullK3 + csCullK4 (qa_kernel_cs.h csCullPad: what the
  // fp32 node transforms can move a hit point by, with margin).  csCullOn = 0: every instance is visited (option "cs_cull").
  const CsCull *csCull;        // [num_inst]
  float csCullS1, csCullS2, csCullK3, csCullK4;
  uint32_t csCullOn;
  float *csSurf;               // scenes with more than QA_CS_LIGHT_BATCH shadow-casting lights: [13][grid lanes] surface columns (qa_kernel_cs.h), else nullptr
  uint32_t walkZeroTerms;      // tests (option "walk_zero_terms"): shadow rays of lights whose unshadowed term is zero are walked too (same frame, slower)
  uint32_t csForceExact;       // tests (option "cs_force_exact"): bit 0 = every closest-hit query, bit 1 = every shadow query goes to the exact walks
  uint32_t csItems, csSlots;   // per-wave pool capacity (items) and ray slots of the LDS layout
  // resident scenes only: the tables themselves, in the kernel-argument segment
  qa_instance instv[QA_KARG_INST];
  DMesh meshv[QA_KARG_MESH];
};

struct DCounters {
  unsigned long long samples, casts_normal, casts_shadow, bvh_nodes, tri_tests, pixels;
#ifdef QA_STAMPS
  // Diagnostic builds (make hip EXTRA=-DQA_STAMPS, tools/gpu_stamps.py): shader-clock cycles of qa_integrate's sections summed
  // over waves, printed by qa_get_counters: 0 kernel, 1 fetch + sample start, 2 closest-hit queries, 3 of which mesh walks,
  // 4 shadeSurface, 5 direct light, 6 of which shadow mesh walks, 7 sample end, 8 loop iterations, 9 waves, 10 miss branch,
  // 11 hit before shading, 12 spawn.
  // qa_integrate_cs also: 13 closest-hit sweep without the rounds, 14 details of the winners, 15 lanes sent to the exact closest-hit
  // walk, 16 (lane, light) pairs sent to the exact shadow walk, 17 shadow sweeps without the rounds, 18 light terms
#define QA_NSTAMPS 20
  unsigned long long stamp[QA_NSTAMPS];
  unsigned long long *sl;   // device side: the wave's accumulators in LDS (one elected lane adds: a section entered by part of the wave counts in full)
#endif
};

// One balanced photon map in HBM: [0] unused, [1..count] the kd-tree in heap order
// (cyPhotonMap.h:272-292).  Next to the byte-compatible qa_photon records the build keeps what the
// gather needs in three 16-byte tables, with everything cy::PhotonMap::Photon decodes on every
// access (GetDirection's integer square root and divisions, GetPower's colour * power) evaluated
// once per photon on the host with the same fp32 operations:
//   node[i]  = (position.xyz, split axis)            read at every visited node
//   dir[i]   = (direction.xyz, GetMaxPower())        read for photons inside the search radius
//   power[i] = (GetPower() rgb, -)                   read for the <= 100 photons that are summed
struct DPhotonMap {
  const uint4 *node;
  const float4 *dir;
  const float4 *power;
  int32_t half;       // PhotonMap::halfStoredPhotons
  uint32_t count;
  float radius;
  uint32_t pad;
};

struct RenderParams {
  int32_t x0, y0, x1, y1;      // region
  int32_t spp_min, spp_max, max_bounce;
  uint32_t seed;
  int32_t tile_row0, tile_row_step, own_tile_rows, pad;  // which 8-row strips of the region
  int32_t sync_samples;        // 1: a wave starts its lanes' next samples together (coherent primary rays)
  float *rgb;                  // region-local outputs
  float *depth;
  uint32_t *ns;
  const uint32_t *tile_order;  // launch order of the tiles (centre of the region first), or nullptr
  unsigned int *work_counter;  // next work item (pixel) of this launch
  const volatile int *stop_flag;
  DCounters *counters;
  // Tiles in sample chunks (qa_integrate, section A): chunk_spp > 0 = a work item is (chunk, tile) - a tile's pixels for chunk_spp
  // samples; a pixel's state between chunks (RNG state, samples taken | finished flag, running mean and variance: 8 words) waits in
  // pix_state[8 x output index], tile_progress[tile] = chunks of the tile that are complete (zeroed before the launch)
  uint32_t chunk_spp, chunk_tail, num_chunks, chunk_pad;   // the first chunk's samples, every further chunk's, how many chunks
  uint32_t *tile_progress;
  uint32_t *pix_state;
  // PHOTON kernel variants (Scene::usePhotonMap): [0] photon map, [1] caustics map, and the per-lane
  // nearest-photon heaps ([grid threads][QA_PHOTON_GATHER + 1] x (distance^2, photon index))
  DPhotonMap pm[2];
  uint2 *heap;
};

}  // namespace qa
