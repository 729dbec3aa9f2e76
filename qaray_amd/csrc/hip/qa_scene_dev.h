// qa_scene_dev.h — device-side view of an uploaded scene.
//
// qa_scene_upload keeps the flat blob (include/qa_flat_scene.h) in HBM verbatim and derives from
// it the arrays the traversal actually streams:
//
//   DNode   32 B   cy::BVH node padded from 28 B so a sibling pair (children are adjacent:
//                  src/ext/cyBVH.h:92-116) is one aligned 64-byte read
//   DTri    64 B   per-triangle intersection record IN LEAF (element) ORDER, so a leaf's
//                  triangles are contiguous: precomputed face normal, projection axis and
//                  1/area - the values TriObj::IntersectTriangle recomputes for every test
//                  (src/objects/objects.cpp:212-246); precomputing them on the host with the same
//                  fp32 operations yields the same bits
//   DTriShade      per-triangle shading inputs (vertex normals, uv, material id), read once per
//                  accepted hit
//   halton         (Halton(s,11), Halton(s,13)) per sample index - identical for every pixel
//                  (src/scene/scene.cpp:99-102)
#pragma once
#include <stdint.h>

#include "qa_flat_scene.h"

namespace qa {

struct alignas(32) DNode {
  float box[6];
  uint32_t data;  // qa_bvh_node::data
  uint32_t pad;
};

struct alignas(64) DTri {
  float N[3];      // normalize(cross(B-A, C-A))
  float A[3];
  float au, av, bu, bv, cu, cv;  // vertices projected on the plane that drops `axis`
  float s;         // 1 / TriangleArea(axis, A, B, C)
  uint32_t axis;   // 0,1,2: dominant axis of N
  uint32_t face;   // original face id
  uint32_t pad;
};

struct DTriShade {
  float n0[3], n1[3], n2[3];   // vertex normals (TriMesh::GetNormal interpolates them)
  float t0[2], t1[2], t2[2];   // texture vertices (valid when hasVT)
  int32_t mtl;
  int32_t hasVT;
};

struct DMesh {
  float bmin[3], bmax[3];
  const DNode *nodes;          // [num_nodes], root = 1
  const DTri *tris;            // [num_faces] in element order
  const DTriShade *shade;      // [num_faces] in element order
  uint32_t num_faces, num_nodes;
  uint32_t rootData;           // nodes[1].data
  uint32_t pad;
};

struct DCamera {
  float screenA[3], screenU[3], screenV[3], screenX[3], screenY[3], pos[3];
  float dof;
  int32_t width, height;
};

struct DScene {
  const unsigned char *blob;
  const qa_instance *inst;
  const qa_mtlset *mtlset;
  const qa_material *mtl;
  const qa_light *light;
  const qa_texmap *texmap;
  const qa_texture *tex;
  const DMesh *mesh;
  const float *halton;         // 2 floats per sample index, [halton_count]
  DCamera cam;
  qa_texcolor background, environment;
  int32_t num_inst, num_lights, halton_count, pad;
};

struct DCounters {
  unsigned long long samples, casts_normal, casts_shadow, bvh_nodes, tri_tests, pixels;
};

struct RenderParams {
  int32_t x0, y0, x1, y1;      // region
  int32_t spp_min, spp_max, max_bounce;
  uint32_t seed;
  int32_t tile_row0, tile_row_step, own_tile_rows, pad;  // which 8-row strips of the region
  float *rgb;                  // region-local outputs
  float *depth;
  uint32_t *ns;
  unsigned int *work_counter;  // next work item (pixel) of this launch
  const volatile int *stop_flag;
  DCounters *counters;
};

}  // namespace qa
