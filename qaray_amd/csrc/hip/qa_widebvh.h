// qa_widebvh.h — the HIP library's own 4-wide search tree for meshes that live in global memory (host code,
// used by qa_scene_upload).
//
// The reference searches a mesh with cy::BVH (binary, split at the centre of the longest axis, up to four -
// sometimes eight - triangles per leaf, src/ext/cyBVH.h:318-421): 30 - 60 dependent node reads and ~20 triangle
// tests per ray on the BASELINE meshes.  The closest hit does not depend on the tree it is searched with, as long as
// (a) every triangle the reference can ACCEPT at or before the distance held is also tested here, and (b) the answer
// is checked against the reference's own rules afterwards (qa_kernel.h hitMesh, qa_wf.h: the found triangle's leaf
// in the reference tree must pass the reference's strict box test, no tie may have been seen).  For (a) the boxes of
// this tree are unions of TRIANGLE bounds (vertex floats, exact), tested non-strictly and widened per ray by the
// fp32 slack of the reference's inside test (ComputeMeshSlack below): an accepted hit point lies within that slack
// of its triangle, hence inside every widened box above the triangle, so the walk cannot prune it.
// Inner structure: binned surface-area heuristic over the triangles down to leaves of at most `leafMax` triangles,
// collapsed to four children per node, numbered breadth-first; one node = 64 bytes: the four child boxes quantised
// outwards to 8 bits per plane on the node's own grid, and four child words (DWideNode, qa_scene_dev.h).  The
// triangles are stored once more in this tree's leaf order (DMesh::wtris), each record carrying its element id.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "qa_scene_dev.h"

namespace qa {

struct WideBvh {
  std::vector<DWideNode> nodes;   // [0] = root (when the mesh has more than one leaf)
  std::vector<uint32_t> order;    // position in this tree's leaf order -> element (triangle in the reference's order)
  uint32_t rootWord = QA_DONE;    // child word of the root: inner index, or a leaf word when the mesh is one leaf
  uint32_t depth = 0;             // wide nodes on the longest root-to-leaf path
};

class WideBvhBuilder {
 public:
  // triBox: 6 floats (min, max) per element; skip[e] != 0 leaves element e out (degenerate: never accepted)
  WideBvhBuilder(const float *triBox, const unsigned char *skip, uint32_t numElements, uint32_t leafMax)
      : box_(triBox), skip_(skip), numElements_(numElements), leafMax_(std::min(std::max(leafMax, 1u), QA_BVH_COUNT_MASK + 1u)) {}

  void Run(WideBvh &out)
  {
    out_ = &out;
    out.nodes.clear();
    out.depth = 0;
    out.order.clear();
    for (uint32_t e = 0; e < numElements_; ++e) if (!skip_ || !skip_[e]) prims_.push_back(e);
    const uint32_t n = (uint32_t) prims_.size();
    if (n == 0) { out.rootWord = QA_DONE; return; }
    order_.resize(n);
    for (uint32_t i = 0; i < n; ++i) order_[i] = i;
    cen_.resize(3 * (size_t) n);
    for (uint32_t i = 0; i < n; ++i)
      for (int k = 0; k < 3; ++k) cen_[3 * (size_t) i + k] = 0.5f * (Box(i)[k] + Box(i)[3 + k]);
    // binary SAH tree over the leaves
    bin_.clear();
    bin_.reserve(2 * (size_t) n);
    BuildBinary(0, n, 0);
    // collapse to four children per node, then number breadth-first and quantise
    tmp_.clear();
    tmp_.reserve(n / 2 + 4);
    out.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) out.order[i] = prims_[order_[i]];
    if (bin_[0].left == ~0u) { out.rootWord = bin_[0].prim; return; }   // the whole mesh is one leaf
    const uint32_t root = Collapse(0, 1);
    Finish(root);
    out.rootWord = 0;   // breadth-first: the root is node 0
  }

 private:
  struct BinNode { float box[6]; uint32_t left, right; uint32_t prim; };   // leaf: left = right = ~0u, prim = its child word
  const float *Box(uint32_t primIdx) const { return box_ + 6 * (size_t) prims_[primIdx]; }

  static float HalfArea(const float *b)
  {
    const float x = b[3] - b[0], y = b[4] - b[1], z = b[5] - b[2];
    return x * y + y * z + z * x;
  }
  static void Grow(float *b, const float *o)
  {
    for (int k = 0; k < 3; ++k) { b[k] = std::min(b[k], o[k]); b[3 + k] = std::max(b[3 + k], o[3 + k]); }
  }
  static void Empty(float *b) { b[0] = b[1] = b[2] = 1e30f; b[3] = b[4] = b[5] = -1e30f; }

  uint32_t BuildBinary(uint32_t first, uint32_t count, int level)
  {
    const uint32_t id = (uint32_t) bin_.size();
    bin_.push_back(BinNode{});
    float box[6];
    Empty(box);
    for (uint32_t i = 0; i < count; ++i) Grow(box, Box(order_[first + i]));
    memcpy(bin_[id].box, box, sizeof(box));
    if (count <= leafMax_) {
      bin_[id].left = bin_[id].right = ~0u;
      bin_[id].prim = QA_BVH_LEAF_BIT | ((count - 1) << QA_BVH_COUNT_SHIFT) | first;   // range of DMesh::wtris
      return id;
    }
    // binned SAH over the centroid bounds (16 bins per axis); median split when no bin boundary separates
    float cb[6];
    Empty(cb);
    for (uint32_t i = 0; i < count; ++i) {
      const float *c = &cen_[3 * (size_t) order_[first + i]];
      for (int k = 0; k < 3; ++k) { cb[k] = std::min(cb[k], c[k]); cb[3 + k] = std::max(cb[3 + k], c[k]); }
    }
    const int kBins = 16;
    float bestCost = 1e30f;
    int bestAxis = -1, bestBin = -1;
    if (level < 48) {
      for (int axis = 0; axis < 3; ++axis) {
        const float lo = cb[axis], ext = cb[3 + axis] - lo;
        if (!(ext > 0)) continue;
        float bb[kBins][6];
        uint32_t bc[kBins];
        for (int q = 0; q < kBins; ++q) { Empty(bb[q]); bc[q] = 0; }
        const float scale = (float) kBins / ext;
        for (uint32_t i = 0; i < count; ++i) {
          const uint32_t p = order_[first + i];
          int q = (int) ((cen_[3 * (size_t) p + axis] - lo) * scale);
          q = std::min(std::max(q, 0), kBins - 1);
          Grow(bb[q], Box(p));
          bc[q]++;
        }
        float ra[kBins];
        uint32_t rc[kBins];
        float acc[6];
        Empty(acc);
        uint32_t cnt = 0;
        for (int q = kBins - 1; q > 0; --q) {
          if (bc[q]) Grow(acc, bb[q]);
          cnt += bc[q];
          ra[q] = cnt ? HalfArea(acc) : 0.f;
          rc[q] = cnt;
        }
        Empty(acc);
        cnt = 0;
        for (int q = 0; q < kBins - 1; ++q) {
          if (bc[q]) Grow(acc, bb[q]);
          cnt += bc[q];
          if (cnt == 0 || rc[q + 1] == 0) continue;
          const float cost = HalfArea(acc) * (float) cnt + ra[q + 1] * (float) rc[q + 1];
          if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = q; }
        }
      }
    }
    uint32_t nLeft = 0;
    if (bestAxis >= 0) {
      const float lo = cb[bestAxis], scale = (float) kBins / (cb[3 + bestAxis] - lo);
      uint32_t *e = order_.data() + first;
      uint32_t i = 0, j = count;
      while (i < j) {
        int q = (int) ((cen_[3 * (size_t) e[i] + bestAxis] - lo) * scale);
        q = std::min(std::max(q, 0), kBins - 1);
        if (q <= bestBin) ++i;
        else { --j; std::swap(e[i], e[j]); }
      }
      nLeft = i;
    }
    if (nLeft == 0 || nLeft == count) {
      // all centroids alike (or a very deep tree): halve along the longest axis of the node
      int axis = 0;
      for (int k = 1; k < 3; ++k) if (box[3 + k] - box[k] > box[3 + axis] - box[axis]) axis = k;
      uint32_t *e = order_.data() + first;
      std::nth_element(e, e + count / 2, e + count, [&](uint32_t a, uint32_t b) { return cen_[3 * (size_t) a + axis] < cen_[3 * (size_t) b + axis]; });
      nLeft = count / 2;
    }
    const uint32_t l = BuildBinary(first, nLeft, level + 1);
    const uint32_t r = BuildBinary(first + nLeft, count - nLeft, level + 1);
    bin_[id].left = l;
    bin_[id].right = r;
    return id;
  }

  struct TmpNode { float lo[4][3], hi[4][3]; uint32_t child[4]; bool inner[4]; int n; };

  // -> index (in tmp_) of the wide node made from binary node `b` (an inner one)
  uint32_t Collapse(uint32_t b, uint32_t level)
  {
    if (level > out_->depth) out_->depth = level;
    uint32_t kids[4];
    int nk = 2;
    kids[0] = bin_[b].left;
    kids[1] = bin_[b].right;
    while (nk < 4) {
      int pick = -1;
      float best = -1.f;
      for (int i = 0; i < nk; ++i)
        if (bin_[kids[i]].left != ~0u) {
          const float a = HalfArea(bin_[kids[i]].box);
          if (a > best) { best = a; pick = i; }
        }
      if (pick < 0) break;
      const uint32_t k = kids[pick];
      kids[pick] = bin_[k].left;
      kids[nk++] = bin_[k].right;
    }
    const uint32_t w = (uint32_t) tmp_.size();
    tmp_.push_back(TmpNode{});
    TmpNode nd;
    nd.n = nk;
    for (int i = 0; i < 4; ++i) { nd.child[i] = QA_DONE; nd.inner[i] = false; }
    for (int i = 0; i < nk; ++i) {
      const BinNode &c = bin_[kids[i]];
      for (int k = 0; k < 3; ++k) { nd.lo[i][k] = c.box[k]; nd.hi[i][k] = c.box[3 + k]; }
      if (c.left == ~0u) nd.child[i] = c.prim;
      else { nd.child[i] = Collapse(kids[i], level + 1); nd.inner[i] = true; }
    }
    tmp_[w] = nd;
    return w;
  }

  static float Decode(float origin, float scale, uint32_t q) { return std::fmaf((float) q, scale, origin); }

  // breadth-first numbering + outward quantisation (checked with the very fma the device evaluates)
  void Finish(uint32_t root)
  {
    std::vector<uint32_t> order, newId(tmp_.size(), 0);
    order.reserve(tmp_.size());
    order.push_back(root);
    for (size_t h = 0; h < order.size(); ++h) {
      const TmpNode &t = tmp_[order[h]];
      for (int i = 0; i < t.n; ++i) if (t.inner[i]) order.push_back(t.child[i]);
    }
    for (size_t i = 0; i < order.size(); ++i) newId[order[i]] = (uint32_t) i;
    out_->nodes.assign(order.size(), DWideNode{});
    for (size_t i = 0; i < order.size(); ++i) {
      const TmpNode &t = tmp_[order[i]];
      DWideNode &d = out_->nodes[i];
      for (int k = 0; k < 3; ++k) {
        float lo = 1e30f, hi = -1e30f;
        for (int c = 0; c < t.n; ++c) { lo = std::min(lo, t.lo[c][k]); hi = std::max(hi, t.hi[c][k]); }
        d.origin[k] = lo;
        int e;
        const float ext = hi - lo;
        float scale = 1.17549435e-38f;                              // degenerate extent: any grid will do
        if (ext > 0) { std::frexp(ext / 255.0f, &e); scale = std::ldexp(1.0f, e); }   // power of two >= ext / 255
        while (Decode(lo, scale, 255) < hi) scale *= 2.f;
        d.scale[k] = scale;
        d.lo[k] = d.hi[k] = 0;
        for (int c = 0; c < t.n; ++c) {
          int ql = (int) std::floor((t.lo[c][k] - lo) / scale), qh = (int) std::ceil((t.hi[c][k] - lo) / scale);
          ql = std::min(std::max(ql, 0), 255);
          qh = std::min(std::max(qh, 0), 255);
          while (ql > 0 && Decode(lo, scale, (uint32_t) ql) > t.lo[c][k]) --ql;
          while (qh < 255 && Decode(lo, scale, (uint32_t) qh) < t.hi[c][k]) ++qh;
          d.lo[k] |= (uint32_t) ql << (8 * c);
          d.hi[k] |= (uint32_t) qh << (8 * c);
        }
      }
      for (int c = 0; c < 4; ++c) d.child[c] = (c < t.n) ? (t.inner[c] ? newId[t.child[c]] : t.child[c]) : QA_DONE;
    }
  }

  std::vector<TmpNode> tmp_;
  const float *box_;
  const unsigned char *skip_;
  uint32_t numElements_, leafMax_;
  WideBvh *out_ = nullptr;
  std::vector<uint32_t> prims_, order_;
  std::vector<float> cen_;
  std::vector<BinNode> bin_;
};

// fp32 slack of the reference's inside test (TriObj::IntersectTriangle, src/objects/objects.cpp:212-306), per mesh.
//   nearPad: how far outside a triangle a point can lie and still be accepted, as long as the areas do not cancel:
//            the barycentrics are 2-D signed areas u1*v1 - u2*v2 of magnitude <= (L + d)^2 scaled by 1 / (2 A): each is off
//            by <= 8 eps (L + d)^2 / (2 A), i.e. a point up to 8 eps L^2 / |edge| outside an edge passes; three edges,
//            the third barycentric is a difference of the other two, and the dropped axis shortens in-plane lengths by
//            at most sqrt(3): 48 eps L^2 / (shortest projected edge), maximised over the triangles.
//   cancelDist: a point farther than |edge| / (8 eps) - 2 L from a triangle can pass the test by cancellation of the
//            areas.  That matters to a pruned search only when such a "hit" lies BEFORE the triangle's leaf box on the
//            ray, i.e. between the origin and the mesh: no farther from the triangle than sqrt(3) (|origin| + 2 |mesh|)
//            (largest coordinates).  Rays for which that reach is not below half the smallest cancellation distance
//            of the mesh keep the reference tree (hitMesh).
// Degenerate triangles (NaN normal: never accepted) are left out.
struct MeshSlack { float nearPad, cancelDist; };
inline MeshSlack ComputeMeshSlack(const DTri *tris, uint32_t n, const float *vertsOfElement /* 9 floats per element */)
{
  const double eps = 5.9604644775390625e-8;   // 2^-24
  double pad = 0, cancel = 1e300;
  for (uint32_t e = 0; e < n; ++e) {
    if (!(tris[e].N[0] == tris[e].N[0])) continue;
    const float *v = vertsOfElement + 9 * (size_t) e;
    const uint32_t axis = tris[e].axis;
    const int iu = axis == 0 ? 1 : 0, iv = axis == 2 ? 1 : 2;
    double L = 0, e2 = 1e300;
    for (int a = 0; a < 3; ++a) {
      const float *p = v + 3 * a, *q = v + 3 * ((a + 1) % 3);
      const double dx = (double) q[0] - p[0], dy = (double) q[1] - p[1], dz = (double) q[2] - p[2];
      L = std::max(L, std::sqrt(dx * dx + dy * dy + dz * dz));
      const double du = (double) q[iu] - p[iu], dv = (double) q[iv] - p[iv];
      e2 = std::min(e2, std::sqrt(du * du + dv * dv));
    }
    if (!(e2 > 0) || !(L > 0)) { cancel = 0; continue; }
    pad = std::max(pad, 48.0 * eps * L * L / e2);
    cancel = std::min(cancel, 0.5 * (e2 / (8.0 * eps) - 2.0 * L));
  }
  MeshSlack s;
  s.nearPad = (float) std::min(pad * 1.0000001 + 1e-30, 1e30);
  s.cancelDist = (float) std::min(cancel, 1e30);
  return s;
}

}  // namespace qa
