// qa_fastbvh.h — the HIP library's own search tree over a mesh (host code, used by qa_scene_upload).
//
// The reference searches a mesh with cy::BVH built by splitting at the centre of the longest box
// axis, up to four (sometimes up to eight) triangles per leaf (src/ext/cyBVH.h:318-421): 15 triangle
// tests per ray on the Cornell box.  The closest hit does not depend on the tree, so the kernels
// that do not count traversal steps search a tree built here instead - binary, surface-area
// heuristic, at most two triangles per leaf - and validate the answer against the reference's
// rules afterwards (qa_kernel.h hitMesh).  Same node format and numbering as the reference tree
// (DNode, children adjacent on an even slot, root = 1), so one traversal routine walks both.
//
// Boxes are tested non-strictly (flat boxes - a wall's two triangles - can be entered) and widened
// per ray by the distance at which the reference's fp32 inside test can still accept a point that is
// geometrically outside a triangle (qa_kernel.h hitMesh), so every triangle the reference can
// accept inside the mesh bounds is also reached here.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "qa_scene_dev.h"

namespace qa {

struct FastBvh {
  std::vector<DNode> nodes;       // [0] unused, root = 1
  std::vector<uint32_t> order;    // element of this tree -> element of the reference tree
  uint32_t rootData = QA_BVH_LEAF_BIT;
  uint32_t depth = 1;             // nodes on the longest root-to-leaf path
};

// Expected work of one random ray on a tree (surface-area metric): every node is visited with the
// probability that a ray through the root box also pierces the node's box; an inner visit costs one
// unit, a leaf visit one unit per triangle.  Used to decide per mesh which of the two trees the
// non-counting kernels search.
inline double TreeCost(const DNode *nodes, uint32_t rootData, const float *rootBox)
{
  auto half = [](const float *b) { const double x = b[3] - b[0], y = b[4] - b[1], z = b[5] - b[2]; return x * y + y * z + z * x; };
  const double rootArea = half(rootBox);
  if (!(rootArea > 0)) return 0;
  double cost = 0;
  std::vector<std::pair<uint32_t, double>> st;   // (data word, probability)
  st.push_back({rootData, 1.0});
  while (!st.empty()) {
    const auto [data, p] = st.back();
    st.pop_back();
    if (data & QA_BVH_LEAF_BIT) { cost += p * (double) (((data >> QA_BVH_COUNT_SHIFT) & QA_BVH_COUNT_MASK) + 1); continue; }
    cost += p;
    const uint32_t ch = data & QA_BVH_CHILD_MASK;
    for (uint32_t k = 0; k < 2; ++k) st.push_back({nodes[ch + k].data, std::min(1.0, half(nodes[ch + k].box) / rootArea)});
  }
  return cost;
}

class FastBvhBuilder {
 public:
  // bounds: 6 floats (min xyz, max xyz) per reference element; n elements
  FastBvhBuilder(const float *bounds, uint32_t n, unsigned maxPerLeaf = 2) : b_(bounds), n_(n), leafMax_(maxPerLeaf) {}

  void Run(FastBvh &out)
  {
    out_ = &out;
    out.nodes.assign(2, DNode{});
    out.order.resize(n_);
    for (uint32_t i = 0; i < n_; ++i) out.order[i] = i;
    out.depth = 1;
    if (n_ == 0) { out.rootData = QA_BVH_LEAF_BIT; return; }
    cen_.resize(3 * (size_t) n_);
    for (uint32_t i = 0; i < n_; ++i)
      for (int k = 0; k < 3; ++k) cen_[3 * (size_t) i + k] = 0.5f * (b_[6 * (size_t) i + k] + b_[6 * (size_t) i + 3 + k]);
    right_.resize(n_);
    Emit(1, 0, n_, 1);
    out.rootData = out.nodes[1].data;
  }

 private:
  struct Box {
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    void Grow(const float *b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b[k]); hi[k] = std::max(hi[k], b[3 + k]); } }
    float HalfArea() const
    {
      const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
      return x * y + y * z + z * x;
    }
  };
  Box BoundsOf(uint32_t first, uint32_t count) const
  {
    Box r;
    for (uint32_t i = 0; i < count; ++i) r.Grow(b_ + 6 * (size_t) out_->order[first + i]);
    return r;
  }
  void SortAxis(uint32_t first, uint32_t count, int axis)
  {
    uint32_t *e = out_->order.data() + first;
    std::stable_sort(e, e + count, [&](uint32_t a, uint32_t b) { return cen_[3 * (size_t) a + axis] < cen_[3 * (size_t) b + axis]; });
  }
  // Number of elements of [first, first+count) that go to the first child (the range is left sorted
  // along the chosen axis); 0 = make a leaf.
  uint32_t Split(uint32_t first, uint32_t count, const Box &box)
  {
    if (count <= 1) return 0;
    const float leafCost = (count <= leafMax_) ? box.HalfArea() * (float) count : 1e30f;
    float bestCost = leafCost;
    int bestAxis = -1;
    uint32_t bestN = 0;
    for (int axis = 0; axis < 3; ++axis) {
      SortAxis(first, count, axis);
      Box r;
      for (uint32_t i = count; i-- > 1;) { r.Grow(b_ + 6 * (size_t) out_->order[first + i]); right_[i] = r.HalfArea(); }
      Box l;
      for (uint32_t i = 1; i < count; ++i) {
        l.Grow(b_ + 6 * (size_t) out_->order[first + i - 1]);
        // one node visit costs about as much as one triangle test
        const float cost = box.HalfArea() + l.HalfArea() * (float) i + right_[i] * (float) (count - i);
        if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestN = i; }
      }
    }
    if (bestAxis < 0) {
      if (count <= leafMax_) return 0;
      bestAxis = 0;          // degenerate input (all boxes alike): halve
      bestN = count / 2;
    }
    if (bestAxis != 2) SortAxis(first, count, bestAxis);
    return bestN;
  }
  void Store(uint32_t id, const Box &tight, uint32_t data)
  {
    DNode &n = out_->nodes[id];
    for (int k = 0; k < 3; ++k) {
      // the ray-dependent widening is applied by the traversal (hitMesh); this only absorbs the rounding of
      // the slab arithmetic itself at the box faces
      const float pad = 4e-7f * std::max(std::fabs(tight.lo[k]), std::fabs(tight.hi[k])) + 1e-30f;
      n.box[k] = tight.lo[k] - pad;
      n.box[3 + k] = tight.hi[k] + pad;
    }
    n.data = data;
    n.pad = 0;
  }
  void Emit(uint32_t id, uint32_t first, uint32_t count, uint32_t level)
  {
    if (level > out_->depth) out_->depth = level;
    const Box box = BoundsOf(first, count);
    uint32_t nFirst;
    if (level >= 40 && count > leafMax_) {
      // a pathological input drove the heuristic into a very deep tree: halve along the longest axis
      // from here on, so that the traversal stack (one entry per level, in LDS) stays bounded
      int axis = 0;
      for (int k = 1; k < 3; ++k) if (box.hi[k] - box.lo[k] > box.hi[axis] - box.lo[axis]) axis = k;
      SortAxis(first, count, axis);
      nFirst = count / 2;
    } else nFirst = Split(first, count, box);
    if (nFirst == 0) {
      Store(id, box, (first & QA_BVH_OFFSET_MASK) | ((count - 1) << QA_BVH_COUNT_SHIFT) | QA_BVH_LEAF_BIT);
      return;
    }
    const uint32_t child = (uint32_t) out_->nodes.size();   // always even: slots are handed out in pairs from 2
    out_->nodes.resize(out_->nodes.size() + 2);
    Store(id, box, child & QA_BVH_CHILD_MASK);
    Emit(child, first, nFirst, level + 1);
    Emit(child + 1, first + nFirst, count - nFirst, level + 1);
  }

  const float *b_;
  uint32_t n_;
  unsigned leafMax_;
  FastBvh *out_ = nullptr;
  std::vector<float> cen_, right_;
};

}  // namespace qa
