// qa_capi.hip — extern "C" surface of libqaray_hip.so (include/qaray_hip.h): context, scene
// upload (blob -> device tables), launches of the integrator kernel, counters and timing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "qa_kernel_cs.h"
#include "qa_ctx.h"
#include "qa_fastbvh.h"
#include "qa_widebvh.h"

// core/sampler.cpp:31-40, evaluated on the host in the reference's fp32 order
static float HaltonF(int index, int base)
{
  float r = 0;
  float f = 1.0f / (float) base;
  for (int i = index; i > 0; i /= base) {
    r += f * (i % base);
    f /= (float) base;
  }
  return r;
}

static int EnsureHalton(qa_ctx *c, int count)
{
  if (count <= c->haltonCount) return QA_OK;
  int n = 64;
  while (n < count) n *= 2;
  std::vector<float> t(2 * (size_t) n);
  for (int s = 0; s < n; ++s) { t[2 * s] = HaltonF(s, 11); t[2 * s + 1] = HaltonF(s, 13); }
  if (c->dHalton) (void) hipFree(c->dHalton);
  c->dHalton = nullptr;
  HIP_TRY(hipMalloc((void **) &c->dHalton, t.size() * sizeof(float)));
  HIP_TRY(hipMemcpy(c->dHalton, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
  c->haltonCount = n;
  return QA_OK;
}

static const size_t kMaxLdsPerBlock = 64 * 1024;      // dynamic LDS a workgroup may ask for without opt-in
static const size_t kResidentLdsBudget = 40 * 1024;   // image + stacks: keeps 4 workgroups per CU (160 KB LDS)

// variants: scene memory (LDS-resident | global) x shading (no lights | lights | + textures | + area
// lights | + both) x stats
template <bool RES, bool STATS>
static KernelFn PickShading(bool lights, bool tex, bool area)
{
  if (area) return tex ? (KernelFn) qa_integrate<RES, true, true, true, STATS> : (KernelFn) qa_integrate<RES, true, false, true, STATS>;
  if (tex) return (KernelFn) qa_integrate<RES, true, true, false, STATS>;
  if (lights) return (KernelFn) qa_integrate<RES, true, false, false, STATS>;
  return (KernelFn) qa_integrate<RES, false, false, false, STATS>;
}
static KernelFn PickKernel(bool resident, bool lights, bool tex, bool area, bool stats)
{
  if (resident) return stats ? PickShading<true, true>(lights, tex, area) : PickShading<true, false>(lights, tex, area);
  return stats ? PickShading<false, true>(lights, tex, area) : PickShading<false, false>(lights, tex, area);
}

namespace qa {
// qa_debug_scrub_scratch: every lane fills its private segment (2 KB here, more than any kernel of this library uses) with one
// pattern and lingers, so that all wave slots of the chip are taken at once.  A frame that depends on the pattern reads scratch it
// never wrote (the compiler's spill-before-mask-restore hazard).
__global__ __launch_bounds__(256, 8) void qa_scrub_scratch(uint32_t pattern, uint32_t *never)
{
  volatile uint32_t a[512];
  for (int i = 0; i < 512; ++i) a[i] = pattern;
  for (int i = 0; i < 300; ++i) __builtin_amdgcn_s_sleep(127);
  if (a[threadIdx.x] == 0x12345u && pattern != 0x12345u) never[0] = 1;
}
}  // namespace qa

static const char *kStagedName = "staged: wf_logic + wf_cull + wf_trace + wf_redo";
static std::string MegaName(const qa_ctx *c, bool cs)
{
  char name[160];
  if (cs) snprintf(name, sizeof(name), "qa_integrate_cs<LIGHTS=%d,TEX=%d,CULL=%d%s>", (int) (c->ds.num_lights > 0), (int) c->textured, (int) c->csCullVariant, c->area ? ",AREA=1" : "");
  else snprintf(name, sizeof(name), "qa_integrate<RES=%d,LIGHTS=%d,TEX=%d,AREA=%d>", (int) c->resident, (int) (c->ds.num_lights > 0), (int) c->textured, (int) c->area);
  return name;
}
// The integrator the next plain frame is planned to run on.  What a frame really ran on (photon-map variants, counting
// kernels, frames the staged integrator refused) is recorded at launch: qa_get_kernel_name returns that once a frame has run.
static void SetKernelName(qa_ctx *c)
{
  const WfHost &w = c->wf;
  if (w.eligible && w.mode == QA_PIPE_STAGED) {
    char buf[64];
    snprintf(buf, sizeof(buf), " (%d tile group%s)", w.numGroups, w.numGroups == 1 ? "" : "s");
    c->kernelName = std::string(kStagedName) + buf;
  } else c->kernelName = MegaName(c, c->kernelCs != nullptr);
  c->launchedName.clear();
}

// Choose the kernel variant for the uploaded scene and size the persistent grid to what is
// resident at once (VGPR / LDS-limited workgroups per CU x CUs).
static int SelectKernel(qa_ctx *c)
{
  const bool lights = c->ds.num_lights > 0;
  c->kernel = PickKernel(c->resident, lights, c->textured, c->area, false);
  c->kernelStats = PickKernel(c->resident, lights, c->textured, c->area, true);
  int resident = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, (const void *) c->kernel, QA_BLOCK, c->ldsBytes) != hipSuccess || resident < 1)
    resident = 2;
  c->blocksPerCUAuto = resident > 8 ? 8 : resident;
  // Cooperative mesh walks (qa_kernel_cs.h): scenes in global memory without area lights.  QA_COOP=0: off.
  // (any number of lights: their shadow queries are pooled QA_CS_LIGHT_BATCH = 4 lights at a time; with more than one batch the
  // surface waits in the slab DScene::csSurf between batches, qa_kernel_cs.h; area lights: the AREA variants)
  c->kernelCs = nullptr;
  c->csMany = false;
  {
    const char *e = DevEnv("QA_COOP");
    int shadowLights = 0;
    {
      const qa_flat_header *fh = reinterpret_cast<const qa_flat_header *>(c->hostBlob.data());
      const qa_light *hl = QA_BLOB_PTR(qa_light, c->hostBlob.data(), fh->off_lights);
      for (uint32_t i = 0; i < fh->num_lights; ++i) shadowLights += hl[i].type != QA_LIGHT_AMBIENT;
    }
    if (!c->resident && c->csFits && (shadowLights <= QA_CS_LIGHT_BATCH || c->ds.csSurf) && c->ldsBytesCs <= kMaxLdsPerBlock && c->optCoop && !(e && !strcmp(e, "0"))) {
      // instance culling (qa_kernel_cs.h): the textured variants always (it pays from a handful of nodes on: C3, 9 nodes, + 4 %), the
      // untextured ones on scenes of more than 12 nodes (their register budget: see the kernel's comment)
      c->csCullVariant = c->csCullOk && (c->textured || c->ds.num_inst > 12);
      const bool many = shadowLights > QA_CS_LIGHT_BATCH && !c->area;   // (those variants always test the nodes' bounds)
      c->csMany = many;
      if (many || c->area) c->csCullVariant = c->csCullOk;
      if (c->area)   // every light is evaluated when the path has ended, by the whole wave (qa_kernel_cs.h, AREA)
        c->kernelCs = c->textured ? (KernelFn) qa_integrate_cs<true, true, true, false, true> : (KernelFn) qa_integrate_cs<true, false, true, false, true>;
      else if (many)
        c->kernelCs = c->textured ? (KernelFn) qa_integrate_cs<true, true, true, true> : (KernelFn) qa_integrate_cs<true, false, true, true>;
      else if (c->csCullVariant)
        c->kernelCs = lights ? (c->textured ? (KernelFn) qa_integrate_cs<true, true, true, false> : (KernelFn) qa_integrate_cs<true, false, true, false>)
                             : (c->textured ? (KernelFn) qa_integrate_cs<false, true, true, false> : (KernelFn) qa_integrate_cs<false, false, true, false>);
      else
        c->kernelCs = lights ? (c->textured ? (KernelFn) qa_integrate_cs<true, true, false, false> : (KernelFn) qa_integrate_cs<true, false, false, false>)
                             : (c->textured ? (KernelFn) qa_integrate_cs<false, true, false, false> : (KernelFn) qa_integrate_cs<false, false, false, false>);
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *) c->kernelCs, QA_BLOCK, c->ldsBytesCs) != hipSuccess || n < 1) n = 2;
      c->blocksPerCUCs = n > 8 ? 8 : n;
    }
  }
  SetKernelName(c);
  if (c->optVerbose || DevEnv("QA_FAST_VERBOSE"))
    fprintf(stderr, "kernel %s: dynamic LDS per workgroup: megakernel %zu B (stack depth %u), cooperative %zu B (%u pool items, %u ray slots per wave); workgroups per CU: megakernel %d, cooperative %d\n",
            c->kernelName.c_str(), c->ldsBytes, c->stackDepth, c->ldsBytesCs, c->ds.csItems, c->ds.csSlots, c->blocksPerCUAuto, c->kernelCs ? c->blocksPerCUCs : 0);
  return QA_OK;
}

// Validate the blob and build the device tables.
static int PrepareScene(qa_ctx *c)
{
  const unsigned char *blob = c->hostBlob.data();
  const size_t nbytes = c->hostBlob.size();
  if (nbytes < sizeof(qa_flat_header)) return Fail(QA_EINVAL, "blob smaller than its header");
  const qa_flat_header *h = reinterpret_cast<const qa_flat_header *>(blob);
  if (h->magic != QA_FLAT_MAGIC || h->version != QA_FLAT_VERSION) return Fail(QA_EINVAL, "not a qaray flat scene (magic/version)");
  if (h->total_bytes != nbytes) return Fail(QA_EINVAL, "blob size does not match its header");
  auto inside = [&](uint64_t off, uint64_t bytes) { return off <= nbytes && bytes <= nbytes - off; };
  if (!inside(h->off_instances, (uint64_t) h->num_instances * sizeof(qa_instance)) ||
      !inside(h->off_meshes, (uint64_t) h->num_meshes * sizeof(qa_mesh)) ||
      !inside(h->off_mtlsets, (uint64_t) h->num_mtlsets * sizeof(qa_mtlset)) ||
      !inside(h->off_materials, (uint64_t) h->num_materials * sizeof(qa_material)) ||
      !inside(h->off_lights, (uint64_t) h->num_lights * sizeof(qa_light)) ||
      !inside(h->off_texmaps, (uint64_t) h->num_texmaps * sizeof(qa_texmap)) ||
      !inside(h->off_textures, (uint64_t) h->num_textures * sizeof(qa_texture)))
    return Fail(QA_EINVAL, "table outside the blob");
  if (h->num_instances == 0 || h->width == 0 || h->height == 0) return Fail(QA_EINVAL, "empty scene");

  const qa_instance *inst = QA_BLOB_PTR(qa_instance, blob, h->off_instances);
  const qa_mesh *mesh = QA_BLOB_PTR(qa_mesh, blob, h->off_meshes);
  const qa_mtlset *mtlset = QA_BLOB_PTR(qa_mtlset, blob, h->off_mtlsets);
  const qa_light *light = QA_BLOB_PTR(qa_light, blob, h->off_lights);
  for (uint32_t k = 0; k < h->num_instances; ++k) {
    const qa_instance &in = inst[k];
    if (in.depth > QA_MAX_NODE_DEPTH) return Fail(QA_EUNSUPPORTED, "node nesting deeper than QA_MAX_NODE_DEPTH");
    if (in.parent >= (int) k || (k > 0 && in.parent < 0)) return Fail(QA_EINVAL, "instances are not in pre-order");
    if (in.obj_type == QA_OBJ_MESH && (in.mesh < 0 || in.mesh >= (int) h->num_meshes)) return Fail(QA_EINVAL, "bad mesh index");
    if (in.mtlset >= (int) h->num_mtlsets) return Fail(QA_EINVAL, "bad material index");
  }
  for (uint32_t i = 0; i < h->num_mtlsets; ++i)
    if (mtlset[i].first < 0 || mtlset[i].count < 0 || (uint32_t) (mtlset[i].first + mtlset[i].count) > h->num_materials)
      return Fail(QA_EINVAL, "bad material range");
  bool area = false;
  for (uint32_t i = 0; i < h->num_lights; ++i)
    if ((light[i].type == QA_LIGHT_POINT || light[i].type == QA_LIGHT_SPOT) && light[i].size > 0.01f) area = true;
  const qa_texmap *texmaps = QA_BLOB_PTR(qa_texmap, blob, h->off_texmaps);
  const qa_texture *textures = QA_BLOB_PTR(qa_texture, blob, h->off_textures);
  for (uint32_t i = 0; i < h->num_texmaps; ++i)
    if (texmaps[i].texture < -1 || texmaps[i].texture >= (int) h->num_textures) return Fail(QA_EINVAL, "bad texture index");
  if (h->background.texmap < -1 || h->background.texmap >= (int) h->num_texmaps || h->environment.texmap < -1 ||
      h->environment.texmap >= (int) h->num_texmaps)
    return Fail(QA_EINVAL, "bad background / environment texmap index");
  // the tables are read in place (4- and 8-byte fields): offsets must be 8-byte aligned
  for (uint64_t off : {h->off_instances, h->off_meshes, h->off_mtlsets, h->off_materials, h->off_lights, h->off_texmaps, h->off_textures})
    if (off % 8) return Fail(QA_EINVAL, "table offset is not 8-byte aligned");
  for (uint32_t i = 0; i < h->num_textures; ++i)
    if (textures[i].type == QA_TEX_FILE &&
        (textures[i].width < 0 || textures[i].height < 0 ||
         !inside(textures[i].off_texels, (uint64_t) textures[i].width * (uint64_t) textures[i].height * 3)))
      return Fail(QA_EINVAL, "texel array outside the blob");
  const bool textured = h->num_texmaps > 0;

  // ---- derived per-mesh arrays --------------------------------------------------------------
  std::vector<DMesh> dmeshes(h->num_meshes);
  std::vector<std::vector<DNode>> allNodes(h->num_meshes);
  std::vector<std::vector<DTri>> allTris(h->num_meshes);
  std::vector<std::vector<DTriShade>> allShade(h->num_meshes);
  std::vector<std::vector<DNode>> allFNodes(h->num_meshes);     // the library's own trees (qa_fastbvh.h)
  bool csFits = true;   // qa_kernel_cs.h: pool items hold 22 bits of node index / triangle offset
  std::vector<std::vector<DTri>> allFTris(h->num_meshes);
  std::vector<std::vector<uint32_t>> allFMap(h->num_meshes);
  std::vector<WideBvh> allWide(h->num_meshes);                  // 4-wide trees over the triangles (qa_widebvh.h)
  std::vector<std::vector<DTri>> allWTris(h->num_meshes);       // triangle records in their leaf order
  std::vector<MeshSlack> meshSlack(h->num_meshes, MeshSlack{0.f, 0.f});
  std::vector<std::pair<double, double>> fastCost(h->num_meshes, {0.0, 0.0});   // expected ray cost: reference tree, own tree
  std::vector<float> meshInvH(h->num_meshes, 0.f), meshAbsMax(h->num_meshes, 0.f);
  std::vector<std::vector<float>> meshNormals(h->num_meshes);   // x, y, z, 0 per distinct face normal; empty = too many
  uint32_t stackNeedMax = 1;
  uint64_t totalFaces = 0;
  for (uint32_t mi = 0; mi < h->num_meshes; ++mi) totalFaces += mesh[mi].num_faces;
  for (uint32_t mi = 0; mi < h->num_meshes; ++mi) {
    const qa_mesh &m = mesh[mi];
    if (!inside(m.off_bvh_nodes, (uint64_t) m.num_bvh_nodes * sizeof(qa_bvh_node)) ||
        !inside(m.off_elements, (uint64_t) m.num_faces * 4) || !inside(m.off_faces, (uint64_t) m.num_faces * sizeof(qa_face)) ||
        !inside(m.off_vertices, (uint64_t) m.num_vertices * 12) || !inside(m.off_normals, (uint64_t) m.num_normals * 12) ||
        !inside(m.off_texcoords, (uint64_t) m.num_texcoords * 8))
      return Fail(QA_EINVAL, "mesh array outside the blob");
    if (m.off_bvh_nodes % 4 || m.off_elements % 4 || m.off_faces % 4 || m.off_vertices % 4 || m.off_normals % 4 || m.off_texcoords % 4)
      return Fail(QA_EINVAL, "mesh array offset is not 4-byte aligned");
    const qa_bvh_node *nodes = QA_BLOB_PTR(qa_bvh_node, blob, m.off_bvh_nodes);
    const uint32_t *elements = QA_BLOB_PTR(uint32_t, blob, m.off_elements);
    const qa_face *faces = QA_BLOB_PTR(qa_face, blob, m.off_faces);
    const float *V = QA_BLOB_PTR(float, blob, m.off_vertices);
    const float *VN = QA_BLOB_PTR(float, blob, m.off_normals);
    std::vector<DNode> &dn = allNodes[mi];
    dn.resize(m.num_bvh_nodes + (m.num_bvh_nodes & 1));  // even count: sibling pairs are 64-byte units
    memset(dn.data(), 0, dn.size() * sizeof(DNode));
    for (uint32_t i = 0; i < m.num_bvh_nodes; ++i) {
      memcpy(dn[i].box, nodes[i].box, sizeof(dn[i].box));
      dn[i].data = nodes[i].data;
      if (i >= 1 && m.num_faces > 0) {
        if (!(nodes[i].data & QA_BVH_LEAF_BIT)) {
          const uint32_t ch = nodes[i].data & QA_BVH_CHILD_MASK;
          if (ch + 1 >= m.num_bvh_nodes || ch <= i || (ch & 1)) return Fail(QA_EINVAL, "BVH child index out of range");
        } else {
          const uint32_t cnt = ((nodes[i].data >> QA_BVH_COUNT_SHIFT) & QA_BVH_COUNT_MASK) + 1;
          if ((nodes[i].data & QA_BVH_OFFSET_MASK) + cnt > m.num_faces) return Fail(QA_EINVAL, "BVH leaf range out of range");
          if (nodes[i].data == QA_DONE) return Fail(QA_EUNSUPPORTED, "leaf word collides with the traversal sentinel");
        }
      }
    }
    // deepest stack the traversal can need = BVH depth (one pending sibling per level); children
    // always have larger indices than their parent, so a forward sweep computes node depths
    uint32_t stackNeed = 1;
    if (m.num_faces > 0 && m.num_bvh_nodes > 1) {
      std::vector<uint32_t> level(m.num_bvh_nodes, 0);
      level[1] = 1;
      for (uint32_t i = 1; i < m.num_bvh_nodes; ++i) {
        if (level[i] == 0) continue;
        if (!(nodes[i].data & QA_BVH_LEAF_BIT)) {
          const uint32_t ch = nodes[i].data & QA_BVH_CHILD_MASK;
          level[ch] = level[ch + 1] = level[i] + 1;
          if (level[i] + 1 > stackNeed) stackNeed = level[i] + 1;
        }
      }
    }
    if (stackNeed > stackNeedMax) stackNeedMax = stackNeed;
    const uint32_t stackNeedRef = stackNeed;   // depth of the reference tree alone
    std::vector<DTri> &dt = allTris[mi];
    std::vector<DTriShade> &dsh = allShade[mi];
    dt.resize(m.num_faces);
    dsh.resize(m.num_faces);
    for (uint32_t e = 0; e < m.num_faces; ++e) {
      const uint32_t fid = elements[e];
      if (fid >= m.num_faces) return Fail(QA_EINVAL, "BVH element out of range");
      const qa_face &f = faces[fid];
      for (int k = 0; k < 3; ++k) {
        if (f.v[k] < 0 || (uint32_t) f.v[k] >= m.num_vertices) return Fail(QA_EINVAL, "vertex index out of range");
        if (f.vn[k] < 0 || (uint32_t) f.vn[k] >= m.num_normals) return Fail(QA_EINVAL, "normal index out of range");
      }
      const f3 A = ld3(V + 3 * f.v[0]), B = ld3(V + 3 * f.v[1]), C = ld3(V + 3 * f.v[2]);
      // src/objects/objects.cpp:220-246
      const f3 N = normalize(cross(B - A, C - A));
      uint32_t axis;
      const float ax = qabs(N.x), ay = qabs(N.y), az = qabs(N.z);
      if (ax > ay && ax > az) axis = 0;
      else if (ay > az) axis = 1;
      else axis = 2;
      auto U = [&](f3 p) { return axis == 0 ? p.y : p.x; };
      auto W = [&](f3 p) { return axis == 2 ? p.y : p.z; };
      DTri &t = dt[e];
      t.N[0] = N.x; t.N[1] = N.y; t.N[2] = N.z;
      t.A[0] = A.x; t.A[1] = A.y; t.A[2] = A.z;
      t.bu = U(B); t.bv = W(B); t.cu = U(C); t.cv = W(C);
      // TriangleArea(axis, A, B, C) (objects.cpp:30-41)
      const float area = (t.bu - U(A)) * (t.cv - W(A)) - (t.cu - U(A)) * (t.bv - W(A));
      t.s = 1.f / area;
      t.axis = axis;
      DTriShade &s = dsh[e];
      memcpy(s.n0, VN + 3 * f.vn[0], 12);
      memcpy(s.n1, VN + 3 * f.vn[1], 12);
      memcpy(s.n2, VN + 3 * f.vn[2], 12);
      s.mtl = f.mtl;
      s.face = fid;
      s.pad = 0;
    }
    // ---- what the non-counting kernels need to search their own tree and to validate the answer
    // against the reference's (qa_kernel.h hitMesh): element -> leaf links of the reference tree, and the
    // SAH tree over the same triangles
    for (uint32_t i = 1; i < m.num_bvh_nodes; ++i) {
      if (nodes[i].data & QA_BVH_LEAF_BIT) {
        const uint32_t cnt = ((nodes[i].data >> QA_BVH_COUNT_SHIFT) & QA_BVH_COUNT_MASK) + 1, off = nodes[i].data & QA_BVH_OFFSET_MASK;
        for (uint32_t q = 0; q < cnt; ++q) dsh[off + q].pad = i;
      }
    }
    {
      std::vector<float> bounds(6 * (size_t) m.num_faces);
      for (uint32_t e = 0; e < m.num_faces; ++e) {
        const qa_face &f = faces[elements[e]];
        float *bb = &bounds[6 * (size_t) e];
        for (int k = 0; k < 3; ++k) { bb[k] = 1e30f; bb[3 + k] = -1e30f; }
        for (int v = 0; v < 3; ++v)
          for (int k = 0; k < 3; ++k) {
            const float x = V[3 * (size_t) f.v[v] + k];
            if (x < bb[k]) bb[k] = x;
            if (x > bb[3 + k]) bb[3 + k] = x;
          }
      }
      // smallest altitude over all triangles: 2 * area / longest edge (degenerate triangles never pass the
      // reference's test - their normal is NaN - and are left out)
      double hMin = 1e300;
      for (uint32_t e = 0; e < m.num_faces; ++e) {
        const qa_face &f = faces[elements[e]];
        const float *A = V + 3 * (size_t) f.v[0], *B = V + 3 * (size_t) f.v[1], *C = V + 3 * (size_t) f.v[2];
        const double ab[3] = {(double) B[0] - A[0], (double) B[1] - A[1], (double) B[2] - A[2]};
        const double ac[3] = {(double) C[0] - A[0], (double) C[1] - A[1], (double) C[2] - A[2]};
        const double bc[3] = {(double) C[0] - B[0], (double) C[1] - B[1], (double) C[2] - B[2]};
        const double cr[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
        const double area2 = std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
        const double L = std::sqrt(std::max({ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2], ac[0] * ac[0] + ac[1] * ac[1] + ac[2] * ac[2],
                                             bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]}));
        if (area2 > 0 && L > 0) hMin = std::min(hMin, area2 / L);
      }
      float absMax = 0;
      for (int k = 0; k < 3; ++k) absMax = std::max(absMax, std::max(std::fabs(m.bmin[k]), std::fabs(m.bmax[k])));
      // distinct face normals up to sign, merged within 1e-5 (the DTri records hold the reference's own normalize(cross()))
      {
        std::vector<float> &nl = meshNormals[mi];
        bool overflow = false;
        for (uint32_t e = 0; e < m.num_faces && !overflow; ++e) {
          const float *N = dt[e].N;
          if (!(N[0] == N[0])) continue;   // degenerate triangle: NaN normal, never accepted
          bool seen = false;
          for (size_t q = 0; q + 3 < nl.size() + 1 && !seen; q += 4)
          {
            // same direction up to sign within 1e-5 (the kernel's parallelism threshold allows for it)
            const float dp = std::fabs(nl[q] * N[0] + nl[q + 1] * N[1] + nl[q + 2] * N[2]);
            const float cx = nl[q + 1] * N[2] - nl[q + 2] * N[1], cy = nl[q + 2] * N[0] - nl[q] * N[2], cz = nl[q] * N[1] - nl[q + 1] * N[0];
            seen = dp > 0.5f && std::sqrt(cx * cx + cy * cy + cz * cz) < 1e-5f;
          }
          if (!seen) {
            if (nl.size() >= 4 * 24) overflow = true;
            else { nl.push_back(N[0]); nl.push_back(N[1]); nl.push_back(N[2]); nl.push_back(0.f); }
          }
        }
        if (overflow) nl.clear();
      }
      meshInvH[mi] = hMin < 1e300 ? (float) (1.0 / hMin) : 0.f;
      meshAbsMax[mi] = absMax;
      FastBvh fb;
      // only LDS-resident scenes search their own trees, and residency needs the whole image within
      // 40 KB (~140 B per triangle before the own tree): skip the build where that is out of reach
      if (totalFaces > 512) {
        fb.nodes.assign(2, DNode{});
        fb.order.clear();
        allFNodes[mi] = fb.nodes;
        fastCost[mi] = {0.0, 0.0};
        // global-memory scene: the 4-wide tree over the reference leaves, and the inside test's fp32 slack
        if (m.num_faces > 0 && m.num_bvh_nodes > 1 && !(DevEnv("QA_WIDE") && atoi(DevEnv("QA_WIDE")) == 0)) {
          try {
            std::vector<float> ev(9 * (size_t) m.num_faces), tb(6 * (size_t) m.num_faces);
            std::vector<unsigned char> skip(m.num_faces, 0);
            for (uint32_t e = 0; e < m.num_faces; ++e) {
              const qa_face &f = faces[elements[e]];
              for (int v = 0; v < 3; ++v) memcpy(&ev[9 * (size_t) e + 3 * v], V + 3 * (size_t) f.v[v], 12);
              const float *p = &ev[9 * (size_t) e];
              for (int k = 0; k < 3; ++k) {
                tb[6 * (size_t) e + k] = std::min(p[k], std::min(p[3 + k], p[6 + k]));
                tb[6 * (size_t) e + 3 + k] = std::max(p[k], std::max(p[3 + k], p[6 + k]));
              }
              skip[e] = !(dt[e].N[0] == dt[e].N[0]);   // degenerate: NaN normal, the inside test never accepts it
            }
            // triangles per leaf: 3 (same-box A/B of 1 / 2 / 3 / 4 / 6 / 8: C3 288 / 363 / 373 / 382 / 389 / 371, C5 391 / 612 / 614 / 594 /
            // 565 / 539 Msamples/s on the megakernel; the staged integrator is flat between 2 and 4)
            const unsigned wideLeaf = DevEnv("QA_WIDE_LEAF") ? (unsigned) atoi(DevEnv("QA_WIDE_LEAF")) : 3u;
            WideBvhBuilder(tb.data(), skip.data(), m.num_faces, wideLeaf).Run(allWide[mi]);
            // the triangle records once more in the wide tree's leaf order; the element id rides above the 2-bit axis
            allWTris[mi].resize(allWide[mi].order.size());
            for (size_t i = 0; i < allWide[mi].order.size(); ++i) {
              allWTris[mi][i] = dt[allWide[mi].order[i]];
              allWTris[mi][i].axis |= allWide[mi].order[i] << 2;
            }
            meshSlack[mi] = ComputeMeshSlack(dt.data(), m.num_faces, ev.data());
          } catch (const std::bad_alloc &) { return Fail(QA_ENOMEM, "out of memory"); }
          const uint32_t need = 3 * allWide[mi].depth + 2;
          if (need > stackNeed) stackNeed = need;
          if (stackNeed > stackNeedMax) stackNeedMax = stackNeed;
          if ((c->optVerbose || DevEnv("QA_FAST_VERBOSE")))
            fprintf(stderr, "mesh %u: %u triangles, reference tree %u nodes depth %u; wide tree %zu nodes depth %u; inside-test slack %g, cancel distance %g, |coord| <= %g\n",
                    mi, m.num_faces, m.num_bvh_nodes, stackNeed, allWide[mi].nodes.size(), allWide[mi].depth, (double) meshSlack[mi].nearPad,
                    (double) meshSlack[mi].cancelDist, (double) absMax);
        }
      } else {
      const unsigned leafMax = DevEnv("QA_FAST_LEAF") ? (unsigned) atoi(DevEnv("QA_FAST_LEAF")) : 2u;
      try { FastBvhBuilder(bounds.data(), m.num_faces, leafMax).Run(fb); } catch (const std::bad_alloc &) { return Fail(QA_ENOMEM, "out of memory"); }
      if (fb.nodes.size() & 1) fb.nodes.push_back(DNode{});
      {
        float rootBox[6];
        memcpy(rootBox, m.bmin, 12);
        memcpy(rootBox + 3, m.bmax, 12);
        const double costRef = m.num_bvh_nodes > 1 ? TreeCost(dn.data(), nodes[1].data, rootBox) : 0;
        const double costFast = TreeCost(fb.nodes.data(), fb.rootData, rootBox);
        fastCost[mi] = {costRef, costFast};
        if ((c->optVerbose || DevEnv("QA_FAST_VERBOSE")))
          fprintf(stderr, "mesh %u: %u triangles, expected ray cost reference tree %.2f, own tree %.2f (depth %u), smallest altitude %g, |coord| <= %g, %zu distinct normals\n",
                  mi, m.num_faces, costRef, costFast, fb.depth, hMin, (double) absMax, meshNormals[mi].size() / 4);
      }
      allFNodes[mi] = fb.nodes;
      allFMap[mi] = fb.order;
      allFTris[mi].resize(m.num_faces);
      // DTri::axis of the own tree's copies = axis | element << 2 | reference-tree leaf << 17: the walk hands back the
      // element and its leaf (refReaches) with the accepted record itself instead of through two more dependent reads
      // (fmap, DTriShade::pad) after it.  Meshes beyond 15 bits of either keep the reference tree (useFast below).
      for (uint32_t i = 0; i < m.num_faces; ++i) {
        const uint32_t e = fb.order[i];
        allFTris[mi][i] = dt[e];
        allFTris[mi][i].axis = (dt[e].axis & 3u) | ((e & 0x7FFFu) << 2) | ((dsh[e].pad & 0x7FFFu) << 17);
      }
      if (fb.depth > stackNeed) stackNeed = fb.depth;
      if (stackNeed > stackNeedMax) stackNeedMax = stackNeed;
      }
    }
    // texture vertices per triangle (element order); a mesh must have them on every face or none
    const float *VT = QA_BLOB_PTR(float, blob, m.off_texcoords);
    std::vector<float> vts;
    uint32_t withVT = 0;
    for (uint32_t e = 0; e < m.num_faces; ++e) {
      const qa_face &f = faces[elements[e]];
      if (f.vt[0] >= 0 && f.vt[1] >= 0 && f.vt[2] >= 0) {
        for (int k = 0; k < 3; ++k) if ((uint32_t) f.vt[k] >= m.num_texcoords) return Fail(QA_EINVAL, "texcoord index out of range");
        ++withVT;
      }
    }
    if (withVT != 0 && withVT != m.num_faces)
      return Fail(QA_EUNSUPPORTED, "mesh with texture vertices on only some faces");
    if (withVT && textured) {
      vts.resize(6 * (size_t) m.num_faces);
      for (uint32_t e = 0; e < m.num_faces; ++e) {
        const qa_face &f = faces[elements[e]];
        for (int k = 0; k < 3; ++k) { vts[6 * e + 2 * k] = VT[2 * f.vt[k]]; vts[6 * e + 2 * k + 1] = VT[2 * f.vt[k] + 1]; }
      }
    }
    DMesh &dm = dmeshes[mi];
    memset(&dm, 0, sizeof(dm));
    dm.hasVT = (withVT && textured) ? 1 : 0;
    {
      int rcv;
      if ((rcv = DeviceCopy(c, vts, &dm.vt)) != QA_OK) return rcv;
    }
    memcpy(dm.bmin, m.bmin, 12);
    memcpy(dm.bmax, m.bmax, 12);
    dm.num_faces = m.num_faces;
    dm.num_nodes = m.num_bvh_nodes;
    dm.rootData = m.num_bvh_nodes > 1 ? nodes[1].data : QA_DONE;
    dm.frootData = (m.num_faces && allFNodes[mi].size() > 1) ? allFNodes[mi][1].data : QA_DONE;
    dm.useFast = (totalFaces <= 512 && m.num_bvh_nodes < 0x8000u && m.num_faces <= (DevEnv("QA_FAST_MAXFACES") ? (uint32_t) atoi(DevEnv("QA_FAST_MAXFACES")) : 0xFFFFFFFFu)) ? 1u : 0u;
    dm.invH = meshInvH[mi];
    dm.absMax = meshAbsMax[mi];
    {
      // needle-like triangles would widen the own tree's boxes (200 eps P^2 / h, see hitMesh) to a sizeable
      // part of the mesh: such a mesh keeps the reference tree
      const double P = 2.0 * meshAbsMax[mi] + 1e-30, diag = std::sqrt((double) (m.bmax[0] - m.bmin[0]) * (m.bmax[0] - m.bmin[0]) +
                                                                       (double) (m.bmax[1] - m.bmin[1]) * (m.bmax[1] - m.bmin[1]) +
                                                                       (double) (m.bmax[2] - m.bmin[2]) * (m.bmax[2] - m.bmin[2]));
      if (!(1.2e-5 * meshInvH[mi] * P * P < 0.01 * diag)) dm.useFast = 0;
    }
    // refReaches tests a leaf's box only: valid when every inner box contains its children's boxes
    // (true for cy::BVH, whose inner boxes are unions); a blob that breaks this keeps the reference tree
    for (uint32_t i = 1; i < m.num_bvh_nodes && dm.useFast; ++i) {
      if (nodes[i].data & QA_BVH_LEAF_BIT) continue;
      const uint32_t ch = nodes[i].data & QA_BVH_CHILD_MASK;
      for (uint32_t q = ch; q < ch + 2; ++q)
        for (int k = 0; k < 3; ++k)
          if (!(nodes[q].box[k] >= nodes[i].box[k] && nodes[q].box[k + 3] <= nodes[i].box[k + 3])) dm.useFast = 0;
    }
    dm.stackNeed = stackNeed;
    dm.wrootWord = allWide[mi].rootWord;
    dm.wideStack = 3 * allWide[mi].depth + 2;
    dm.wnodeCount = (uint32_t) allWide[mi].nodes.size();
    if (m.num_faces > QA_CS_INDEX_MASK) csFits = false;          // a key holds instance << 20 | element (qa_kernel_cs.h)
    if (stackNeedRef > QA_CS_EXACT_STACK) csFits = false;        // private stacks of the exact walks
    if (textured && m.num_faces > 0 && !dm.hasVT) csFits = false;   // a hit there keeps the uvw of an earlier, farther hit: history only a sequential walk has
    dm.nearPad = meshSlack[mi].nearPad;
    dm.cancelDist = meshSlack[mi].cancelDist;
    {
      const double diag = std::sqrt((double) (m.bmax[0] - m.bmin[0]) * (m.bmax[0] - m.bmin[0]) + (double) (m.bmax[1] - m.bmin[1]) * (m.bmax[1] - m.bmin[1]) +
                                    (double) (m.bmax[2] - m.bmin[2]) * (m.bmax[2] - m.bmin[2]));
      // needle triangles would widen every box by a sizeable part of the mesh: such a mesh keeps the reference tree
      dm.useWide = (allWide[mi].rootWord != QA_DONE && meshSlack[mi].nearPad < 0.01 * diag) ? 1u : 0u;
    }
    // the order check tests a leaf's box only: valid when every inner box of the reference tree contains its children's
    for (uint32_t i = 1; i < m.num_bvh_nodes && dm.useWide; ++i) {
      if (nodes[i].data & QA_BVH_LEAF_BIT) continue;
      const uint32_t ch = nodes[i].data & QA_BVH_CHILD_MASK;
      for (uint32_t q = ch; q < ch + 2; ++q)
        for (int k = 0; k < 3; ++k)
          if (!(nodes[q].box[k] >= nodes[i].box[k] && nodes[q].box[k + 3] <= nodes[i].box[k + 3])) dm.useWide = 0;
    }
    // QA_SLACK_SCALE is 1 except in the test-only library lib_noslack (qa_scene_dev.h)
    if (QA_SLACK_SCALE != 1.0f) {
      dm.nearPad *= QA_SLACK_SCALE;
      dm.cancelDist = QA_SLACK_SCALE > 0 ? dm.cancelDist / QA_SLACK_SCALE : 1e30f;
    }
    dm.gateIsRoot = (m.num_bvh_nodes > 1 && memcmp(nodes[1].box, m.bmin, 12) == 0 && memcmp(nodes[1].box + 3, m.bmax, 12) == 0) ? 1u : 0u;
    int rc;
    if ((rc = DeviceCopy(c, dn, &dm.nodes)) != QA_OK) return rc;
    if ((rc = DeviceCopy(c, dt, &dm.tris)) != QA_OK) return rc;
    if ((rc = DeviceCopy(c, dsh, &dm.shade)) != QA_OK) return rc;
    if ((rc = DeviceCopy(c, allFNodes[mi], &dm.fnodes)) != QA_OK) return rc;
    if ((rc = DeviceCopy(c, allFTris[mi], &dm.ftris)) != QA_OK) return rc;
    if ((rc = DeviceCopy(c, allFMap[mi], &dm.fmap)) != QA_OK) return rc;
    if ((rc = DeviceCopy(c, allWide[mi].nodes, &dm.wnodes)) != QA_OK) return rc;
    if ((rc = DeviceCopy(c, allWTris[mi], &dm.wtris)) != QA_OK) return rc;
  }

  // ---- material table (plain colours) -----------------------------------------------------------
  const qa_material *mats = QA_BLOB_PTR(qa_material, blob, h->off_materials);
  std::vector<DMaterial> dmat(h->num_materials);
  bool anySpecularLobes = false;
  for (uint32_t i = 0; i < h->num_materials; ++i) {
    const qa_material &m = mats[i];
    DMaterial &d = dmat[i];
    memcpy(d.diffuse, m.diffuse.color, 12);       d.kill = m.kill;
    memcpy(d.specular, m.specular.color, 12);     d.gloss_spec = m.gloss_spec;
    memcpy(d.emission, m.emission.color, 12);     d.ior = m.ior;
    memcpy(d.reflection, m.reflection.color, 12); d.gloss_refl = m.gloss_refl;
    memcpy(d.refraction, m.refraction.color, 12); d.gloss_refr = m.gloss_refr;
    memcpy(d.absorption, m.absorption, 12);
    d.flags = 0;
    for (int k = 0; k < 3; ++k) {
      if (m.reflection.color[k] != 0.f || m.refraction.color[k] != 0.f) anySpecularLobes = true;
      if (m.reflection.color[k] != 0.f || m.refraction.color[k] != 0.f) d.flags |= QA_MTL_SPECULAR_LOBES;
      if (m.specular.color[k] != 0.f) d.flags |= QA_MTL_HAS_SPECULAR;
    }
  }

  // ---- resident image: [nodes | tris | shade] per mesh, then materials, in 16-byte units --------
  std::vector<uint4> image;
  auto append = [&](const void *p, size_t bytes) {
    const uint32_t off = (uint32_t) image.size();
    const size_t n = (bytes + 15) / 16;
    image.resize(image.size() + n, uint4{0, 0, 0, 0});
    if (bytes) memcpy(image.data() + off, p, bytes);
    return off;
  };
  for (uint32_t mi = 0; mi < h->num_meshes; ++mi) {
    while (image.size() % 4) image.push_back(uint4{0, 0, 0, 0});  // node pairs on 64-byte boundaries
    dmeshes[mi].resNodes = append(allNodes[mi].data(), allNodes[mi].size() * sizeof(DNode));
    dmeshes[mi].resTris = append(allTris[mi].data(), allTris[mi].size() * sizeof(DTri));
    dmeshes[mi].resShade = append(allShade[mi].data(), allShade[mi].size() * sizeof(DTriShade));
    while (image.size() % 4) image.push_back(uint4{0, 0, 0, 0});
    dmeshes[mi].resFNodes = append(allFNodes[mi].data(), allFNodes[mi].size() * sizeof(DNode));
    dmeshes[mi].resFTris = append(allFTris[mi].data(), allFTris[mi].size() * sizeof(DTri));
    dmeshes[mi].resFMap = append(allFMap[mi].data(), allFMap[mi].size() * sizeof(uint32_t));
    dmeshes[mi].resNormals = append(meshNormals[mi].data(), meshNormals[mi].size() * sizeof(float));
    dmeshes[mi].numNormals = (uint32_t) (meshNormals[mi].size() / 4);
  }
  const uint32_t resMaterials = append(dmat.data(), dmat.size() * sizeof(DMaterial));

  DScene &ds = c->ds;
  memset(&ds, 0, sizeof(ds));
  ds.blob = c->dBlob;
  ds.inst = QA_BLOB_PTR(qa_instance, c->dBlob, h->off_instances);
  ds.mtlset = QA_BLOB_PTR(qa_mtlset, c->dBlob, h->off_mtlsets);
  ds.light = QA_BLOB_PTR(qa_light, c->dBlob, h->off_lights);
  int rc;
  // ---- qa_integrate_cs: the 4-wide trees of all meshes in one node array and one triangle array (qa_kernel_cs.h) -------------
  {
    std::vector<DWideNode> csNodes;
    std::vector<DTri> csTris;
    std::vector<float> csLeafBox;   // 8 floats per triangle of csTris: box of its leaf in the reference tree, 1.0f = that leaf is the root
    try {
      for (uint32_t mi = 0; mi < h->num_meshes; ++mi) {
        const WideBvh &wb = allWide[mi];
        const uint32_t nodeBase = (uint32_t) csNodes.size(), triBase = (uint32_t) csTris.size();
        auto rebase = [&](uint32_t w) -> uint32_t {
          if (w == QA_DONE) return w;
          if (w & QA_BVH_LEAF_BIT) return (w & ~QA_BVH_OFFSET_MASK) | ((w & QA_BVH_OFFSET_MASK) + triBase);
          return w + nodeBase;
        };
        for (const DWideNode &nd : wb.nodes) {
          DWideNode d = nd;
          for (int q = 0; q < 4; ++q) d.child[q] = rebase(nd.child[q]);
          csNodes.push_back(d);
        }
        const qa_bvh_node *rnodes = QA_BLOB_PTR(qa_bvh_node, blob, mesh[mi].off_bvh_nodes);
        for (size_t i = 0; i < allWTris[mi].size(); ++i) {
          csTris.push_back(allWTris[mi][i]);
          const uint32_t leaf = allShade[mi][wb.order[i]].pad;
          float rec[8] = {0, 0, 0, 0, 0, 0, leaf <= 1 ? 1.0f : 0.0f, 0};
          if (leaf < mesh[mi].num_bvh_nodes) memcpy(rec, rnodes[leaf].box, 24);
          csLeafBox.insert(csLeafBox.end(), rec, rec + 8);
        }
        dmeshes[mi].csRootWord = rebase(wb.rootWord);
      }
    } catch (const std::bad_alloc &) { return Fail(QA_ENOMEM, "out of memory"); }
    if (csNodes.size() > QA_CS_INDEX_MASK || csTris.size() > QA_CS_INDEX_MASK || h->num_instances > 256) csFits = false;
    if (h->width > 0xFFFFu || h->height > 0xFFFFu || h->num_materials > 0xFFFEu) csFits = false;   // pixel and material ride in 16-bit halves of the kernel's state words
    if (csFits && !csTris.empty()) {
      const DWideNode *dn = nullptr;
      const DTri *dtr = nullptr;
      const float *dlb = nullptr;
      if (csNodes.empty()) csNodes.push_back(DWideNode{});
      if ((rc = DeviceCopy(c, csNodes, &dn)) != QA_OK) return rc;
      if ((rc = DeviceCopy(c, csTris, &dtr)) != QA_OK) return rc;
      if ((rc = DeviceCopy(c, csLeafBox, &dlb)) != QA_OK) return rc;
      c->csNodesDev = reinterpret_cast<const uint4 *>(dn);
      c->csTrisDev = reinterpret_cast<const uint4 *>(dtr);
      c->csLeafBoxDev = reinterpret_cast<const uint4 *>(dlb);
    } else csFits = false;
  }
  // one flat record per scene-graph node for qa_integrate_cs's sweeps
  {
    static const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z3[3] = {0, 0, 0};
    if (!(memcmp(inst[0].tm, I9, 36) == 0 && memcmp(inst[0].itm, I9, 36) == 0 && memcmp(inst[0].pos, Z3, 12) == 0)) csFits = false;   // (XML scenes: always the identity)
    std::vector<CsInst> ci(h->num_instances);
    memset(ci.data(), 0, ci.size() * sizeof(CsInst));
    std::vector<CsCull> cull(h->num_instances);
    for (CsCull &cb : cull) { cb.lo[0] = cb.lo[1] = cb.lo[2] = 3e38f; cb.hi[0] = cb.hi[1] = cb.hi[2] = -3e38f; cb.pad0 = cb.pad1 = 0.f; }   // empty: never entered
    double cullS1 = 1, cullS2 = 1, cullK3 = 0, cullK4 = 0;
    bool cullOk = true;
    for (uint32_t k = 1; k < h->num_instances; ++k) {
      const qa_instance &in = inst[k];
      CsInst &r = ci[k];
      r.type = in.obj_type;
      r.depth = in.depth;
      r.parent = in.parent;
      if (in.obj_type == QA_OBJ_NONE) continue;
      if (in.depth < 1 || in.depth > 2) { csFits = false; continue; }
      const qa_instance &a = in.depth == 2 ? inst[in.parent] : in;
      memcpy(r.itmA, a.itm, 36); memcpy(r.posA, a.pos, 12); memcpy(r.tmA, a.tm, 36);
      if (in.depth == 2) { memcpy(r.itmB, in.itm, 36); memcpy(r.posB, in.pos, 12); memcpy(r.tmB, in.tm, 36); }
      double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};   // sphere: the unit ball; plane: the unit square at z = 0
      if (in.obj_type == QA_OBJ_PLANE) lo[2] = hi[2] = 0;
      if (in.obj_type == QA_OBJ_MESH) {
        const DMesh &dm = dmeshes[in.mesh];
        r.mesh = (uint32_t) in.mesh;
        r.useWide = dm.useWide;
        r.csRootWord = dm.csRootWord;
        r.num_faces = dm.num_faces;
        memcpy(r.bmin, dm.bmin, 12); memcpy(r.bmax, dm.bmax, 12);
        r.nearPad = dm.nearPad; r.absMax = dm.absMax; r.cancelDist = dm.cancelDist;
        for (int q = 0; q < 3; ++q) { lo[q] = dm.bmin[q]; hi[q] = dm.bmax[q]; }
      }
      // bounds in root space: the eight corners through tm * p + pos of every level (double), padded below
      double wlo[3] = {1e300, 1e300, 1e300}, whi[3] = {-1e300, -1e300, -1e300};
      for (int corner = 0; corner < 8; ++corner) {
        double pnt[3] = {(corner & 1) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 4) ? hi[2] : lo[2]};
        for (int lvl = in.depth; lvl >= 1; --lvl) {
          const qa_instance &t = (lvl == in.depth) ? in : inst[in.parent];
          double o[3];
          for (int rr = 0; rr < 3; ++rr) o[rr] = (double) t.tm[rr] * pnt[0] + (double) t.tm[3 + rr] * pnt[1] + (double) t.tm[6 + rr] * pnt[2] + (double) t.pos[rr];
          memcpy(pnt, o, sizeof(o));
        }
        for (int q = 0; q < 3; ++q) { wlo[q] = std::min(wlo[q], pnt[q]); whi[q] = std::max(whi[q], pnt[q]); }
      }
      for (int q = 0; q < 3; ++q) { r.wmin[q] = (float) wlo[q]; r.wmax[q] = (float) whi[q]; }
      // instance culling (qa_kernel_cs.h csCullRay): bounds rounded outwards, and this node's share of the scene's widening constants
      CsCull &cb = cull[k];
      double boxAbs = 0;
      for (int q = 0; q < 3; ++q) {
        cb.lo[q] = std::nextafterf((float) wlo[q], -INFINITY);
        cb.hi[q] = std::nextafterf((float) whi[q], INFINITY);
        boxAbs = std::max({boxAbs, std::fabs(wlo[q]), std::fabs(whi[q])});
      }
      auto normInf = [](const float *m) { double n = 0; for (int rr = 0; rr < 3; ++rr) n = std::max(n, (double) std::fabs(m[rr]) + std::fabs(m[3 + rr]) + std::fabs(m[6 + rr])); return n; };
      auto vecInf = [](const float *v) { return std::max({(double) std::fabs(v[0]), (double) std::fabs(v[1]), (double) std::fabs(v[2])}); };
      double cond = normInf(a.tm) * normInf(a.itm), tmNorm = normInf(a.tm), posAbs = vecInf(a.pos);
      if (in.depth == 2) {
        cond *= normInf(in.tm) * normInf(in.itm);
        posAbs += normInf(a.tm) * vecInf(in.pos);
        tmNorm *= normInf(in.tm);
      }
      cullS1 = std::max(cullS1, posAbs + 1.0);
      cullS2 = std::max(cullS2, boxAbs + 1.0);
      cullK3 = std::max(cullK3, 2e-5 * cond);
      cullK4 = std::max(cullK4, 2.0 * tmNorm * (in.obj_type == QA_OBJ_MESH ? (double) r.nearPad : 0.0) + 1e-5 * (boxAbs + 1.0));
      if (!std::isfinite(cond) || !std::isfinite(boxAbs) || !std::isfinite(posAbs) || !std::isfinite(tmNorm)) cullOk = false;
    }
    c->csCullS1 = (float) cullS1; c->csCullS2 = (float) cullS2; c->csCullK3 = (float) cullK3; c->csCullK4 = (float) cullK4;
    if (!std::isfinite(c->csCullS1) || !std::isfinite(c->csCullS2) || !std::isfinite(c->csCullK3) || !std::isfinite(c->csCullK4)) cullOk = false;
    c->csCullOk = cullOk;
    if ((rc = DeviceCopy(c, cull, &c->csCullDev)) != QA_OK) return rc;
    if ((rc = DeviceCopy(c, ci, &c->csInstDev)) != QA_OK) return rc;
  }
  if ((rc = DeviceCopy(c, dmeshes, &ds.mesh)) != QA_OK) return rc;
  c->hostMeshes = dmeshes;
  if ((rc = DeviceCopy(c, dmat, &ds.mtl)) != QA_OK) return rc;
  // ---- texture-side tables (TEX kernel variants) -------------------------------------------------
  ds.texmap = QA_BLOB_PTR(qa_texmap, c->dBlob, h->off_texmaps);
  ds.tex = QA_BLOB_PTR(qa_texture, c->dBlob, h->off_textures);
  ds.bgTexmap = h->background.texmap;
  ds.envTexmap = h->environment.texmap;
  c->textured = textured;
  c->area = area;
  // Without reflective / refractive lobes a path is at most camera ray + one diffuse bounce: starting
  // the samples of a wave together keeps its coherent camera rays apart from the incoherent bounce
  // rays (+21 % on the Cornell box).  Long specular chains would make lanes wait for the longest path.
  // Textured scenes also start samples together: the 32-tap filtered lookups of camera hits are the
  // expensive part of their shading and stay coherent that way (+18 % on project7_object, whereas the
  // untextured glossy-caustics scene loses 14 % to waiting for its long specular chains).
  c->syncAuto = (!anySpecularLobes || textured) ? 1 : 0;
  if (area) {
    // hit log of the AREA variants: QA_MAX_PATH x 19 floats per thread of the largest grid
    const size_t threads = (size_t) c->numCUs * 8 * QA_BLOCK;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, threads * QA_MAX_PATH * QA_REC_FLOATS * sizeof(float)));
    c->sceneAllocs.push_back(p);
    ds.areaScratch = static_cast<float *>(p);
  }
  {
    // more shadow-casting lights than qa_integrate_cs takes in one batch: the slab its surface waits in between batches
    int shadowLights = 0;
    for (uint32_t i = 0; i < h->num_lights; ++i) shadowLights += light[i].type != QA_LIGHT_AMBIENT;
    if (shadowLights > QA_CS_LIGHT_BATCH && !area) {
      const size_t threads = (size_t) c->numCUs * 8 * QA_BLOCK;
      void *p = nullptr;
      HIP_TRY(hipMalloc(&p, threads * 13 * sizeof(float)));
      c->sceneAllocs.push_back(p);
      ds.csSurf = static_cast<float *>(p);
    }
  }
  if (textured) {
    std::vector<int32_t> mtex(8 * (size_t) h->num_materials, -1);
    for (uint32_t i = 0; i < h->num_materials; ++i) {
      mtex[8 * i + 0] = mats[i].diffuse.texmap;
      mtex[8 * i + 1] = mats[i].specular.texmap;
      mtex[8 * i + 2] = mats[i].emission.texmap;
      mtex[8 * i + 3] = mats[i].reflection.texmap;
      mtex[8 * i + 4] = mats[i].refraction.texmap;
      for (int k = 0; k < 5; ++k) if (mtex[8 * i + k] >= (int) h->num_texmaps) return Fail(QA_EINVAL, "bad texmap index");
    }
    if ((rc = DeviceCopy(c, mtex, &ds.mtlTex)) != QA_OK) return rc;
    // file textures as float RGB: TextureFile::Sample divides every byte it reads by 255.0f (src/textures/texture.cpp:120-131) - 12
    // divisions per bilinear tap, 384 per filtered lookup; the same IEEE division once per texel here gives the same bits
    {
      std::vector<uint32_t> toff(std::max<uint32_t>(h->num_textures, 1u), 0u);
      std::vector<float> tex4;
      try {
        for (uint32_t i = 0; i < h->num_textures; ++i) {
          const qa_texture &tx = textures[i];
          toff[i] = (uint32_t) (tex4.size() / 4);
          if (tx.type == QA_TEX_CHECKER || tx.width <= 0 || tx.height <= 0) continue;
          const size_t n = (size_t) tx.width * (size_t) tx.height;
          if (!inside(tx.off_texels, 3 * n)) return Fail(QA_EINVAL, "texture texels outside the blob");
          const unsigned char *px = blob + tx.off_texels;
          const size_t at = tex4.size();
          tex4.resize(at + 4 * n);
          for (size_t t = 0; t < n; ++t) {
            tex4[at + 4 * t + 0] = px[3 * t + 0] / 255.0f;
            tex4[at + 4 * t + 1] = px[3 * t + 1] / 255.0f;
            tex4[at + 4 * t + 2] = px[3 * t + 2] / 255.0f;
            tex4[at + 4 * t + 3] = 0.f;
          }
        }
      } catch (const std::bad_alloc &) { return Fail(QA_ENOMEM, "out of memory"); }
      if (tex4.size() / 4 > 0xFFFFFFFFull) return Fail(QA_EUNSUPPORTED, "more than 2^32 texels");
      if (tex4.empty()) tex4.assign(4, 0.f);
      const float *dt = nullptr;
      if ((rc = DeviceCopy(c, tex4, &dt)) != QA_OK) return rc;
      ds.texels = reinterpret_cast<const float4 *>(dt);
      if ((rc = DeviceCopy(c, toff, &ds.texOff)) != QA_OK) return rc;
    }
    // Texture::Sample's elliptical taps (src/core/texture.cpp:39-46), i = 1..31, host libm
    std::vector<float> taps(62);
    for (int i = 1; i < 32; ++i) {
      float x = HaltonF(i, 2), y = HaltonF(i, 3);
      const float r = sqrtf(x) * 0.5f;
      x = r * sinf(y * (float) M_PI * 2);
      y = r * cosf(y * (float) M_PI * 2);
      taps[2 * (i - 1)] = x;
      taps[2 * (i - 1) + 1] = y;
    }
    if ((rc = DeviceCopy(c, taps, &ds.texFilter)) != QA_OK) return rc;
  }
  ds.stackNeed = stackNeedMax;
  c->stackDepth = stackNeedMax < 8 ? 8 : stackNeedMax;
  // LDS budget per workgroup: resident image + stacks; small scenes stay entirely on the CU
  ds.stackDepth = c->stackDepth;
  // traversal stacks + 6 accumulator floats per lane (mean, variance)
  const size_t stackBytes = ((size_t) c->stackDepth + QA_LANE_SLOTS) * QA_BLOCK * sizeof(uint32_t);
  const size_t imageBytes = image.size() * sizeof(uint4);
  if (stackBytes > kMaxLdsPerBlock) return Fail(QA_EUNSUPPORTED, "BVH too deep for the LDS traversal stack");
  // workgroups of a resident scene also keep the cold path state in LDS columns (QA_LANE_SLOTS_RES)
  const size_t stackBytesRes = ((size_t) c->stackDepth + QA_LANE_SLOTS_RES) * QA_BLOCK * sizeof(uint32_t);
  c->resident = (imageBytes > 0 && imageBytes + stackBytesRes <= kResidentLdsBudget &&
                 h->num_instances <= QA_KARG_INST && h->num_meshes <= QA_KARG_MESH);
  if (c->resident) {
    const uint4 *dimg = nullptr;
    if ((rc = DeviceCopy(c, image, &dimg)) != QA_OK) return rc;
    ds.resident = dimg;
    ds.residentVec4 = (uint32_t) image.size();
    ds.resMaterials = resMaterials;
    for (uint32_t k = 0; k < h->num_instances; ++k) ds.instv[k] = inst[k];
    for (uint32_t k = 0; k < h->num_meshes; ++k) ds.meshv[k] = dmeshes[k];
  }
  c->ldsBytes = c->resident ? stackBytesRes + imageBytes : stackBytes;
  // qa_integrate_cs: per wave [ray slots | results | flags | pool items | accumulators]; four workgroups per CU (160 KB LDS)
  ds.csNodes = c->csNodesDev;
  ds.csTris = c->csTrisDev;
  ds.csLeafBox = c->csLeafBoxDev;
  ds.csInst = c->csInstDev;
  ds.csCull = c->csCullDev;
  ds.csCullS1 = c->csCullS1; ds.csCullS2 = c->csCullS2; ds.csCullK3 = c->csCullK3; ds.csCullK4 = c->csCullK4;
  ds.csItems = DevEnv("QA_CS_ITEMS") ? (uint32_t) atoi(DevEnv("QA_CS_ITEMS")) : 576u;
  ds.csSlots = DevEnv("QA_CS_SLOTS") ? (uint32_t) atoi(DevEnv("QA_CS_SLOTS")) : 80u;
  if (ds.csSlots < 64u) ds.csSlots = 64u;     // an instance enters up to 64 rays at once
  if (ds.csSlots > 256u) ds.csSlots = 256u;   // 8 bits of an item
  if (ds.csItems < 128u) ds.csItems = 128u;
  c->ldsBytesCs = (size_t) CsLdsWords(ds.csItems, ds.csSlots) * (QA_BLOCK / 64) * sizeof(uint32_t);
  memcpy(ds.cam.screenA, h->screenA, 12);
  memcpy(ds.cam.screenU, h->screenU, 12);
  memcpy(ds.cam.screenV, h->screenV, 12);
  memcpy(ds.cam.screenX, h->screenX, 12);
  memcpy(ds.cam.screenY, h->screenY, 12);
  memcpy(ds.cam.pos, h->cam_pos, 12);
  ds.cam.dof = h->dof;
  ds.cam.width = (int) h->width;
  ds.cam.height = (int) h->height;
  memcpy(ds.background, h->background.color, 12);
  memcpy(ds.environment, h->environment.color, 12);
  ds.num_inst = (int) h->num_instances;
  ds.num_lights = (int) h->num_lights;
  ds.num_materials = (int) h->num_materials;
  {
    static const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z[3] = {0, 0, 0};
    ds.rootIdentity = (memcmp(inst[0].tm, I, 36) == 0 && memcmp(inst[0].itm, I, 36) == 0 && memcmp(inst[0].pos, Z, 12) == 0) ? 1 : 0;
  }
  c->haveScene = true;
  c->csFits = csFits;
  SelectStaged(c);
  return SelectKernel(c);
}

static int DrainEvents(qa_ctx *c);

static int OwnTileRows(int y0, int y1, int tile_row0, int tile_row_step)
{
  const int tilesY = (y1 - y0 + 7) / 8;
  if (tile_row0 >= tilesY) return 0;
  return (tilesY - tile_row0 + tile_row_step - 1) / tile_row_step;
}

static int Render(qa_ctx *c, int x0, int y0, int x1, int y1, int tile_row0, int tile_row_step, int spp_min, int spp_max,
                  int max_bounce, uint32_t seed, uint32_t flags, float *d_rgb, float *d_depth, uint32_t *d_ns, hipStream_t s)
{
  if (!c->haveScene) return Fail(QA_ENOSCENE, "no scene uploaded");
  if (x0 < 0 || y0 < 0 || x1 > c->ds.cam.width || y1 > c->ds.cam.height || x1 <= x0 || y1 <= y0)
    return Fail(QA_EINVAL, "region outside the image");
  // sppMin = 0 would mean "no sample at all" (SuperSamplerHalton::Loop, src/scene/scene.cpp:92-97): refused
  if (spp_min < 1 || spp_max < spp_min || max_bounce < 0) return Fail(QA_EINVAL, "bad spp / bounce");
  if (c->area && max_bounce + 1 > QA_MAX_PATH) return Fail(QA_EUNSUPPORTED, "area lights: maxBounce must be <= 7");
  if (!d_rgb || !d_depth || !d_ns) return Fail(QA_EINVAL, "null output buffer");
  int rc = EnsureHalton(c, spp_max);
  if (rc != QA_OK) return rc;
  c->ds.halton = c->dHalton;
  c->ds.halton_count = c->haltonCount;

  if (tile_row0 < 0 || tile_row_step < 1) return Fail(QA_EINVAL, "bad strip partition");
  const int ownRows = OwnTileRows(y0, y1, tile_row0, tile_row_step);
  if (ownRows == 0) return QA_OK;  // nothing to do for this rank
  const bool whole = (tile_row0 == 0 && tile_row_step == 1);
  const size_t npix = (size_t) (x1 - x0) * (whole ? (size_t) (y1 - y0) : (size_t) ownRows * 8);
  // pixels skipped by a stop request (and the padding rows of a ragged last strip) read as "not rendered"
  HIP_TRY(hipMemsetAsync(d_ns, 0, npix * sizeof(uint32_t), s));
  unsigned int *work = c->dWork + c->workNext;
  c->workNext = (c->workNext + 1) % qa_ctx::kCounterRing;
  HIP_TRY(hipMemsetAsync(work, 0, sizeof(unsigned int), s));

  // one frame at a time per context (its device slabs are one per context): a frame on another stream than the last one waits for it
  if (!c->chunkEv) HIP_TRY(hipEventCreateWithFlags(&c->chunkEv, hipEventDisableTiming));
  if (c->chunkEvSet && s != c->lastStream) HIP_TRY(hipStreamWaitEvent(s, c->chunkEv, 0));

  RenderParams rp;
  rp.x0 = x0; rp.y0 = y0; rp.x1 = x1; rp.y1 = y1;
  rp.spp_min = spp_min; rp.spp_max = spp_max; rp.max_bounce = max_bounce;
  rp.seed = seed;
  rp.tile_row0 = tile_row0; rp.tile_row_step = tile_row_step; rp.own_tile_rows = ownRows; rp.pad = 0;
  rp.sync_samples = c->syncSamples < 0 ? c->syncAuto : c->syncSamples;
  rp.rgb = d_rgb; rp.depth = d_depth; rp.ns = d_ns;
  rp.work_counter = work;
  rp.tile_order = nullptr;
  {
    // Tiles are handed out centre-first: the cheap ones (rays that leave the scene at the image
    // border) end up last, so the end-of-frame tail is made of short tiles instead of long ones.
    const int tx = (x1 - x0 + 7) / 8;
    const uint64_t key = ((uint64_t) tx << 40) ^ ((uint64_t) ownRows << 20) ^ ((uint64_t) tile_row0 << 8) ^ (uint64_t) tile_row_step ^
                         ((uint64_t) (y1 - y0) << 50);
    if (key != c->orderKey || !c->dOrder) {
      const size_t n = (size_t) tx * ownRows;
      std::vector<std::pair<float, uint32_t>> v(n);
      const float cx = 0.5f * (x1 - x0), cy = 0.5f * (y1 - y0);
      for (int r = 0; r < ownRows; ++r)
        for (int i = 0; i < tx; ++i) {
          const float px = i * 8 + 4 - cx, py = (tile_row0 + r * tile_row_step) * 8 + 4 - cy;
          v[(size_t) r * tx + i] = {px * px + py * py, (uint32_t) (r * tx + i)};
        }
      std::stable_sort(v.begin(), v.end(), [](const std::pair<float, uint32_t> &a, const std::pair<float, uint32_t> &b) { return a.first < b.first; });
      std::vector<uint32_t> order(n);
      for (size_t i = 0; i < n; ++i) order[i] = v[i].second;
      if (c->dOrder) { HIP_TRY(hipStreamSynchronize(s)); (void) hipFree(c->dOrder); c->dOrder = nullptr; }
      HIP_TRY(hipMalloc((void **) &c->dOrder, n * sizeof(uint32_t)));
      HIP_TRY(hipMemcpyAsync(c->dOrder, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
      HIP_TRY(hipStreamSynchronize(s));
      c->orderKey = key;
    }
    if (c->tileOrder) rp.tile_order = c->dOrder;
  }
  rp.stop_flag = c->dStopAlias;
  rp.counters = c->dCounters;
  // Scene::usePhotonMap: once qa_photon_maps_build has run, frames gather from the maps
  const bool pmOn = c->photonReady;
  memset(rp.pm, 0, sizeof(rp.pm));
  rp.heap = nullptr;
  if (pmOn) {
    for (int k = 0; k < 2; ++k) {
      const qa_photon_map_params &mp = k ? c->photonParams.caustics : c->photonParams.photon;
      rp.pm[k].node = static_cast<const uint4 *>(c->dPmTables[k][0]);
      rp.pm[k].dir = static_cast<const float4 *>(c->dPmTables[k][1]);
      rp.pm[k].power = static_cast<const float4 *>(c->dPmTables[k][2]);
      rp.pm[k].count = mp.size;
      rp.pm[k].half = (int32_t) (mp.size / 2) - 1;   // halfStoredPhotons = (photons.size() - 1) / 2 - 1, cyPhotonMap.h:291
      rp.pm[k].radius = mp.radius;
    }
    rp.heap = static_cast<uint2 *>(c->dHeap);
  }
  DScene ds = c->ds;
  if (pmOn) ds.stackDepth = c->stackDepthPm;
  const bool cs = c->kernelCs && !pmOn && !(flags & QA_RENDER_STATS);
  // the cooperative kernel's third way between "a lane starts its next sample at once" (0) and "when the whole wave is between samples"
  // (1): a finished path waits until 32 of the wave's have gathered, then those lanes finish and start samples together.  Where 1 was
  // the per-scene choice, and on scenes of many lights (an iteration is mostly their shadow batches), it beats both (experiments.txt 22)
  if (cs && c->syncSamples < 0 && ((c->syncAuto && c->textured) || c->csMany)) rp.sync_samples = 32;   // (the variants that carry the code)
  if (cs && c->area) rp.sync_samples = 1;   // the cooperative AREA variants evaluate a wave's lights between its samples
  ds.csCullOn = (c->optCsCull && c->csCullOk) ? 1u : 0u;
  ds.csForceExact = c->optCsForceExact;
  ds.walkZeroTerms = c->optWalkZeroTerms;
  ds.csPoolLimit = DevEnv("QA_CS_POOL") ? (uint32_t) std::max(64, atoi(DevEnv("QA_CS_POOL"))) : c->optCsPool;
  const size_t ldsBytes = pmOn ? c->ldsBytesPm : (cs ? c->ldsBytesCs : c->ldsBytes);
  const KernelFn kernel = pmOn ? ((flags & QA_RENDER_STATS) ? c->kernelPmStats : c->kernelPm)
                               : ((flags & QA_RENDER_STATS) ? c->kernelStats : (cs ? c->kernelCs : c->kernel));

  const unsigned tiles = (unsigned) ((x1 - x0 + 7) / 8) * (unsigned) ownRows;
  const long long needBlocks = ((long long) tiles * 64 + QA_BLOCK - 1) / QA_BLOCK;
  long long blocks = (long long) c->numCUs * (c->blocksPerCU > 0 ? c->blocksPerCU : (pmOn ? c->blocksPerCUPm : (cs ? c->blocksPerCUCs : c->blocksPerCUAuto)));
  if (pmOn && blocks > (long long) c->numCUs * 8) blocks = (long long) c->numCUs * 8;   // the heap scratch is sized for this
  if (blocks > needBlocks) blocks = needBlocks;
  if (blocks < 1) blocks = 1;

  // ---- tiles in sample chunks (qa_kernel.h, section A): the per-lane kernels and the cooperative kernel's textured variants (in the
  // untextured ones the code costs more than their 4K frames' tails: 31 tiles per wave).  Per frame: when a wave gets fewer than 16 tiles, a tile's samples are handed out in chunks, so that
  // the frame ends on work items an eighth the size: half of them first, then eighths, where a wave's lanes start their samples
  // together (they also reach a chunk's end together); three quarters first where they do not (every hand-over then waits for the
  // tile's slowest pixel).  Cornell box 1080p @ 512 spp: 81.3 -> 72.5 ms (profiles/round03/chunk_sweep.txt).
  rp.chunk_spp = 0; rp.chunk_tail = 0; rp.num_chunks = 1; rp.chunk_pad = 0; rp.tile_progress = nullptr; rp.pix_state = nullptr;
  if ((!cs || c->textured) && !(c->wf.mode == QA_PIPE_STAGED) && c->optChunkSpp != 0) {   // (cooperative kernel: the textured variants carry the code)
    uint32_t chunk = 0, tail = 0;
    if (c->optChunkSpp > 0) chunk = (uint32_t) c->optChunkSpp;
    else if ((long long) tiles < 16 * blocks * (QA_BLOCK / 64) && (long long) tiles >= blocks * (QA_BLOCK / 64) && spp_max >= 64)
      chunk = rp.sync_samples ? (uint32_t) spp_max / 2u : (uint32_t) spp_max - (uint32_t) spp_max / 4u;
    tail = c->optChunkTail > 0 ? (uint32_t) c->optChunkTail : std::max(16u, (uint32_t) spp_max / 8u);
    if (chunk > 0 && chunk < (uint32_t) spp_max) {
      const uint32_t nChunks = 1u + ((uint32_t) spp_max - chunk + tail - 1) / tail;
      if ((unsigned long long) tiles * 64ull * nChunks < 0xF0000000ull) {   // (the work counter is 32 bits; every exiting wave adds 64 more)
        const size_t needState = (size_t) tiles * 64 * 8, needProg = tiles;
        if (needState > c->pixStateWords) {
          if (c->dPixState) { HIP_TRY(hipDeviceSynchronize()); (void) hipFree(c->dPixState); c->dPixState = nullptr; c->pixStateWords = 0; }
          HIP_TRY(hipMalloc((void **) &c->dPixState, needState * sizeof(uint32_t)));
          c->pixStateWords = needState;
        }
        if (needProg > c->tileProgressWords) {
          if (c->dTileProgress) { HIP_TRY(hipDeviceSynchronize()); (void) hipFree(c->dTileProgress); c->dTileProgress = nullptr; c->tileProgressWords = 0; }
          HIP_TRY(hipMalloc((void **) &c->dTileProgress, needProg * sizeof(uint32_t)));
          c->tileProgressWords = needProg;
        }
        HIP_TRY(hipMemsetAsync(c->dTileProgress, 0, needProg * sizeof(uint32_t), s));
        rp.chunk_spp = chunk; rp.chunk_tail = tail; rp.num_chunks = nChunks; rp.tile_progress = c->dTileProgress; rp.pix_state = c->dPixState;
      }
    }
  }

  // ---- which integrator: both return the same bits.  The staged one (qa_wf.h) runs on request only (QA_PIPE_STAGED): since
  // the cooperative walks the megakernel is the faster one on every scene measured, and round 2's timed probe between the
  // two is gone.
  const bool staged = c->wf.mode == QA_PIPE_STAGED && StagedTakes(c, flags, spp_max, max_bounce, (size_t) tiles * 64);

  EventPair ev;
  if (!c->freeEvents.empty()) { ev = c->freeEvents.back(); c->freeEvents.pop_back(); }
  else { HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b)); }
  HIP_TRY(hipEventRecord(ev.a, s));
  if (staged) {
    // one event pair around the whole frame of the staged integrator (qa_wf.h)
    rc = RenderStaged(c, ds, rp, s, rp.counters);
    if (rc != QA_OK) { c->freeEvents.push_back(ev); return rc; }
  } else {
    hipLaunchKernelGGL(kernel, dim3((unsigned) blocks), dim3(QA_BLOCK), (unsigned) ldsBytes, s, ds, rp);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(c->chunkEv, s));
  c->chunkEvSet = true;
  c->lastStream = s;
  HIP_TRY(hipEventRecord(ev.b, s));
  {
    // the kernel this frame really ran on
    if (staged) c->launchedName = c->kernelName;
    else {
      c->launchedName = MegaName(c, cs);
      if (pmOn) c->launchedName += " + photon-map gathers (PHOTON=1)";
      if (flags & QA_RENDER_STATS) c->launchedName += " counting variant (STATS=1, reference tree)";
    }
  }
  c->pending.push_back(ev);
  c->launches++;
  // a caller that never asks for timers or counters must not grow the event list without bound
  if (c->pending.size() > 256) return DrainEvents(c);
  return QA_OK;
}

static int DrainEvents(qa_ctx *c)
{
  for (EventPair &ev : c->pending) {
    HIP_TRY(hipEventSynchronize(ev.b));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
    c->totalMs += ms;
    c->freeEvents.push_back(ev);
  }
  c->pending.clear();
  return QA_OK;
}

__global__ void qa_sincos_probe(const float *x, int n, float *s, float *c)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { s[i] = qsinf(x[i]); c[i] = qcosf(x[i]); }
}

__global__ void qa_math_probe(int fn, const float *x, const float *y, int n, float *out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  switch (fn) {
    case 0: out[i] = qsinf(x[i]); break;
    case 1: out[i] = qcosf(x[i]); break;
    case 2: out[i] = qpowf(x[i], y[i]); break;
    default: out[i] = qexpf(x[i]); break;
  }
}

extern "C" {

// the device build of qa_device_math.h: fn 0 sinf, 1 cosf, 2 powf(x, y), 3 expf (host arrays in / out)
int qa_test_math_device(int fn, const float *x, const float *y, int n, float *out)
{
  if (!x || !out || n <= 0 || fn < 0 || fn > 3 || (fn == 2 && !y)) return Fail(QA_EINVAL, "bad argument");
  float *dx = nullptr, *dy = nullptr, *dout = nullptr;
  HIP_TRY(hipMalloc((void **) &dx, n * sizeof(float)));
  HIP_TRY(hipMalloc((void **) &dy, n * sizeof(float)));
  HIP_TRY(hipMalloc((void **) &dout, n * sizeof(float)));
  HIP_TRY(hipMemcpy(dx, x, n * sizeof(float), hipMemcpyHostToDevice));
  if (y) HIP_TRY(hipMemcpy(dy, y, n * sizeof(float), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(qa_math_probe, dim3((n + 255) / 256), dim3(256), 0, 0, fn, dx, dy, n, dout);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout, n * sizeof(float), hipMemcpyDeviceToHost));
  (void) hipFree(dx); (void) hipFree(dy); (void) hipFree(dout);
  return QA_OK;
}

// Self-test hooks: the device math next to the host libm (tests/test_gpu_parity.py, tests/test_device_math.py)
int qa_test_sincosf_device(const float *x, int n, float *s, float *c)
{
  if (!x || !s || !c || n <= 0) return Fail(QA_EINVAL, "bad argument");
  float *dx = nullptr, *dsn = nullptr, *dcs = nullptr;
  HIP_TRY(hipMalloc((void **) &dx, n * sizeof(float)));
  HIP_TRY(hipMalloc((void **) &dsn, n * sizeof(float)));
  HIP_TRY(hipMalloc((void **) &dcs, n * sizeof(float)));
  HIP_TRY(hipMemcpy(dx, x, n * sizeof(float), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(qa_sincos_probe, dim3((n + 255) / 256), dim3(256), 0, 0, dx, n, dsn, dcs);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(s, dsn, n * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(c, dcs, n * sizeof(float), hipMemcpyDeviceToHost));
  (void) hipFree(dx); (void) hipFree(dsn); (void) hipFree(dcs);
  return QA_OK;
}
// the same source compiled for the host (no GPU needed)
int qa_test_math_host(int fn, const float *x, const float *y, int n, float *out)
{
  if (!x || !out || n <= 0) return QA_EINVAL;
  for (int i = 0; i < n; ++i) {
    switch (fn) {
      case 0: out[i] = qsinf(x[i]); break;
      case 1: out[i] = qcosf(x[i]); break;
      case 2: out[i] = qpowf(x[i], y ? y[i] : 1.f); break;
      case 3: out[i] = qexpf(x[i]); break;
      default: return QA_EINVAL;
    }
  }
  return QA_OK;
}

const char *qa_last_error(void) { return g_err.c_str(); }

int qa_ctx_create(int device_id, qa_ctx **out)
{
  if (!out) return Fail(QA_EINVAL, "null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return Fail(QA_EHIP, "no HIP device: the qaray HIP path has no CPU fallback");
  if (device_id < 0 || device_id >= n) return Fail(QA_EINVAL, "device id out of range");
  HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  qa_ctx *c = new (std::nothrow) qa_ctx;
  if (!c) return Fail(QA_ENOMEM, "out of memory");
  c->device = device_id;
  c->numCUs = prop.multiProcessorCount;
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipMalloc((void **) &c->dWork, qa_ctx::kCounterRing * sizeof(unsigned int))) != hipSuccess ||
      (e = hipMalloc((void **) &c->dCounters, sizeof(DCounters))) != hipSuccess ||
      (e = hipMemset(c->dCounters, 0, sizeof(DCounters))) != hipSuccess ||
      (e = hipHostMalloc((void **) &c->hStop, sizeof(int), hipHostMallocMapped)) != hipSuccess) {
    qa_ctx_destroy(c);
    return Fail(QA_EHIP, std::string("context setup: ") + hipGetErrorString(e));
  }
  *c->hStop = 0;
  if (const char *e = DevEnv("QA_SYNC")) c->syncSamples = atoi(e);
  c->tileOrder = DevEnv("QA_NO_TILE_ORDER") == nullptr;
  if (const char *e = DevEnv("QA_WF_BUDGET")) c->wf.budget = atoi(e) > 0 ? (uint32_t) atoi(e) : 512u;
  if (const char *e = DevEnv("QA_WF_GATE")) c->wf.gate = (uint32_t) std::max(1, atoi(e));
  // (tile groups of the staged integrator: one unless qa_set_option("staged_groups") says otherwise - several groups only pay
  // when the process gave the HIP runtime a hardware queue per group stream, GPU_MAX_HW_QUEUES >= 8 before its first call)
  if (const char *e = DevEnv("QA_WF_REDO_ASYNC")) c->wf.redoAsync = atoi(e) != 0;
  if (const char *e = DevEnv("QA_WF_GROUPS")) c->wf.numGroups = std::max(1, std::min(atoi(e), (int) WfHost::kMaxGroups));
  if (const char *e = DevEnv("QA_WF_STACK")) c->wf.stackCap = atoi(e) > 1 ? (uint32_t) atoi(e) : 24u;
  if (const char *e = DevEnv("QA_WF_BLOCKS")) c->wf.traceBlocksPerCU = atoi(e);
  if ((e = hipHostGetDevicePointer((void **) &c->dStopAlias, c->hStop, 0)) != hipSuccess) {
    qa_ctx_destroy(c);
    return Fail(QA_EHIP, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
  }
  *out = c;
  return QA_OK;
}

int qa_ctx_destroy(qa_ctx *c)
{
  if (!c) return QA_OK;
  (void) hipSetDevice(c->device);
  if (c->stream) (void) hipStreamSynchronize(c->stream);
  FreeScene(c);
  FreeStaged(c);
  for (EventPair &ev : c->pending) { (void) hipEventDestroy(ev.a); (void) hipEventDestroy(ev.b); }
  for (EventPair &ev : c->freeEvents) { (void) hipEventDestroy(ev.a); (void) hipEventDestroy(ev.b); }
  if (c->dHalton) (void) hipFree(c->dHalton);
  if (c->dOrder) (void) hipFree(c->dOrder);
  if (c->dWork) (void) hipFree(c->dWork);
  if (c->dPixState) (void) hipFree(c->dPixState);
  if (c->dTileProgress) (void) hipFree(c->dTileProgress);
  if (c->chunkEv) (void) hipEventDestroy(c->chunkEv);
  if (c->dCounters) (void) hipFree(c->dCounters);
  if (c->hStop) (void) hipHostFree(c->hStop);
  if (c->dRgb) (void) hipFree(c->dRgb);
  if (c->dDepth) (void) hipFree(c->dDepth);
  if (c->dNs) (void) hipFree(c->dNs);
  if (c->stream) (void) hipStreamDestroy(c->stream);
  delete c;
  return QA_OK;
}

int qa_scene_upload(qa_ctx *c, const void *host_blob, uint64_t nbytes)
{
  if (!c || !host_blob || nbytes == 0) return Fail(QA_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  FreeScene(c);
  try {
    c->hostBlob.assign((const unsigned char *) host_blob, (const unsigned char *) host_blob + nbytes);
  } catch (const std::bad_alloc &) { return Fail(QA_ENOMEM, "out of memory"); }
  HIP_TRY(hipMalloc((void **) &c->dBlob, nbytes));
  HIP_TRY(hipMemcpy(c->dBlob, host_blob, nbytes, hipMemcpyHostToDevice));
  const int rc = PrepareScene(c);
  if (rc != QA_OK) FreeScene(c);
  return rc;
}

int qa_scene_upload_device(qa_ctx *c, const void *device_blob, uint64_t nbytes)
{
  if (!c || !device_blob || nbytes == 0) return Fail(QA_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  FreeScene(c);
  try { c->hostBlob.resize(nbytes); } catch (const std::bad_alloc &) { return Fail(QA_ENOMEM, "out of memory"); }
  HIP_TRY(hipMalloc((void **) &c->dBlob, nbytes));
  HIP_TRY(hipMemcpy(c->dBlob, device_blob, nbytes, hipMemcpyDeviceToDevice));
  HIP_TRY(hipMemcpy(c->hostBlob.data(), device_blob, nbytes, hipMemcpyDeviceToHost));
  const int rc = PrepareScene(c);
  if (rc != QA_OK) FreeScene(c);
  return rc;
}

int qa_render_region_device(qa_ctx *c, int x0, int y0, int x1, int y1, int spp_min, int spp_max, int max_bounce,
                            uint32_t seed, uint32_t flags, float *d_rgb, float *d_depth, uint32_t *d_ns, void *hip_stream)
{
  if (!c) return Fail(QA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = hip_stream ? (hipStream_t) hip_stream : c->stream;
  return Render(c, x0, y0, x1, y1, 0, 1, spp_min, spp_max, max_bounce, seed, flags, d_rgb, d_depth, d_ns, s);
}

int qa_render_strips_device(qa_ctx *c, int x0, int y0, int x1, int y1, int first_strip, int strip_step, int spp_min,
                            int spp_max, int max_bounce, uint32_t seed, uint32_t flags, float *d_rgb, float *d_depth,
                            uint32_t *d_ns, void *hip_stream)
{
  if (!c) return Fail(QA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t s = hip_stream ? (hipStream_t) hip_stream : c->stream;
  return Render(c, x0, y0, x1, y1, first_strip, strip_step, spp_min, spp_max, max_bounce, seed, flags, d_rgb, d_depth, d_ns, s);
}

int qa_strip_count(int y0, int y1, int first_strip, int strip_step)
{
  if (y1 <= y0 || first_strip < 0 || strip_step < 1) return 0;
  return OwnTileRows(y0, y1, first_strip, strip_step);
}

int qa_render_region(qa_ctx *c, int x0, int y0, int x1, int y1, int spp_min, int spp_max, int max_bounce,
                     uint32_t seed, uint32_t flags, float *rgb, float *depth, uint32_t *ns)
{
  if (!c || !rgb || !depth || !ns) return Fail(QA_EINVAL, "null argument");
  if (x1 <= x0 || y1 <= y0) return Fail(QA_EINVAL, "empty region");
  HIP_TRY(hipSetDevice(c->device));
  const size_t npix = (size_t) (x1 - x0) * (y1 - y0);
  if (npix > c->stagePixels) {
    if (c->dRgb) (void) hipFree(c->dRgb);
    if (c->dDepth) (void) hipFree(c->dDepth);
    if (c->dNs) (void) hipFree(c->dNs);
    c->dRgb = c->dDepth = nullptr;
    c->dNs = nullptr;
    c->stagePixels = 0;
    HIP_TRY(hipMalloc((void **) &c->dRgb, npix * 3 * sizeof(float)));
    HIP_TRY(hipMalloc((void **) &c->dDepth, npix * sizeof(float)));
    HIP_TRY(hipMalloc((void **) &c->dNs, npix * sizeof(uint32_t)));
    c->stagePixels = npix;
  }
  const int rc = Render(c, x0, y0, x1, y1, 0, 1, spp_min, spp_max, max_bounce, seed, flags, c->dRgb, c->dDepth, c->dNs, c->stream);
  if (rc != QA_OK) return rc;
  HIP_TRY(hipMemcpyAsync(rgb, c->dRgb, npix * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(depth, c->dDepth, npix * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(ns, c->dNs, npix * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DrainEvents(c);   // the frame is complete: fold its event pair into the kernel time
}

int qa_synchronize(qa_ctx *c)
{
  if (!c) return Fail(QA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DrainEvents(c);
}

int qa_request_stop(qa_ctx *c)
{
  if (!c) return Fail(QA_EINVAL, "null context");
  __atomic_store_n(c->hStop, 1, __ATOMIC_SEQ_CST);
  return QA_OK;
}
int qa_clear_stop(qa_ctx *c)
{
  if (!c) return Fail(QA_EINVAL, "null context");
  __atomic_store_n(c->hStop, 0, __ATOMIC_SEQ_CST);
  return QA_OK;
}

int qa_get_counters(qa_ctx *c, qa_counters *out)
{
  if (!c || !out) return Fail(QA_EINVAL, "null argument");
  int rc = qa_synchronize(c);
  if (rc != QA_OK) return rc;
  DCounters h;
  HIP_TRY(hipMemcpy(&h, c->dCounters, sizeof(h), hipMemcpyDeviceToHost));
  out->samples = h.samples;
  out->casts_normal = h.casts_normal;
  out->casts_shadow = h.casts_shadow;
  out->bvh_nodes = h.bvh_nodes;
  out->tri_tests = h.tri_tests;
  out->pixels = h.pixels;
#ifdef QA_STAMPS
  {
    const double w = (double) std::max<unsigned long long>(h.stamp[9], 1), k = (double) std::max<unsigned long long>(h.stamp[0], 1);
    fprintf(stderr, "[stamps] waves %llu, iterations/wave %.0f, cycles/wave %.3e | share of wave time: fetch+start %.3f, closest %.3f (mesh walks %.3f), shade %.3f, "
            "direct light %.3f (shadow mesh walks %.3f), sample end %.3f, miss branch %.3f, hit before shading %.3f, spawn %.3f\n", h.stamp[9], h.stamp[8] / w, k / w, h.stamp[1] / k, h.stamp[2] / k, h.stamp[3] / k,
            h.stamp[4] / k, h.stamp[5] / k, h.stamp[6] / k, h.stamp[7] / k, h.stamp[10] / k, h.stamp[11] / k, h.stamp[12] / k);
    if (c->kernelCs && h.stamp[11])   // qa_integrate_cs reuses slots 10 / 11: items taken from the pool / rounds of the cooperative walks
      fprintf(stderr, "[stamps] cooperative walks: %llu rounds, %.1f of 64 lanes hold an item on average (lane occupancy of the walks %.3f); %.3f of the rounds are leaf rounds; 'mesh walks' above = the rounds alone\n", h.stamp[11],
              (double) h.stamp[10] / (double) h.stamp[11], (double) h.stamp[10] / (64.0 * (double) h.stamp[11]), (double) h.stamp[12] / (double) h.stamp[11]);
    if (c->kernelCs)
      fprintf(stderr, "[stamps] closest-hit sweeps without their rounds %.3f, winners' details %.3f, shadow sweeps without their rounds %.3f; lanes sent to the exact walks: %llu closest, %llu shadow (of %llu + %llu casts)\n",
              (h.stamp[13] - (double) h.stamp[3]) / k, h.stamp[14] / k, (h.stamp[17] - (double) h.stamp[6]) / k, h.stamp[15], h.stamp[16], (unsigned long long) h.casts_normal, (unsigned long long) h.casts_shadow);
  }
#endif
  return QA_OK;
}
int qa_reset_counters(qa_ctx *c)
{
  if (!c) return Fail(QA_EINVAL, "null context");
  int rc = qa_synchronize(c);
  if (rc != QA_OK) return rc;
  HIP_TRY(hipMemset(c->dCounters, 0, sizeof(DCounters)));
  if (c->wf.dStats) HIP_TRY(hipMemset(c->wf.dStats, 0, sizeof(WfStats)));
  c->wf.iterations = c->wf.raysClosest = c->wf.raysShadow = c->wf.jobs = c->wf.redo = 0;
  return QA_OK;
}

int qa_get_staged_stats(qa_ctx *c, uint64_t out[QA_STAGED_STATS])
{
  if (!c || !out) return Fail(QA_EINVAL, "null argument");
  int rc = qa_synchronize(c);
  if (rc != QA_OK) return rc;
  WfStats st;
  memset(&st, 0, sizeof(st));
  if (c->wf.dStats) HIP_TRY(hipMemcpy(&st, c->wf.dStats, sizeof(st), hipMemcpyDeviceToHost));
  const uint64_t v[QA_STAGED_STATS] = {c->wf.iterations, c->wf.raysClosest, c->wf.raysShadow, c->wf.jobs, c->wf.redo, st.jobs, st.nodeSteps,
                                       st.leafSteps, st.triTests, st.redo, st.suspended, st.laneSlots, st.waveRounds};
  memcpy(out, v, sizeof(v));
  return QA_OK;
}

int qa_set_pipeline(qa_ctx *c, int mode)
{
  if (!c) return Fail(QA_EINVAL, "null context");
  if (mode < QA_PIPE_MEGA || mode > QA_PIPE_AUTO) return Fail(QA_EINVAL, "pipeline mode must be QA_PIPE_MEGA, QA_PIPE_STAGED or QA_PIPE_AUTO");
  c->wf.mode = mode;
  c->wf.modeSet = true;
  if (c->haveScene) SetKernelName(c);
  return QA_OK;
}

const char *qa_get_kernel_name(qa_ctx *c)
{
  if (!c || !c->haveScene) return "";
  return c->launchedName.empty() ? c->kernelName.c_str() : c->launchedName.c_str();
}

int qa_get_kernel_time(qa_ctx *c, double *total_ms, uint64_t *launches)
{
  if (!c || !total_ms || !launches) return Fail(QA_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  int rc = DrainEvents(c);
  if (rc != QA_OK) return rc;
  *total_ms = c->totalMs;
  *launches = c->launches;
  return QA_OK;
}
int qa_reset_kernel_time(qa_ctx *c)
{
  if (!c) return Fail(QA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  int rc = DrainEvents(c);
  if (rc != QA_OK) return rc;
  c->totalMs = 0;
  c->launches = 0;
  return QA_OK;
}

int qa_set_option(qa_ctx *c, const char *name, long long value)
{
  if (!c || !name) return Fail(QA_EINVAL, "null argument");
  const std::string n(name);
  if (n == "coop") {
    c->optCoop = value != 0;
    if (c->haveScene) return SelectKernel(c);
  } else if (n == "cs_cull") c->optCsCull = value != 0;
  else if (n == "cs_force_exact") c->optCsForceExact = (uint32_t) (value & 3);
  else if (n == "walk_zero_terms") c->optWalkZeroTerms = value ? 1u : 0u;
  else if (n == "chunk_spp") c->optChunkSpp = value < 0 ? -1 : (int) (value > 65535 ? 65535 : value);
  else if (n == "chunk_tail") c->optChunkTail = value < 0 ? 0 : (int) (value > 65535 ? 65535 : value);
  else if (n == "cs_pool_limit") c->optCsPool = value > 0 ? (uint32_t) std::max<long long>(64, value) : 0u;
  else if (n == "sync_samples") c->syncSamples = value < 0 ? -1 : (value > 64 ? 64 : (int) value);
  else if (n == "tile_order") c->tileOrder = value != 0;
  else if (n == "staged_groups") {
    c->wf.numGroups = (int) std::max<long long>(1, std::min<long long>(value, WfHost::kMaxGroups));
    if (c->haveScene) SetKernelName(c);
  } else if (n == "verbose") c->optVerbose = value != 0;
  else return Fail(QA_EINVAL, "unknown option '" + n + "'");
  return QA_OK;
}

int qa_debug_scrub_scratch(qa_ctx *c, uint32_t pattern)
{
  if (!c) return Fail(QA_EINVAL, "null context");
  HIP_TRY(hipSetDevice(c->device));
  // eight waves per SIMD on every CU: every wave slot of the chip - and with it every private segment the next launch can get -
  // holds a wave of this kernel at the same time (each lingers until the grid has been placed)
  hipLaunchKernelGGL(qa::qa_scrub_scratch, dim3((unsigned) c->numCUs * 8), dim3(256), 0, c->stream, pattern, reinterpret_cast<uint32_t *>(c->dCounters));
  HIP_TRY(hipGetLastError());
  for (int g = 0; g < c->wf.numGroups; ++g)
    if (c->wf.groups[g].stream) {
      hipLaunchKernelGGL(qa::qa_scrub_scratch, dim3((unsigned) c->numCUs * 8), dim3(256), 0, c->wf.groups[g].stream, pattern, reinterpret_cast<uint32_t *>(c->dCounters));
      HIP_TRY(hipGetLastError());
    }
  if (c->wf.redoStream) {
    hipLaunchKernelGGL(qa::qa_scrub_scratch, dim3((unsigned) c->numCUs * 8), dim3(256), 0, c->wf.redoStream, pattern, reinterpret_cast<uint32_t *>(c->dCounters));
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipDeviceSynchronize());
  return QA_OK;
}

int qa_set_launch_config(qa_ctx *c, int blocks_per_cu, int threads_per_block)
{
  if (!c) return Fail(QA_EINVAL, "null context");
  if (threads_per_block != 0 && threads_per_block != QA_BLOCK) return Fail(QA_EINVAL, "this build supports 256-thread workgroups only");
  if (blocks_per_cu < 0 || blocks_per_cu > 8) return Fail(QA_EINVAL, "blocks_per_cu must be in 0..8");
  c->blocksPerCU = blocks_per_cu;
  return QA_OK;
}

}  // extern "C"
