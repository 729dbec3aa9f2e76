// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY.
//
// Driver linked against the REAL reference (wilsonCernWq/qaray) objects, compiled from the
// reference's own sources by oracle/Makefile.  It exists because the reference
//   * only writes 8-bit sRGB pixels (src/renderers/renderer.cpp:347-365),
//   * seeds its RNG from time() (src/samplers/Sampler_Marsaglia.cpp:32-42),
//   * has no resolution / seed / crop flags (src/main.cpp:17-44),
// none of which allows a float parity check.  The harness therefore
//   1. interposes rand()/srand() so that the reference's own Sampler_Marsaglia::Init() produces
//      seed[0] = qa_pixel_seed(seed, pixel) (include/qa_seed.h), and resets the worker thread's
//      sampler before every pixel => one reproducible xorshift32 stream per pixel;
//   2. subclasses qaray::Renderer and drives the per-pixel sample loop through the reference's
//      public API (SuperSamplerHalton, Scene::TraceNodeNormal, Material::Shade,
//      TexturedColor::Sample) following src/renderers/renderer.cpp:302-346, but keeps the
//      LINEAR FLOAT mean radiance, the first-sample depth and the sample count;
//   3. counts top-level ray casts with linker --wrap on Scene::TraceNodeNormal/Shadow;
//   5. with --photon-map, fills Scene::photonmap / causticsmap through the reference's own
//      Light::RandomPhoton, Scene::TraceNodeNormal, Material::RandomPhotonBounce and cyPhotonMap
//      (store, scale, balance), one emission at a time as src/renderers/renderer.cpp:146-271
//      does, but on one RNG stream per emission (include/qa_photon.h), sets Scene::usePhotonMap
//      and dumps the balanced arrays (<out>.photonmap.bin, <out>.caustics.bin);
//   4. can dump what the reference's loader built (node tree, mesh faces, BVH) as JSON so the
//      repo's own loader / BVH builder / flattener can be compared field by field.
// All tracing and shading arithmetic is executed by the reference's code.
//
// Output: <out>.rgb.f32 (h*w*3 float32), <out>.depth.f32 (h*w float32), <out>.ns.u32 (h*w
// uint32 sample counts), <out>.json (metadata, timing, counters).

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <chrono>
#include <atomic>
#include <omp.h>

#define private public   // test-only: BVH arrays of TriObj are private (objects.h:76)
#define protected public
#include "objects/objects.h"
#undef private
#undef protected
#include "renderers/renderer.h"
#include "materials/materials.h"
#include "lights/lights.h"
#include "parser/xmlload.h"

#include "qa_seed.h"
#include "qa_photon.h"

// ---------------------------------------------------------------------------------------------
// rand()/srand() interposition (the executable's definitions win over libc's)
// ---------------------------------------------------------------------------------------------
static uint32_t g_seed = QA_DEFAULT_SEED;
static thread_local uint32_t tl_pixel = 0;
static thread_local uint32_t tl_stream = 0;   // 0: pixel streams; QA_STREAM_PHOTON / QA_STREAM_CAUSTICS: emissions
namespace qaray { float LinearToSRGB(const float c); }   // src/renderers/renderer.cpp:34
extern "C" int rand(void) { return (int) qa_pixel_rand(g_seed ^ tl_stream, tl_pixel); }
extern "C" void srand(unsigned) {}

// ---------------------------------------------------------------------------------------------
// cast counters through --wrap (only calls from other translation units are redirected, i.e.
// exactly the top-level casts; the in-file recursion over child nodes is not)
// ---------------------------------------------------------------------------------------------
static thread_local uint64_t tl_casts_normal = 0, tl_casts_shadow = 0;
#define QA_SYM_NORMAL _ZN5qaray5Scene15TraceNodeNormalERNS_4NodeERNS_7DiffRayERNS_11DiffHitInfoE
#define QA_SYM_SHADOW _ZN5qaray5Scene15TraceNodeShadowERNS_4NodeERNS_3RayERNS_7HitInfoE
#define QA_CAT2(a, b) a##b
#define QA_CAT(a, b) QA_CAT2(a, b)
extern "C" {
bool QA_CAT(__real_, QA_SYM_NORMAL)(qaray::Scene *, qaray::Node &, qaray::DiffRay &,
                                    qaray::DiffHitInfo &);
bool QA_CAT(__real_, QA_SYM_SHADOW)(qaray::Scene *, qaray::Node &, qaray::Ray &, qaray::HitInfo &);
bool QA_CAT(__wrap_, QA_SYM_NORMAL)(qaray::Scene *s, qaray::Node &n, qaray::DiffRay &r,
                                    qaray::DiffHitInfo &h)
{
  ++tl_casts_normal;
  return QA_CAT(__real_, QA_SYM_NORMAL)(s, n, r, h);
}
bool QA_CAT(__wrap_, QA_SYM_SHADOW)(qaray::Scene *s, qaray::Node &n, qaray::Ray &r,
                                    qaray::HitInfo &h)
{
  ++tl_casts_shadow;
  return QA_CAT(__real_, QA_SYM_SHADOW)(s, n, r, h);
}
}

// ---------------------------------------------------------------------------------------------
struct Options {
  const char *sceneFile = nullptr;
  std::string out = "ref_out";
  bool eightBit = false, useSRGB = true;   // --eight-bit <srgb 0|1>: also dump the reference's 8-bit products
  int width = -1, height = -1;
  int crop[4] = {0, 0, -1, -1};
  int sppMin = 1, sppMax = 1;
  int bounce = 5;
  int threads = 1;
  bool photonMap = false;
  qa_photon_params pm = {{10000, 20, 0.2f}, {1000, 20, 1.0f}};   // RendererParam defaults, renderer.h:51-57
  const char *dumpScene = nullptr;
  bool render = true;
};

class HarnessRenderer : public qaray::Renderer {
 public:
  explicit HarnessRenderer(RendererParam &p) : qaray::Renderer(p) {}
  void Render() override {}

  // Fill one of the two maps.  The bookkeeping (which hits are stored, when the map is full, what
  // counts as an emitted ray) is that of renderer.cpp:146-197 / :217-271; everything numeric is a
  // call into the reference.
  void FillMap(qaray::PhotonMap &pm, bool caustics, const qa_photon_map_params &mp, uint64_t &emitted, uint64_t &emissions)
  {
    std::vector<Light *> sources;
    for (auto l : scene->lights) if (l->IsPhotonSource()) sources.push_back(l);
    if (sources.empty()) { fprintf(stderr, "photon map: the scene has no photon source\n"); exit(3); }
    const qaFLOAT lightScale = 1.f / static_cast<qaFLOAT>(sources.size());
    pm.size = mp.size; pm.radius = mp.radius; pm.bounce = mp.bounce;
    pm.map.CreateAllPhotons(mp.size);
    tl_stream = caustics ? QA_STREAM_CAUSTICS : QA_STREAM_PHOTON;
    size_t stored = 0;
    emitted = 0;
    bool full = false;
    uint32_t e = 0;
    for (; !full; ++e) {
      if (e >= QA_PHOTON_MAX_EMISSIONS(mp.size)) { fprintf(stderr, "photon map: not full after %u emissions\n", e); exit(4); }
      tl_pixel = e;
      qaray::rng->local() = qaray::Sampler_Marsaglia();
      Light *light = sources[0];
      if (sources.size() > 1) {
        qaFLOAT r;
        qaray::rng->local().Get1f(r);
        size_t id = caustics ? MIN(static_cast<size_t>(CEIL(r * sources.size())), sources.size() - 1)
                             : MIN(FLOOR(r * sources.size()), sources.size() - 1);
        light = sources[id];
      }
      DiffRay ray = light->RandomPhoton();
      ray.Normalize();
      DiffHitInfo hit;
      hit.Init();
      Color3f power = light->GetPhotonIntensity(ray.c.dir) * lightScale;
      bool any = false;
      for (size_t bounce = 0; bounce < mp.bounce;) {
        if (!scene->TraceNodeNormal(scene->rootNode, ray, hit)) break;
        const Material *mtl = hit.c.node->GetMaterial();
        if (!mtl) break;
        const bool surface = mtl->IsPhotonSurface(0);
        if (surface && bounce != 0 && !(caustics && hit.c.hasDiffuseHit)) {
          if (stored >= mp.size) { full = true; break; }
          pm.map[stored].position = hit.c.p;
          pm.map[stored].SetDirection(ray.c.dir);
          pm.map[stored].SetPower(power);
          ++stored;
          any = true;
        }
        if (!mtl->RandomPhotonBounce(ray, power, hit)) break;
        const bool wasDiffuse = hit.c.hasDiffuseHit;
        ++bounce;
        ray.Normalize();
        hit.Init();
        if (caustics) hit.c.hasDiffuseHit = (wasDiffuse || surface);
      }
      if (any) ++emitted;
    }
    emissions = e;
    pm.map.ScalePhotonPowers(1.f / static_cast<qaUINT>(emitted));
    pm.map.PrepareForIrradianceEstimation();
    tl_stream = 0;
  }
  void BuildPhotonMaps(const qa_photon_params &pp, uint64_t emitted[2], uint64_t emissions[2])
  {
    FillMap(scene->photonmap, false, pp.photon, emitted[0], emissions[0]);
    FillMap(scene->causticsmap, true, pp.caustics, emitted[1], emissions[1]);
    scene->usePhotonMap = true;
  }

  // One pixel: the sample loop of renderer.cpp:302-346 with float outputs.
  void Pixel(int i, int j, float *rgb, float *depthOut, uint32_t *nsOut)
  {
    tl_stream = 0;
    tl_pixel = (uint32_t) j * (uint32_t) pixelW + (uint32_t) i;
    qaray::rng->local() = qaray::Sampler_Marsaglia();  // fresh, un-initialised stream
    SuperSamplerHalton pix(Color3f(0.005f, 0.001f, 0.005f), (int) param.sppMin, (int) param.sppMax);
    float zFirst = 0.f;
    for (; pix.Loop(); pix.Increment()) {
      const Point3 jit = pix.NewPixelSample() + Point3(i, j, 0.f);
      const Point3 onC = screenA + jit.x * screenU + jit.y * screenV;
      const Point3 onX = screenA + (jit.x + DiffRay::dx) * screenU + jit.y * screenV;
      const Point3 onY = screenA + jit.x * screenU + (jit.y + DiffRay::dy) * screenV;
      Point3 eye = scene->camera.pos;
      if (dof > 0.1f) {
        const Point3 lens = pix.NewDofSample(dof);
        eye += lens.x * screenX + lens.y * screenY;
      }
      DiffRay cam(eye, onC - eye, eye, onX - eye, eye, onY - eye);
      cam.Normalize();
      DiffHitInfo hit;
      hit.c.z = BIGFLOAT;
      Color3f L;
      const bool found = scene->TraceNodeNormal(scene->rootNode, cam, hit);
      if (found) {
        L = hit.c.node->GetMaterial()->Shade(cam, hit, scene->lights, Material::maxBounce);
      } else {
        L = scene->background.Sample(Point3(jit.x / pixelW, jit.y / pixelH, 0.f));
      }
      if (pix.GetSampleID() == 0) zFirst = found ? hit.c.z : BIGFLOAT;
      pix.Accumulate(L);
    }
    const Color3f mean = pix.GetColor();
    rgb[0] = mean.r; rgb[1] = mean.g; rgb[2] = mean.b;
    *depthOut = zFirst;
    *nsOut = (uint32_t) pix.GetSampleID();
  }
  size_t W() const { return pixelW; }
  size_t H() const { return pixelH; }
  void CameraFrame(float out[18]) const
  {
    const Point3 *v[6] = {&screenA, &screenU, &screenV, &screenX, &screenY, &screenZ};
    for (int k = 0; k < 6; ++k) { out[3 * k] = v[k]->x; out[3 * k + 1] = v[k]->y; out[3 * k + 2] = v[k]->z; }
  }
};

// ---------------------------------------------------------------------------------------------
// Scene dump (hex floats => exact round trip)
// ---------------------------------------------------------------------------------------------
static void jf(FILE *f, float v) { uint32_t u; memcpy(&u, &v, 4); fprintf(f, "%u", u); }
static void jv3(FILE *f, const Point3 &p) { fputc('[', f); jf(f, p.x); fputc(',', f); jf(f, p.y); fputc(',', f); jf(f, p.z); fputc(']', f); }
static void jm3(FILE *f, const Matrix3 &m)
{
  // column-major, 9 floats as bit patterns
  fputc('[', f);
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { if (c || r) fputc(',', f); jf(f, m[c][r]); }
  fputc(']', f);
}
static std::vector<const TriObj *> g_meshes;
static int MeshIndex(const TriObj *t)
{
  for (size_t k = 0; k < g_meshes.size(); ++k) if (g_meshes[k] == t) return (int) k;
  g_meshes.push_back(t);
  return (int) g_meshes.size() - 1;
}
static void DumpNode(FILE *f, const qaray::Node *n)
{
  fprintf(f, "{\"name\":\"%s\",", n->GetName() ? n->GetName() : "");
  fprintf(f, "\"tm\":"); jm3(f, n->GetTransform());
  fprintf(f, ",\"itm\":"); jm3(f, n->GetInverseTransform());
  fprintf(f, ",\"pos\":"); jv3(f, n->GetPosition());
  const qaray::Object *o = n->GetNodeObj();
  const char *type = "none";
  int mesh = -1;
  if (o == &theSphere) type = "sphere";
  else if (o == &thePlane) type = "plane";
  else if (o) { type = "obj"; mesh = MeshIndex(static_cast<const TriObj *>(o)); }
  fprintf(f, ",\"type\":\"%s\",\"mesh\":%d", type, mesh);
  const qaray::Material *m = n->GetMaterial();
  fprintf(f, ",\"material\":\"%s\"", (m && m->GetName()) ? m->GetName() : "");
  fprintf(f, ",\"children\":[");
  for (int c = 0; c < n->GetNumChild(); ++c) { if (c) fputc(',', f); DumpNode(f, n->GetChild(c)); }
  fprintf(f, "]}");
}
static void DumpScene(const char *path, const HarnessRenderer &r)
{
  FILE *f = fopen(path, "w");
  if (!f) { perror(path); exit(2); }
  fprintf(f, "{\"float_encoding\":\"u32 bit pattern\",\n\"root\":");
  DumpNode(f, &qaray::scene.rootNode);
  float cam[18]; r.CameraFrame(cam);
  fprintf(f, ",\n\"camera_frame\":[");
  for (int k = 0; k < 18; ++k) { if (k) fputc(',', f); jf(f, cam[k]); }
  fprintf(f, "],\n\"width\":%zu,\"height\":%zu,\n\"meshes\":[", r.W(), r.H());
  for (size_t k = 0; k < g_meshes.size(); ++k) {
    const TriObj *t = g_meshes[k];
    if (k) fputc(',', f);
    fprintf(f, "{\"nv\":%zu,\"nvn\":%zu,\"nvt\":%zu,\"nf\":%zu,\"nm\":%zu,", t->NV(), t->NVN(), t->NVT(), t->NF(), t->NM());
    fprintf(f, "\"bmin\":"); jv3(f, t->GetBoundMin());
    fprintf(f, ",\"bmax\":"); jv3(f, t->GetBoundMax());
    fprintf(f, ",\"v\":[");
    for (size_t i = 0; i < t->NV(); ++i) { if (i) fputc(',', f); jv3(f, t->V((int) i)); }
    fprintf(f, "],\"vn\":[");
    for (size_t i = 0; i < t->NVN(); ++i) { if (i) fputc(',', f); jv3(f, t->VN((int) i)); }
    fprintf(f, "],\"vt\":[");
    for (size_t i = 0; i < t->NVT(); ++i) { if (i) fputc(',', f); fputc('[', f); jf(f, t->VT((int) i).x); fputc(',', f); jf(f, t->VT((int) i).y); fputc(']', f); }
    fprintf(f, "],\"faces\":[");
    for (size_t i = 0; i < t->NF(); ++i) {
      const auto &fc = t->F(i);
      if (i) fputc(',', f);
      fprintf(f, "[%d,%d,%d,%d,%d,%d,%d,%d,%d,%d]", fc.v[0]->vertex_index, fc.v[1]->vertex_index,
              fc.v[2]->vertex_index, fc.v[0]->normal_index, fc.v[1]->normal_index,
              fc.v[2]->normal_index, fc.v[0]->texcoord_index, fc.v[1]->texcoord_index,
              fc.v[2]->texcoord_index, fc.mtl);
    }
    // BVH: walk from the root to find the node count (children are allocated densely, 1-based)
    unsigned maxNode = 1;
    {
      std::vector<unsigned> st{t->bvh.GetRootNodeID()};
      while (!st.empty()) {
        unsigned id = st.back(); st.pop_back();
        if (id > maxNode) maxNode = id;
        if (!t->bvh.IsLeafNode(id)) { unsigned a, b; t->bvh.GetChildNodes(id, a, b); st.push_back(a); st.push_back(b); }
      }
    }
    fprintf(f, "],\"bvh_nodes\":[");
    for (unsigned id = 1; id <= maxNode; ++id) {
      const float *b = t->bvh.GetNodeBounds(id);
      if (id > 1) fputc(',', f);
      fputc('[', f);
      for (int q = 0; q < 6; ++q) { jf(f, b[q]); fputc(',', f); }
      if (t->bvh.IsLeafNode(id)) {
        const unsigned *el = t->bvh.GetNodeElements(id);
        const unsigned base = (unsigned) (el - t->bvh.GetNodeElements(t->bvh.GetRootNodeID()) );
        (void) base;
        fprintf(f, "1,%u", t->bvh.GetNodeElementCount(id));
        for (unsigned q = 0; q < t->bvh.GetNodeElementCount(id); ++q) fprintf(f, ",%u", el[q]);
      } else {
        fprintf(f, "0,%u", t->bvh.GetFirstChildNode(id));
      }
      fputc(']', f);
    }
    fprintf(f, "]}");
  }
  fprintf(f, "],\n\"num_lights\":%zu,\"num_materials\":%zu}\n", qaray::scene.lights.size(), qaray::scene.materials.size());
  fclose(f);
}

// ---------------------------------------------------------------------------------------------
static void Usage()
{
  fprintf(stderr,
          "usage: ref_harness scene.xml [--size W H] [--crop x0 y0 x1 y1] [--spp N | --spp-min A --spp-max B]\n"
          "                   [--bounce B] [--seed S] [--threads T] [--out prefix] [--dump-scene file.json] [--no-render] [--eight-bit srgb]\n"
          "                   [--photon-map N_PHOTON N_CAUSTICS] [--photon-bounce B B] [--photon-radius R R]\n"
          "       (run with the scene's asset root as the working directory)\n");
}

int main(int argc, char **argv)
{
  Options o;
  for (int a = 1; a < argc; ++a) {
    std::string s(argv[a]);
    auto need = [&](int n) { if (a + n >= argc) { Usage(); exit(2); } };
    if (s == "--size") { need(2); o.width = atoi(argv[++a]); o.height = atoi(argv[++a]); }
    else if (s == "--crop") { need(4); for (int k = 0; k < 4; ++k) o.crop[k] = atoi(argv[++a]); }
    else if (s == "--spp") { need(1); o.sppMin = o.sppMax = atoi(argv[++a]); }
    else if (s == "--spp-min") { need(1); o.sppMin = atoi(argv[++a]); }
    else if (s == "--spp-max") { need(1); o.sppMax = atoi(argv[++a]); }
    else if (s == "--bounce") { need(1); o.bounce = atoi(argv[++a]); }
    else if (s == "--seed") { need(1); g_seed = (uint32_t) strtoul(argv[++a], nullptr, 0); }
    else if (s == "--threads") { need(1); o.threads = atoi(argv[++a]); }
    else if (s == "--out") { need(1); o.out = argv[++a]; }
    else if (s == "--eight-bit") { need(1); o.eightBit = true; o.useSRGB = atoi(argv[++a]) != 0; }
    else if (s == "--photon-map") { need(2); o.photonMap = true; o.pm.photon.size = (uint32_t) atoi(argv[++a]); o.pm.caustics.size = (uint32_t) atoi(argv[++a]); }
    else if (s == "--photon-bounce") { need(2); o.pm.photon.bounce = (uint32_t) atoi(argv[++a]); o.pm.caustics.bounce = (uint32_t) atoi(argv[++a]); }
    else if (s == "--photon-radius") { need(2); o.pm.photon.radius = (float) atof(argv[++a]); o.pm.caustics.radius = (float) atof(argv[++a]); }
    else if (s == "--dump-scene") { need(1); o.dumpScene = argv[++a]; }
    else if (s == "--no-render") { o.render = false; }
    else if (s[0] == '-') { Usage(); return 2; }
    else o.sceneFile = argv[a];
  }
  if (!o.sceneFile) { Usage(); return 2; }

  RendererParam param;
  param.SetSPPMin(o.sppMin);
  param.SetSPPMax(o.sppMax);
  param.SetSRGBFlag(false);
  Material::maxBounce = o.bounce;
  HarnessRenderer R(param);
  LoadSceneInSilentMode(true);
  if (!LoadScene(o.sceneFile)) { fprintf(stderr, "cannot load %s\n", o.sceneFile); return 1; }
  if (o.width > 0) { qaray::scene.camera.imgWidth = o.width; qaray::scene.camera.imgHeight = o.height; }
  R.ComputeScene(qaray::renderImage, qaray::scene);
  const int W = (int) R.W(), H = (int) R.H();
  if (o.crop[2] < 0) { o.crop[0] = 0; o.crop[1] = 0; o.crop[2] = W; o.crop[3] = H; }
  const int cw = o.crop[2] - o.crop[0], ch = o.crop[3] - o.crop[1];
  if (cw <= 0 || ch <= 0 || o.crop[0] < 0 || o.crop[1] < 0 || o.crop[2] > W || o.crop[3] > H) {
    fprintf(stderr, "bad crop\n"); return 2;
  }
  if (o.dumpScene) DumpScene(o.dumpScene, R);
  uint64_t pmEmitted[2] = {0, 0}, pmEmissions[2] = {0, 0};
  if (o.photonMap) {
    static_assert(sizeof(qa_photon) == sizeof(cyPhotonMap::Photon), "qa_photon must mirror cy::PhotonMap::Photon");
    R.BuildPhotonMaps(o.pm, pmEmitted, pmEmissions);
    const struct { const char *ext; qaray::PhotonMap *m; } maps[2] = {{".photonmap.bin", &qaray::scene.photonmap},
                                                                    {".caustics.bin", &qaray::scene.causticsmap}};
    for (auto &mm : maps) {
      FILE *pf = fopen((o.out + mm.ext).c_str(), "wb");
      if (!pf) { perror(mm.ext); return 2; }
      fwrite(mm.m->map.GetPhotons(), sizeof(cyPhotonMap::Photon), mm.m->map.NumPhotons(), pf);
      fclose(pf);
    }
    printf("ref_harness: photon map %u photons from %llu emitted rays (%llu emissions), caustics map %u from %llu (%llu)\n",
           o.pm.photon.size, (unsigned long long) pmEmitted[0], (unsigned long long) pmEmissions[0], o.pm.caustics.size,
           (unsigned long long) pmEmitted[1], (unsigned long long) pmEmissions[1]);
  }
  if (!o.render) return 0;

  const size_t maxThreads = qaray::tasking::get_num_of_threads();
  if (o.threads < 1) o.threads = 1;
  if ((size_t) o.threads > maxThreads) o.threads = (int) maxThreads;
  omp_set_num_threads(o.threads);

  std::vector<float> rgb((size_t) cw * ch * 3), depth((size_t) cw * ch);
  std::vector<uint32_t> ns((size_t) cw * ch);
  std::atomic<uint64_t> castsN(0), castsS(0), samples(0);
  const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel
  {
    tl_casts_normal = tl_casts_shadow = 0;
    uint64_t mySamples = 0;
#pragma omp for schedule(dynamic, 16)
    for (int q = 0; q < cw * ch; ++q) {
      const int i = o.crop[0] + q % cw, j = o.crop[1] + q / cw;
      R.Pixel(i, j, &rgb[3 * (size_t) q], &depth[q], &ns[q]);
      mySamples += ns[q];
    }
    castsN += tl_casts_normal; castsS += tl_casts_shadow; samples += mySamples;
  }
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  auto dump = [&](const std::string &name, const void *p, size_t bytes) {
    FILE *f = fopen(name.c_str(), "wb");
    if (!f) { perror(name.c_str()); exit(2); }
    fwrite(p, 1, bytes, f); fclose(f);
  };
  if (o.eightBit) {
    // The reference's own 8-bit products of this frame: the tail of Renderer::PixelRender (src/renderers/renderer.cpp:
    // 347-365) with the reference's LinearToSRGB and MAX / MIN macros, written into the reference's FrameBuffer, whose
    // ComputeZBufferImage / ComputeSampleCountImage (src/fb/framebuffer.cpp:62-107) then run unchanged.
    ::FrameBuffer fb;
    fb.Init((qaUINT) cw, (qaUINT) ch);
    for (int q = 0; q < cw * ch; ++q) {
      Color3f color(rgb[3 * (size_t) q], rgb[3 * (size_t) q + 1], rgb[3 * (size_t) q + 2]);
      if (o.useSRGB) {
        color.r = qaray::LinearToSRGB(color.r);
        color.g = qaray::LinearToSRGB(color.g);
        color.b = qaray::LinearToSRGB(color.b);
      }
      color.r = MAX(0.f, MIN(1.f, color.r));
      color.g = MAX(0.f, MIN(1.f, color.g));
      color.b = MAX(0.f, MIN(1.f, color.b));
      fb.GetPixels()[q].r = static_cast<qaUCHAR>(roundf(color.r * 255.f));
      fb.GetPixels()[q].g = static_cast<qaUCHAR>(roundf(color.g * 255.f));
      fb.GetPixels()[q].b = static_cast<qaUCHAR>(roundf(color.b * 255.f));
      fb.GetZBuffer()[q] = depth[q];
      fb.GetSampleCount()[q] = static_cast<qaUCHAR>(255.f * ns[q] / static_cast<qaFLOAT>(o.sppMax));
    }
    fb.ComputeZBufferImage();
    fb.ComputeSampleCountImage();
    dump(o.out + ".color.u8", fb.GetPixels(), (size_t) cw * ch * 3);
    dump(o.out + ".zimg.u8", fb.GetZBufferImage(), (size_t) cw * ch);
    dump(o.out + ".count.u8", fb.GetSampleCount(), (size_t) cw * ch);
    dump(o.out + ".countimg.u8", fb.GetSampleCountImage(), (size_t) cw * ch);
  }
  dump(o.out + ".rgb.f32", rgb.data(), rgb.size() * 4);
  dump(o.out + ".depth.f32", depth.data(), depth.size() * 4);
  dump(o.out + ".ns.u32", ns.data(), ns.size() * 4);
  FILE *f = fopen((o.out + ".json").c_str(), "w");
  fprintf(f,
          "{\"producer\":\"oracle/ref_harness (reference code, float outputs)\",\"scene\":\"%s\","
          "\"width\":%d,\"height\":%d,\"crop\":[%d,%d,%d,%d],\"spp_min\":%d,\"spp_max\":%d,"
          "\"bounce\":%d,\"seed\":%u,\"threads\":%d,\"seconds\":%.6f,\"samples\":%llu,"
          "\"casts_normal\":%llu,\"casts_shadow\":%llu,\"msamples_per_s\":%.6f,"
          "\"photon_map\":%d,\"photon_emitted\":[%llu,%llu],\"photon_emissions\":[%llu,%llu]}\n",
          o.sceneFile, W, H, o.crop[0], o.crop[1], o.crop[2], o.crop[3], o.sppMin, o.sppMax,
          o.bounce, g_seed, o.threads, sec, (unsigned long long) samples.load(),
          (unsigned long long) castsN.load(), (unsigned long long) castsS.load(),
          samples.load() / sec * 1e-6, o.photonMap ? 1 : 0, (unsigned long long) pmEmitted[0],
          (unsigned long long) pmEmitted[1], (unsigned long long) pmEmissions[0], (unsigned long long) pmEmissions[1]);
  fclose(f);
  printf("ref_harness: %dx%d crop %dx%d spp %d..%d threads %d: %.3f s, %.4f Msamples/s, casts/sample %.3f normal + %.3f shadow\n",
         W, H, cw, ch, o.sppMin, o.sppMax, o.threads, sec, samples.load() / sec * 1e-6,
         (double) castsN.load() / samples.load(), (double) castsS.load() / samples.load());
  return 0;
}
