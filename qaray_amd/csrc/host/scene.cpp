// scene.cpp — scene-graph plumbing and the scene flattener.  See scene.h.
#include "scene.h"

#include <cmath>
#include <cstring>
#include <map>

namespace qaray_hip {

static const float PI = static_cast<float>(M_PI);  // src/math/math.cpp:13

Sphere theSphere;
Plane thePlane;
int Material::maxBounce = 5;  // src/core/material.cpp:31

// ---------------------------------------------------------------------------------------------
// Transformation — src/core/transform.h:62-74, transform.cpp:36-48
void Transformation::Rotate(Point3 axis, float degree) { Transform(rotation(degree * PI / 180.0f, axis)); }
void Transformation::Scale(float sx, float sy, float sz) { Transform(scaling(sx, sy, sz)); }
void Transformation::Transform(const Mat3 &m)
{
  tm = m * tm;
  pos = m * pos;
  itm = inverse(tm);
}
void Transformation::InitTransform()
{
  pos = Point3(0, 0, 0);
  tm = Mat3(1.f);
  itm = Mat3(1.f);
}

// Box — src/core/box.cpp:70-86
void Box::operator+=(const Point3 &p)
{
  for (int i = 0; i < 3; i++) {
    if (pmin[i] > p[i]) pmin[i] = p[i];
    if (pmax[i] < p[i]) pmax[i] = p[i];
  }
}
void Box::operator+=(const Box &b)
{
  for (int i = 0; i < 3; i++) {
    if (pmin[i] > b.pmin[i]) pmin[i] = b.pmin[i];
    if (pmax[i] < b.pmax[i]) pmax[i] = b.pmax[i];
  }
}

// Node — src/core/node.cpp:85-100.  Computed like the reference does at load time; the tracing
// code never consults it, it is kept for host-side tools.
const Box &Node::ComputeChildBoundBox()
{
  childBoundBox = Box();
  for (auto &c : child) {
    Box cb = c->ComputeChildBoundBox();
    if (const Object *o = c->GetNodeObj()) cb += o->GetBoundBox();
    if (!cb.IsEmpty())
      for (int j = 0; j < 8; j++) childBoundBox += c->TransformFrom(cb.Corner(j));
  }
  return childBoundBox;
}

// Camera — src/core/camera.cpp:31-41
void Camera::Init()
{
  pos = Point3(0, 0, 0);
  dir = Point3(0, 0, -1);
  up = Point3(0, 1, 0);
  fovy = 40;
  focalDistance = 1;
  depthOfField = 0;
  imgWidth = 200;
  imgHeight = 150;
}

// Renderer::ComputeScene — src/renderers/renderer.cpp:76-93
CameraFrame ComputeCameraFrame(const Camera &cam)
{
  CameraFrame f;
  f.dof = cam.depthOfField;
  f.focal = cam.focalDistance;
  const float aspect = cam.imgWidth / static_cast<float>(cam.imgHeight);
  f.screenH = 2.f * f.focal * std::tan(cam.fovy * PI / 2.f / 180.f);
  f.screenW = aspect * f.screenH;
  const Point3 X = normalize(cross(cam.dir, cam.up));
  const Point3 Y = normalize(cross(X, cam.dir));
  const Point3 Z = normalize(-cam.dir);
  f.screenU = X * (f.screenW / cam.imgWidth);
  f.screenV = -Y * (f.screenH / cam.imgHeight);
  f.screenA = cam.pos - Z * f.focal + Y * f.screenH / 2.f - X * f.screenW / 2.f;
  f.screenX = X;
  f.screenY = Y;
  f.screenZ = Z;
  f.width = cam.imgWidth;
  f.height = cam.imgHeight;
  return f;
}

// Lights — src/lights/lights.cpp:110-127
void SpotLight::SetRotation(float degree, Point3 axis)
{
  // normalize(glm::rotate(mat4(1), radians(degree), axis) * vec4(0,0,-1,0)): column 2 negated.
  // mat4*vec4 evaluates m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3]*v.w with v = (0,0,-1,0).
  const Mat3 R = rotation(degree * (0.01745329251994329576923690768489f), axis);
  Point3 d;
  for (int i = 0; i < 3; ++i) {
    const float a = R.c[0][i] * 0.f + R.c[1][i] * 0.f;
    const float b = R.c[2][i] * -1.f + 0.f * 0.f;
    d[i] = a + b;
  }
  direction = normalize(d);
}
void SpotLight::SetAngle(float s)
{
  const float a = s / 2.f;
  const float lo = (a > 1.f) ? a : 1.f;          // MAX(s/2, 1)
  s = ((lo < 89.f) ? lo : 89.f) / 180.f * PI;    // MIN(.., 89)
  outer = std::tan(s);
}
void SpotLight::SetBlend(float s)
{
  const float lo = (s < 1.f) ? s : 1.f;          // MIN(s, 1)
  blend = (lo > 0.f) ? lo : 0.f;                 // MAX(.., 0)
  inner = std::sqrt(outer * outer * (1.f - blend));
}

static void put3(float *dst, const Vec3 &v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

qa_light AmbientLight::Flatten() const
{
  qa_light l;
  memset(&l, 0, sizeof(l));
  l.type = QA_LIGHT_AMBIENT;
  put3(l.intensity, intensity);
  return l;
}
qa_light DirectLight::Flatten() const
{
  qa_light l;
  memset(&l, 0, sizeof(l));
  l.type = QA_LIGHT_DIRECT;
  put3(l.intensity, intensity);
  put3(l.direction, direction);
  return l;
}
qa_light PointLight::Flatten() const
{
  qa_light l;
  memset(&l, 0, sizeof(l));
  l.type = QA_LIGHT_POINT;
  put3(l.intensity, intensity);
  put3(l.position, position);
  l.size = size;
  return l;
}
qa_light SpotLight::Flatten() const
{
  qa_light l;
  memset(&l, 0, sizeof(l));
  l.type = QA_LIGHT_SPOT;
  put3(l.intensity, intensity);
  put3(l.position, position);
  put3(l.direction, direction);
  l.size = size;
  l.inner = inner;
  l.outer = outer;
  return l;
}

// Scene lookups — src/core/material.cpp:32-40, src/core/items.h:64-72
Material *Scene::FindMaterial(const char *name) const
{
  for (auto &m : materials) if (m && strcmp(name, m->GetName()) == 0) return m.get();
  return nullptr;
}
TriObj *Scene::FindObject(const char *name) const
{
  for (auto &o : objList) if (o.first == name) return o.second.get();
  return nullptr;
}
Texture *Scene::FindTexture(const char *name) const
{
  for (auto &t : textureList) if (t.first == name) return t.second.get();
  return nullptr;
}
void Scene::Clear()
{
  rootNode.Init();
  materials.clear();
  lights.clear();
  objList.clear();
  textureList.clear();
  background = TexturedColor();
  environment = TexturedColor();
}

// ---------------------------------------------------------------------------------------------
// Flattener
// ---------------------------------------------------------------------------------------------
class FlatBuilder {
 public:
  std::vector<qa_instance> instances;
  std::vector<qa_mesh> meshes;
  std::vector<qa_mtlset> mtlsets;
  std::vector<qa_material> materials;
  std::vector<qa_light> lights;
  std::vector<qa_texmap> texmaps;
  std::vector<qa_texture> textures;
  std::vector<unsigned char> payload;  // variable-size arrays; offsets patched at Serialize()

  std::map<const TriObj *, int> meshIndex;
  std::map<const Material *, int> mtlsetIndex;
  std::map<const Texture *, int> textureIndex;

  uint64_t AppendPayload(const void *p, size_t bytes)
  {
    while (payload.size() % 16) payload.push_back(0);
    const uint64_t off = payload.size();
    const unsigned char *b = static_cast<const unsigned char *>(p);
    payload.insert(payload.end(), b, b + bytes);
    return off;  // relative to the payload start
  }
  int AddTexture(const Texture *t)
  {
    if (!t) return -1;
    auto it = textureIndex.find(t);
    if (it != textureIndex.end()) return it->second;
    qa_texture rec;
    memset(&rec, 0, sizeof(rec));
    t->Flatten(*this, &rec);
    textures.push_back(rec);
    return textureIndex[t] = (int) textures.size() - 1;
  }
  int AddMesh(const TriObj *t)
  {
    auto it = meshIndex.find(t);
    if (it != meshIndex.end()) return it->second;
    qa_mesh m;
    memset(&m, 0, sizeof(m));
    put3(m.bmin, t->boundMin);
    put3(m.bmax, t->boundMax);
    m.num_faces = (uint32_t) t->NF();
    m.num_vertices = (uint32_t) t->NV();
    m.num_normals = (uint32_t) t->NVN();
    m.num_texcoords = (uint32_t) t->NVT();
    m.num_bvh_nodes = (uint32_t) t->bvhNodes.size();
    m.off_bvh_nodes = AppendPayload(t->bvhNodes.data(), t->bvhNodes.size() * sizeof(qa_bvh_node));
    m.off_elements = AppendPayload(t->bvhElements.data(), t->bvhElements.size() * sizeof(uint32_t));
    m.off_faces = AppendPayload(t->faces.data(), t->faces.size() * sizeof(qa_face));
    m.off_vertices = AppendPayload(t->vertices.data(), t->vertices.size() * sizeof(float));
    m.off_normals = AppendPayload(t->normals.data(), t->normals.size() * sizeof(float));
    m.off_texcoords = AppendPayload(t->texcoords.data(), t->texcoords.size() * sizeof(float));
    meshes.push_back(m);
    return meshIndex[t] = (int) meshes.size() - 1;
  }
  int AddMaterial(const Material *m)
  {
    if (!m) return -1;
    auto it = mtlsetIndex.find(m);
    if (it != mtlsetIndex.end()) return it->second;
    const qa_mtlset s = m->Flatten(*this);
    mtlsets.push_back(s);
    return mtlsetIndex[m] = (int) mtlsets.size() - 1;
  }
  void AddNode(const Node *n, int parent, int depth)
  {
    const int me = (int) instances.size();
    qa_instance in;
    memset(&in, 0, sizeof(in));
    memcpy(in.tm, n->GetTransform().data(), sizeof(in.tm));
    memcpy(in.itm, n->GetInverseTransform().data(), sizeof(in.itm));
    put3(in.pos, n->GetPosition());
    const Object *o = n->GetNodeObj();
    in.obj_type = o ? o->FlatType() : QA_OBJ_NONE;
    in.mesh = (o && o->FlatType() == QA_OBJ_MESH) ? AddMesh(static_cast<const TriObj *>(o)) : -1;
    in.mtlset = AddMaterial(n->GetMaterial());
    in.parent = parent;
    in.depth = depth;
    instances.push_back(in);
    for (int c = 0; c < n->GetNumChild(); ++c) AddNode(n->GetChild(c), me, depth + 1);
    instances[me].subtree_end = (int) instances.size();
  }
};

void TextureFile::Flatten(FlatBuilder &fb, qa_texture *out) const
{
  out->type = QA_TEX_FILE;
  out->width = width;
  out->height = height;
  out->off_texels = fb.AppendPayload(data.data(), data.size());
}
void TextureChecker::Flatten(FlatBuilder &, qa_texture *out) const
{
  out->type = QA_TEX_CHECKER;
  put3(out->color1, color1);
  put3(out->color2, color2);
}
qa_texcolor TexturedColor::Flatten(FlatBuilder &fb) const
{
  qa_texcolor tc;
  put3(tc.color, color);
  tc.texmap = -1;
  if (map) {
    qa_texmap m;
    memset(&m, 0, sizeof(m));
    memcpy(m.itm, map->GetInverseTransform().data(), sizeof(m.itm));
    put3(m.pos, map->GetPosition());
    m.texture = fb.AddTexture(map->GetTexture());
    fb.texmaps.push_back(m);
    tc.texmap = (int) fb.texmaps.size() - 1;
  }
  return tc;
}
qa_material MtlBlinn::FlattenRecord(FlatBuilder &fb) const
{
  qa_material m;
  memset(&m, 0, sizeof(m));
  m.diffuse = diffuse.Flatten(fb);
  m.specular = specular.Flatten(fb);
  m.reflection = reflection.Flatten(fb);
  m.refraction = refraction.Flatten(fb);
  m.emission = emission.Flatten(fb);
  put3(m.absorption, absorption);
  m.ior = ior;
  m.kill = kill;
  m.gloss_spec = specularGlossiness;
  m.gloss_refl = reflectionGlossiness;
  m.gloss_refr = refractionGlossiness;
  return m;
}
qa_mtlset MtlBlinn::Flatten(FlatBuilder &fb) const
{
  qa_mtlset s = {(int32_t) fb.materials.size(), 1, 0, 0};
  const qa_material rec = FlattenRecord(fb);
  fb.materials.push_back(rec);
  return s;
}
qa_mtlset MultiMtl::Flatten(FlatBuilder &fb) const
{
  std::vector<qa_material> recs;
  for (auto &m : mtls) recs.push_back(m->FlattenRecord(fb));
  qa_mtlset s = {(int32_t) fb.materials.size(), (int32_t) recs.size(), 1, 0};
  fb.materials.insert(fb.materials.end(), recs.begin(), recs.end());
  return s;
}

template <class T>
static uint64_t PlaceTable(std::vector<unsigned char> &blob, const std::vector<T> &v)
{
  while (blob.size() % 16) blob.push_back(0);
  const uint64_t off = blob.size();
  const unsigned char *b = reinterpret_cast<const unsigned char *>(v.data());
  blob.insert(blob.end(), b, b + v.size() * sizeof(T));
  return off;
}

std::vector<unsigned char> FlattenScene(const Scene &scene)
{
  FlatBuilder fb;
  fb.AddNode(&scene.rootNode, -1, 0);
  for (auto &l : scene.lights) fb.lights.push_back(l->Flatten());

  qa_flat_header h;
  memset(&h, 0, sizeof(h));
  h.magic = QA_FLAT_MAGIC;
  h.version = QA_FLAT_VERSION;
  const CameraFrame cf = ComputeCameraFrame(scene.camera);
  put3(h.screenA, cf.screenA);
  put3(h.screenU, cf.screenU);
  put3(h.screenV, cf.screenV);
  put3(h.screenX, cf.screenX);
  put3(h.screenY, cf.screenY);
  put3(h.cam_pos, scene.camera.pos);
  h.dof = cf.dof;
  h.width = (uint32_t) cf.width;
  h.height = (uint32_t) cf.height;
  h.background = scene.background.Flatten(fb);
  h.environment = scene.environment.Flatten(fb);

  std::vector<unsigned char> blob(sizeof(qa_flat_header), 0);
  h.num_instances = (uint32_t) fb.instances.size();
  h.num_meshes = (uint32_t) fb.meshes.size();
  h.num_mtlsets = (uint32_t) fb.mtlsets.size();
  h.num_materials = (uint32_t) fb.materials.size();
  h.num_lights = (uint32_t) fb.lights.size();
  h.num_texmaps = (uint32_t) fb.texmaps.size();
  h.num_textures = (uint32_t) fb.textures.size();
  // payload first, so table entries can be patched to absolute offsets before they are placed
  while (blob.size() % 16) blob.push_back(0);
  const uint64_t payloadBase = blob.size();
  blob.insert(blob.end(), fb.payload.begin(), fb.payload.end());
  for (auto &m : fb.meshes) {
    m.off_bvh_nodes += payloadBase;
    m.off_elements += payloadBase;
    m.off_faces += payloadBase;
    m.off_vertices += payloadBase;
    m.off_normals += payloadBase;
    m.off_texcoords += payloadBase;
  }
  for (auto &t : fb.textures) if (t.type == QA_TEX_FILE) t.off_texels += payloadBase;
  h.off_instances = PlaceTable(blob, fb.instances);
  h.off_meshes = PlaceTable(blob, fb.meshes);
  h.off_mtlsets = PlaceTable(blob, fb.mtlsets);
  h.off_materials = PlaceTable(blob, fb.materials);
  h.off_lights = PlaceTable(blob, fb.lights);
  h.off_texmaps = PlaceTable(blob, fb.texmaps);
  h.off_textures = PlaceTable(blob, fb.textures);
  while (blob.size() % 16) blob.push_back(0);
  h.total_bytes = blob.size();
  memcpy(blob.data(), &h, sizeof(h));
  return blob;
}

}  // namespace qaray_hip
