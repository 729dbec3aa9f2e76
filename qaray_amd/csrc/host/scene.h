// scene.h — host-side scene graph and plugin surface.
//
// Mirrors the extension surface of the reference: Node/Transformation, the
// Object / Material / Light / Texture plugin families, Camera and Scene, with the same names and
// setter vocabulary as the reference's classes so its XML loader logic carries over
// (src/core/{node,transform,object,material,light,texture,camera}.h, src/objects/objects.h,
// src/materials/MtlBlinn_PhotonMap.h, src/materials/materials.h, src/lights/lights.h,
// src/textures/texture.h, src/scene/scene.h).  Unlike the reference, plugins do not trace or
// shade on the host: every plugin type instead knows how to flatten itself into the POD tables
// of include/qa_flat_scene.h, which is what the HIP kernels consume.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "qa_flat_scene.h"
#include "qa_math.h"

namespace qaray_hip {

class FlatBuilder;

// ---------------------------------------------------------------------------------------------
class ItemBase {
 public:
  virtual ~ItemBase() = default;
  const char *GetName() const { return name_.c_str(); }
  void SetName(const char *n) { name_ = n ? n : ""; }
 private:
  std::string name_;
};

// src/core/transform.h:36-79
class Transformation {
 public:
  const Point3 &GetPosition() const { return pos; }
  const Mat3 &GetTransform() const { return tm; }
  const Mat3 &GetInverseTransform() const { return itm; }
  Point3 TransformTo(const Point3 &p) const { return itm * (p - pos); }
  Point3 TransformFrom(const Point3 &p) const { return tm * p + pos; }
  void Translate(Point3 p) { pos += p; }
  void Rotate(Point3 axis, float degree);
  void Scale(float sx, float sy, float sz);
  void Transform(const Mat3 &m);
  void InitTransform();
 private:
  Point3 pos;
  Mat3 tm, itm;
};

// src/core/box.h
struct Box {
  Point3 pmin{QA_BIGFLOAT, QA_BIGFLOAT, QA_BIGFLOAT}, pmax{-QA_BIGFLOAT, -QA_BIGFLOAT, -QA_BIGFLOAT};
  Box() = default;
  Box(float x0, float y0, float z0, float x1, float y1, float z1) : pmin(x0, y0, z0), pmax(x1, y1, z1) {}
  bool IsEmpty() const { return pmin.x > pmax.x || pmin.y > pmax.y || pmin.z > pmax.z; }
  Point3 Corner(int i) const { return {(i & 1) ? pmax.x : pmin.x, (i & 2) ? pmax.y : pmin.y, (i & 4) ? pmax.z : pmin.z}; }
  void operator+=(const Point3 &p);
  void operator+=(const Box &b);
};

// ---------------------------------------------------------------------------------------------
// Objects (src/core/object.h, src/objects/objects.h)
class Object {
 public:
  virtual ~Object() = default;
  virtual Box GetBoundBox() const = 0;
  virtual int FlatType() const = 0;  // QA_OBJ_*
};
class Sphere : public Object {
 public:
  Box GetBoundBox() const override { return Box(-1, -1, -1, 1, 1, 1); }
  int FlatType() const override { return QA_OBJ_SPHERE; }
};
class Plane : public Object {
 public:
  Box GetBoundBox() const override { return Box(-1, -1, 0, 1, 1, 0); }
  int FlatType() const override { return QA_OBJ_PLANE; }
};
extern Sphere theSphere;
extern Plane thePlane;

// What tinyobj::material_t carries that the reference reads (src/parser/xmlload.cpp:236-265)
struct ObjMaterial {
  std::string name;
  float diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0}, transmittance[3] = {0, 0, 0};
  float shininess = 1.f, ior = 1.f;
  int illum = 0;
  std::string diffuse_texname, specular_texname;
};

// src/mesh/TriMesh.h + src/mesh/TriBVH.h + src/objects/objects.h:57-93
class TriObj : public Object {
 public:
  Box GetBoundBox() const override;
  int FlatType() const override { return QA_OBJ_MESH; }
  // LoadFromFileObj + ComputeNormals (if the file has none) + ComputeBoundingBox + BVH (<=4/leaf)
  bool Load(const char *filename, bool loadMtl, std::string *err = nullptr);

  size_t NF() const { return faces.size(); }
  size_t NV() const { return vertices.size() / 3; }
  size_t NVN() const { return normals.size() / 3; }
  size_t NVT() const { return texcoords.size() / 2; }
  size_t NM() const { return materials.size(); }
  const ObjMaterial &M(size_t i) const { return materials[i]; }
  const std::string &GetDirectoryName() const { return path; }
  // directory of the OBJ relative to the scene's asset root (texture names in its .mtl are
  // resolved against it, xmlload.cpp:246-256)
  void SetRelDirectory(const std::string &objName)
  {
    const size_t s = objName.find_last_of("/\\");
    relDir = (s == std::string::npos) ? "" : objName.substr(0, s + 1);
  }
  const std::string &GetDirectoryNameRel() const { return relDir; }

  std::vector<float> vertices, normals, texcoords;
  std::vector<qa_face> faces;            // sorted by material like TriMesh.cpp:107-114
  std::vector<ObjMaterial> materials;
  std::vector<qa_bvh_node> bvhNodes;     // [0] unused, root = 1
  std::vector<uint32_t> bvhElements;
  Point3 boundMin{1, 1, 1}, boundMax{0, 0, 0};

 private:
  bool LoadFromFileObj(const char *filename, std::string *err);
  void ComputeNormals();
  void ComputeBoundingBox();
  void BuildBVH(unsigned maxElementsPerNode);
  std::string path, name, file, relDir;
};

// ---------------------------------------------------------------------------------------------
// Textures (src/core/texture.h, src/textures/texture.h)
class Texture : public ItemBase {
 public:
  virtual void Flatten(FlatBuilder &fb, qa_texture *out) const = 0;
};
class TextureFile : public Texture {
 public:
  bool Load();  // PNG or PPM by extension, name = file path
  void Flatten(FlatBuilder &fb, qa_texture *out) const override;
  int width = 0, height = 0;
  std::vector<unsigned char> data;  // RGB8
};
class TextureChecker : public Texture {
 public:
  void SetColor1(const Color3f &c) { color1 = c; }
  void SetColor2(const Color3f &c) { color2 = c; }
  void Flatten(FlatBuilder &fb, qa_texture *out) const override;
 private:
  Color3f color1{0, 0, 0}, color2{1, 1, 1};
};
class TextureMap : public Transformation {
 public:
  TextureMap() = default;
  explicit TextureMap(const Texture *t) : texture(t) {}
  const Texture *GetTexture() const { return texture; }
 private:
  const Texture *texture = nullptr;
};
class TexturedColor {
 public:
  TexturedColor() = default;
  TexturedColor(float r, float g, float b) : color(r, g, b) {}
  void SetColor(const Color3f &c) { color = c; }
  void SetTexture(TextureMap *m) { map.reset(m); }
  Color3f GetColor() const { return color; }
  const TextureMap *GetTexture() const { return map.get(); }
  qa_texcolor Flatten(FlatBuilder &fb) const;
 private:
  Color3f color{0, 0, 0};
  std::unique_ptr<TextureMap> map;
};

// ---------------------------------------------------------------------------------------------
// Materials (src/core/material.h, src/materials/MtlBlinn_PhotonMap.h, src/materials/materials.h)
class Material : public ItemBase {
 public:
  static int maxBounce;
  // Appends this material's record(s) to the flat material table, returns the qa_mtlset.
  virtual qa_mtlset Flatten(FlatBuilder &fb) const = 0;
};
class MtlBlinn : public Material {
 public:
  void SetDiffuse(Color3f c) { diffuse.SetColor(c); }
  void SetSpecular(Color3f c) { specular.SetColor(c); }
  void SetGlossiness(float g) { specularGlossiness = g; }
  void SetEmission(Color3f c) { emission.SetColor(c); }
  void SetReflection(Color3f c) { reflection.SetColor(c); }
  void SetRefraction(Color3f c) { refraction.SetColor(c); }
  void SetAbsorption(Color3f c) { absorption = c; }
  void SetRefractionIndex(float i) { ior = i; }
  void SetDiffuseTexture(TextureMap *m) { diffuse.SetTexture(m); }
  void SetSpecularTexture(TextureMap *m) { specular.SetTexture(m); }
  void SetEmissionTexture(TextureMap *m) { emission.SetTexture(m); }
  void SetReflectionTexture(TextureMap *m) { reflection.SetTexture(m); }
  void SetRefractionTexture(TextureMap *m) { refraction.SetTexture(m); }
  void SetReflectionGlossiness(float g) { reflectionGlossiness = g > 0.00001f ? g : -1.f; }
  void SetRefractionGlossiness(float g) { refractionGlossiness = g > 0.00001f ? g : -1.f; }
  qa_mtlset Flatten(FlatBuilder &fb) const override;
  qa_material FlattenRecord(FlatBuilder &fb) const;
 private:
  TexturedColor diffuse{0.5f, 0.5f, 0.5f}, specular{0.7f, 0.7f, 0.7f};
  TexturedColor reflection{0, 0, 0}, refraction{0, 0, 0}, emission{0, 0, 0};
  Color3f absorption{0, 0, 0};
  float kill = 0.1f, ior = 1.f;
  float specularGlossiness = 20.f, reflectionGlossiness = 0.f, refractionGlossiness = 0.f;
};
class MultiMtl : public Material {
 public:
  void AppendMaterial(MtlBlinn *m) { mtls.emplace_back(m); }
  qa_mtlset Flatten(FlatBuilder &fb) const override;
 private:
  std::vector<std::unique_ptr<MtlBlinn>> mtls;
};

// ---------------------------------------------------------------------------------------------
// Lights (src/core/light.h, src/lights/lights.h)
class Light : public ItemBase {
 public:
  virtual qa_light Flatten() const = 0;
  virtual bool IsAmbient() const { return false; }
};
class AmbientLight : public Light {
 public:
  void SetIntensity(Color3f c) { intensity = c; }
  bool IsAmbient() const override { return true; }
  qa_light Flatten() const override;
 private:
  Color3f intensity{0, 0, 0};
};
class DirectLight : public Light {
 public:
  void SetIntensity(Color3f c) { intensity = c; }
  void SetDirection(Point3 d) { direction = normalize(d); }
  qa_light Flatten() const override;
 private:
  Color3f intensity{0, 0, 0};
  Point3 direction{0, 0, 1};
};
class PointLight : public Light {
 public:
  void SetIntensity(Color3f c) { intensity = c; }
  void SetPosition(Point3 p) { position = p; }
  void SetSize(float s) { size = s; }
  qa_light Flatten() const override;
 private:
  Color3f intensity{0, 0, 0};
  Point3 position{0, 0, 0};
  float size = 0;
};
class SpotLight : public Light {
 public:
  SpotLight() { SetAngle(45); SetBlend(1.f); }
  void SetIntensity(Color3f c) { intensity = c; }
  void SetPosition(Point3 p) { position = p; }
  void SetRotation(float degree, Point3 axis);
  void SetSize(float s) { size = s; }
  void SetAngle(float s);
  void SetBlend(float s);
  qa_light Flatten() const override;
 private:
  Color3f intensity{0, 0, 0};
  Point3 position{0, 0, 0}, direction{1, 0, 0};
  float size = 0, blend = 1, inner = 0, outer = 0;
};

// ---------------------------------------------------------------------------------------------
// src/core/node.h
class Node : public ItemBase, public Transformation {
 public:
  int GetNumChild() const { return (int) child.size(); }
  const Node *GetChild(int i) const { return child[i].get(); }
  Node *GetChild(int i) { return child[i].get(); }
  void AppendChild(Node *n) { child.emplace_back(n); }
  void DeleteAllChildNodes() { child.clear(); }
  void Init() { DeleteAllChildNodes(); obj = nullptr; mtl = nullptr; SetName(nullptr); InitTransform(); }
  const Object *GetNodeObj() const { return obj; }
  void SetNodeObj(const Object *o) { obj = o; }
  const Material *GetMaterial() const { return mtl; }
  void SetMaterial(const Material *m) { mtl = m; }
  const Box &ComputeChildBoundBox();
  const Box &GetChildBoundBox() const { return childBoundBox; }
 private:
  std::vector<std::unique_ptr<Node>> child;
  const Object *obj = nullptr;
  const Material *mtl = nullptr;
  Box childBoundBox;
};

// src/core/camera.h
struct Camera {
  Point3 pos, dir, up;
  float fovy, focalDistance, depthOfField;
  int imgWidth, imgHeight;
  void Init();
};

// src/scene/scene.h:55-71 (photon maps are out of scope)
class Scene {
 public:
  Node rootNode;
  Camera camera;
  std::vector<std::unique_ptr<Material>> materials;
  std::vector<std::unique_ptr<Light>> lights;
  std::vector<std::pair<std::string, std::unique_ptr<TriObj>>> objList;
  std::vector<std::pair<std::string, std::unique_ptr<Texture>>> textureList;
  TexturedColor background, environment;
  std::string assetRoot;  // prefix for relative obj/texture paths ("" = working directory)

  Material *FindMaterial(const char *name) const;
  TriObj *FindObject(const char *name) const;
  Texture *FindTexture(const char *name) const;
  void Clear();
};

// src/parser/xmlload.h
void LoadSceneInSilentMode(bool flag);
int LoadScene(const char *filename, Scene &scene);

// Camera frame of Renderer::ComputeScene (src/renderers/renderer.cpp:71-99)
struct CameraFrame {
  Point3 screenA, screenU, screenV, screenX, screenY, screenZ;
  float dof, focal, screenW, screenH;
  int width, height;
};
CameraFrame ComputeCameraFrame(const Camera &cam);

// Flatten the whole scene into one relocatable blob (include/qa_flat_scene.h).
std::vector<unsigned char> FlattenScene(const Scene &scene);

}  // namespace qaray_hip
