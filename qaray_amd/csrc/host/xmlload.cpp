// xmlload.cpp — qaray XML scene loader for the host scene graph.
//
// Accepts the reference's scene files unchanged and builds the same scene: tag and attribute
// vocabulary, defaults, the order in which transforms are applied (children first, then the
// node's own scale/rotate/translate list), OBJ multi-materials derived from the .mtl, and the
// late binding of material names all follow src/parser/xmlload.cpp:71-630 (each function below
// cites its counterpart).  Element names are compared case-insensitively like the reference's
// COMPARE macro (xmlload.cpp:25-28).
#include <strings.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "image.h"
#include "scene.h"
#include "xml.h"

namespace qaray_hip {

namespace {

bool silentmode = true;
#define QA_PRINTF(...) do { if (!silentmode) printf(__VA_ARGS__); } while (0)

inline bool Is(const XmlElement *e, const char *name) { return strcasecmp(e->Value().c_str(), name) == 0; }
inline bool Same(const char *a, const char *b) { return strcasecmp(a, b) == 0; }

// xmlload.cpp:559-569
void ReadFloat(const XmlElement *e, float &f, const char *name = "value")
{
  double d = (double) f;
  e->QueryDoubleAttribute(name, &d);
  f = (float) d;
}
// xmlload.cpp:523-538
void ReadVector(const XmlElement *e, Point3 &v)
{
  double x = (double) v.x, y = (double) v.y, z = (double) v.z;
  e->QueryDoubleAttribute("x", &x);
  e->QueryDoubleAttribute("y", &y);
  e->QueryDoubleAttribute("z", &z);
  v.x = (float) x; v.y = (float) y; v.z = (float) z;
  float f = 1;
  ReadFloat(e, f);
  v *= f;
}
// xmlload.cpp:541-556
void ReadColor(const XmlElement *e, Color3f &c)
{
  double r = (double) c.x, g = (double) c.y, b = (double) c.z;
  e->QueryDoubleAttribute("r", &r);
  e->QueryDoubleAttribute("g", &g);
  e->QueryDoubleAttribute("b", &b);
  c.x = (float) r; c.y = (float) g; c.z = (float) b;
  float f = 1;
  ReadFloat(e, f);
  c *= f;
}

struct Loader {
  Scene &scene;
  struct NodeMtl { Node *node; std::string mtlName; };
  std::vector<NodeMtl> nodeMtlList;

  explicit Loader(Scene &s) : scene(s) {}

  std::string AssetPath(const std::string &rel) const { return scene.assetRoot + rel; }

  // xmlload.cpp:291-320
  void LoadTransform(Transformation *trans, const XmlElement *element)
  {
    for (const XmlElement *c = element->FirstChildElement(); c; c = c->NextSiblingElement()) {
      if (Is(c, "scale")) {
        Point3 s(1, 1, 1);
        ReadVector(c, s);
        trans->Scale(s.x, s.y, s.z);
      } else if (Is(c, "rotate")) {
        Point3 s(0, 0, 0);
        ReadVector(c, s);
        s = normalize(s);
        float a = 0;  // the reference leaves `a` uninitialised when the attribute is missing
        ReadFloat(c, a, "angle");
        trans->Rotate(s, a);
      } else if (Is(c, "translate")) {
        Point3 t(0, 0, 0);
        ReadVector(c, t);
        trans->Translate(t);
      }
    }
  }

  // xmlload.cpp:607-630
  Texture *ReadTextureFile(const char *texName)
  {
    Texture *tex = scene.FindTexture(texName);
    if (!tex) {
      std::unique_ptr<TextureFile> ft(new TextureFile);
      ft->SetName(AssetPath(texName).c_str());
      if (!ft->Load()) {
        QA_PRINTF(" -- Error loading texture file \"%s\"!\n", texName);
        return nullptr;
      }
      tex = ft.get();
      scene.textureList.emplace_back(texName, std::move(ft));
    }
    return tex;
  }

  // xmlload.cpp:573-604
  TextureMap *ReadTexture(const XmlElement *element)
  {
    const char *texName = element->Attribute("texture");
    if (!texName) return nullptr;
    Texture *tex = nullptr;
    if (Same(texName, "checkerboard")) {
      std::unique_ptr<TextureChecker> ct(new TextureChecker);
      for (const XmlElement *c = element->FirstChildElement(); c; c = c->NextSiblingElement()) {
        if (Is(c, "color1")) { Color3f col(0, 0, 0); ReadColor(c, col); ct->SetColor1(col); }
        else if (Is(c, "color2")) { Color3f col(0, 0, 0); ReadColor(c, col); ct->SetColor2(col); }
      }
      tex = ct.get();
      scene.textureList.emplace_back(texName, std::move(ct));
    } else {
      tex = ReadTextureFile(texName);
    }
    TextureMap *map = new TextureMap(tex);
    LoadTransform(map, element);
    return map;
  }

  // xmlload.cpp:232-272: one MtlBlinn per .mtl entry, gathered into a MultiMtl named after the file
  MultiMtl *MakeMultiMtl(const TriObj *tobj)
  {
    MultiMtl *mm = new MultiMtl;
    for (size_t i = 0; i < tobj->NM(); ++i) {
      MtlBlinn *m = new MtlBlinn;
      const ObjMaterial &mtl = tobj->M(i);
      m->SetDiffuse(Color3f(mtl.diffuse[0], mtl.diffuse[1], mtl.diffuse[2]));
      m->SetSpecular(Color3f(mtl.specular[0], mtl.specular[1], mtl.specular[2]));
      m->SetGlossiness(mtl.shininess);
      m->SetRefractionIndex(mtl.ior);
      if (!mtl.diffuse_texname.empty())
        m->SetDiffuseTexture(new TextureMap(ReadTextureFile((tobj->GetDirectoryNameRel() + mtl.diffuse_texname).c_str())));
      if (!mtl.specular_texname.empty())  // the reference puts the specular map into the diffuse slot (xmlload.cpp:249-252)
        m->SetDiffuseTexture(new TextureMap(ReadTextureFile((tobj->GetDirectoryNameRel() + mtl.specular_texname).c_str())));
      if (mtl.illum > 2 && mtl.illum <= 7) {
        m->SetReflection(Color3f(mtl.specular[0], mtl.specular[1], mtl.specular[2]));
        if (!mtl.specular_texname.empty())
          m->SetReflectionTexture(new TextureMap(ReadTextureFile((tobj->GetDirectoryNameRel() + mtl.specular_texname).c_str())));
        if (mtl.illum >= 6)
          m->SetRefraction(Color3f(1.f - mtl.transmittance[0], 1.f - mtl.transmittance[1], 1.f - mtl.transmittance[2]));
      }
      mm->AppendMaterial(m);
    }
    return mm;
  }

  // xmlload.cpp:188-289
  void LoadNode(Node *parent, const XmlElement *element, int level = 0)
  {
    Node *node = new Node;
    parent->AppendChild(node);
    const char *name = element->Attribute("name");
    node->SetName(name);
    const char *mtlName = element->Attribute("material");
    if (mtlName) nodeMtlList.push_back({node, mtlName});
    const char *type = element->Attribute("type");
    if (type) {
      if (Same(type, "sphere")) node->SetNodeObj(&theSphere);
      else if (Same(type, "plane")) node->SetNodeObj(&thePlane);
      else if (Same(type, "obj")) {
        const char *objName = name ? name : "";
        TriObj *obj = scene.FindObject(objName);
        if (!obj) {
          std::unique_ptr<TriObj> tobj(new TriObj);
          std::string err;
          if (!tobj->Load(AssetPath(objName).c_str(), mtlName == nullptr, &err)) {
            QA_PRINTF(" -- ERROR: Cannot load file \"%s\" (%s)\n", objName, err.c_str());
          } else {
            tobj->SetRelDirectory(objName);
            obj = tobj.get();
            scene.objList.emplace_back(objName, std::move(tobj));
            if (mtlName == nullptr && obj->NM() > 0) {
              if (scene.FindMaterial(objName) == nullptr) {
                MultiMtl *mm = MakeMultiMtl(obj);
                mm->SetName(objName);
                scene.materials.emplace_back(mm);
                nodeMtlList.push_back({node, objName});
              }
            }
          }
        }
        node->SetNodeObj(obj);
      } else {
        QA_PRINTF(" - UNKNOWN TYPE %s\n", type);
      }
    }
    for (const XmlElement *c = element->FirstChildElement(); c; c = c->NextSiblingElement())
      if (Is(c, "object")) LoadNode(node, c, level + 1);
    LoadTransform(node, element);
  }

  // xmlload.cpp:324-399
  void LoadMaterial(const XmlElement *element)
  {
    const char *name = element->Attribute("name");
    const char *type = element->Attribute("type");
    if (!type || !Same(type, "blinn")) return;
    MtlBlinn *m = new MtlBlinn();
    for (const XmlElement *c = element->FirstChildElement(); c; c = c->NextSiblingElement()) {
      Color3f col(1, 1, 1);
      float f = 1;
      if (Is(c, "diffuse")) { ReadColor(c, col); m->SetDiffuse(col); m->SetDiffuseTexture(ReadTexture(c)); }
      else if (Is(c, "specular")) { ReadColor(c, col); m->SetSpecular(col); m->SetSpecularTexture(ReadTexture(c)); }
      else if (Is(c, "glossiness")) { ReadFloat(c, f); m->SetGlossiness(f); }
      else if (Is(c, "emission")) { ReadColor(c, col); m->SetEmission(col); m->SetEmissionTexture(ReadTexture(c)); }
      else if (Is(c, "reflection")) {
        ReadColor(c, col);
        m->SetReflection(col);
        m->SetReflectionTexture(ReadTexture(c));
        f = 0;
        ReadFloat(c, f, "glossiness");
        m->SetReflectionGlossiness(f);
      } else if (Is(c, "refraction")) {
        ReadColor(c, col);
        m->SetRefraction(col);
        ReadFloat(c, f, "index");
        m->SetRefractionIndex(f);
        m->SetRefractionTexture(ReadTexture(c));
        f = 0;
        ReadFloat(c, f, "glossiness");
        m->SetRefractionGlossiness(f);
      } else if (Is(c, "absorption")) { ReadColor(c, col); m->SetAbsorption(col); }
    }
    m->SetName(name);
    scene.materials.emplace_back(m);
  }

  // xmlload.cpp:403-519
  void LoadLight(const XmlElement *element)
  {
    const char *name = element->Attribute("name");
    const char *type = element->Attribute("type");
    if (!type) return;
    Light *light = nullptr;
    if (Same(type, "ambient")) {
      AmbientLight *l = new AmbientLight();
      light = l;
      for (const XmlElement *c = element->FirstChildElement(); c; c = c->NextSiblingElement())
        if (Is(c, "intensity")) { Color3f col(1, 1, 1); ReadColor(c, col); l->SetIntensity(col); }
    } else if (Same(type, "direct")) {
      DirectLight *l = new DirectLight();
      light = l;
      for (const XmlElement *c = element->FirstChildElement(); c; c = c->NextSiblingElement()) {
        if (Is(c, "intensity")) { Color3f col(1, 1, 1); ReadColor(c, col); l->SetIntensity(col); }
        else if (Is(c, "direction")) { Point3 v(1, 1, 1); ReadVector(c, v); l->SetDirection(v); }
      }
    } else if (Same(type, "point")) {
      PointLight *l = new PointLight();
      light = l;
      for (const XmlElement *c = element->FirstChildElement(); c; c = c->NextSiblingElement()) {
        if (Is(c, "intensity")) { Color3f col(1, 1, 1); ReadColor(c, col); l->SetIntensity(col); }
        else if (Is(c, "position")) { Point3 v(0, 0, 0); ReadVector(c, v); l->SetPosition(v); }
        else if (Is(c, "size")) { float f = 0; ReadFloat(c, f); l->SetSize(f); }
      }
    } else if (Same(type, "spot")) {
      SpotLight *l = new SpotLight();
      light = l;
      for (const XmlElement *c = element->FirstChildElement(); c; c = c->NextSiblingElement()) {
        if (Is(c, "intensity")) { Color3f col(1, 1, 1); ReadColor(c, col); l->SetIntensity(col); }
        else if (Is(c, "position")) { Point3 v(0, 0, 0); ReadVector(c, v); l->SetPosition(v); }
        else if (Is(c, "size")) { float f = 0; ReadFloat(c, f); l->SetSize(f); }
        else if (Is(c, "rotation")) { Point3 v(0, 0, 0); float f = 0; ReadVector(c, v); ReadFloat(c, f, "angle"); l->SetRotation(f, v); }
        else if (Is(c, "angle")) { float f = 0; ReadFloat(c, f); l->SetAngle(f); }
        else if (Is(c, "blend")) { float f = 0; ReadFloat(c, f); l->SetBlend(f); }
      }
    }
    if (light) {
      light->SetName(name);
      scene.lights.emplace_back(light);
    }
  }

  // xmlload.cpp:160-185
  void LoadSceneElement(const XmlElement *element)
  {
    for (const XmlElement *c = element->FirstChildElement(); c; c = c->NextSiblingElement()) {
      if (Is(c, "background")) {
        Color3f col(1, 1, 1);
        ReadColor(c, col);
        scene.background.SetColor(col);
        scene.background.SetTexture(ReadTexture(c));
      } else if (Is(c, "environment")) {
        Color3f col(1, 1, 1);
        ReadColor(c, col);
        scene.environment.SetColor(col);
        scene.environment.SetTexture(ReadTexture(c));
      } else if (Is(c, "object")) LoadNode(&scene.rootNode, c);
      else if (Is(c, "material")) LoadMaterial(c);
      else if (Is(c, "light")) LoadLight(c);
    }
  }
};

}  // namespace

void LoadSceneInSilentMode(bool flag) { silentmode = flag; }

// src/textures/texture.cpp:58-93
bool TextureFile::Load()
{
  data.clear();
  width = height = 0;
  const char *name = GetName();
  const int len = (int) strlen(name);
  if (len < 3) return false;
  const char *ext = name + len - 3;
  if (strncasecmp(ext, "png", 3) == 0) return LoadPNG(name, width, height, data);
  if (strncasecmp(ext, "ppm", 3) == 0) return LoadPPM(name, width, height, data);
  return false;
}

// src/parser/xmlload.cpp:71-149
int LoadScene(const char *filename, Scene &scene)
{
  XmlDocument doc;
  if (!doc.LoadFile(filename)) {
    QA_PRINTF("Failed to load the file \"%s\" (%s)\n", filename, doc.Error().c_str());
    return 0;
  }
  const XmlElement *xml = doc.FirstChildElement("xml");
  if (!xml) { QA_PRINTF("No \"xml\" tag found.\n"); return 0; }
  const XmlElement *sc = xml->FirstChildElement("scene");
  if (!sc) { QA_PRINTF("No \"scene\" tag found.\n"); return 0; }
  const XmlElement *cam = xml->FirstChildElement("camera");
  if (!cam) { QA_PRINTF("No \"camera\" tag found.\n"); return 0; }

  const std::string root = scene.assetRoot;
  scene.Clear();
  scene.assetRoot = root;
  Loader loader(scene);
  loader.LoadSceneElement(sc);
  scene.rootNode.ComputeChildBoundBox();
  for (auto &nm : loader.nodeMtlList) {
    Material *mtl = scene.FindMaterial(nm.mtlName.c_str());
    if (mtl) nm.node->SetMaterial(mtl);
  }

  Camera &c = scene.camera;
  c.Init();
  c.dir += c.pos;
  for (const XmlElement *e = cam->FirstChildElement(); e; e = e->NextSiblingElement()) {
    if (Is(e, "position")) ReadVector(e, c.pos);
    else if (Is(e, "target")) ReadVector(e, c.dir);
    else if (Is(e, "up")) ReadVector(e, c.up);
    else if (Is(e, "fov")) ReadFloat(e, c.fovy);
    else if (Is(e, "focaldist")) ReadFloat(e, c.focalDistance);
    else if (Is(e, "dof")) ReadFloat(e, c.depthOfField);
    else if (Is(e, "width")) e->QueryIntAttribute("value", &c.imgWidth);
    else if (Is(e, "height")) e->QueryIntAttribute("value", &c.imgHeight);
  }
  c.dir -= c.pos;
  c.dir = normalize(c.dir);
  const Point3 x = cross(c.dir, c.up);
  c.up = normalize(cross(x, c.dir));
  return 1;
}

}  // namespace qaray_hip
