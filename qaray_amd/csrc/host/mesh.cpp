// mesh.cpp — triangle meshes: Wavefront OBJ/MTL reading, vertex normals, bounds and the BVH.
//
// The reference loads meshes with tinyobjloader (external/tinyobjloader/tiny_obj_loader.h) into
// TriMesh (src/mesh/TriMesh.cpp:63-116) and builds a cy::BVH over the faces
// (src/ext/cyBVH.h:144-166,318-421, src/mesh/TriBVH.cpp:32-56) with at most 4 triangles per leaf
// (src/objects/objects.h:65-73).  The GPU traversal must visit nodes and triangles in the same
// order as the reference (ties between equal hit distances are resolved by visiting order), so
// this file reproduces the *results* of those steps exactly: face order (file order, fan
// triangulation, then the material sort), float values of the vertices (tinyobj's own decimal
// parser, tiny_obj_loader.h:498-611), vertex-normal accumulation incl. its index quirk
// (TriMesh.cpp:134-158), and the BVH's split rule, element permutation and node numbering.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "scene.h"

namespace qaray_hip {

Box TriObj::GetBoundBox() const
{
  return Box(boundMin.x, boundMin.y, boundMin.z, boundMax.x, boundMax.y, boundMax.z);
}

// ---------------------------------------------------------------------------------------------
// number parsing with tinyobjloader's semantics
// ---------------------------------------------------------------------------------------------
namespace {

inline bool IsDigit(char c) { return c >= '0' && c <= '9'; }

// tiny_obj_loader.h:498-611: digits are accumulated into a double (integer part exactly, each
// decimal digit as digit * 10^-k), an optional decimal exponent e is applied as
// ldexp(m * 5^e, e).  Returns false (leaving *out) when the token is not a number.
bool ParseDecimal(const char *s, const char *end, double *out)
{
  if (s >= end) return false;
  double mant = 0.0;
  int expo = 0;
  bool neg = false, expNeg = false;
  const char *c = s;
  if (*c == '+' || *c == '-') { neg = (*c == '-'); ++c; }
  else if (!IsDigit(*c)) return false;
  int nread = 0;
  while (c != end && IsDigit(*c)) { mant *= 10; mant += (int) (*c - '0'); ++c; ++nread; }
  if (nread == 0) return false;
  if (c != end) {
    bool more = true;
    if (*c == '.') {
      ++c;
      static const double lut[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
      int k = 1;
      while (c != end && IsDigit(*c)) {
        mant += (int) (*c - '0') * (k < 8 ? lut[k] : std::pow(10.0, -k));
        ++k;
        ++c;
      }
    } else if (*c != 'e' && *c != 'E') {
      more = false;
    }
    if (more && c != end && (*c == 'e' || *c == 'E')) {
      ++c;
      if (c != end && (*c == '+' || *c == '-')) { expNeg = (*c == '-'); ++c; }
      else if (!(c != end && IsDigit(*c))) return false;
      int n = 0;
      while (c != end && IsDigit(*c)) { expo *= 10; expo += (int) (*c - '0'); ++c; ++n; }
      if (expNeg) expo = -expo;
      if (n == 0) return false;
    }
  }
  *out = (neg ? -1 : 1) * (expo ? std::ldexp(mant * std::pow(5.0, expo), expo) : mant);
  return true;
}

// tiny_obj_loader.h:613-621: next blank-delimited token -> float, default when unparsable
float ParseReal(const char **tok, double def = 0.0)
{
  *tok += strspn(*tok, " \t");
  const char *end = *tok + strcspn(*tok, " \t\r");
  double v = def;
  ParseDecimal(*tok, end, &v);
  *tok = end;
  return static_cast<float>(v);
}

int ParseInt(const char **tok)
{
  *tok += strspn(*tok, " \t");
  const int i = atoi(*tok);
  *tok += strcspn(*tok, " \t\r");
  return i;
}

// OBJ index: 1-based, negative = relative to the current count, 0 invalid (tiny_obj_loader.h:432-453)
bool FixIndex(int idx, int n, int *ret)
{
  if (idx > 0) { *ret = idx - 1; return true; }
  if (idx == 0) return false;
  *ret = n + idx;
  return true;
}

struct Corner { int v = -1, vt = -1, vn = -1; };

// "i", "i/j", "i//k", "i/j/k" (tiny_obj_loader.h:747-798)
bool ParseCorner(const char **tok, int nv, int nvn, int nvt, Corner *out)
{
  Corner c;
  if (!FixIndex(atoi(*tok), nv, &c.v)) return false;
  *tok += strcspn(*tok, "/ \t\r");
  if ((*tok)[0] != '/') { *out = c; return true; }
  ++*tok;
  if ((*tok)[0] == '/') {
    ++*tok;
    if (!FixIndex(atoi(*tok), nvn, &c.vn)) return false;
    *tok += strcspn(*tok, "/ \t\r");
    *out = c;
    return true;
  }
  if (!FixIndex(atoi(*tok), nvt, &c.vt)) return false;
  *tok += strcspn(*tok, "/ \t\r");
  if ((*tok)[0] != '/') { *out = c; return true; }
  ++*tok;
  if (!FixIndex(atoi(*tok), nvn, &c.vn)) return false;
  *tok += strcspn(*tok, "/ \t\r");
  *out = c;
  return true;
}

inline bool IsSpace(char c) { return c == ' ' || c == '\t'; }

// getline that accepts \n, \r\n and \r (tinyobj's safeGetline)
bool GetLine(std::istream &is, std::string &line)
{
  line.clear();
  if (is.peek() == EOF) return false;
  while (true) {
    const int ch = is.get();
    if (ch == EOF) break;
    if (ch == '\n') break;
    if (ch == '\r') { if (is.peek() == '\n') is.get(); break; }
    line.push_back((char) ch);
  }
  return true;
}

std::string TexName(const char *tok)
{
  // last blank-separated word: texture options (-bm 1 ...) precede the file name
  std::string s(tok);
  const size_t e = s.find_last_not_of(" \t\r");
  if (e == std::string::npos) return "";
  s.erase(e + 1);
  const size_t b = s.find_last_of(" \t");
  return b == std::string::npos ? s : s.substr(b + 1);
}

// tiny_obj_loader.h:1049-1430, restricted to the fields the reference reads
void LoadMtl(std::istream &is, std::map<std::string, int> &byName, std::vector<ObjMaterial> &out)
{
  ObjMaterial cur;
  std::string line;
  while (GetLine(is, line)) {
    const size_t e = line.find_last_not_of(" \t");
    line = (e == std::string::npos) ? "" : line.substr(0, e + 1);
    if (line.empty()) continue;
    const char *t = line.c_str();
    t += strspn(t, " \t");
    if (t[0] == '\0' || t[0] == '#') continue;
    if (strncmp(t, "newmtl", 6) == 0 && IsSpace(t[6])) {
      if (!cur.name.empty()) { byName.insert({cur.name, (int) out.size()}); out.push_back(cur); }
      cur = ObjMaterial();
      cur.name = t + 7;
      continue;
    }
    auto real3 = [&](float *dst) { const char *p = t + 2; dst[0] = ParseReal(&p); dst[1] = ParseReal(&p); dst[2] = ParseReal(&p); };
    if (t[0] == 'K' && t[1] == 'd' && IsSpace(t[2])) { real3(cur.diffuse); continue; }
    if (t[0] == 'K' && t[1] == 's' && IsSpace(t[2])) { real3(cur.specular); continue; }
    if (((t[0] == 'K' && t[1] == 't') || (t[0] == 'T' && t[1] == 'f')) && IsSpace(t[2])) { real3(cur.transmittance); continue; }
    if (t[0] == 'N' && t[1] == 'i' && IsSpace(t[2])) { const char *p = t + 2; cur.ior = ParseReal(&p); continue; }
    if (t[0] == 'N' && t[1] == 's' && IsSpace(t[2])) { const char *p = t + 2; cur.shininess = ParseReal(&p); continue; }
    if (strncmp(t, "illum", 5) == 0 && IsSpace(t[5])) { const char *p = t + 6; cur.illum = ParseInt(&p); continue; }
    if (strncmp(t, "map_Kd", 6) == 0 && IsSpace(t[6])) { cur.diffuse_texname = TexName(t + 7); continue; }
    if (strncmp(t, "map_Ks", 6) == 0 && IsSpace(t[6])) { cur.specular_texname = TexName(t + 7); continue; }
  }
  byName.insert({cur.name, (int) out.size()});  // the last material is flushed unconditionally
  out.push_back(cur);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// TriMesh::LoadFromFileObj — src/mesh/TriMesh.cpp:63-116 over tinyobj::LoadObj(triangulate=true)
// ---------------------------------------------------------------------------------------------
bool TriObj::LoadFromFileObj(const char *filename, std::string *err)
{
  file = filename;
  std::replace(file.begin(), file.end(), '\\', '/');
  const size_t slash = file.find_last_of('/');
  path = (slash == std::string::npos) ? "" : file.substr(0, slash + 1);
  name = (slash == std::string::npos) ? file : file.substr(slash + 1);

  std::ifstream ifs(file.c_str());
  if (!ifs) { if (err) *err = "Cannot open file [" + file + "]"; return false; }

  vertices.clear(); normals.clear(); texcoords.clear(); faces.clear(); materials.clear();
  std::map<std::string, int> mtlByName;
  int curMtl = -1;
  std::string line;
  while (GetLine(ifs, line)) {
    if (line.empty()) continue;
    const char *t = line.c_str();
    t += strspn(t, " \t");
    if (t[0] == '\0' || t[0] == '#') continue;
    if (t[0] == 'v' && IsSpace(t[1])) {
      t += 2;
      const float x = ParseReal(&t), y = ParseReal(&t), z = ParseReal(&t);
      vertices.push_back(x); vertices.push_back(y); vertices.push_back(z);
      continue;
    }
    if (t[0] == 'v' && t[1] == 'n' && IsSpace(t[2])) {
      t += 3;
      const float x = ParseReal(&t), y = ParseReal(&t), z = ParseReal(&t);
      normals.push_back(x); normals.push_back(y); normals.push_back(z);
      continue;
    }
    if (t[0] == 'v' && t[1] == 't' && IsSpace(t[2])) {
      t += 3;
      const float x = ParseReal(&t), y = ParseReal(&t);
      texcoords.push_back(x); texcoords.push_back(y);
      continue;
    }
    if (t[0] == 'f' && IsSpace(t[1])) {
      t += 2;
      t += strspn(t, " \t");
      std::vector<Corner> poly;
      while (t[0] != '\0' && t[0] != '\r' && t[0] != '\n') {
        Corner c;
        if (!ParseCorner(&t, (int) (vertices.size() / 3), (int) (normals.size() / 3), (int) (texcoords.size() / 2), &c)) {
          if (err) *err = "Failed parse `f' line(e.g. zero value for face index).";
          return false;
        }
        poly.push_back(c);
        t += strspn(t, " \t\r");
      }
      // polygon -> triangle fan (tiny_obj_loader.h:992-1015)
      for (size_t k = 2; k < poly.size(); ++k) {
        const Corner &a = poly[0], &b = poly[k - 1], &c = poly[k];
        qa_face f;
        f.v[0] = a.v; f.v[1] = b.v; f.v[2] = c.v;
        f.vn[0] = a.vn; f.vn[1] = b.vn; f.vn[2] = c.vn;
        f.vt[0] = a.vt; f.vt[1] = b.vt; f.vt[2] = c.vt;
        f.mtl = curMtl;
        faces.push_back(f);
      }
      continue;
    }
    if (strncmp(t, "usemtl", 6) == 0 && IsSpace(t[6])) {
      const std::string nm(t + 7);
      auto it = mtlByName.find(nm);
      curMtl = (it != mtlByName.end()) ? it->second : -1;
      continue;
    }
    if (strncmp(t, "mtllib", 6) == 0 && IsSpace(t[6])) {
      std::stringstream ss(std::string(t + 7));
      std::string fn;
      while (std::getline(ss, fn, ' ')) {
        std::ifstream ms((path + fn).c_str());
        if (!ms) continue;
        LoadMtl(ms, mtlByName, materials);
        break;  // first file that opens wins
      }
      continue;
    }
    // g / o / s / t and unknown statements do not change the face order
  }
  // tinyobjloader does not range-check indices (the reference would read out of bounds); refuse
  // such files instead
  {
    const int nv = (int) (vertices.size() / 3), nn = (int) (normals.size() / 3), nt = (int) (texcoords.size() / 2);
    for (const qa_face &f : faces)
      for (int k = 0; k < 3; ++k) {
        const bool bad = f.v[k] < 0 || f.v[k] >= nv || f.vn[k] < -1 || f.vn[k] >= nn || f.vt[k] < -1 || f.vt[k] >= nt ||
                         (nn > 0 && f.vn[k] < 0);
        if (bad) {
          if (err) *err = "face references a vertex / normal / texture vertex that does not exist";
          return false;
        }
      }
  }
  // TriMesh.cpp:107-114: faces are ordered by material with std::sort and this comparator; the
  // sort is not stable, so the same library routine is used to land on the same permutation.
  std::sort(faces.begin(), faces.end(), [](const qa_face &a, const qa_face &b) {
    if (a.mtl >= 0 && b.mtl >= 0) return a.mtl < b.mtl;
    return false;
  });
  return true;
}

// TriMesh::ComputeNormals — src/mesh/TriMesh.cpp:134-158.  The reference clears and finally
// normalises entries [0, NF) of the per-VERTEX normal array (it loops over the face count), so
// vertices with index >= NF keep un-normalised sums; reproduced, clamped to the array size.
void TriObj::ComputeNormals()
{
  normals.assign(3 * NV(), 0.f);
  auto VN = [&](size_t i) { return reinterpret_cast<Vec3 *>(&normals[3 * i]); };
  auto V = [&](int i) { return *reinterpret_cast<const Vec3 *>(&vertices[3 * (size_t) i]); };
  for (size_t i = 0; i < NF(); ++i) {
    const Vec3 N = cross(V(faces[i].v[1]) - V(faces[i].v[0]), V(faces[i].v[2]) - V(faces[i].v[0]));
    for (int k = 0; k < 3; ++k) {
      *VN((size_t) faces[i].v[k]) += N;
      faces[i].vn[k] = faces[i].v[k];
    }
  }
  const size_t n = std::min(NF(), NV());
  for (size_t i = 0; i < n; ++i) *VN(i) = normalize(*VN(i));
}

// TriMesh::ComputeBoundingBox — src/mesh/TriMesh.cpp:117-133
void TriObj::ComputeBoundingBox()
{
  if (NV() == 0) { boundMin = Point3(1, 1, 1); boundMax = Point3(0, 0, 0); return; }
  boundMin = boundMax = Point3(vertices[0], vertices[1], vertices[2]);
  for (size_t i = 1; i < NV(); ++i)
    for (int k = 0; k < 3; ++k) {
      const float v = vertices[3 * i + k];
      if (boundMin[k] > v) boundMin[k] = v;
      if (boundMax[k] < v) boundMax[k] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// BVH — same tree as cy::BVH::Build with MeanSplit (src/ext/cyBVH.h:144-166,318-421)
// ---------------------------------------------------------------------------------------------
namespace {

struct Bounds {
  float b[6] = {1e30f, 1e30f, 1e30f, -1e30f, -1e30f, -1e30f};
  void Grow(const Bounds &o)
  {
    for (int i = 0; i < 3; ++i) {
      if (b[i] > o.b[i]) b[i] = o.b[i];
      if (b[i + 3] < o.b[i + 3]) b[i + 3] = o.b[i + 3];
    }
  }
};

class BvhBuilder {
 public:
  BvhBuilder(const TriObj &m, unsigned maxPerLeaf) : mesh(m), leafMax(maxPerLeaf) {}

  void Run(std::vector<qa_bvh_node> &nodes, std::vector<uint32_t> &elements)
  {
    const unsigned n = (unsigned) mesh.NF();
    elements.resize(n);
    for (unsigned i = 0; i < n; ++i) elements[i] = i;
    nodes.assign(2, qa_bvh_node{});  // slot 0 unused, slot 1 = root
    if (n == 0) return;
    el = elements.data();
    out = &nodes;
    Bounds root;
    for (unsigned i = 0; i < n; ++i) root.Grow(ElementBounds(i));
    Emit(1, 0, n, root);
  }

 private:
  Bounds ElementBounds(unsigned f) const
  {
    // src/mesh/TriBVH.cpp:32-46
    Bounds r;
    const qa_face &fc = mesh.faces[f];
    const float *p0 = &mesh.vertices[3 * (size_t) fc.v[0]];
    for (int k = 0; k < 3; ++k) r.b[k] = r.b[k + 3] = p0[k];
    for (int j = 1; j < 3; ++j) {
      const float *p = &mesh.vertices[3 * (size_t) fc.v[j]];
      for (int k = 0; k < 3; ++k) {
        if (r.b[k] > p[k]) r.b[k] = p[k];
        if (r.b[k + 3] < p[k]) r.b[k + 3] = p[k];
      }
    }
    return r;
  }
  float Center(unsigned f, int dim) const
  {
    // src/mesh/TriBVH.cpp:49-56
    const qa_face &fc = mesh.faces[f];
    return (mesh.vertices[3 * (size_t) fc.v[0] + dim] + mesh.vertices[3 * (size_t) fc.v[1] + dim] +
            mesh.vertices[3 * (size_t) fc.v[2] + dim]) / 3.0f;
  }
  // Number of elements that go to the first child; 0 = keep as a leaf.
  unsigned Split(unsigned first, unsigned count, const Bounds &box)
  {
    if (count <= leafMax) return 0;
    const float d[3] = {box.b[3] - box.b[0], box.b[4] - box.b[1], box.b[5] - box.b[2]};
    unsigned order[3];
    order[0] = d[0] >= d[1] ? (d[0] >= d[2] ? 0 : 2) : (d[1] >= d[2] ? 1 : 2);
    order[1] = (order[0] + 1) % 3;
    order[2] = (order[0] + 2) % 3;
    if (d[order[1]] < d[order[2]]) std::swap(order[1], order[2]);
    uint32_t *e = el + first;
    for (int s = 0; s < 3; ++s) {
      const unsigned dim = order[s];
      const float mid = 0.5f * (box.b[dim] + box.b[dim + 3]);
      unsigned i = 0, j = count;
      while (i < j) {
        if (Center(e[i], (int) dim) <= mid) ++i;
        else { --j; std::swap(e[i], e[j]); }
      }
      if (i < count && i > 0) return i;
    }
    return 0;
  }
  // Writes node `id` covering elements [first, first+count); children get the next two free
  // slots at the moment their parent is written, first child's subtree before the second's.
  void Emit(unsigned id, unsigned first, unsigned count, const Bounds &box)
  {
    unsigned nFirst = Split(first, count, box);
    if (nFirst == 0 || nFirst >= count) {
      if (count > 8) nFirst = count / 2;  // CY_BVH_MAX_ELEMENT_COUNT
      else {
        qa_bvh_node &n = (*out)[id];
        memcpy(n.box, box.b, sizeof(n.box));
        n.data = (first & QA_BVH_OFFSET_MASK) | ((count - 1) << QA_BVH_COUNT_SHIFT) | QA_BVH_LEAF_BIT;
        return;
      }
    }
    Bounds b0, b1;
    for (unsigned i = 0; i < nFirst; ++i) b0.Grow(ElementBounds(el[first + i]));
    for (unsigned i = nFirst; i < count; ++i) b1.Grow(ElementBounds(el[first + i]));
    const unsigned child = (unsigned) out->size();
    out->resize(out->size() + 2);
    {
      qa_bvh_node &n = (*out)[id];
      memcpy(n.box, box.b, sizeof(n.box));
      n.data = child & QA_BVH_CHILD_MASK;
    }
    Emit(child, first, nFirst, b0);
    Emit(child + 1, first + nFirst, count - nFirst, b1);
  }

  const TriObj &mesh;
  unsigned leafMax;
  uint32_t *el = nullptr;
  std::vector<qa_bvh_node> *out = nullptr;
};

}  // namespace

void TriObj::BuildBVH(unsigned maxElementsPerNode)
{
  if (maxElementsPerNode > 8) maxElementsPerNode = 8;
  BvhBuilder(*this, maxElementsPerNode).Run(bvhNodes, bvhElements);
}

// TriObj::Load — src/objects/objects.h:65-73
bool TriObj::Load(const char *filename, bool /*loadMtl*/, std::string *err)
{
  bvhNodes.clear();
  bvhElements.clear();
  if (!LoadFromFileObj(filename, err)) return false;
  if (NVN() == 0) ComputeNormals();
  ComputeBoundingBox();
  BuildBVH(4);
  return true;
}

}  // namespace qaray_hip
