// qa_texture_dev.h — texture coordinates, ray differentials and texture sampling on the device
// (kernel variants with TEX = true; scenes without any TextureMap never instantiate this code).
//
// Reference: per-object texture coordinates and their differentials inside the intersectors
// (src/objects/objects.cpp:48-53,96-136,144-147,168-204,256-304), Texture::Sample's 32-tap
// elliptical filter and TileClamp (src/core/texture.cpp:32-63), TextureMap / TexturedColor
// (src/core/texture.cpp:67-114), TextureFile / TextureChecker (src/textures/texture.cpp:97-137).
// Scene::TraceNodeNormal transforms the ray with Node::ToNodeCoords(DiffRay), which builds a fresh
// DiffRay whose hasDiffRay member is true (src/core/node.cpp:119-126, src/core/ray.h:55): the
// differential branch therefore runs for secondary rays too (their x/y rays equal the central
// one), which is why tiny non-zero duvw - and the 32-tap filter - also occur there.
#pragma once
#include "qa_device_math.h"
#include "qa_flat_scene.h"

namespace qa {

#define QA_RDX (1.f / 0.01f) /* DiffRay::rdx = 1/dx, src/core/ray.cpp:31-34 */
#define QA_RCP_PI (1.f / QA_PI)
#define QA_RCP_2PI (1.f / (2.f * QA_PI))

struct TexHit {     // HitInfo::uvw, duvw[2], hasTexture (src/core/hitinfo.h:36-52)
  f3 uvw, duvw0, duvw1;
  bool hasTexture;
};

struct TexTables {
  const float4 *texels;   // every file texture as float RGB (x / 255.0f evaluated once per texel at upload: the same IEEE division, the
  const uint32_t *texOff; // same bits), 16 bytes per texel: texture i starts at texels[texOff[i]]
  const unsigned char *blob;
  const qa_texmap *texmap;
  const qa_texture *tex;
  const float *filter;  // 31 x (x, y): the elliptical tap offsets, evaluated on the host with glibc
};

// ---- texture coordinates --------------------------------------------------------------------
// Sphere_TexCoord (objects.cpp:48-53): C's double atan2/asin, rounded when the Point3 is built
__device__ __forceinline__ f3 sphereTexCoord(f3 p, float rcp_l)
{
  const double u = 0.5 - atan2((double) p.x, (double) p.y) * (double) QA_RCP_2PI;
  const double v = 0.5 + asin((double) (p.z * rcp_l)) * (double) QA_RCP_PI;
  return F3((float) u, (float) v, 0.f);
}
__device__ __forceinline__ f3 planeTexCoord(f3 p) { return F3((p.x + 1.f) * 0.5f, (p.y + 1.f) * 0.5f, 0.f); }

// Sphere::IntersectRay's texture block (objects.cpp:96-118); p, N: the accepted local hit
__device__ __forceinline__ void texSphere(f3 o, f3 dx, f3 dy, f3 p, f3 N, TexHit &t)
{
  t.hasTexture = true;
  t.uvw = sphereTexCoord(p, 1.f);
  const float pz = dot(o - p, N);
  const float t_x = -pz / dot(dx, N);
  const float t_y = -pz / dot(dy, N);
  const f3 p_x = o + dx * t_x;
  const f3 p_y = o + dy * t_y;
  t.duvw0 = (sphereTexCoord(p_x, 1.f / length(p_x)) - t.uvw) * QA_RDX;
  t.duvw1 = (sphereTexCoord(p_y, 1.f / length(p_y)) - t.uvw) * QA_RDX;
}

// Plane::IntersectRay's texture block (objects.cpp:168-192)
__device__ __forceinline__ void texPlane(f3 o, f3 dx, f3 dy, f3 p, TexHit &t)
{
  const f3 N = F3(0, 0, 1);
  t.hasTexture = true;
  t.uvw = planeTexCoord(p);
  const float pz = dot(o, N);
  const float t_x = -pz / dot(dx, N);
  const float t_y = -pz / dot(dy, N);
  t.duvw0 = (planeTexCoord(o + dx * t_x) - t.uvw) * QA_RDX;
  t.duvw1 = (planeTexCoord(o + dy * t_y) - t.uvw) * QA_RDX;
}

// TriMesh::GetTexCoord (src/mesh/TriMesh.h:207-214)
__device__ __forceinline__ f3 triTexCoord(const float *t0, const float *t1, const float *t2, float a, float b, float c)
{
  return F3(t0[0] * a + t1[0] * b + t2[0] * c, t0[1] * a + t1[1] * b + t2[1] * c, 0.f);
}

// TriObj::IntersectTriangle's texture block (objects.cpp:256-294) for the accepted triangle:
// q0..q2 = its DTri record, (a, b) its barycentrics, vt = 6 floats (three texture vertices).
__device__ __forceinline__ void texTriangle(const uint4 q0, const uint4 q1, const uint4 q2, const float *vt, f3 o, f3 dx,
                                            f3 dy, float a, float b, TexHit &t)
{
  const f3 N = F3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
  const f3 A = F3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
  const uint32_t axis = q2.w;
  const float au = (axis == 0) ? A.y : A.x, av = (axis == 2) ? A.y : A.z;
  const float bu = __uint_as_float(q1.z), bv = __uint_as_float(q1.w), cu = __uint_as_float(q2.x),
              cv = __uint_as_float(q2.y), s = __uint_as_float(q2.z);
  t.hasTexture = true;
  t.uvw = triTexCoord(vt, vt + 2, vt + 4, a, b, 1.f - a - b);
  const float pz = dot(o - A, N);
  const float t_x = -pz / dot(dx, N);
  const float t_y = -pz / dot(dy, N);
  const f3 p_x = o + dx * t_x;
  const f3 p_y = o + dy * t_y;
  auto bary = [&](f3 p, float &ra, float &rb, float &rc) {
    const float pu = (axis == 0) ? p.y : p.x, pv = (axis == 2) ? p.y : p.z;
    ra = ((bu - pu) * (cv - pv) - (cu - pu) * (bv - pv)) * s;
    rb = ((cu - pu) * (av - pv) - (au - pu) * (cv - pv)) * s;
    rc = 1.f - ra - rb;
  };
  float ax, bx, cx, ay, by, cy;
  bary(p_x, ax, bx, cx);
  bary(p_y, ay, by, cy);
  t.duvw0 = (triTexCoord(vt, vt + 2, vt + 4, ax, bx, cx) - t.uvw) * QA_RDX;
  t.duvw1 = (triTexCoord(vt, vt + 2, vt + 4, ay, by, cy) - t.uvw) * QA_RDX;
}

// ---- sampling -----------------------------------------------------------------------------------
// Texture::TileClamp (src/core/texture.cpp:53-63)
__device__ __forceinline__ f3 tileClamp(f3 uvw)
{
  f3 u = F3(uvw.x - (int) uvw.x, uvw.y - (int) uvw.y, uvw.z - (int) uvw.z);
  if (u.x < 0) u.x += 1;
  if (u.y < 0) u.y += 1;
  if (u.z < 0) u.z += 1;
  return u;
}
// (the reference converts a texel with three divisions per bilinear corner, 12 per tap and 384 per filtered lookup,
// src/textures/texture.cpp:120-131; here the quotients are tabulated per texel at upload and a corner is one 16-byte load)
__device__ __forceinline__ f3 texel(const float4 *px) { const float4 t = *px; return F3(t.x, t.y, t.z); }

// TextureChecker::Sample / TextureFile::Sample (src/textures/texture.cpp:97-137)
__device__ __forceinline__ f3 textureSample(const TexTables &tt, int ti, f3 uvw)
{
  const qa_texture &tx = tt.tex[ti];
  if (tx.type == QA_TEX_CHECKER) {
    const f3 u = tileClamp(uvw);
    const bool first = (u.x <= 0.5f) == (u.y <= 0.5f);
    return first ? ld3(tx.color1) : ld3(tx.color2);
  }
  const int width = tx.width, height = tx.height;
  if (width + height == 0) return F3(0, 0, 0);
  const float4 *data = tt.texels + tt.texOff[ti];
  const f3 u = tileClamp(F3(uvw.x, 1.f - uvw.y, uvw.z));
  const float x = width * u.x, y = height * u.y;
  int ix = (int) x, iy = (int) y;
  const float fx = x - ix, fy = y - iy;
  if (ix < 0) ix -= (ix / width - 1) * width;
  if (ix >= width) ix -= (ix / width) * width;
  int ixp = ix + 1;
  if (ixp >= width) ixp -= width;
  if (iy < 0) iy -= (iy / height - 1) * height;
  if (iy >= height) iy -= (iy / height) * height;
  int iyp = iy + 1;
  if (iyp >= height) iyp -= height;
  f3 r = texel(data + (iy * width + ix)) * ((1 - fx) * (1 - fy));
  r = r + texel(data + (iy * width + ixp)) * (fx * (1 - fy));
  r = r + texel(data + (iyp * width + ix)) * ((1 - fx) * fy);
  r = r + texel(data + (iyp * width + ixp)) * (fx * fy);
  return r;
}

// Texture::Sample(uvw, duvw, elliptic = true) (src/core/texture.cpp:32-52): the lookup itself plus 31 taps on the ellipse the
// differentials span.  Everything that does not change from tap to tap is read ONCE here - the texture's kind, its size, where its
// texels start (textureSample() above reads them per call: three dependent memory round trips per tap before the four texel loads) -
// the kind is branched on outside the tap loop, the wrap-around of a texel index (integer divisions) sits behind one rarely taken
// branch, and the tap offsets come through the scalar cache (the tap number is wave-uniform).  Same arithmetic per tap, same order
// of the sum: same bits.
typedef const __attribute__((address_space(4))) float *QaTapPtr;   // constant address space: uniform loads become s_load

__device__ __forceinline__ f3 texCheckerAt(f3 c1, f3 c2, f3 uvw)
{
  const f3 u = tileClamp(uvw);
  return ((u.x <= 0.5f) == (u.y <= 0.5f)) ? c1 : c2;
}

__device__ __forceinline__ f3 texBilinearAt(const float4 *data, int width, int height, f3 uvw)
{
  const f3 u = tileClamp(F3(uvw.x, 1.f - uvw.y, uvw.z));
  const float x = width * u.x, y = height * u.y;
  int ix = (int) x, iy = (int) y;
  const float fx = x - ix, fy = y - iy;
  if (__builtin_expect((unsigned) ix >= (unsigned) width || (unsigned) iy >= (unsigned) height, 0)) {
    if (ix < 0) ix -= (ix / width - 1) * width;
    if (ix >= width) ix -= (ix / width) * width;
    if (iy < 0) iy -= (iy / height - 1) * height;
    if (iy >= height) iy -= (iy / height) * height;
  }
  int ixp = ix + 1;
  if (ixp >= width) ixp -= width;
  int iyp = iy + 1;
  if (iyp >= height) iyp -= height;
  f3 r = texel(data + (iy * width + ix)) * ((1 - fx) * (1 - fy));
  r = r + texel(data + (iy * width + ixp)) * (fx * (1 - fy));
  r = r + texel(data + (iyp * width + ix)) * ((1 - fx) * fy);
  r = r + texel(data + (iyp * width + ixp)) * (fx * fy);
  return r;
}

__device__ __forceinline__ f3 textureSampleFiltered(const TexTables &tt, int ti, f3 uvw, f3 d0, f3 d1)
{
  const qa_texture &tx = tt.tex[ti];
  const bool filtered = !(dot(d0, d0) + dot(d1, d1) == 0);
  const QaTapPtr taps = (QaTapPtr) tt.filter;
  f3 c;
  if (tx.type == QA_TEX_CHECKER) {
    const f3 c1 = ld3(tx.color1), c2 = ld3(tx.color2);
    c = texCheckerAt(c1, c2, uvw);
    if (filtered)
      for (int i = 0; i < 31; ++i) c = c + texCheckerAt(c1, c2, (uvw + d0 * taps[2 * i]) + d1 * taps[2 * i + 1]);
  } else {
    const int width = tx.width, height = tx.height;
    if (width + height == 0) return F3(0, 0, 0);    // (filtered: 32 zeros over 32)
    const float4 *data = tt.texels + tt.texOff[ti];
    c = texBilinearAt(data, width, height, uvw);
    if (filtered)
      for (int i = 0; i < 31; ++i) c = c + texBilinearAt(data, width, height, (uvw + d0 * taps[2 * i]) + d1 * taps[2 * i + 1]);
  }
  return filtered ? c / 32.f : c;
}

__device__ __forceinline__ f3 xformTo(const qa_texmap &m, f3 p) { return mulMV(m.itm, p - ld3(m.pos)); }

// TexturedColor::Sample(uvw) (src/core/texture.cpp:67-70,95-98)
__device__ __forceinline__ f3 texColorSample(const TexTables &tt, f3 color, int texmap, f3 uvw)
{
  if (texmap < 0) return color;
  const qa_texmap &m = tt.texmap[texmap];
  if (m.texture < 0) return color * F3(0, 0, 0);
  return color * textureSample(tt, m.texture, xformTo(m, uvw));
}

// static Sample(hInfo, TexturedColor) (src/materials/MtlBlinn_PhotonMap.cpp:34-39) over
// TexturedColor::Sample(uvw, duvw) (src/core/texture.cpp:71-81,99-104)
__device__ __forceinline__ f3 mtlSample(const TexTables &tt, const TexHit &h, f3 color, int texmap)
{
  if (!h.hasTexture || texmap < 0) return color;
  const qa_texmap &m = tt.texmap[texmap];
  if (m.texture < 0) return color * F3(0, 0, 0);
  const f3 u = xformTo(m, h.uvw);
  const f3 d0 = xformTo(m, h.duvw0 + h.uvw) - u;
  const f3 d1 = xformTo(m, h.duvw1 + h.uvw) - u;
  return color * textureSampleFiltered(tt, m.texture, u, d0, d1);
}

// TexturedColor::SampleEnvironment (src/core/texture.cpp:106-114)
__device__ __forceinline__ f3 sampleEnvironment(const TexTables &tt, f3 color, int texmap, f3 dir)
{
  const float z = qasinf(-dir.z) / QA_PI + 0.5f;
  const float x = dir.x / (qabs(dir.x) + qabs(dir.y));
  const float y = dir.y / (qabs(dir.x) + qabs(dir.y));
  const f3 a = F3(0.5f, 0.5f, 0) * x;
  const f3 b = F3(-0.5f, 0.5f, 0) * y;
  return texColorSample(tt, color, texmap, F3(0.5f, 0.5f, 0.0f) + (a + b) * z);
}

}  // namespace qa
