/* oracle/qa_oracle.c — TEST INFRASTRUCTURE ONLY.  See qa_oracle.h.
 *
 * Plain-C restatement of the reference (wilsonCernWq/qaray) hot path.  Every function names the
 * reference file:line it follows.  Arithmetic is written in the reference's evaluation order
 * (GLM 0.9.8.4 vec3/mat3 operators expand to scalar expressions evaluated left to right;
 * glm::dot = (x*x' + y*y') + z*z', glm::normalize = v * (1/sqrt(dot(v,v)))) and this file must be
 * compiled with -ffp-contract=off, so that results match the reference build bit for bit.
 * libm entry points are the ones the reference objects import (nm -u): sinf/cosf (sincosf),
 * powf, expf, tanf, sqrtf, asinf, and the DOUBLE asin/atan2 in the sphere texture coordinates
 * (src/objects/objects.cpp:48-53 calls the unqualified C functions).
 */
#define _GNU_SOURCE
#include "qa_oracle.h"
#include "qa_flat_scene.h"
#include "qa_seed.h"
#include "qa_photon.h"

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* vec3 / mat3 helpers in GLM's evaluation order                                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3p(const float *p) { return V3(p[0], p[1], p[2]); }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vdiv(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* glm/detail/func_geometric.inl:54-61 */
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* glm/detail/func_geometric.inl:74-85 */
static inline v3 vcross(v3 a, v3 b)
{
  return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }
/* glm/detail/func_geometric.inl:88-96 + func_exponential.inl:130-133 */
static inline v3 vnormalize(v3 a) { return vscale(a, 1.f / sqrtf(vdot(a, a))); }
/* glm/detail/type_mat3x3.inl:428-434, m column-major */
static inline v3 m3mulv(const float *m, v3 v)
{
  return V3(m[0] * v.x + m[3] * v.y + m[6] * v.z,
            m[1] * v.x + m[4] * v.y + m[7] * v.z,
            m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
/* src/core/transform.cpp:49-56 */
static inline v3 m3tmulv(const float *m, v3 d)
{
  return V3(vdot(V3(m[0], m[1], m[2]), d), vdot(V3(m[3], m[4], m[5]), d),
            vdot(V3(m[6], m[7], m[8]), d));
}

#define QMIN(x, y) ((x) < (y) ? (x) : (y))   /* src/math/math.h:104-107 */
#define QMAX(x, y) ((x) > (y) ? (x) : (y))
#define QABS(x) ((x) > 0 ? (x) : -(x))

static const float kPI = (float) M_PI;           /* src/math/math.cpp:13-15 */
#define kRCP_PI (1.f / kPI)
#define kRCP_2PI (1.f / (2.f * kPI))
#define kBIAS 0.005f                             /* src/objects/objects.cpp:19 */
#define kDX 0.01f                                /* src/core/ray.cpp:31-34 */
#define kRDX (1.f / kDX)

/* ------------------------------------------------------------------------------------------ */
typedef struct { v3 p, dir; } ray_t;
typedef struct { ray_t c, x, y; int hasDiffRay; } diffray_t;
typedef struct {                                 /* src/core/hitinfo.h:36-52 */
  float z; v3 p, N, uvw, duvw[2];
  int mtlID, node, hasFrontHit, hasTexture, hasDiffuseHit;
} hit_t;
typedef struct { float z; v3 p, N; } hitcore_t;
typedef struct { hit_t c; hitcore_t x, y; } diffhit_t;

typedef struct {
  const unsigned char *blob;
  const qa_flat_header *h;
  const qa_instance *inst;
  const qa_mesh *mesh;
  const qa_mtlset *mtlset;
  const qa_material *mtl;
  const qa_light *light;
  const qa_texmap *texmap;
  const qa_texture *tex;
  int max_bounce;
  /* Scene::usePhotonMap, photonmap, causticsmap (src/scene/scene.h:48-67); [0] photon, [1] caustics */
  int use_pm;
  struct { const qa_photon *photons; uint32_t count; int half; float radius; } pm[2];
} scene_t;

typedef struct {
  uint32_t rng;
  qa_oracle_counters cnt;
} tls_t;

/* src/core/hitinfo.cpp:31-42, hitinfo.h:60-72 */
static void hit_init(diffhit_t *h)
{
  h->c.z = QA_BIGFLOAT;
  h->c.p = V3(0, 0, 0);
  h->c.N = V3(0, 0, 0);
  h->c.uvw = V3(0.5f, 0.5f, 0.5f);
  h->c.duvw[0] = V3(0, 0, 0);
  h->c.duvw[1] = V3(0, 0, 0);
  h->c.node = -1;
  h->c.mtlID = 0;
  h->c.hasFrontHit = 1;
  h->c.hasTexture = 0;
  h->c.hasDiffuseHit = 0;
  h->x.z = QA_BIGFLOAT; h->x.p = V3(0, 0, 0); h->x.N = V3(0, 0, 0);
  h->y.z = QA_BIGFLOAT; h->y.p = V3(0, 0, 0); h->y.N = V3(0, 0, 0);
}

/* ------------------------------------------------------------------------------------------ */
/* RNG + samplers                                                                              */
/* ------------------------------------------------------------------------------------------ */
/* src/samplers/Sampler_Marsaglia.cpp:43-53: x / (float)(2^32 - 1) with the divisor rounding to 2^32 */
static inline float rng1(tls_t *t)
{
  uint32_t x = t->rng;
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  t->rng = x;
  return (float) x / 4294967296.0f;
}

/* src/core/sampler.cpp:31-40 */
float qa_oracle_halton(int index, int base)
{
  float r = 0;
  float f = 1.0f / (float) base;
  for (int i = index; i > 0; i /= base) {
    r += f * (i % base);
    f /= (float) base;
  }
  return r;
}

void qa_oracle_rng_stream(uint32_t seed, uint32_t pixel, int n, float *out)
{
  tls_t t;
  t.rng = qa_pixel_seed(seed, pixel);
  for (int i = 0; i < n; ++i) out[i] = rng1(&t);
}

/* src/core/sampler.cpp:42-53 (z deliberately uses r2, as the reference does) */
static v3 uniform_ball(tls_t *t, float radius)
{
  v3 p;
  do {
    float r1 = rng1(t), r2 = rng1(t), r3 = rng1(t);
    (void) r3;
    p.x = (2.f * r1 - 1.f) * radius;
    p.y = (2.f * r2 - 1.f) * radius;
    p.z = (2.f * r2 - 1.f) * radius;
  } while (vlength(p) > radius);
  return p;
}

/* src/core/sampler.cpp:87-103 */
static v3 cos_weighted_hemisphere(tls_t *t)
{
  float r1 = rng1(t), r2 = rng1(t);
  const float cosTheta = sqrtf(r1);
  const float sinTheta = sqrtf(1 - r1);
  const float phi = 2 * kPI * r2;
  const float x = sinTheta * cosf(phi);
  const float y = sinTheta * sinf(phi);
  return V3(x, y, cosTheta);
}

/* src/math/math.cpp:37-46 */
static v3 to_local_frame(v3 N, v3 sample)
{
  const v3 Z = N;
  const v3 Y = (QABS(Z.x) > QABS(Z.y)) ? vnormalize(V3(Z.z, 0, -Z.x)) : vnormalize(V3(0, -Z.z, Z.y));
  const v3 X = vnormalize(vcross(Y, Z));
  const v3 unit = vnormalize(sample);
  return vadd(vadd(vscale(X, unit.x), vscale(Y, unit.y)), vscale(Z, unit.z));
}

/* src/math/math.h:128-131 */
static inline float luma(v3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }

/* ------------------------------------------------------------------------------------------ */
/* Textures                                                                                    */
/* ------------------------------------------------------------------------------------------ */
/* src/core/texture.cpp:53-63 */
static v3 tile_clamp(v3 uvw)
{
  v3 u;
  u.x = uvw.x - (int) uvw.x;
  u.y = uvw.y - (int) uvw.y;
  u.z = uvw.z - (int) uvw.z;
  if (u.x < 0) u.x += 1;
  if (u.y < 0) u.y += 1;
  if (u.z < 0) u.z += 1;
  return u;
}

/* src/math/math.h:118-121 */
static inline v3 texel(const unsigned char *px)
{
  return V3(px[0] / 255.0f, px[1] / 255.0f, px[2] / 255.0f);
}

/* src/textures/texture.cpp:97-137 */
static v3 texture_sample(const scene_t *s, const qa_texture *tx, v3 uvw)
{
  if (tx->type == QA_TEX_CHECKER) {
    v3 u = tile_clamp(uvw);
    if (u.x <= 0.5f) return u.y <= 0.5f ? v3p(tx->color1) : v3p(tx->color2);
    else return u.y <= 0.5f ? v3p(tx->color2) : v3p(tx->color1);
  }
  const int width = tx->width, height = tx->height;
  if (width + height == 0) return V3(0, 0, 0);
  const unsigned char *data = s->blob + tx->off_texels;
  v3 fl = V3(uvw.x, 1.f - uvw.y, uvw.z);
  v3 u = tile_clamp(fl);
  float x = width * u.x;
  float y = height * u.y;
  int ix = (int) x;
  int iy = (int) y;
  float fx = x - ix;
  float fy = y - iy;
  if (ix < 0) ix -= (ix / width - 1) * width;
  if (ix >= width) ix -= (ix / width) * width;
  int ixp = ix + 1;
  if (ixp >= width) ixp -= width;
  if (iy < 0) iy -= (iy / height - 1) * height;
  if (iy >= height) iy -= (iy / height) * height;
  int iyp = iy + 1;
  if (iyp >= height) iyp -= height;
  v3 r = vscale(texel(data + 3 * (iy * width + ix)), (1 - fx) * (1 - fy));
  r = vadd(r, vscale(texel(data + 3 * (iy * width + ixp)), fx * (1 - fy)));
  r = vadd(r, vscale(texel(data + 3 * (iyp * width + ix)), (1 - fx) * fy));
  r = vadd(r, vscale(texel(data + 3 * (iyp * width + ixp)), fx * fy));
  return r;
}

/* src/core/texture.cpp:32-52 (elliptic = true is the only value ever passed) */
static v3 texture_sample_filtered(const scene_t *s, const qa_texture *tx, v3 uvw, const v3 duvw[2])
{
  v3 c = texture_sample(s, tx, uvw);
  if (vdot(duvw[0], duvw[0]) + vdot(duvw[1], duvw[1]) == 0) return c;
  for (int i = 1; i < 32; i++) {
    float x = qa_oracle_halton(i, 2);
    float y = qa_oracle_halton(i, 3);
    float r = sqrtf(x) * 0.5f;
    x = r * sinf(y * (float) M_PI * 2);
    y = r * cosf(y * (float) M_PI * 2);
    c = vadd(c, texture_sample(s, tx, vadd(vadd(uvw, vscale(duvw[0], x)), vscale(duvw[1], y))));
  }
  return vdivs(c, 32.f);
}

/* Transformation::TransformTo, src/core/transform.h:47 */
static inline v3 xform_to(const float *itm, const float *pos, v3 p) { return m3mulv(itm, vsub(p, v3p(pos))); }

/* TexturedColor::Sample(uvw), src/core/texture.cpp:67-70,95-98 */
static v3 texcolor_sample(const scene_t *s, const qa_texcolor *tc, v3 uvw)
{
  v3 color = v3p(tc->color);
  if (tc->texmap < 0) return color;
  const qa_texmap *m = &s->texmap[tc->texmap];
  if (m->texture < 0) return vmul(color, V3(0, 0, 0));
  return vmul(color, texture_sample(s, &s->tex[m->texture], xform_to(m->itm, m->pos, uvw)));
}

/* TexturedColor::Sample(uvw, duvw), src/core/texture.cpp:71-81,99-104 */
static v3 texcolor_sample_d(const scene_t *s, const qa_texcolor *tc, v3 uvw, const v3 duvw[2])
{
  v3 color = v3p(tc->color);
  if (tc->texmap < 0) return color;
  const qa_texmap *m = &s->texmap[tc->texmap];
  if (m->texture < 0) return vmul(color, V3(0, 0, 0));
  v3 u = xform_to(m->itm, m->pos, uvw);
  v3 d[2];
  d[0] = vsub(xform_to(m->itm, m->pos, vadd(duvw[0], uvw)), u);
  d[1] = vsub(xform_to(m->itm, m->pos, vadd(duvw[1], uvw)), u);
  return vmul(color, texture_sample_filtered(s, &s->tex[m->texture], u, d));
}

/* src/core/texture.cpp:106-114 */
static v3 sample_environment(const scene_t *s, const qa_texcolor *tc, v3 dir)
{
  float z = asinf(-dir.z) / (float) M_PI + 0.5f;
  float x = dir.x / (QABS(dir.x) + QABS(dir.y));
  float y = dir.y / (QABS(dir.x) + QABS(dir.y));
  v3 a = vscale(V3(0.5f, 0.5f, 0), x);
  v3 b = vscale(V3(-0.5f, 0.5f, 0), y);
  return texcolor_sample(s, tc, vadd(V3(0.5f, 0.5f, 0.0f), vscale(vadd(a, b), z)));
}

/* static Sample(hInfo, TexturedColor), src/materials/MtlBlinn_PhotonMap.cpp:34-39 */
static v3 mtl_sample(const scene_t *s, const diffhit_t *h, const qa_texcolor *tc)
{
  return h->c.hasTexture ? texcolor_sample_d(s, tc, h->c.uvw, h->c.duvw) : v3p(tc->color);
}

/* ------------------------------------------------------------------------------------------ */
/* Objects                                                                                     */
/* ------------------------------------------------------------------------------------------ */
/* src/objects/objects.cpp:48-53: computed in double (C asin/atan2), rounded on construction */
static v3 sphere_texcoord(v3 p, float rcp_l)
{
  return V3((float) (0.5f - atan2(p.x, p.y) * kRCP_2PI), (float) (0.5f + asin(p.z * rcp_l) * kRCP_PI), 0.f);
}

/* src/objects/objects.cpp:55-141; dr == NULL is a shadow query */
static int sphere_intersect(const ray_t *ray, hit_t *hc, const diffray_t *dr, diffhit_t *dh)
{
  const float a = vdot(ray->dir, ray->dir);
  const float b = 2.f * vdot(ray->p, ray->dir);
  const float c = vdot(ray->p, ray->p) - 1;
  const float rcp2a = 1.f / (2.f * a);
  const float delta = b * b - 4 * a * c;
  float t = QA_BIGFLOAT;
  if (delta < 0) return 0;
  if (delta == 0) {
    const float t0 = -b * rcp2a;
    if (t0 <= kBIAS) return 0;
    else t = t0;
  } else {
    const float sq = sqrtf(delta);
    const float t1 = (-b - sq) * rcp2a;
    const float t2 = (-b + sq) * rcp2a;
    if (t1 <= kBIAS && t2 <= kBIAS) return 0;
    else if (t1 > kBIAS) t = QMIN(t, t1);
    else if (t2 > kBIAS) t = QMIN(t, t2);
  }
  if (hc->z > t) {
    const v3 p = vadd(ray->p, vscale(ray->dir, t));
    const v3 N = vnormalize(p);
    const int front = (vdot(N, ray->dir) <= 0);
    hc->z = t;
    if (dr != NULL && dh != NULL) {
      hc->p = p;
      hc->N = N;
      hc->hasFrontHit = front;
      hc->hasTexture = 1;
      hc->uvw = sphere_texcoord(p, 1.f);
      if (dr->hasDiffRay) {
        const float pz_x = vdot(vsub(dr->x.p, p), N);
        const float pz_y = vdot(vsub(dr->y.p, p), N);
        const float dz_x = vdot(dr->x.dir, N);
        const float dz_y = vdot(dr->y.dir, N);
        const float t_x = -pz_x / dz_x;
        const float t_y = -pz_y / dz_y;
        const v3 p_x = vadd(dr->x.p, vscale(dr->x.dir, t_x));
        const v3 p_y = vadd(dr->y.p, vscale(dr->y.dir, t_y));
        dh->x.z = t_x; dh->x.p = p_x; dh->x.N = vnormalize(p_x);
        dh->y.z = t_y; dh->y.p = p_y; dh->y.N = vnormalize(p_y);
        hc->duvw[0] = vscale(vsub(sphere_texcoord(p_x, 1.f / vlength(p_x)), hc->uvw), kRDX);
        hc->duvw[1] = vscale(vsub(sphere_texcoord(p_y, 1.f / vlength(p_y)), hc->uvw), kRDX);
      } else {
        dh->x.z = t; dh->x.p = p; dh->x.N = N;
        dh->y.z = t; dh->y.p = p; dh->y.N = N;
        hc->duvw[0] = V3(0, 0, 0);
        hc->duvw[1] = V3(0, 0, 0);
      }
    }
    return 1;
  }
  return 0;
}

/* src/objects/objects.cpp:144-147 */
static inline v3 plane_texcoord(v3 p) { return V3((p.x + 1.f) * 0.5f, (p.y + 1.f) * 0.5f, 0.f); }

/* src/objects/objects.cpp:149-208 */
static int plane_intersect(const ray_t *ray, hit_t *hc, const diffray_t *dr, diffhit_t *dh)
{
  const v3 N = V3(0, 0, 1);
  const float dz = vdot(ray->dir, N);
  if (QABS(dz) < 1e-7f) return 0;
  const float pz = vdot(ray->p, N);
  const float t = -pz / dz;
  if (t <= kBIAS) return 0;
  if (hc->z > t) {
    const v3 p = vadd(ray->p, vscale(ray->dir, t));
    if (QABS(p.x) > 1.f || QABS(p.y) > 1.f) return 0;
    const int front = (vdot(N, ray->dir) <= 0);
    hc->z = t;
    if (dr != NULL && dh != NULL) {
      hc->p = p;
      hc->N = N;
      hc->hasFrontHit = front;
      hc->hasTexture = 1;
      hc->uvw = plane_texcoord(p);
      if (dr->hasDiffRay) {
        const float pz_x = vdot(dr->x.p, N);
        const float pz_y = vdot(dr->y.p, N);
        const float dz_x = vdot(dr->x.dir, N);
        const float dz_y = vdot(dr->y.dir, N);
        const float t_x = -pz_x / dz_x;
        const float t_y = -pz_y / dz_y;
        const v3 p_x = vadd(dr->x.p, vscale(dr->x.dir, t_x));
        const v3 p_y = vadd(dr->y.p, vscale(dr->y.dir, t_y));
        dh->x.z = t_x; dh->x.p = p_x; dh->x.N = N;
        dh->y.z = t_y; dh->y.p = p_y; dh->y.N = N;
        hc->duvw[0] = vscale(vsub(plane_texcoord(p_x), hc->uvw), kRDX);
        hc->duvw[1] = vscale(vsub(plane_texcoord(p_y), hc->uvw), kRDX);
      } else {
        dh->x.z = t; dh->x.p = p; dh->x.N = N;
        dh->y.z = t; dh->y.p = p; dh->y.N = N;
        hc->duvw[0] = V3(0, 0, 0);
        hc->duvw[1] = V3(0, 0, 0);
      }
    }
    return 1;
  }
  return 0;
}

/* src/objects/objects.cpp:30-41 */
static inline float tri_area(int i, v3 A, v3 B, v3 C)
{
  switch (i) {
    case 0: return (B.y - A.y) * (C.z - A.z) - (C.y - A.y) * (B.z - A.z);
    case 1: return (B.x - A.x) * (C.z - A.z) - (C.x - A.x) * (B.z - A.z);
    default: return (B.x - A.x) * (C.y - A.y) - (C.x - A.x) * (B.y - A.y);
  }
}

typedef struct {
  const qa_mesh *m;
  const qa_bvh_node *nodes;
  const uint32_t *elements;
  const qa_face *faces;
  const float *V, *VN, *VT;
} meshview_t;

static meshview_t mesh_view(const scene_t *s, int mi)
{
  meshview_t v;
  v.m = &s->mesh[mi];
  v.nodes = QA_BLOB_PTR(qa_bvh_node, s->blob, v.m->off_bvh_nodes);
  v.elements = QA_BLOB_PTR(uint32_t, s->blob, v.m->off_elements);
  v.faces = QA_BLOB_PTR(qa_face, s->blob, v.m->off_faces);
  v.V = QA_BLOB_PTR(float, s->blob, v.m->off_vertices);
  v.VN = QA_BLOB_PTR(float, s->blob, v.m->off_normals);
  v.VT = QA_BLOB_PTR(float, s->blob, v.m->off_texcoords);
  return v;
}

/* TriMesh::GetTexCoord, src/mesh/TriMesh.h:207-214 */
static inline v3 tri_texcoord(const meshview_t *mv, const qa_face *f, v3 bc)
{
  const float *t0 = mv->VT + 2 * f->vt[0], *t1 = mv->VT + 2 * f->vt[1], *t2 = mv->VT + 2 * f->vt[2];
  return V3(t0[0] * bc.x + t1[0] * bc.y + t2[0] * bc.z, t0[1] * bc.x + t1[1] * bc.y + t2[1] * bc.z, 0.f);
}

/* src/objects/objects.cpp:212-306 */
/* debugging aid: QA_ORACLE_TRACE="i,j,sample" prints the casts of that sample (rays, triangle tests) to stderr */
static int g_trace_on = 0;
static int g_trace_px = -1, g_trace_py = -1, g_trace_s = -1;

static int triangle_intersect(const meshview_t *mv, const ray_t *ray, hit_t *hc, uint32_t faceID,
                              const diffray_t *dr, diffhit_t *dh, tls_t *tl)
{
  tl->cnt.tri_tests++;
  if (g_trace_on) {
    const qa_face *f_ = &mv->faces[faceID];
    const v3 A_ = v3p(mv->V + 3 * f_->v[0]), B_ = v3p(mv->V + 3 * f_->v[1]), C_ = v3p(mv->V + 3 * f_->v[2]);
    const v3 N_ = vnormalize(vcross(vsub(B_, A_), vsub(C_, A_)));
    const float dz_ = vdot(ray->dir, N_), pz_ = vdot(vsub(ray->p, A_), N_);
    fprintf(stderr, "    tri face %u: t %a (%.9g) held %a dz %g\n", faceID, -pz_ / dz_, -pz_ / dz_, hc->z, dz_);
  }
  const qa_face *f = &mv->faces[faceID];
  const v3 A = v3p(mv->V + 3 * f->v[0]);
  const v3 B = v3p(mv->V + 3 * f->v[1]);
  const v3 C = v3p(mv->V + 3 * f->v[2]);
  const v3 N = vnormalize(vcross(vsub(B, A), vsub(C, A)));
  const float dz = vdot(ray->dir, N);
  if (QABS(dz) < 1e-7f) return 0;
  const float pz = vdot(vsub(ray->p, A), N);
  const float t = -pz / dz;
  if (t <= kBIAS) return 0;
  if (hc->z > t) {
    const int front = (dz <= 0);
    const v3 p = vadd(ray->p, vscale(ray->dir, t));
    int axis;
    const float ax_ = QABS(N.x), ay_ = QABS(N.y), az_ = QABS(N.z);
    if (ax_ > ay_ && ax_ > az_) axis = 0;
    else if (ay_ > az_) axis = 1;
    else axis = 2;
    const float s = 1.f / tri_area(axis, A, B, C);
    const float a = tri_area(axis, p, B, C) * s;
    const float b = tri_area(axis, p, C, A) * s;
    const float c = 1.f - a - b;
    if (a < 0 || b < 0 || c < 0) return 0;
    const v3 bc = V3(a, b, c);
    if (g_trace_on) fprintf(stderr, "      ACCEPT face %u t %a bary %g %g %g\n", faceID, t, a, b, c);
    hc->z = t;
    if (dr != NULL && dh != NULL) {
      const int hasVT = (f->vt[0] >= 0) && (f->vt[1] >= 0) && (f->vt[2] >= 0);
      hc->p = p;
      /* TriMesh::GetNormal, src/mesh/TriMesh.h:196-204 (not normalised here) */
      hc->N = vadd(vadd(vscale(v3p(mv->VN + 3 * f->vn[0]), bc.x), vscale(v3p(mv->VN + 3 * f->vn[1]), bc.y)),
                   vscale(v3p(mv->VN + 3 * f->vn[2]), bc.z));
      hc->hasFrontHit = front;
      hc->mtlID = f->mtl;
      if (hasVT) {
        hc->hasTexture = 1;
        hc->uvw = tri_texcoord(mv, f, bc);
      }
      if (dr->hasDiffRay) {
        const float pz_x = vdot(vsub(dr->x.p, A), N);
        const float pz_y = vdot(vsub(dr->y.p, A), N);
        const float dz_x = vdot(dr->x.dir, N);
        const float dz_y = vdot(dr->y.dir, N);
        const float t_x = -pz_x / dz_x;
        const float t_y = -pz_y / dz_y;
        const v3 p_x = vadd(dr->x.p, vscale(dr->x.dir, t_x));
        const v3 p_y = vadd(dr->y.p, vscale(dr->y.dir, t_y));
        const float axx = tri_area(axis, p_x, B, C) * s;
        const float bxx = tri_area(axis, p_x, C, A) * s;
        const float cxx = 1.f - axx - bxx;
        const float ayy = tri_area(axis, p_y, B, C) * s;
        const float byy = tri_area(axis, p_y, C, A) * s;
        const float cyy = 1.f - ayy - byy;
        dh->x.z = t_x; dh->x.p = p_x; dh->x.N = hc->N;
        dh->y.z = t_y; dh->y.p = p_y; dh->y.N = hc->N;
        if (hasVT) {
          hc->duvw[0] = vscale(vsub(tri_texcoord(mv, f, V3(axx, bxx, cxx)), hc->uvw), kRDX);
          hc->duvw[1] = vscale(vsub(tri_texcoord(mv, f, V3(ayy, byy, cyy)), hc->uvw), kRDX);
        }
      } else {
        dh->x.z = t; dh->x.p = p; dh->x.N = hc->N;
        dh->y.z = t; dh->y.p = p; dh->y.N = hc->N;
        hc->duvw[0] = V3(0, 0, 0);
        hc->duvw[1] = V3(0, 0, 0);
      }
    }
    return 1;
  }
  return 0;
}

/* one axis of the slab test, src/objects/objects.cpp:360-395 / src/core/box.cpp:103-123 */
static inline void slab(float d, float p0, float p1, float *t0, float *t1)
{
  if (QABS(d) < 1e-7f) { *t0 = -QA_BIGFLOAT; *t1 = QA_BIGFLOAT; }
  else { *t0 = QMIN(p0, p1); *t1 = QMAX(p0, p1); }
}

static inline void box_entry_exit(v3 rpos, v3 rdir, v3 drcp, const float *box, float *entry, float *exit_)
{
  const v3 p0 = vmul(vneg(vsub(rpos, v3p(box))), drcp);
  const v3 p1 = vmul(vneg(vsub(rpos, v3p(box + 3))), drcp);
  v3 t0, t1;
  slab(rdir.x, p0.x, p1.x, &t0.x, &t1.x);
  slab(rdir.y, p0.y, p1.y, &t0.y, &t1.y);
  slab(rdir.z, p0.z, p1.z, &t0.z, &t1.z);
  *entry = QMAX(t0.x, QMAX(t0.y, t0.z));
  *exit_ = QMIN(t1.x, QMIN(t1.y, t1.z));
}

/* TriObj::IntersectRay + TraceBVHNode, src/objects/objects.cpp:310-420 */
static int mesh_intersect(const scene_t *s, int mi, const ray_t *ray, hit_t *hc, const diffray_t *dr,
                          diffhit_t *dh, tls_t *tl)
{
  const meshview_t mv = mesh_view(s, mi);
  const v3 drcp = vdiv(V3(1.f, 1.f, 1.f), ray->dir);
  {
    /* Box::IntersectRay, src/core/box.cpp:94-128 */
    float box[6] = {mv.m->bmin[0], mv.m->bmin[1], mv.m->bmin[2], mv.m->bmax[0], mv.m->bmax[1], mv.m->bmax[2]};
    float entry, exit_;
    box_entry_exit(ray->p, ray->dir, drcp, box, &entry, &exit_);
    if (entry > hc->z || entry > exit_) return 0;
  }
  if (mv.m->num_faces == 0) return 0;
  uint32_t stack[256];
  int sp = 0;
  int hasHit = 0;
  stack[sp++] = 1;
  while (sp != 0) {
    const uint32_t id = stack[--sp];
    const qa_bvh_node *n = &mv.nodes[id];
    tl->cnt.bvh_nodes++;
    if (n->data & QA_BVH_LEAF_BIT) {
      const uint32_t count = ((n->data >> QA_BVH_COUNT_SHIFT) & QA_BVH_COUNT_MASK) + 1;
      const uint32_t *el = mv.elements + (n->data & QA_BVH_OFFSET_MASK);
      for (uint32_t i = 0; i < count; ++i)
        if (triangle_intersect(&mv, ray, hc, el[i], dr, dh, tl)) hasHit = 1;
    } else {
      const uint32_t child0 = n->data & QA_BVH_CHILD_MASK, child1 = child0 + 1;
      float entry0, exit0, entry1, exit1;
      box_entry_exit(ray->p, ray->dir, drcp, mv.nodes[child0].box, &entry0, &exit0);
      box_entry_exit(ray->p, ray->dir, drcp, mv.nodes[child1].box, &entry1, &exit1);
      const float t_max = hc->z;
      const int hit0 = (entry0 < t_max && entry0 < exit0);
      const int hit1 = (entry1 < t_max && entry1 < exit1);
      if (hit0 && hit1) {
        if (entry0 < entry1) { stack[sp++] = child1; stack[sp++] = child0; }
        else { stack[sp++] = child0; stack[sp++] = child1; }
      } else if (hit0 && !hit1) stack[sp++] = child0;
      else if (hit1 && !hit0) stack[sp++] = child1;
      if (sp > 254) return hasHit; /* the reference's 40-entry stack would have overflowed long ago */
    }
  }
  return hasHit;
}

static int object_intersect(const scene_t *s, const qa_instance *in, const ray_t *ray, hit_t *hc,
                            const diffray_t *dr, diffhit_t *dh, tls_t *tl)
{
  switch (in->obj_type) {
    case QA_OBJ_SPHERE: return sphere_intersect(ray, hc, dr, dh);
    case QA_OBJ_PLANE: return plane_intersect(ray, hc, dr, dh);
    case QA_OBJ_MESH: return mesh_intersect(s, in->mesh, ray, hc, dr, dh, tl);
    default: return 0;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Scene graph                                                                                 */
/* ------------------------------------------------------------------------------------------ */
/* Node::ToNodeCoords(Ray), src/core/node.cpp:112-118 */
static inline ray_t to_node(const qa_instance *in, const ray_t *r)
{
  ray_t o;
  o.p = xform_to(in->itm, in->pos, r->p);
  o.dir = vsub(xform_to(in->itm, in->pos, vadd(r->p, r->dir)), o.p);
  return o;
}

/* Node::FromNodeCoords, src/core/node.cpp:127-139 */
static inline void from_node_core(const qa_instance *in, v3 *p, v3 *N)
{
  *p = vadd(m3mulv(in->tm, *p), v3p(in->pos));
  *N = vnormalize(m3tmulv(in->itm, *N));
}

/* Scene::TraceNodeShadow, src/scene/scene.cpp:35-46 */
static int trace_shadow_node(const scene_t *s, int k, const ray_t *ray, hit_t *hc, tls_t *tl)
{
  const qa_instance *in = &s->inst[k];
  ray_t nr = to_node(in, ray);
  if (in->obj_type != QA_OBJ_NONE) {
    if (object_intersect(s, in, &nr, hc, NULL, NULL, tl)) return 1;
  }
  for (int c = k + 1; c < in->subtree_end; c = s->inst[c].subtree_end)
    if (trace_shadow_node(s, c, &nr, hc, tl)) return 1;
  return 0;
}

/* Scene::TraceNodeNormal, src/scene/scene.cpp:50-74.  Node::ToNodeCoords(DiffRay) builds a fresh
 * DiffRay whose hasDiffRay member defaults to true (src/core/node.cpp:119-126, ray.h:55). */
static int trace_normal_node(const scene_t *s, int k, const diffray_t *ray, diffhit_t *h, tls_t *tl)
{
  const qa_instance *in = &s->inst[k];
  int hasHit = 0;
  diffray_t nr;
  nr.c = to_node(in, &ray->c);
  nr.x = to_node(in, &ray->x);
  nr.y = to_node(in, &ray->y);
  nr.hasDiffRay = 1;
  if (in->obj_type != QA_OBJ_NONE) {
    if (object_intersect(s, in, &nr.c, &h->c, &nr, h, tl)) {
      h->c.node = k;
      hasHit = 1;
    }
  }
  for (int c = k + 1; c < in->subtree_end; c = s->inst[c].subtree_end)
    if (trace_normal_node(s, c, &nr, h, tl)) hasHit = 1;
  if (hasHit) {
    from_node_core(in, &h->c.p, &h->c.N);
    from_node_core(in, &h->x.p, &h->x.N);
    from_node_core(in, &h->y.p, &h->y.N);
  }
  return hasHit;
}

static int trace_normal(const scene_t *s, const diffray_t *ray, diffhit_t *h, tls_t *tl)
{
  tl->cnt.casts_normal++;
  if (g_trace_on) fprintf(stderr, "  cast p %a %a %a  d %a %a %a\n", ray->c.p.x, ray->c.p.y, ray->c.p.z, ray->c.dir.x, ray->c.dir.y, ray->c.dir.z);
  return trace_normal_node(s, 0, ray, h, tl);
}

/* GenLight::Shadow, src/lights/lights.cpp:39-48 */
static float shadow(const scene_t *s, ray_t ray, float t_max, tls_t *tl)
{
  diffhit_t h;
  hit_init(&h);
  h.c.z = t_max;
  tl->cnt.casts_shadow++;
  return trace_shadow_node(s, 0, &ray, &h.c, tl) ? 0.0f : 1.0f;
}

/* ------------------------------------------------------------------------------------------ */
/* Lights                                                                                      */
/* ------------------------------------------------------------------------------------------ */
/* src/lights/lights.cpp:23-30 */
static inline float inverse_square_falloff(v3 v) { return QMIN(1.f, 1.f / vdot(v, v)); }

/* shared body of Point/SpotLight::Illuminate, src/lights/lights.cpp:50-74,85-106 */
static v3 point_illuminate(const scene_t *s, const qa_light *l, v3 p, tls_t *tl)
{
  const v3 position = v3p(l->position), intensity = v3p(l->intensity);
  if (l->size > 0.01f) {
    int spp = 16, n = 0;
    float inshadow = 0.0f;
    while (n < spp) {
      const v3 dir = vsub(vadd(position, uniform_ball(tl, l->size)), p);
      ray_t r; r.p = p; r.dir = vnormalize(dir);
      inshadow += (shadow(s, r, vlength(dir), tl) - inshadow) * inverse_square_falloff(dir) / (float) (n + 1);
      n++;
      if (inshadow > 0.f && inshadow < 1.f) spp = 64;
    }
    return vscale(intensity, inshadow);
  } else {
    const v3 dir = vsub(position, p);
    ray_t r; r.p = p; r.dir = vnormalize(dir);
    return vscale(vscale(intensity, shadow(s, r, vlength(dir), tl)), inverse_square_falloff(dir));
  }
}

/* SpotLight::GetAttenuation, src/lights/lights.cpp:128-143 */
static float spot_attenuation(const qa_light *l, v3 dir)
{
  const float cosTheta = vdot(dir, v3p(l->direction));
  if (cosTheta < 0) return 0;
  const float r = sqrtf(1.f - cosTheta * cosTheta) / cosTheta;
  if (r > l->outer) return 0;
  return r < l->inner ? 1.f : powf((l->outer - r) / (l->outer - l->inner), 2.f);
}

/* Light::Direction */
static v3 light_direction(const qa_light *l, v3 p)
{
  switch (l->type) {
    case QA_LIGHT_DIRECT: return v3p(l->direction);
    case QA_LIGHT_POINT:
    case QA_LIGHT_SPOT: return vnormalize(vsub(p, v3p(l->position)));
    default: return V3(0, 0, 0);
  }
}

/* Light::Illuminate */
static v3 light_illuminate(const scene_t *s, const qa_light *l, v3 p, tls_t *tl)
{
  switch (l->type) {
    case QA_LIGHT_DIRECT: {                      /* src/lights/lights.h:66-71 */
      ray_t r; r.p = p; r.dir = vnormalize(vneg(v3p(l->direction)));
      return vscale(v3p(l->intensity), shadow(s, r, QA_BIGFLOAT, tl));
    }
    case QA_LIGHT_POINT: return point_illuminate(s, l, p, tl);
    case QA_LIGHT_SPOT: {                        /* src/lights/lights.cpp:83-109 */
      v3 I = point_illuminate(s, l, p, tl);
      return vscale(I, spot_attenuation(l, light_direction(l, p)));
    }
    default: return v3p(l->intensity);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Material (MtlBlinn_PhotonMap, non-photon-map branch) + MultiMtl                             */
/* ------------------------------------------------------------------------------------------ */
static v3 shade(const scene_t *s, const diffray_t *ray, const diffhit_t *h, int bounce, tls_t *tl);

/* src/materials/materials.cpp:32-38 */
static v3 attenuation(const float *absorption, float l)
{
  return V3(expf(-absorption[0] * l), expf(-absorption[1] * l), expf(-absorption[2] * l));
}

/* ComputeSecondaryRay, src/materials/MtlBlinn_PhotonMap.cpp:226-254 */
static v3 secondary(const scene_t *s, const qa_material *m, v3 pos, v3 dir, v3 BxDF, float PDF,
                    int bounce, int hasDiffuseHit, tls_t *tl)
{
  diffray_t r;
  r.c.p = pos; r.c.dir = dir;
  r.x = r.c; r.y = r.c;
  r.hasDiffRay = 0;
  r.c.dir = vnormalize(r.c.dir);
  r.x.dir = vnormalize(r.x.dir);
  r.y.dir = vnormalize(r.y.dir);
  diffhit_t sh;
  hit_init(&sh);
  sh.c.hasDiffuseHit = hasDiffuseHit;
  v3 incoming;
  if (trace_normal(s, &r, &sh, tl)) {
    incoming = shade(s, &r, &sh, bounce - 1, tl);
    if (!sh.c.hasFrontHit) incoming = vmul(incoming, attenuation(m->absorption, sh.c.z));
  } else {
    incoming = sample_environment(s, &s->h->environment, r.c.dir);
  }
  return vdivs(vmul(incoming, BxDF), PDF);
}

/* ------------------------------------------------------------------------------------------ */
/* Photon maps (cy::PhotonMap, src/ext/cyPhotonMap.h)                                          */
/* ------------------------------------------------------------------------------------------ */
/* Photon::SetPower, cyPhotonMap.h:214-220 (Color24(255.f * c / power): float -> uchar truncation) */
static void photon_set_power(qa_photon *ph, v3 c)
{
  float power = c.x;
  if (power < c.y) power = c.y;
  if (power < c.z) power = c.z;
  ph->power = power;
  const v3 q = vdivs(vscale(c, 255.f), power);
  ph->rgb[0] = (uint8_t) (int) q.x;
  ph->rgb[1] = (uint8_t) (int) q.y;
  ph->rgb[2] = (uint8_t) (int) q.z;
}
/* Photon::SetDirection, cyPhotonMap.h:222-231 */
static void photon_set_direction(qa_photon *ph, v3 dir)
{
  ph->dirx = (int16_t) (dir.x * 0x7FFF);
  ph->diry = (int16_t) (dir.y * 0x7FFF);
  if (dir.z > 0) ph->plane_dirz &= 0x7;
  else ph->plane_dirz = (uint8_t) (0x8 | (ph->plane_dirz & 0x7));
}
/* Photon::GetDirection, cyPhotonMap.h:233-254: z from x only ("dirX*dirX + dirY - dirY"), by a
 * digit-by-digit integer square root */
static v3 photon_get_direction(const qa_photon *ph)
{
  v3 dir;
  dir.x = (float) ph->dirx / (float) 0x7FFF;
  dir.y = (float) ph->diry / (float) 0x7FFF;
  int xy2 = ph->dirx * ph->dirx + ph->diry - ph->diry;
  if (xy2 > 0x3FFF0001) xy2 = 0x3FFF0001;
  const int z2 = 0x3FFF0001 - xy2;
  int root = 0, bit = 0x40000000, rem = z2;
  while (bit > rem) bit >>= 2;
  while (bit) {
    if (rem >= root + bit) {
      rem = rem - root - bit;
      root = root + (bit << 1);
    }
    root >>= 1;
    bit >>= 2;
  }
  dir.z = (float) root / (float) 0x7FFF;
  if (ph->plane_dirz & 0x8) dir.z = -dir.z;
  return dir;
}
static inline float axis_of(const float *p, int axis) { return p[axis]; }

/* PhotonMap::BalanceSegment, cyPhotonMap.h:295-372: left-balanced kd-tree in heap order; the
 * median is found with the reference's own quick-select (its swaps decide how ties are split) */
static void balance_segment(qa_photon *photons, qa_photon *balanced, v3 boxMin, v3 boxMax, uint32_t index,
                            uint32_t start, uint32_t end)
{
  uint32_t median = 1;
  while ((4 * median) <= (end - start + 1)) median += median;
  if ((3 * median) <= (end - start + 1)) {
    median += median;
    median += start - 1;
  } else {
    median = end - median + 1;
  }
  int axis = 2;
  const v3 d = vsub(boxMax, boxMin);
  if (d.x > d.y) {
    if (d.x > d.z) axis = 0;
  } else if (d.y > d.z) axis = 1;

  uint32_t left = start, right = end;
  while (right > left) {
    const float v = photons[right].pos[axis];
    uint32_t i = left - 1, j = right;
    while (photons[++i].pos[axis] < v) {}
    while (photons[--j].pos[axis] > v && j > left) {}
    while (i < j) {
      qa_photon t = photons[i]; photons[i] = photons[j]; photons[j] = t;
      while (photons[++i].pos[axis] < v) {}
      while (photons[--j].pos[axis] > v && j > left) {}
    }
    { qa_photon t = photons[i]; photons[i] = photons[right]; photons[right] = t; }
    if (i >= median) right = i - 1;
    if (i <= median) left = i + 1;
  }
  balanced[index] = photons[median];
  balanced[index].plane_dirz = (uint8_t) ((balanced[index].plane_dirz & 0x8) | (axis & 0x3));
  if (median > start) {
    if (start < median - 1) {
      v3 tmax = boxMax;
      ((float *) &tmax)[axis] = balanced[index].pos[axis];
      balance_segment(photons, balanced, boxMin, tmax, 2 * index, start, median - 1);
    } else {
      balanced[2 * index] = photons[start];
    }
  }
  if (median < end) {
    if (median + 1 < end) {
      v3 tmin = boxMin;
      ((float *) &tmin)[axis] = balanced[index].pos[axis];
      balance_segment(photons, balanced, tmin, boxMax, 2 * index + 1, median + 1, end);
    } else {
      balanced[2 * index + 1] = photons[end];
    }
  }
}

/* PhotonMap::PrepareForIrradianceEstimation, cyPhotonMap.h:272-292.  photons: count+1 records,
 * [0] is the value-initialised dummy the reference keeps (all zero: it takes part in the box). */
static int balance(qa_photon *photons, uint32_t count)
{
  v3 bmin = v3p(photons[0].pos), bmax = bmin;
  for (uint32_t i = 1; i <= count; ++i) {
    const float *q = photons[i].pos;
    if (bmin.x > q[0]) bmin.x = q[0];
    if (bmax.x < q[0]) bmax.x = q[0];
    if (bmin.y > q[1]) bmin.y = q[1];
    if (bmax.y < q[1]) bmax.y = q[1];
    if (bmin.z > q[2]) bmin.z = q[2];
    if (bmax.z < q[2]) bmax.z = q[2];
  }
  qa_photon *balanced = (qa_photon *) calloc((size_t) count + 1, sizeof(qa_photon));
  if (!balanced) return -1;
  balance_segment(photons, balanced, bmin, bmax, 1, 1, count);
  memcpy(photons, balanced, ((size_t) count + 1) * sizeof(qa_photon));
  free(balanced);
  return 0;
}

typedef struct {   /* PhotonMap::NearestPhotons, cyPhotonMap.h:189-198 (photon copies -> indices) */
  v3 pos, normal;
  int found;
  float dist2[QA_PHOTON_GATHER + 1];
  uint32_t photon[QA_PHOTON_GATHER + 1];
} nearest_t;

/* PhotonMap::LocatePhotons, cyPhotonMap.h:437-501 (normal given, ellipticity 1 => normScale 0) */
static void locate_photons(const qa_photon *photons, int half, nearest_t *np, int index)
{
  const qa_photon *p = &photons[index];
  const int axis = p->plane_dirz & 0x3;
  if (index < half) {
    const float dist = ((const float *) &np->pos)[axis] - p->pos[axis];
    if (dist > 0) {
      locate_photons(photons, half, np, 2 * index + 1);
      if (dist * dist < np->dist2[0]) locate_photons(photons, half, np, 2 * index);
    } else {
      locate_photons(photons, half, np, 2 * index);
      if (dist * dist < np->dist2[0]) locate_photons(photons, half, np, 2 * index + 1);
    }
  }
  const v3 dif = vsub(v3p(p->pos), np->pos);
  const float dist2 = vdot(dif, dif);
  if (dist2 < np->dist2[0]) {
    const v3 dir = photon_get_direction(p);
    if (vdot(dir, np->normal) >= 0) return;
    if (np->found < QA_PHOTON_GATHER) {
      np->found++;
      np->dist2[np->found] = dist2;
      np->photon[np->found] = (uint32_t) index;
      if (np->found == QA_PHOTON_GATHER) {  /* build the max-heap */
        const int half_found = np->found >> 1;
        for (int k = half_found; k >= 1; k--) {
          int parent = k;
          const uint32_t tp = np->photon[k];
          const float td2 = np->dist2[k];
          while (parent <= half_found) {
            int j = parent + parent;
            if (j < np->found && np->dist2[j] < np->dist2[j + 1]) j++;
            if (td2 >= np->dist2[j]) break;
            np->dist2[parent] = np->dist2[j];
            np->photon[parent] = np->photon[j];
            parent = j;
          }
          np->photon[parent] = tp;
          np->dist2[parent] = td2;
        }
      }
    } else {
      int parent = 1, j = 2;
      while (j <= np->found) {
        if (j < np->found && np->dist2[j] < np->dist2[j + 1]) j++;
        if (dist2 > np->dist2[j]) break;
        np->dist2[parent] = np->dist2[j];
        np->photon[parent] = np->photon[j];
        parent = j;
        j <<= 1;
      }
      np->photon[parent] = (uint32_t) index;
      np->dist2[parent] = dist2;
      np->dist2[0] = np->dist2[1];
    }
  }
}

/* PhotonMap::EstimateIrradiance<100>(..., &N, 1.f, FILTER_TYPE_QUADRATIC), cyPhotonMap.h:375-433 */
static void estimate_irradiance(const scene_t *s, int which, v3 pos, v3 N, v3 *irrad, v3 *direction)
{
  *irrad = V3(0, 0, 0);
  *direction = V3(0, 0, 0);
  const qa_photon *photons = s->pm[which].photons;
  nearest_t np;
  np.pos = pos;
  np.normal = N;
  np.found = 0;
  np.dist2[0] = s->pm[which].radius * s->pm[which].radius;
  locate_photons(photons, s->pm[which].half, &np, 1);
  for (int i = 1; i <= np.found; i++) {
    const qa_photon *ph = &photons[np.photon[i]];
    const v3 power = vscale(V3(ph->rgb[0] / 255.0f, ph->rgb[1] / 255.0f, ph->rgb[2] / 255.0f), ph->power);
    const float filter = 1 - np.dist2[i] / np.dist2[0];
    *irrad = vadd(*irrad, vscale(power, filter));
    const v3 dir = photon_get_direction(ph);
    *direction = vadd(*direction, vscale(dir, filter * ph->power));
  }
  if (np.found > 0) {
    const float area = ((float) M_PI * 0.5f) * np.dist2[0];
    if (area > 0) {
      const float one_over_area = 1.0f / area;
      *irrad = vscale(*irrad, one_over_area);
    }
    *direction = vnormalize(*direction);
  }
}

/* the gather term of Shade, MtlBlinn_PhotonMap.cpp:426-458 */
static v3 gather(const scene_t *s, int which, v3 p, v3 N, v3 V, v3 kd, v3 ks, float gloss)
{
  v3 I, D;
  estimate_irradiance(s, which, p, N, &I, &D);
  if (luma(I) > 0.00001f) {
    const v3 L = vneg(vnormalize(D));
    const v3 H = vnormalize(vadd(V, L));
    const float cosNL = QMAX(0.f, vdot(N, L));
    const float cosNH = QMAX(0.f, vdot(N, H));
    return vmul(vscale(I, cosNL), vadd(kd, vscale(ks, powf(cosNH, gloss))));
  }
  return V3(0, 0, 0);
}

/* MtlBlinn_PhotonMap::Shade, src/materials/MtlBlinn_PhotonMap.cpp:256-500 */
static v3 shade_blinn(const scene_t *s, const qa_material *m, const diffray_t *ray, const diffhit_t *h,
                      int bounce, tls_t *tl)
{
  v3 color = mtl_sample(s, h, &m->emission);
  const v3 V = vneg(ray->c.dir);
  const v3 N = h->c.N;
  const v3 Y = vdot(N, V) > 0.f ? N : vneg(N);
  const v3 p = h->c.p;
  /* ComputeFresnel, :65-105 */
  v3 tDir, rDir;
  float tC, rC;
  int totReflection;
  {
    const v3 Z = vcross(V, Y);
    const v3 X = vnormalize(vcross(Y, Z));
    const float nIOR = h->c.hasFrontHit ? 1.f / m->ior : m->ior;
    const float cosI = vdot(N, V);
    const float sinI = sqrtf(1 - cosI * cosI);
    const float sinO = QMAX(0.f, QMIN(1.f, sinI * nIOR));
    const float cosO = sqrtf(1.f - sinO * sinO);
    tDir = vsub(vscale(vneg(X), sinO), vscale(Y, cosO));
    rDir = vsub(vscale(vscale(N, 2.f), vdot(N, V)), V);
    totReflection = (nIOR * sinI) > 1.001f;
    const float C = (nIOR - 1.f) * (nIOR - 1.f) / ((nIOR + 1.f) * (nIOR + 1.f));
    rC = C + (1.f - C) * powf(1.f - QABS(cosI), 5.f);
    tC = 1.f - rC;
  }
  const v3 tK = mtl_sample(s, h, &m->refraction);
  const v3 rK = mtl_sample(s, h, &m->reflection);
  const v3 sampleTransmission = totReflection ? V3(0, 0, 0) : vscale(tK, tC);
  const v3 sampleReflection = totReflection ? vadd(rK, tK) : vadd(rK, vscale(tK, rC));
  const v3 sampleSpecular = mtl_sample(s, h, &m->specular);
  const v3 sampleDiffuse = mtl_sample(s, h, &m->diffuse);
  /* RandomSelectMtl, :107-150 */
  enum { TRANSMIT, REFLECT, DIFFUSE, ABSORB } select;
  {
    const float lumaT = luma(sampleTransmission), lumaR = luma(sampleReflection), lumaD = luma(sampleDiffuse);
    const float r = rng1(tl);
    const float coefTransmit = lumaT;
    const float coefReflection = coefTransmit + lumaR;
    const float coefDiffuse = coefReflection + lumaD;
    const float coefSum = coefDiffuse + m->kill;
    const float sel = r * coefSum;
    if (sel < coefTransmit && lumaT > 0.00001f) select = TRANSMIT;
    else if (sel < coefReflection && lumaR > 0.00001f) select = REFLECT;
    else if (sel < coefDiffuse && lumaD > 0.00001f) select = DIFFUSE;
    else select = ABSORB;
  }
  const int doReflect = select == REFLECT;
  const int doTransmit = select == TRANSMIT;
  int doDiffuse = 0, doGatherPhoton = 0, doGatherCaustics = 0;
  if (select == DIFFUSE) {
    if (!h->c.hasDiffuseHit) doDiffuse = 1;
    if (s->use_pm) {                     /* :349-359 */
      doGatherPhoton = h->c.hasDiffuseHit;
      doGatherCaustics = 1;
    }
  }
  if (bounce > 0) {
    if (luma(sampleReflection) > 0.00001f) {
      if (doReflect) {
        /* SampleReflectionBxDF, :175-198 */
        v3 sampleDir;
        if (m->gloss_refl > 0.f) {
          do {
            sampleDir = vnormalize(vadd(vnormalize(rDir), uniform_ball(tl, 2.f * m->gloss_refl)));
          } while (vdot(sampleDir, Y) < 0);
        } else sampleDir = rDir;
        color = vadd(color, secondary(s, m, p, sampleDir, sampleReflection, 1.f, bounce, 0, tl));
      }
    }
    if (select == TRANSMIT && luma(sampleTransmission) > 0.00001f) {
      if (doTransmit) {
        /* SampleTransmitBxDF, :152-174 */
        v3 sampleDir;
        if (m->gloss_refr > 0.f) {
          do {
            sampleDir = vnormalize(vadd(vnormalize(tDir), uniform_ball(tl, 2.f * m->gloss_refr)));
          } while (vdot(sampleDir, Y) > 0);
        } else sampleDir = tDir;
        color = vadd(color, secondary(s, m, p, sampleDir, sampleTransmission, 1.f, bounce, 0, tl));
      }
    }
  }
  if (luma(sampleDiffuse) > 0.00001f) {
    if (doGatherPhoton) color = vadd(color, gather(s, 0, p, N, V, sampleDiffuse, sampleSpecular, m->gloss_spec));
    if (doGatherCaustics) color = vadd(color, gather(s, 1, p, N, V, sampleDiffuse, sampleSpecular, m->gloss_spec));
    if (bounce > 0) {
      if (doDiffuse) {
        if (h->c.hasFrontHit) {
          /* SampleDiffuseBxDF, :199-224 */
          const v3 sampleDir = to_local_frame(N, cos_weighted_hemisphere(tl));
          const v3 L = vnormalize(sampleDir);
          const v3 H = vnormalize(vadd(V, L));
          const float cosNH = QMAX(0.f, vdot(N, H));
          const v3 BxDF = vadd(sampleDiffuse, vscale(sampleSpecular, powf(cosNH, m->gloss_spec)));
          color = vadd(color, vscale(secondary(s, m, p, sampleDir, BxDF, 1.f, bounce, 1, tl), 1.f / 1));
        }
      }
    }
  }
  /* direct lighting, :481-498 */
  {
    const int nl = (int) s->h->num_lights;
    const float normCoefDI = (nl == 0 ? 1.f : 1.f / nl);
    for (int li = 0; li < nl; ++li) {
      const qa_light *l = &s->light[li];
      if (l->type == QA_LIGHT_AMBIENT) continue;
      const v3 intensity = vscale(light_illuminate(s, l, p, tl), normCoefDI);
      const v3 L = vnormalize(vneg(light_direction(l, p)));
      const v3 H = vnormalize(vadd(V, L));
      const float cosNL = QMAX(0.f, vdot(N, L));
      const float cosNH = QMAX(0.f, vdot(N, H));
      color = vadd(color, vmul(vscale(intensity, cosNL),
                               vadd(sampleDiffuse, vscale(sampleSpecular, powf(cosNH, m->gloss_spec)))));
    }
  }
  return color;
}

/* Node::GetMaterial()->Shade with MultiMtl dispatch, src/materials/materials.h:70-76 */
static v3 shade(const scene_t *s, const diffray_t *ray, const diffhit_t *h, int bounce, tls_t *tl)
{
  const qa_instance *in = &s->inst[h->c.node];
  if (in->mtlset < 0) return V3(0, 0, 0); /* the reference dereferences a null Material here */
  const qa_mtlset *ms = &s->mtlset[in->mtlset];
  if (ms->multi) {
    if (h->c.mtlID < ms->count && h->c.mtlID >= 0) return shade_blinn(s, &s->mtl[ms->first + h->c.mtlID], ray, h, bounce, tl);
    return V3(1, 1, 1);
  }
  return shade_blinn(s, &s->mtl[ms->first], ray, h, bounce, tl);
}

/* ------------------------------------------------------------------------------------------ */
/* Photon tracing                                                                              */
/* ------------------------------------------------------------------------------------------ */
static int scene_bind(scene_t *s, const void *blob);

/* Sampler::UniformSphere / UniformHemisphere, src/core/sampler.cpp:55-85 */
static v3 uniform_sphere(tls_t *t)
{
  float r1 = rng1(t), r2 = rng1(t);
  r1 = r1 * 2.f - 1.f;
  const float cosTheta = r1;
  const float sinTheta = sqrtf(1 - r1 * r1);
  const float phi = 2 * kPI * r2;
  return V3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
}
static v3 uniform_hemisphere(tls_t *t)
{
  float r1 = rng1(t), r2 = rng1(t);
  const float cosTheta = r1;
  const float sinTheta = sqrtf(1 - r1 * r1);
  const float phi = 2 * kPI * r2;
  return V3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
}

/* Material of a hit for the photon extensions: 0 = none, 1 = MtlBlinn (*out), 2 = MultiMtl, which
 * inherits Material::IsPhotonSurface (true) and RandomPhotonBounce (false), src/core/material.h:53-63 */
static int photon_material(const scene_t *s, const diffhit_t *h, const qa_material **out)
{
  const qa_instance *in = &s->inst[h->c.node];
  if (in->mtlset < 0) return 0;
  const qa_mtlset *ms = &s->mtlset[in->mtlset];
  if (ms->multi) return 2;
  *out = &s->mtl[ms->first];
  return 1;
}

/* MtlBlinn_PhotonMap::RandomPhotonBounce, src/materials/MtlBlinn_PhotonMap.cpp:503-578 */
static int photon_bounce(const scene_t *s, const qa_material *m, diffray_t *ray, v3 *c, const diffhit_t *h, tls_t *tl)
{
  (void) mtl_sample(s, h, &m->emission);
  const v3 V = vneg(ray->c.dir);
  const v3 N = h->c.N;
  const v3 Y = vdot(N, V) > 0.f ? N : vneg(N);
  v3 tDir, rDir;
  float tC, rC;
  int totReflection;
  {
    const v3 Z = vcross(V, Y);
    const v3 X = vnormalize(vcross(Y, Z));
    const float nIOR = h->c.hasFrontHit ? 1.f / m->ior : m->ior;
    const float cosI = vdot(N, V);
    const float sinI = sqrtf(1 - cosI * cosI);
    const float sinO = QMAX(0.f, QMIN(1.f, sinI * nIOR));
    const float cosO = sqrtf(1.f - sinO * sinO);
    tDir = vsub(vscale(vneg(X), sinO), vscale(Y, cosO));
    rDir = vsub(vscale(vscale(N, 2.f), vdot(N, V)), V);
    totReflection = (nIOR * sinI) > 1.001f;
    const float C = (nIOR - 1.f) * (nIOR - 1.f) / ((nIOR + 1.f) * (nIOR + 1.f));
    rC = C + (1.f - C) * powf(1.f - QABS(cosI), 5.f);
    tC = 1.f - rC;
  }
  const v3 tK = mtl_sample(s, h, &m->refraction);
  const v3 rK = mtl_sample(s, h, &m->reflection);
  const v3 sampleTransmission = totReflection ? V3(0, 0, 0) : vscale(tK, tC);
  const v3 sampleReflection = totReflection ? vadd(rK, tK) : vadd(rK, vscale(tK, rC));
  const v3 sampleDiffuse = mtl_sample(s, h, &m->diffuse);
  const v3 sampleSpecular = mtl_sample(s, h, &m->specular);
  /* RandomSelectMtl with its scale output, :107-150 */
  const float lumaT = luma(sampleTransmission), lumaR = luma(sampleReflection), lumaD = luma(sampleDiffuse);
  const float r = rng1(tl);
  const float coefTransmit = lumaT;
  const float coefReflection = coefTransmit + lumaR;
  const float coefDiffuse = coefReflection + lumaD;
  const float coefAbsorb = coefDiffuse + m->kill;
  const float rcpCoefSum = 1.f / coefAbsorb;
  const float sel = r * coefAbsorb;
  v3 sampleDir = V3(0, 0, 0), BxDF = V3(0, 0, 0);
  float PDF = 1.f, scale;
  int doShade = 0;
  if (sel < coefTransmit && lumaT > 0.00001f) {
    scale = lumaT * rcpCoefSum;
    if (m->gloss_refr > 0.f) {
      do {
        sampleDir = vnormalize(vadd(vnormalize(tDir), uniform_ball(tl, 2.f * m->gloss_refr)));
      } while (vdot(sampleDir, Y) > 0);
    } else sampleDir = tDir;
    BxDF = sampleTransmission;
    doShade = 1;
  } else if (sel < coefReflection && lumaR > 0.00001f) {
    scale = lumaR * rcpCoefSum;
    if (m->gloss_refl > 0.f) {
      do {
        sampleDir = vnormalize(vadd(vnormalize(rDir), uniform_ball(tl, 2.f * m->gloss_refl)));
      } while (vdot(sampleDir, Y) < 0);
    } else sampleDir = rDir;
    BxDF = sampleReflection;
    doShade = 1;
  } else if (sel < coefDiffuse && lumaD > 0.00001f) {
    scale = lumaD * rcpCoefSum;
    if (h->c.hasFrontHit) {
      /* SampleDiffuseBxDF(..., photonMap = true), :203-224 */
      sampleDir = to_local_frame(N, uniform_hemisphere(tl));
      const v3 L = vnormalize(sampleDir);
      const v3 H = vnormalize(vadd(V, L));
      const float cosNH = QMAX(0.f, vdot(N, H));
      BxDF = vadd(sampleDiffuse, vscale(sampleSpecular, powf(cosNH, m->gloss_spec)));
      PDF = 0.5f;
      doShade = 1;
    }
  } else {
    scale = coefAbsorb * rcpCoefSum;
  }
  if (!doShade) return 0;
  /* DiffRay(p, dir).Normalize() here and ray.Normalize() once more in the emission loop (renderer.cpp:183,253) */
  ray->c.p = h->c.p; ray->c.dir = vnormalize(vnormalize(sampleDir));
  ray->x = ray->c; ray->y = ray->c;
  ray->hasDiffRay = 0;
  *c = vdivs(vmul(*c, BxDF), PDF * scale);
  if (!h->c.hasFrontHit) *c = vmul(*c, attenuation(m->absorption, h->c.z));
  return 1;
}

/* One iteration of the emission loops of Renderer::ComputeScene (src/renderers/renderer.cpp:146-197
 * photon map, :217-271 caustics map) on the emission's own stream (include/qa_photon.h).
 * Writes the photons this emission would store, in order, to out[0..]; returns their number
 * (at most bounce - 1). */
static int emit_photon(const scene_t *s, int caustics, uint32_t max_bounce, const int *photonLights, int numPhotonLights,
                       uint32_t seed, uint32_t emission, qa_photon *out, tls_t *tl)
{
  tl->rng = qa_photon_seed(seed, caustics ? QA_STREAM_CAUSTICS : QA_STREAM_PHOTON, emission);
  const float lightScale = 1.f / (float) numPhotonLights;
  const qa_light *light;
  if (numPhotonLights == 1) light = &s->light[photonLights[0]];
  else {
    const float r = rng1(tl);
    size_t id;
    if (!caustics) {
      const float fl = floorf(r * (float) (size_t) numPhotonLights);
      const float lim = (float) (size_t) (numPhotonLights - 1);
      id = (size_t) (fl < lim ? fl : lim);
    } else {
      const size_t ce = (size_t) ceilf(r * (float) (size_t) numPhotonLights);
      id = ce < (size_t) (numPhotonLights - 1) ? ce : (size_t) (numPhotonLights - 1);
    }
    light = &s->light[photonLights[id]];
  }
  /* PointLight::RandomPhoton, src/lights/lights.cpp:76-80 */
  diffray_t ray;
  ray.c.p = v3p(light->position);
  ray.c.dir = vnormalize(uniform_sphere(tl));
  ray.x = ray.c; ray.y = ray.c;
  ray.hasDiffRay = 0;
  diffhit_t h;
  hit_init(&h);
  v3 intensity = vscale(v3p(light->intensity), lightScale);
  int stored = 0;
  uint32_t bounce = 0;
  while (bounce < max_bounce) {
    if (!trace_normal(s, &ray, &h, tl)) break;
    const qa_material *m = NULL;
    const int kind = photon_material(s, &h, &m);
    if (kind == 0) break;  /* the reference dereferences a null Material here */
    const int photonSurface = kind == 2 ? 1 : (luma(v3p(m->diffuse.color)) > 0);
    if (photonSurface && bounce != 0 && !(caustics && h.c.hasDiffuseHit)) {
      qa_photon *ph = &out[stored++];
      memset(ph, 0, sizeof(*ph));
      ph->pos[0] = h.c.p.x; ph->pos[1] = h.c.p.y; ph->pos[2] = h.c.p.z;
      photon_set_direction(ph, ray.c.dir);
      photon_set_power(ph, intensity);
    }
    if (kind != 1 || !photon_bounce(s, m, &ray, &intensity, &h, tl)) break;
    const int diffuseHit = h.c.hasDiffuseHit;
    ++bounce;
    hit_init(&h);
    if (caustics) h.c.hasDiffuseHit = (diffuseHit || photonSurface);
  }
  return stored;
}

/* Both maps of Renderer::ComputeScene.  photon / caustics: size + 1 records each ([0] = dummy).
 * emitted[2]: numOfEmittedRays of each map; emissions[2]: loop iterations until the map was full.
 * Returns 0; -3 when the scene has no photon source (the reference divides by zero and reads an
 * uninitialised light there); -4 when a map is not full after QA_PHOTON_MAX_EMISSIONS emissions (the
 * reference loops forever). */
int qa_oracle_photon_build(const void *blob, const qa_photon_params *pp, uint32_t seed, qa_photon *photon,
                           qa_photon *caustics, uint64_t emitted[2], uint64_t emissions[2])
{
  scene_t s;
  if (scene_bind(&s, blob) != 0) return -1;
  memset(s.pm, 0, sizeof(s.pm));
  s.use_pm = 0;
  s.max_bounce = 0;
  int lights[256], nl = 0;
  for (uint32_t i = 0; i < s.h->num_lights && nl < 256; ++i)
    if (s.light[i].type == QA_LIGHT_POINT) lights[nl++] = (int) i;   /* IsPhotonSource, lights.h:114,156 */
  if (nl == 0) return -3;
  tls_t tl;
  memset(&tl, 0, sizeof(tl));
  for (int which = 0; which < 2; ++which) {
    const qa_photon_map_params *mp = which ? &pp->caustics : &pp->photon;
    qa_photon *map = which ? caustics : photon;
    memset(map, 0, ((size_t) mp->size + 1) * sizeof(qa_photon));
    qa_photon *tmp = (qa_photon *) malloc(((size_t) mp->bounce + 1) * sizeof(qa_photon));
    uint64_t recorded = 0, numEmitted = 0, e = 0;
    int finished = 0;
    while (!finished) {
      if (e >= QA_PHOTON_MAX_EMISSIONS(mp->size)) { free(tmp); return -4; }
      const int n = emit_photon(&s, which, mp->bounce, lights, nl, seed, (uint32_t) e, tmp, &tl);
      int any = 0;
      for (int k = 0; k < n; ++k) {
        if (recorded >= mp->size) { finished = 1; break; }
        map[1 + recorded++] = tmp[k];
        any = 1;
      }
      if (any) ++numEmitted;
      ++e;
    }
    free(tmp);
    /* ScalePhotonPowers(1.f / numOfEmittedRays), cyPhotonMap.h:128-132 */
    const float scale = 1.f / (float) (uint32_t) numEmitted;
    for (uint32_t i = 1; i <= mp->size; ++i) map[i].power *= scale;
    if (balance(map, mp->size) != 0) return -5;
    emitted[which] = numEmitted;
    emissions[which] = e;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Pixel loop                                                                                  */
/* ------------------------------------------------------------------------------------------ */
/* Renderer::PixelRender, src/renderers/renderer.cpp:302-346 + SuperSamplerHalton, src/scene/scene.cpp:83-123 */
static void render_pixel(const scene_t *s, int i, int j, int spp_min, int spp_max, uint32_t seed,
                         float *rgb, float *depthOut, uint32_t *nsOut, tls_t *tl)
{
  const qa_flat_header *h = s->h;
  const v3 screenA = v3p(h->screenA), screenU = v3p(h->screenU), screenV = v3p(h->screenV);
  const v3 th = V3(0.005f, 0.001f, 0.005f);
  v3 color = V3(0, 0, 0), color_std = V3(0, 0, 0);
  int sidx = 0;
  float depth = 0.0f;
  tl->rng = qa_pixel_seed(seed, (uint32_t) j * h->width + (uint32_t) i);
  while (sidx < spp_min ||
         (sidx >= spp_min && sidx < spp_max && (color_std.x > th.x || color_std.y > th.y || color_std.z > th.z))) {
    const v3 texpos = vadd(V3(qa_oracle_halton(sidx, 11), qa_oracle_halton(sidx, 13), 0.f), V3((float) i, (float) j, 0.f));
    const v3 cpt = vadd(vadd(screenA, vscale(screenU, texpos.x)), vscale(screenV, texpos.y));
    const v3 xpt = vadd(vadd(screenA, vscale(screenU, texpos.x + kDX)), vscale(screenV, texpos.y));
    const v3 ypt = vadd(vadd(screenA, vscale(screenU, texpos.x)), vscale(screenV, texpos.y + kDX));
    v3 campos = v3p(h->cam_pos);
    if (h->dof > 0.1f) {
      /* SuperSamplerHalton::NewDofSample, src/scene/scene.cpp:104-111 */
      float r1 = rng1(tl), r2 = rng1(tl);
      const float r = h->dof * sqrtf(r1);
      const float t = r2 * 2.f * kPI;
      const v3 ds = V3(r * cosf(t), r * sinf(t), 0.f);
      campos = vadd(campos, vadd(vscale(v3p(h->screenX), ds.x), vscale(v3p(h->screenY), ds.y)));
    }
    g_trace_on = (i == g_trace_px && j == g_trace_py && sidx == g_trace_s);
    if (g_trace_on) fprintf(stderr, "pixel %d %d sample %d\n", i, j, sidx);
    diffray_t ray;
    ray.c.p = campos; ray.c.dir = vnormalize(vsub(cpt, campos));
    ray.x.p = campos; ray.x.dir = vnormalize(vsub(xpt, campos));
    ray.y.p = campos; ray.y.dir = vnormalize(vsub(ypt, campos));
    ray.hasDiffRay = 1;
    diffhit_t hit;
    hit_init(&hit);
    hit.c.z = QA_BIGFLOAT;
    const int hasHit = trace_normal(s, &ray, &hit, tl);
    v3 local;
    if (hasHit) {
      local = shade(s, &ray, &hit, s->max_bounce, tl);
    } else {
      const float u = texpos.x / (float) h->width;
      const float v = texpos.y / (float) h->height;
      local = texcolor_sample(s, &h->background, V3(u, v, 0.f));
    }
    if (sidx == 0) depth = hasHit ? hit.c.z : QA_BIGFLOAT;
    /* Accumulate, src/scene/scene.cpp:113-121 */
    {
      const v3 dc = vdivs(vsub(local, color), (float) (sidx + 1));
      color = vadd(color, dc);
      if (sidx > 0)
        color_std = vadd(color_std, vsub(vscale(vmul(dc, dc), (float) (sidx + 1)), vdivs(color_std, (float) sidx)));
      else
        color_std = vadd(color_std, V3(0.0f, 0.0f, 0.0f));
    }
    g_trace_on = 0;
    ++sidx;
    tl->cnt.samples++;
  }
  rgb[0] = color.x; rgb[1] = color.y; rgb[2] = color.z;
  *depthOut = depth;
  *nsOut = (uint32_t) sidx;
}

static int scene_bind(scene_t *s, const void *blob)
{
  const qa_flat_header *h = (const qa_flat_header *) blob;
  if (h->magic != QA_FLAT_MAGIC || h->version != QA_FLAT_VERSION) return -1;
  s->blob = (const unsigned char *) blob;
  s->h = h;
  s->inst = QA_BLOB_PTR(qa_instance, blob, h->off_instances);
  s->mesh = QA_BLOB_PTR(qa_mesh, blob, h->off_meshes);
  s->mtlset = QA_BLOB_PTR(qa_mtlset, blob, h->off_mtlsets);
  s->mtl = QA_BLOB_PTR(qa_material, blob, h->off_materials);
  s->light = QA_BLOB_PTR(qa_light, blob, h->off_lights);
  s->texmap = QA_BLOB_PTR(qa_texmap, blob, h->off_texmaps);
  s->tex = QA_BLOB_PTR(qa_texture, blob, h->off_textures);
  return 0;
}

int qa_oracle_render(const void *blob, int x0, int y0, int x1, int y1, int spp_min, int spp_max,
                     int max_bounce, uint32_t seed, float *rgb, float *depth, uint32_t *ns,
                     int threads, qa_oracle_counters *counters)
{
  return qa_oracle_render_pm(blob, x0, y0, x1, y1, spp_min, spp_max, max_bounce, seed, rgb, depth, ns, threads,
                             counters, NULL, NULL, NULL);
}

int qa_oracle_render_pm(const void *blob, int x0, int y0, int x1, int y1, int spp_min, int spp_max,
                        int max_bounce, uint32_t seed, float *rgb, float *depth, uint32_t *ns,
                        int threads, qa_oracle_counters *counters, const qa_photon_params *pp,
                        const qa_photon *photon, const qa_photon *caustics)
{
  scene_t s;
  if (scene_bind(&s, blob) != 0) return -1;
  memset(s.pm, 0, sizeof(s.pm));
  s.use_pm = pp != NULL;
  if (pp) {
    /* halfStoredPhotons = (photons.size() - 1) / 2 - 1, cyPhotonMap.h:291 */
    s.pm[0].photons = photon;   s.pm[0].count = pp->photon.size;   s.pm[0].radius = pp->photon.radius;
    s.pm[0].half = (int) (pp->photon.size / 2) - 1;
    s.pm[1].photons = caustics; s.pm[1].count = pp->caustics.size; s.pm[1].radius = pp->caustics.radius;
    s.pm[1].half = (int) (pp->caustics.size / 2) - 1;
  }
  if (x0 < 0 || y0 < 0 || x1 > (int) s.h->width || y1 > (int) s.h->height || x1 < x0 || y1 < y0) return -2;
  s.max_bounce = max_bounce;
  if (getenv("QA_ORACLE_TRACE")) {
    if (sscanf(getenv("QA_ORACLE_TRACE"), "%d,%d,%d", &g_trace_px, &g_trace_py, &g_trace_s) == 3) threads = 1;
  }
  const int cw = x1 - x0, ch = y1 - y0;
  qa_oracle_counters total;
  memset(&total, 0, sizeof(total));
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#else
  threads = 1;
#endif
#pragma omp parallel num_threads(threads)
  {
    tls_t tl;
    memset(&tl, 0, sizeof(tl));
#pragma omp for schedule(dynamic, 16)
    for (int q = 0; q < cw * ch; ++q) {
      const int i = x0 + q % cw, j = y0 + q / cw;
      render_pixel(&s, i, j, spp_min, spp_max, seed, rgb + 3 * (size_t) q, depth + q, ns + q, &tl);
    }
#pragma omp critical
    {
      total.samples += tl.cnt.samples;
      total.casts_normal += tl.cnt.casts_normal;
      total.casts_shadow += tl.cnt.casts_shadow;
      total.bvh_nodes += tl.cnt.bvh_nodes;
      total.tri_tests += tl.cnt.tri_tests;
    }
  }
  if (counters) *counters = total;
  return 0;
}
