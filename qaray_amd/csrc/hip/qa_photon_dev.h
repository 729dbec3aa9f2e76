// qa_photon_dev.h — device side of the photon / caustics maps (-use-photon-map): the gather
// (cy::PhotonMap::EstimateIrradiance<100>, src/ext/cyPhotonMap.h:375-501, as called from
// MtlBlinn_PhotonMap::Shade, src/materials/MtlBlinn_PhotonMap.cpp:426-458) and the pieces of
// photon tracing that run per hit (Photon::SetPower / SetDirection / GetDirection,
// cyPhotonMap.h:214-254).  Included by qa_kernel.h; the emission kernel lives in qa_photon.hip.
#pragma once
#include "qa_device_math.h"
#include "qa_scene_dev.h"

namespace qa {

// Photon::SetPower + SetDirection packed into the record's last three dwords
__device__ __forceinline__ void photonPack(f3 power, f3 dir, float &maxPower, uint32_t &w4, uint32_t &w5)
{
  float pw = power.x;
  if (pw < power.y) pw = power.y;
  if (pw < power.z) pw = power.z;
  maxPower = pw;
  const f3 q = (power * 255.f) / pw;
  const uint32_t r = (uint32_t) (int) q.x & 0xFFu, g = (uint32_t) (int) q.y & 0xFFu, b = (uint32_t) (int) q.z & 0xFFu;
  const uint32_t zbit = dir.z > 0 ? 0u : 0x8u;
  w4 = r | (g << 8) | (b << 16) | (zbit << 24);
  const int dx = (int) (dir.x * (float) 0x7FFF), dy = (int) (dir.y * (float) 0x7FFF);
  w5 = ((uint32_t) dx & 0xFFFFu) | ((uint32_t) dy << 16);
}

// PhotonMap::LocatePhotons + EstimateIrradiance<100> with a normal, ellipticity 1 and the quadratic
// filter.  The reference recurses (far child, near child, then the node itself) and keeps copies of
// the found photons; here the recursion is an explicit stack of (node, phase) words in the lane's
// LDS traversal stack, and the heap keeps photon indices: the lane's 101 consecutive 8-byte
// (distance, index) elements of a global scratch slab.  Directions and powers come decoded from the tables
// of DPhotonMap (same values as Photon::GetDirection / GetPower return).
__device__ __forceinline__ void photonEstimate(const DPhotonMap &pm, f3 pos, f3 N, uint32_t *stack, uint2 *heap, f3 &irrad,
                                               f3 &direction)
{
  // heap element k of this lane: (distance^2 bits, photon index)
  auto H = [&](int k) -> uint2 & { return heap[k]; };
  irrad = F3(0, 0, 0);
  direction = F3(0, 0, 0);
  float d2max = pm.radius * pm.radius;  // np.dist2[0]
  int found = 0;
  // Stack entries are two words: (node << 2 | phase, payload).
  //   phase 0     first visit;
  //   phase 1, 3  the child on pos' side is done; payload = signed distance to the splitting plane;
  //               3 = the node itself was outside the search radius at its first visit: d2max only
  //               ever shrinks, so it stays outside and its record is not read a second time;
  //   phase 2     both subtrees done, the node itself is next.
  stack[0] = 1u << 2;
  int sp = 2;
  while (sp) {
    sp -= 2;
    const uint32_t e = stack[sp * QA_BLOCK];
    const uint32_t index = e >> 2, phase = e & 3u;
    if (phase & 1u) {
      const float dist = asF(stack[(sp + 1) * QA_BLOCK]);
      if (phase == 1) { stack[sp * QA_BLOCK] = (index << 2) | 2u; sp += 2; }
      if (dist * dist < d2max) { stack[sp * QA_BLOCK] = (dist > 0 ? 2 * index : 2 * index + 1) << 2; sp += 2; }
      continue;
    }
    const uint4 nd = pm.node[index];
    if (phase == 0 && (int) index < pm.half) {
      const uint32_t axis = nd.w;
      const float dist = (axis == 0 ? pos.x - asF(nd.x) : axis == 1 ? pos.y - asF(nd.y) : pos.z - asF(nd.z));
      const f3 dif0 = F3(asF(nd.x), asF(nd.y), asF(nd.z)) - pos;
      const bool inside = dot(dif0, dif0) < d2max;
      stack[sp * QA_BLOCK] = (index << 2) | (inside ? 1u : 3u);
      stack[(sp + 1) * QA_BLOCK] = __float_as_uint(dist);
      stack[(sp + 2) * QA_BLOCK] = (dist > 0 ? 2 * index + 1 : 2 * index) << 2;
      sp += 4;
      continue;
    }
    // the node itself
    const f3 dif = F3(asF(nd.x), asF(nd.y), asF(nd.z)) - pos;
    const float dist2 = dot(dif, dif);
    if (!(dist2 < d2max)) continue;
    const float4 pd = pm.dir[index];
    if (dot(F3(pd.x, pd.y, pd.z), N) >= 0) continue;
    if (found < QA_PHOTON_GATHER) {
      found++;
      H(found) = make_uint2(__float_as_uint(dist2), index);
      if (found == QA_PHOTON_GATHER) {  // build the max-heap
        const int half_found = found >> 1;
        for (int k = half_found; k >= 1; k--) {
          int parent = k;
          const uint2 t = H(k);
          const float td2 = asF(t.x);
          while (parent <= half_found) {
            int j = parent + parent;
            uint2 ej = H(j);
            if (j < found) {
              const uint2 ej1 = H(j + 1);
              if (asF(ej.x) < asF(ej1.x)) { j++; ej = ej1; }
            }
            if (td2 >= asF(ej.x)) break;
            H(parent) = ej;
            parent = j;
          }
          H(parent) = t;
        }
      }
    } else {
      int parent = 1, j = 2;
      while (j <= found) {
        uint2 ej = H(j);
        if (j < found) {
          const uint2 ej1 = H(j + 1);
          if (asF(ej.x) < asF(ej1.x)) { j++; ej = ej1; }
        }
        if (dist2 > asF(ej.x)) break;
        H(parent) = ej;
        parent = j;
        j <<= 1;
      }
      H(parent) = make_uint2(__float_as_uint(dist2), index);
      d2max = asF(H(1).x);
    }
  }
  for (int i = 1; i <= found; i++) {
    const uint2 e = H(i);
    const float4 pw = pm.power[e.y], pd = pm.dir[e.y];
    const float filter = 1 - asF(e.x) / d2max;
    irrad = irrad + F3(pw.x, pw.y, pw.z) * filter;
    direction = direction + F3(pd.x, pd.y, pd.z) * (filter * pd.w);
  }
  if (found > 0) {
    const float area = (QA_PI * 0.5f) * d2max;
    if (area > 0) irrad = irrad * (1.0f / area);
    direction = normalize(direction);
  }
}

// the gather term of Shade (MtlBlinn_PhotonMap.cpp:426-458).  Inlined by default (20 % faster than a
// call); -DQA_PM_CALL keeps it a real call, which was the remedy while an earlier version of the
// walk - one that indexed a private float[3] with the split axis - returned different pixels from
// run to run when inlined at 128 VGPRs.  tests/test_gpu_photon.py renders every
// photon frame three times and requires identical bits.
#ifdef QA_PM_CALL
#define QA_PM_GATHER_ATTR __attribute__((noinline))
#else
#define QA_PM_GATHER_ATTR __forceinline__
#endif
__device__ QA_PM_GATHER_ATTR f3 photonGather(const DPhotonMap &pm, f3 p, f3 N, f3 V, f3 kd, f3 ks, float gloss, uint32_t *stack,
                                             uint2 *heap)
{
  f3 I, D;
  photonEstimate(pm, p, N, stack, heap, I, D);
  if (luma(I) > 0.00001f) {
    const f3 L = -normalize(D);
    const f3 H = normalize(V + L);
    const float cosNL = qmax(0.f, dot(N, L));
    const float cosNH = qmax(0.f, dot(N, H));
    return (I * cosNL) * (kd + ks * qpowf(cosNH, gloss));
  }
  return F3(0, 0, 0);
}

}  // namespace qa
