// qa_photon_dev.h — device side of the photon / caustics maps (-use-photon-map): the gather
// (cy::PhotonMap::EstimateIrradiance<100>, src/ext/cyPhotonMap.h:375-501, as called from
// MtlBlinn_PhotonMap::Shade, src/materials/MtlBlinn_PhotonMap.cpp:426-458) and the pieces of
// photon tracing that run per hit (Photon::SetPower / SetDirection / GetDirection,
// cyPhotonMap.h:214-254).  Included by qa_kernel.h; the emission kernel lives in qa_photon.hip.
#pragma once
#include "qa_device_math.h"
#include "qa_scene_dev.h"

namespace qa {

// Photon::GetDirection (cyPhotonMap.h:233-254): x, y from the two shorts; z from x ALONE - the
// reference computes "dirX*dirX + dirY - dirY" - through a digit-by-digit integer square root.
// w5 = dirx | diry << 16, planeDirZ = byte 3 of w4.
__device__ __forceinline__ f3 photonDirection(uint32_t w4, uint32_t w5)
{
  const int dirX = (int) (short) (w5 & 0xFFFFu), dirY = (int) (short) (w5 >> 16);
  f3 dir;
  dir.x = (float) dirX / (float) 0x7FFF;
  dir.y = (float) dirY / (float) 0x7FFF;
  int xy2 = dirX * dirX;
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
  if ((w4 >> 24) & 0x8u) dir.z = -dir.z;
  return dir;
}

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
// LDS traversal stack, and the heap keeps photon indices.  hd / hi: the lane's heap columns in
// global scratch, element k at [k * stride].
__device__ __forceinline__ void photonEstimate(const DPhotonMap &pm, f3 pos, f3 N, uint32_t *stack, float *hd, uint32_t *hi,
                                               size_t stride, f3 &irrad, f3 &direction)
{
  irrad = F3(0, 0, 0);
  direction = F3(0, 0, 0);
  float d2max = pm.radius * pm.radius;  // np.dist2[0]
  int found = 0;
  const float posv[3] = {pos.x, pos.y, pos.z};
  int sp = 0;
  stack[(sp++) * QA_BLOCK] = 1u << 2;   // (index 1, phase 0)
  while (sp) {
    const uint32_t e = stack[(--sp) * QA_BLOCK];
    const uint32_t index = e >> 2, phase = e & 3u;
    const uint32_t *rec = pm.photons + 6 * (size_t) index;
    if (phase < 2 && (int) index < pm.half) {
      const uint32_t axis = (rec[4] >> 24) & 0x3u;
      const float dist = posv[axis] - asF(rec[axis]);
      if (phase == 0) {
        // first the child on pos' side, then (phase 1) maybe the other one, then (phase 2) this node
        stack[(sp++) * QA_BLOCK] = (index << 2) | 1u;
        stack[(sp++) * QA_BLOCK] = (dist > 0 ? 2 * index + 1 : 2 * index) << 2;
      } else {
        stack[(sp++) * QA_BLOCK] = (index << 2) | 2u;
        if (dist * dist < d2max) stack[(sp++) * QA_BLOCK] = (dist > 0 ? 2 * index : 2 * index + 1) << 2;
      }
      continue;
    }
    // the node itself
    const f3 dif = F3(asF(rec[0]), asF(rec[1]), asF(rec[2])) - pos;
    const float dist2 = dot(dif, dif);
    if (!(dist2 < d2max)) continue;
    if (dot(photonDirection(rec[4], rec[5]), N) >= 0) continue;
    if (found < QA_PHOTON_GATHER) {
      found++;
      hd[(size_t) found * stride] = dist2;
      hi[(size_t) found * stride] = index;
      if (found == QA_PHOTON_GATHER) {  // build the max-heap
        const int half_found = found >> 1;
        for (int k = half_found; k >= 1; k--) {
          int parent = k;
          const uint32_t tp = hi[(size_t) k * stride];
          const float td2 = hd[(size_t) k * stride];
          while (parent <= half_found) {
            int j = parent + parent;
            float dj = hd[(size_t) j * stride];
            if (j < found) {
              const float dj1 = hd[(size_t) (j + 1) * stride];
              if (dj < dj1) { j++; dj = dj1; }
            }
            if (td2 >= dj) break;
            hd[(size_t) parent * stride] = dj;
            hi[(size_t) parent * stride] = hi[(size_t) j * stride];
            parent = j;
          }
          hi[(size_t) parent * stride] = tp;
          hd[(size_t) parent * stride] = td2;
        }
      }
    } else {
      int parent = 1, j = 2;
      while (j <= found) {
        float dj = hd[(size_t) j * stride];
        if (j < found) {
          const float dj1 = hd[(size_t) (j + 1) * stride];
          if (dj < dj1) { j++; dj = dj1; }
        }
        if (dist2 > dj) break;
        hd[(size_t) parent * stride] = dj;
        hi[(size_t) parent * stride] = hi[(size_t) j * stride];
        parent = j;
        j <<= 1;
      }
      hi[(size_t) parent * stride] = index;
      hd[(size_t) parent * stride] = dist2;
      d2max = hd[stride];
    }
  }
  for (int i = 1; i <= found; i++) {
    const uint32_t *rec = pm.photons + 6 * (size_t) hi[(size_t) i * stride];
    const float pw = asF(rec[3]);
    const uint32_t w4 = rec[4];
    const f3 power = F3((float) (w4 & 0xFFu) / 255.0f, (float) ((w4 >> 8) & 0xFFu) / 255.0f, (float) ((w4 >> 16) & 0xFFu) / 255.0f) * pw;
    const float filter = 1 - hd[(size_t) i * stride] / d2max;
    irrad = irrad + power * filter;
    direction = direction + photonDirection(w4, rec[5]) * (filter * pw);
  }
  if (found > 0) {
    const float area = (QA_PI * 0.5f) * d2max;
    if (area > 0) irrad = irrad * (1.0f / area);
    direction = normalize(direction);
  }
}

// the gather term of Shade (MtlBlinn_PhotonMap.cpp:426-458).  A real call, not inlined: the kd-tree
// walk and the heap need ~60 registers of their own, which would otherwise be taken from the
// integrator loop around it on every iteration, not only at the hits that gather.
__device__ __attribute__((noinline)) f3 photonGather(const DPhotonMap &pm, f3 p, f3 N, f3 V, f3 kd, f3 ks, float gloss, uint32_t *stack,
                                           float *hd, uint32_t *hi, size_t stride)
{
  f3 I, D;
  photonEstimate(pm, p, N, stack, hd, hi, stride, I, D);
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
