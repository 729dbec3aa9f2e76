// qa_wf.hip — host side of the staged integrator (qa_wf.h): eligibility, per-slot state and queues in
// HBM, and the iteration loop that launches the stages until every pixel of the region has finished.
// Third translation unit of libqaray_hip.so.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "qa_wf.h"
#include "qa_ctx.h"

namespace {

const int kChunk = 16;   // iterations between two looks at the "pixels still running" counter

void FreeBuffers(WfHost &w)
{
  for (void *p : w.allocs) (void) hipFree(p);
  w.allocs.clear();
  w.capSlots = 0;
  w.capLights = -1;
  memset(&w.buf, 0, sizeof(w.buf));
}

template <class T>
int Alloc(WfHost &w, T **out, size_t count)
{
  void *p = nullptr;
  HIP_TRY(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
  w.allocs.push_back(p);
  *out = static_cast<T *>(p);
  return QA_OK;
}

int EnsureBuffers(qa_ctx *c, size_t slots, int lights, uint32_t stackDepth, uint32_t traceStack)
{
  WfHost &w = c->wf;
  if (!w.dCtr) {
    HIP_TRY(hipMalloc((void **) &w.dCtr, kChunk * sizeof(WfCounters)));
    HIP_TRY(hipMalloc((void **) &w.dStats, sizeof(WfStats)));
    HIP_TRY(hipMemset(w.dStats, 0, sizeof(WfStats)));
    HIP_TRY(hipHostMalloc((void **) &w.hCtr, kChunk * sizeof(WfCounters), hipHostMallocDefault));
  }
  if (slots <= w.capSlots && lights == w.capLights && stackDepth == w.buf.stackDepth && traceStack == w.buf.traceStack) return QA_OK;
  FreeBuffers(w);
  WfBuf &b = w.buf;
  int rc;
  const size_t nl = (size_t) std::max(lights, 1);
  if ((rc = Alloc(w, &b.P, slots)) || (rc = Alloc(w, &b.D, slots)) || (rc = Alloc(w, &b.T, slots)) || (rc = Alloc(w, &b.L, slots)) ||
      (rc = Alloc(w, &b.mean, slots)) || (rc = Alloc(w, &b.cstd, slots)) || (rc = Alloc(w, &b.Tp, slots)) ||
      (rc = Alloc(w, &b.SH, slots * nl)) || (rc = Alloc(w, &b.C, slots * nl)) || (rc = Alloc(w, &b.key, slots)) ||
      (rc = Alloc(w, &b.vis, slots)) || (rc = Alloc(w, &b.rayq, slots * (1 + nl))) || (rc = Alloc(w, &b.redoq, slots * (1 + nl))) ||
      (rc = Alloc(w, &b.out, slots)) || (rc = Alloc(w, &b.redoFlag, slots)) || (rc = Alloc(w, &b.contCount, 2))) {
    FreeBuffers(w);
    return rc;
  }
  // job queue: a ray enters the bounds of ~1 mesh on average; rays the queue cannot take go to the exact repeat
  b.jobCap = (uint32_t) std::min<size_t>(slots * (1 + nl) * 2, 0x7FFFFFFFu);
  b.contCap = (uint32_t) std::max<size_t>(b.jobCap / 4, 4096);
  b.stackDepth = stackDepth;
  b.traceStack = traceStack;
  if ((rc = Alloc(w, &b.jobA, b.jobCap)) || (rc = Alloc(w, &b.jobB, b.jobCap))) { FreeBuffers(w); return rc; }
  for (int k = 0; k < 2; ++k)
    if ((rc = Alloc(w, &b.contA[k], b.contCap)) || (rc = Alloc(w, &b.contB[k], b.contCap)) || (rc = Alloc(w, &b.contC[k], b.contCap)) ||
        (rc = Alloc(w, &b.contStack[k], (size_t) b.contCap * traceStack))) { FreeBuffers(w); return rc; }
  b.stats = w.dStats;
  w.capSlots = slots;
  w.capLights = lights;
  return QA_OK;
}

}  // namespace

void FreeStaged(qa_ctx *c)
{
  FreeBuffers(c->wf);
  if (c->wf.dCtr) (void) hipFree(c->wf.dCtr);
  if (c->wf.dStats) (void) hipFree(c->wf.dStats);
  if (c->wf.hCtr) (void) hipHostFree(c->wf.hCtr);
  c->wf.dCtr = nullptr;
  c->wf.dStats = nullptr;
  c->wf.hCtr = nullptr;
}

// Which scenes the staged integrator takes (decided once per upload).  Everything else keeps the megakernel.
void SelectStaged(qa_ctx *c)
{
  WfHost &w = c->wf;
  w.eligible = false;
  w.numLights = 0;
  const qa_flat_header *h = reinterpret_cast<const qa_flat_header *>(c->hostBlob.data());
  const qa_light *light = QA_BLOB_PTR(qa_light, c->hostBlob.data(), h->off_lights);
  const qa_instance *inst = QA_BLOB_PTR(qa_instance, c->hostBlob.data(), h->off_instances);
  const qa_mesh *mesh = QA_BLOB_PTR(qa_mesh, c->hostBlob.data(), h->off_meshes);
  int nl = 0;
  for (uint32_t i = 0; i < h->num_lights; ++i)
    if (light[i].type != QA_LIGHT_AMBIENT) {
      if (nl < QA_WF_MAX_LIGHTS) w.lightIdx[nl] = (int32_t) i;
      ++nl;
    }
  bool ok = !c->resident && !c->area && nl <= QA_WF_MAX_LIGHTS && h->num_instances <= 31;
  bool anyMesh = false;
  for (uint32_t k = 0; k < h->num_instances && ok; ++k) {
    if (inst[k].obj_type != QA_OBJ_MESH) continue;
    anyMesh = true;
    const qa_mesh &m = mesh[inst[k].mesh];
    if (m.num_faces >= (1u << 24)) ok = false;
    // a hit on a mesh without texture vertices keeps the uvw of an earlier, farther hit (the intersectors only
    // overwrite what they set): that history lives in the megakernel's sequential walk only
    if (c->textured && m.num_faces > 0) {
      uint32_t withVT = 0;
      const qa_face *faces = QA_BLOB_PTR(qa_face, c->hostBlob.data(), m.off_faces);
      for (uint32_t f = 0; f < m.num_faces; ++f) if (faces[f].vt[0] >= 0 && faces[f].vt[1] >= 0 && faces[f].vt[2] >= 0) ++withVT;
      if (withVT != m.num_faces) ok = false;
    }
  }
  ok = ok && anyMesh;
  // the trace stage searches the 4-wide trees only (qa_widebvh.h)
  for (const DMesh &dm : c->hostMeshes) if (dm.num_faces > 0 && !dm.useWide) ok = false;
  if (!w.modeSet) {     // qa_set_pipeline outlives scene uploads; otherwise the environment decides
    w.mode = QA_PIPE_AUTO;
    if (const char *e = getenv("QA_PIPELINE")) {
      if (!strcmp(e, "mega")) w.mode = QA_PIPE_MEGA;
      else if (!strcmp(e, "staged")) w.mode = QA_PIPE_STAGED;
    }
  }
  w.decision = -1;
  w.eligible = ok;
  w.numLights = ok ? nl : 0;
}

bool StagedTakes(const qa_ctx *c, uint32_t flags, int spp_max, int max_bounce, size_t slots)
{
  return c->wf.eligible && !c->photonReady && !(flags & QA_RENDER_STATS) && spp_max <= 65535 && max_bounce <= 15 &&
         slots <= ((size_t) 1 << QA_WF_SLOT_BITS);
}

int RenderStaged(qa_ctx *c, const DScene &ds, const RenderParams &rp, hipStream_t s, DCounters *frameCounters)
{
  WfHost &w = c->wf;
  const int rw = rp.x1 - rp.x0;
  const size_t tilesX = (size_t) (rw + 7) / 8;
  const size_t slots = tilesX * (size_t) rp.own_tile_rows * 64;
  // wf_trace's stacks: what the wide trees can need, capped (QA_WF_STACK, default 24): nearest-first walks rarely hold
  // more than a dozen entries, and every LDS kilobyte saved is occupancy; a full stack sends the ray to wf_redo
  uint32_t wideNeed = 2;
  for (const DMesh &dm : c->hostMeshes) if (dm.useWide) wideNeed = std::max(wideNeed, dm.wideStack);
  const uint32_t traceStack = std::min(wideNeed, w.stackCap);
  int rc = EnsureBuffers(c, slots, w.numLights, ds.stackDepth, traceStack);
  if (rc != QA_OK) return rc;
  WfBuf b = w.buf;
  b.n = (uint32_t) slots;
  b.numLights = (uint32_t) w.numLights;
  for (int j = 0; j < QA_WF_MAX_LIGHTS; ++j) b.lightIdx[j] = w.lightIdx[j];

  const unsigned blocks = (unsigned) ((slots + QA_BLOCK - 1) / QA_BLOCK);
  hipLaunchKernelGGL(wf_init, dim3(blocks), dim3(QA_BLOCK), 0, s, ds, rp, b);
  HIP_TRY(hipGetLastError());
  const size_t stackLds = (size_t) ds.stackDepth * QA_BLOCK * sizeof(uint32_t);
  const size_t traceLds = (size_t) traceStack * QA_BLOCK * sizeof(uint32_t);
  if (!w.traceBlocksPerCU) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *) wf_trace, QA_BLOCK, traceLds) != hipSuccess || n < 1) n = 2;
    w.traceBlocksPerCU = std::min(n, 8);
  }
  const size_t rays = slots * (1 + b.numLights);
  const unsigned logicBlocks = (unsigned) std::min<size_t>((size_t) c->numCUs * 8, blocks);
  const unsigned cullBlocks = (unsigned) std::min<size_t>((size_t) c->numCUs * 8, (rays + QA_BLOCK - 1) / QA_BLOCK);
  const unsigned traceBlocks = (unsigned) std::min<size_t>((size_t) c->numCUs * w.traceBlocksPerCU, (rays + QA_BLOCK - 1) / QA_BLOCK);
  const unsigned redoBlocks = (unsigned) std::min<size_t>((size_t) c->numCUs * 2, (rays + QA_BLOCK - 1) / QA_BLOCK);
  const uint32_t budget = w.budget;
  b.refillAt = getenv("QA_WF_REFILL") ? (uint32_t) atoi(getenv("QA_WF_REFILL")) : 16u;
  const bool dbg = getenv("QA_WF_DEBUG") != nullptr;
  b.debug = dbg ? (uint32_t) atoi(getenv("QA_WF_DEBUG")) >> 1 : 0u;   // synchronise and report after every stage
  // every pixel advances by at most one path segment per pass; suspended walks add passes
  const long long maxIter = ((long long) rp.spp_max * (rp.max_bounce + 3) + 4) * 64;
  long long iter = 0;
  bool finished = false;
  while (!finished && iter < maxIter) {
    if (__atomic_load_n(c->hStop, __ATOMIC_SEQ_CST)) break;   // tasking::signal_stop: unfinished pixels keep ns = 0
    HIP_TRY(hipMemsetAsync(w.dCtr, 0, kChunk * sizeof(WfCounters), s));
    for (int i = 0; i < kChunk; ++i, ++iter) {
      WfCounters *ctr = w.dCtr + i;
      const uint32_t parity = (uint32_t) (iter & 1);
      if (c->textured) hipLaunchKernelGGL(wf_logic<true>, dim3(logicBlocks), dim3(QA_BLOCK), 0, s, ds, rp, b, ctr, frameCounters, parity);
      else hipLaunchKernelGGL(wf_logic<false>, dim3(logicBlocks), dim3(QA_BLOCK), 0, s, ds, rp, b, ctr, frameCounters, parity);
      if (dbg) { HIP_TRY(hipStreamSynchronize(s)); fprintf(stderr, "[wf] pass %lld logic ok\n", iter); }
      hipLaunchKernelGGL(wf_cull, dim3(cullBlocks), dim3(QA_BLOCK), 0, s, ds, b, ctr);
      if (dbg) { HIP_TRY(hipStreamSynchronize(s)); fprintf(stderr, "[wf] pass %lld cull ok\n", iter); }
      hipLaunchKernelGGL(wf_trace, dim3(traceBlocks), dim3(QA_BLOCK), (unsigned) traceLds, s, ds, b, ctr, parity, budget);
      if (dbg) { HIP_TRY(hipStreamSynchronize(s)); fprintf(stderr, "[wf] pass %lld trace ok\n", iter); }
      hipLaunchKernelGGL(wf_redo, dim3(redoBlocks), dim3(QA_BLOCK), (unsigned) stackLds, s, ds, b, ctr);
      if (dbg) { HIP_TRY(hipStreamSynchronize(s)); fprintf(stderr, "[wf] pass %lld redo ok\n", iter); }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(w.hCtr, w.dCtr, kChunk * sizeof(WfCounters), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < kChunk; ++i) {
      w.raysClosest += w.hCtr[i].nClosest;
      w.raysShadow += w.hCtr[i].nShadow;
      w.jobs += w.hCtr[i].nJobs;
      w.redo += w.hCtr[i].nRedo;
      if (w.hCtr[i].active) w.iterations++;
    }
    if (w.hCtr[kChunk - 1].active == 0) finished = true;
  }
  if (!finished && !__atomic_load_n(c->hStop, __ATOMIC_SEQ_CST)) return Fail(QA_EHIP, "staged integrator did not converge (internal error)");
  return QA_OK;
}
