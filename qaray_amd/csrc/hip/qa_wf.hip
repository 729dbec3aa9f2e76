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

const int kChunk = 16;   // iterations between two looks at the "pixels still running" counters

typedef WfHost::Group Group;

void FreeBuffers(Group &g)
{
  for (void *p : g.allocs) (void) hipFree(p);
  g.allocs.clear();
  g.capSlots = 0;
  g.capLights = -1;
  memset(&g.buf, 0, sizeof(g.buf));
}

template <class T>
int Alloc(Group &g, T **out, size_t count)
{
  void *p = nullptr;
  HIP_TRY(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
  g.allocs.push_back(p);
  *out = static_cast<T *>(p);
  return QA_OK;
}

int EnsureBuffers(qa_ctx *c, Group &g, size_t slots, int lights, uint32_t stackDepth, uint32_t traceStack)
{
  WfHost &w = c->wf;
  if (!g.dCtr) {
    HIP_TRY(hipMalloc((void **) &g.dCtr, kChunk * sizeof(WfCounters)));
    HIP_TRY(hipHostMalloc((void **) &g.hCtr, kChunk * sizeof(WfCounters), hipHostMallocDefault));
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&g.logicDone, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&g.redoDone, hipEventDisableTiming));
  }
  if (slots <= g.capSlots && lights == g.capLights && stackDepth == g.buf.stackDepth && traceStack == g.buf.traceStack) return QA_OK;
  FreeBuffers(g);
  WfBuf &b = g.buf;
  int rc;
  const size_t nl = (size_t) std::max(lights, 1);
  if ((rc = Alloc(g, &b.P, slots)) || (rc = Alloc(g, &b.D, slots)) || (rc = Alloc(g, &b.T, slots)) || (rc = Alloc(g, &b.L, slots)) ||
      (rc = Alloc(g, &b.mean, slots)) || (rc = Alloc(g, &b.cstd, slots)) || (rc = Alloc(g, &b.Tp, slots)) ||
      (rc = Alloc(g, &b.SH, slots * nl)) || (rc = Alloc(g, &b.C, slots * nl)) || (rc = Alloc(g, &b.key, slots)) ||
      (rc = Alloc(g, &b.vis, slots)) || (rc = Alloc(g, &b.rayq, slots * (1 + nl))) || (rc = Alloc(g, &b.redoq, slots * (1 + nl))) ||
      (rc = Alloc(g, &b.out, slots)) || (rc = Alloc(g, &b.redoFlag, slots)) || (rc = Alloc(g, &b.contCount, 2))) {
    FreeBuffers(g);
    return rc;
  }
  // job queue: a ray enters the bounds of ~1 mesh on average; rays the queue cannot take go to the exact repeat
  b.jobCap = (uint32_t) std::min<size_t>(slots * (1 + nl) * 2, 0x7FFFFFFFu);
  b.contCap = (uint32_t) std::max<size_t>(b.jobCap / 4, 4096);
  b.stackDepth = stackDepth;
  b.traceStack = traceStack;
  if ((rc = Alloc(g, &b.jobA, b.jobCap)) || (rc = Alloc(g, &b.jobB, b.jobCap))) { FreeBuffers(g); return rc; }
  for (int k = 0; k < 2; ++k)
    if ((rc = Alloc(g, &b.contA[k], b.contCap)) || (rc = Alloc(g, &b.contB[k], b.contCap)) || (rc = Alloc(g, &b.contC[k], b.contCap)) ||
        (rc = Alloc(g, &b.contStack[k], (size_t) b.contCap * traceStack))) { FreeBuffers(g); return rc; }
  b.stats = w.dStats;
  g.capSlots = slots;
  g.capLights = lights;
  return QA_OK;
}

}  // namespace

void FreeStaged(qa_ctx *c)
{
  WfHost &w = c->wf;
  for (Group &g : w.groups) {
    FreeBuffers(g);
    if (g.dCtr) (void) hipFree(g.dCtr);
    if (g.hCtr) (void) hipHostFree(g.hCtr);
    if (g.stream) (void) hipStreamDestroy(g.stream);
    if (g.done) (void) hipEventDestroy(g.done);
    if (g.logicDone) (void) hipEventDestroy(g.logicDone);
    if (g.redoDone) (void) hipEventDestroy(g.redoDone);
    g.dCtr = nullptr; g.hCtr = nullptr; g.stream = nullptr; g.done = nullptr;
  }
  if (w.dStats) (void) hipFree(w.dStats);
  if (w.start) (void) hipEventDestroy(w.start);
  if (w.redoStream) (void) hipStreamDestroy(w.redoStream);
  w.redoStream = nullptr;
  w.dStats = nullptr;
  w.start = nullptr;
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
    if (const char *e = DevEnv("QA_PIPELINE")) {
      if (!strcmp(e, "mega")) w.mode = QA_PIPE_MEGA;
      else if (!strcmp(e, "staged")) w.mode = QA_PIPE_STAGED;
    }
  }
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
  const size_t tiles = tilesX * (size_t) rp.own_tile_rows;
  // groups: at least ~2000 tiles each, or the frame is too small to be worth splitting
  int G = std::max(1, std::min(w.numGroups, WfHost::kMaxGroups));
  while (G > 1 && tiles / G < 2048) --G;
  // wf_trace's stacks: what the wide trees can need, capped (QA_WF_STACK, default 24): nearest-first walks rarely hold
  // more than a dozen entries, and every LDS kilobyte saved is occupancy; a full stack sends the ray to wf_redo
  uint32_t wideNeed = 2;
  for (const DMesh &dm : c->hostMeshes) if (dm.useWide) wideNeed = std::max(wideNeed, dm.wideStack);
  const uint32_t traceStack = std::min(wideNeed, w.stackCap);

  const size_t stackLds = (size_t) ds.stackDepth * QA_BLOCK * sizeof(uint32_t);
  const size_t traceLds = (size_t) traceStack * QA_BLOCK * sizeof(uint32_t);
  // persistent grids: with several groups every stage kernel takes a slice of the chip (2 workgroups per CU by default),
  // so that kernels of different groups are resident together; a single group takes what fits
  int perCU = w.traceBlocksPerCU;
  if (!perCU) {
    if (G > 1) perCU = 2;
    else {
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *) wf_trace, QA_BLOCK, traceLds) != hipSuccess || n < 1) n = 2;
      perCU = std::min(n, 8);
    }
  }
  const uint32_t budget = w.budget;
  const bool dbg = DevEnv("QA_WF_DEBUG") != nullptr;   // synchronise and report after every stage
  const uint32_t refillAt = DevEnv("QA_WF_REFILL") ? (uint32_t) atoi(DevEnv("QA_WF_REFILL")) : 16u;

  // ---- per-group buffers and launch shapes
  struct Shape { unsigned initBlocks, logicBlocks, cullBlocks, traceBlocks, redoBlocks; };
  Shape shape[WfHost::kMaxGroups];
  if (!w.dStats) {
    HIP_TRY(hipMalloc((void **) &w.dStats, sizeof(WfStats)));
    HIP_TRY(hipMemset(w.dStats, 0, sizeof(WfStats)));
    HIP_TRY(hipEventCreateWithFlags(&w.start, hipEventDisableTiming));
  }
  if (!w.redoStream) HIP_TRY(hipStreamCreateWithFlags(&w.redoStream, hipStreamNonBlocking));
  // wf_redo (a handful of rays per pass, each a long sequential walk: 60 - 120 us of one lane's latency) only needs this
  // pass's wf_logic before it and the next pass's wf_logic after it: it runs on a side stream beside wf_cull and wf_trace
  // (with one group: +4 - 5 %; with four groups the other groups' kernels already fill that time and the extra stream costs 1 - 2 %)
  const bool redoAsync = w.redoAsync && !dbg && G == 1;
  HIP_TRY(hipEventRecord(w.start, s));
  for (int gi = 0; gi < G; ++gi) {
    Group &g = w.groups[gi];
    const size_t gtiles = (tiles - (size_t) gi + (size_t) G - 1) / (size_t) G;   // tiles gi, gi + G, ...
    const size_t slots = gtiles * 64;
    int rc = EnsureBuffers(c, g, std::max<size_t>(slots, 64), w.numLights, ds.stackDepth, traceStack);
    if (rc != QA_OK) return rc;
    WfBuf &b = g.buf;
    b.n = (uint32_t) slots;
    b.groupIndex = (uint32_t) gi;
    b.groupCount = (uint32_t) G;
    b.numLights = (uint32_t) w.numLights;
    for (int j = 0; j < QA_WF_MAX_LIGHTS; ++j) b.lightIdx[j] = w.lightIdx[j];
    b.refillAt = refillAt;
    b.reserve = DevEnv("QA_WF_RESERVE") ? (uint32_t) std::max(64, atoi(DevEnv("QA_WF_RESERVE"))) : 128u;   // 64 / 128 / 256 @ 64 spp: C3 313 / 324 / 316, C4 1522 / 1473 / 1370, C5 810 / 815 / 791 Msamples/s
    b.debug = dbg ? (uint32_t) atoi(DevEnv("QA_WF_DEBUG")) >> 1 : 0u;
    const size_t rays = slots * (1 + b.numLights);
    Shape &sh = shape[gi];
    sh.initBlocks = (unsigned) std::max<size_t>(1, (slots + QA_BLOCK - 1) / QA_BLOCK);
    const size_t slice = (size_t) c->numCUs * (size_t) perCU;
    const size_t logicSlice = DevEnv("QA_WF_LOGIC_BLOCKS") ? (size_t) c->numCUs * (size_t) atoi(DevEnv("QA_WF_LOGIC_BLOCKS")) : slice;
    sh.logicBlocks = (unsigned) std::max<size_t>(1, std::min<size_t>(G > 1 ? logicSlice : (size_t) c->numCUs * 8, sh.initBlocks));
    sh.cullBlocks = (unsigned) std::max<size_t>(1, std::min<size_t>(G > 1 ? slice : (size_t) c->numCUs * 8, (rays + QA_BLOCK - 1) / QA_BLOCK));
    sh.traceBlocks = (unsigned) std::max<size_t>(1, std::min<size_t>(slice, (rays + QA_BLOCK - 1) / QA_BLOCK));
    sh.redoBlocks = (unsigned) std::max<size_t>(1, std::min<size_t>((size_t) c->numCUs * 2, (rays + QA_BLOCK - 1) / QA_BLOCK));
    g.finished = slots == 0;
    HIP_TRY(hipStreamWaitEvent(g.stream, w.start, 0));   // after whatever the caller queued before this frame
    if (slots) hipLaunchKernelGGL(wf_init, dim3(sh.initBlocks), dim3(QA_BLOCK), 0, g.stream, ds, rp, b);
  }
  HIP_TRY(hipGetLastError());

  // every pixel advances by at most one path segment per pass; suspended walks add passes
  const long long maxIter = ((long long) rp.spp_max * (rp.max_bounce + 3) + 4) * 64;
  long long iter = 0;
  bool allDone = false, stopped = false;
  while (!allDone && iter < maxIter) {
    if (__atomic_load_n(c->hStop, __ATOMIC_SEQ_CST)) { stopped = true; break; }   // tasking::signal_stop: unfinished pixels keep ns = 0
    for (int gi = 0; gi < G; ++gi) {
      Group &g = w.groups[gi];
      if (g.finished) continue;
      HIP_TRY(hipMemsetAsync(g.dCtr, 0, kChunk * sizeof(WfCounters), g.stream));
    }
    // the chains of the groups are issued pass by pass, interleaved, so that their kernels reach the GPU side by side
    for (int i = 0; i < kChunk; ++i) {
      const uint32_t parity = (uint32_t) ((iter + i) & 1);
      for (int gi = 0; gi < G; ++gi) {
        Group &g = w.groups[gi];
        if (g.finished) continue;
        const Shape &sh = shape[gi];
        WfCounters *ctr = g.dCtr + i;
        g.buf.gateOpen = ((iter + i) % w.gate) == 0 ? 1u : 0u;
        if (c->textured) hipLaunchKernelGGL(wf_logic<true>, dim3(sh.logicBlocks), dim3(QA_BLOCK), 0, g.stream, ds, rp, g.buf, ctr, frameCounters, parity);
        else hipLaunchKernelGGL(wf_logic<false>, dim3(sh.logicBlocks), dim3(QA_BLOCK), 0, g.stream, ds, rp, g.buf, ctr, frameCounters, parity);
        if (dbg) { HIP_TRY(hipStreamSynchronize(g.stream)); fprintf(stderr, "[wf] group %d pass %lld logic ok\n", gi, iter + i); }
        if (redoAsync) {
          HIP_TRY(hipEventRecord(g.logicDone, g.stream));
          HIP_TRY(hipStreamWaitEvent(w.redoStream, g.logicDone, 0));
          hipLaunchKernelGGL(wf_redo, dim3(sh.redoBlocks), dim3(QA_BLOCK), (unsigned) stackLds, w.redoStream, ds, g.buf, ctr);
          HIP_TRY(hipEventRecord(g.redoDone, w.redoStream));
        }
        hipLaunchKernelGGL(wf_cull, dim3(sh.cullBlocks), dim3(QA_BLOCK), 0, g.stream, ds, g.buf, ctr);
        if (dbg) { HIP_TRY(hipStreamSynchronize(g.stream)); fprintf(stderr, "[wf] group %d pass %lld cull ok\n", gi, iter + i); }
        hipLaunchKernelGGL(wf_trace, dim3(sh.traceBlocks), dim3(QA_BLOCK), (unsigned) traceLds, g.stream, ds, g.buf, ctr, parity, budget);
        if (dbg) { HIP_TRY(hipStreamSynchronize(g.stream)); fprintf(stderr, "[wf] group %d pass %lld trace ok\n", gi, iter + i); }
        if (redoAsync) HIP_TRY(hipStreamWaitEvent(g.stream, g.redoDone, 0));   // before the next wf_logic (and the counters' way back)
        else hipLaunchKernelGGL(wf_redo, dim3(sh.redoBlocks), dim3(QA_BLOCK), (unsigned) stackLds, g.stream, ds, g.buf, ctr);
        if (dbg) { HIP_TRY(hipStreamSynchronize(g.stream)); fprintf(stderr, "[wf] group %d pass %lld redo ok\n", gi, iter + i); }
      }
    }
    iter += kChunk;
    HIP_TRY(hipGetLastError());
    for (int gi = 0; gi < G; ++gi) {
      Group &g = w.groups[gi];
      if (g.finished) continue;
      HIP_TRY(hipMemcpyAsync(g.hCtr, g.dCtr, kChunk * sizeof(WfCounters), hipMemcpyDeviceToHost, g.stream));
    }
    allDone = true;
    for (int gi = 0; gi < G; ++gi) {
      Group &g = w.groups[gi];
      if (g.finished) continue;
      HIP_TRY(hipStreamSynchronize(g.stream));
      for (int i = 0; i < kChunk; ++i) {
        w.raysClosest += g.hCtr[i].nClosest;
        w.raysShadow += g.hCtr[i].nShadow;
        w.jobs += g.hCtr[i].nJobs;
        w.redo += g.hCtr[i].nRedo;
        if (gi == 0 && g.hCtr[i].active) w.iterations++;
      }
      if (g.hCtr[kChunk - 1].active == 0) g.finished = true;
      else allDone = false;
    }
  }
  // the caller's stream continues after every group
  for (int gi = 0; gi < G; ++gi) {
    Group &g = w.groups[gi];
    HIP_TRY(hipEventRecord(g.done, g.stream));
    HIP_TRY(hipStreamWaitEvent(s, g.done, 0));
  }
  if (!allDone && !stopped) return Fail(QA_EHIP, "staged integrator did not converge (internal error)");
  return QA_OK;
}
