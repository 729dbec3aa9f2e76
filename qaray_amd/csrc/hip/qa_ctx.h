// qa_ctx.h — internals shared by the translation units of libqaray_hip.so (qa_capi.hip: context,
// scene, render launches; qa_photon.hip: photon / caustics maps).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "qa_scene_dev.h"
#include "qa_wf_types.h"
#include "qaray_hip.h"

using namespace qa;

typedef void (*KernelFn)(const DScene, const RenderParams);

inline thread_local std::string g_err;
inline int Fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return Fail(QA_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Developer knobs.  The PRODUCT library reads no environment variable of its own: what an embedding application or a test
// may change goes through qa_set_option (include/qaray_hip.h).  Builds made with -DQA_DEV_KNOBS (make hip EXTRA=-DQA_DEV_KNOBS,
// the A/B scripts under tools/) additionally read the QA_* variables named at the call sites.
inline const char *DevEnv(const char *name)
{
#ifdef QA_DEV_KNOBS
  return getenv(name);
#else
  (void) name;
  return nullptr;
#endif
}

struct EventPair { hipEvent_t a, b; };

// Host side of the staged integrator (qa_wf.hip): buffers are kept between frames
struct WfHost {
  bool eligible = false;        // the uploaded scene can run staged (SelectStaged)
  bool modeSet = false;         // qa_set_pipeline was called
  int mode = 2;                 // qa_set_pipeline: 0 mega, 1 staged, 2 auto (= the megakernel: the staged integrator runs on request only)
  int numLights = 0;            // non-ambient lights
  int32_t lightIdx[QA_WF_MAX_LIGHTS] = {0, 0, 0, 0};
  // The frame's 8x8 tiles are dealt round-robin to a few GROUPS; each group has its own slot state, queues and counters
  // and drives its own logic -> cull -> trace -> redo chain on its own stream, so that the (latency-bound, tail-heavy)
  // kernels of different groups overlap on the chip.
  struct Group {
    WfBuf buf{};
    size_t capSlots = 0;
    int capLights = -1;
    std::vector<void *> allocs;
    WfCounters *dCtr = nullptr, *hCtr = nullptr;   // one per iteration of a chunk (device / pinned host)
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t logicDone = nullptr, redoDone = nullptr;   // wf_redo runs on WfHost::redoStream beside the pass's cull and trace stages
    bool finished = false;
  };
  static const int kMaxGroups = 8;
  Group groups[kMaxGroups];
  int numGroups = 1;            // qa_set_option("staged_groups"): several groups only pay when the process has a hardware queue per group stream (GPU_MAX_HW_QUEUES >= 8)
  hipEvent_t start = nullptr;
  hipStream_t redoStream = nullptr;   // shared by the groups
  bool redoAsync = true;              // QA_WF_REDO_ASYNC=0: wf_redo in the group's own chain
  WfStats *dStats = nullptr;
  // diagnostics of the frames rendered since the last reset
  uint64_t iterations = 0, raysClosest = 0, raysShadow = 0, jobs = 0, redo = 0;
  int traceBlocksPerCU = 0;     // QA_WF_BLOCKS: workgroups per CU of every stage kernel of one group (0 = 2 with several groups, what fits with one)
  uint32_t gate = 1;            // new samples start every gate-th pass (QA_WF_GATE).  2 was worth +50 % with the first trace stage; since the
                                // 4-wide tree over triangles and the scalar / global table reads it is equal on long frames and 4 - 10 % behind on short ones
  uint32_t stackCap = 24;       // LDS stack entries per lane of wf_trace (QA_WF_STACK)
  uint32_t budget = 512;        // BVH steps a job may take per pass (QA_WF_BUDGET)
};

struct qa_ctx {
  int device = 0;
  int numCUs = 0;
  hipStream_t stream = nullptr;
  // scene
  std::vector<unsigned char> hostBlob;
  unsigned char *dBlob = nullptr;
  std::vector<void *> sceneAllocs;  // derived arrays
  std::vector<DMesh> hostMeshes;    // host copy of the device mesh table
  DScene ds{};
  bool haveScene = false;
  float *dHalton = nullptr;
  int haltonCount = 0;
  // launch plumbing
  static const int kCounterRing = 64;
  unsigned int *dWork = nullptr;  // ring of work counters
  // tiles in sample chunks (qa_integrate, RenderParams::chunk_spp): per-pixel state between chunks, per-tile progress
  uint32_t *dPixState = nullptr, *dTileProgress = nullptr;
  size_t pixStateWords = 0, tileProgressWords = 0;
  hipEvent_t chunkEv = nullptr;   // end of the last frame: the slabs (this one, the area-light log, the many-light surface slab) are one per context
  bool chunkEvSet = false;
  hipStream_t lastStream = nullptr;   // ... a frame on another stream waits for it
  int optChunkSpp = -1;           // "chunk_spp": -1 per frame (few tiles per wave), 0 off, n samples of a tile's first chunk
  int optChunkTail = 0;           // "chunk_tail": samples of every further chunk (0: an eighth of the frame's spp)
  int workNext = 0;
  int *hStop = nullptr;           // mapped host memory, read by the kernel's wave leaders
  int *dStopAlias = nullptr;
  DCounters *dCounters = nullptr;
  // host-variant staging
  float *dRgb = nullptr, *dDepth = nullptr;
  uint32_t *dNs = nullptr;
  size_t stagePixels = 0;
  // timing
  std::vector<EventPair> pending, freeEvents;
  double totalMs = 0;
  uint64_t launches = 0;
  int blocksPerCU = 0, blocksPerCUAuto = 2, threads = QA_BLOCK;  // 0 = use the occupancy-derived value
  void (*kernel)(const DScene, const RenderParams) = nullptr;
  void (*kernelStats)(const DScene, const RenderParams) = nullptr;
  bool resident = false, textured = false, area = false;
  int syncAuto = 0;
  bool csMany = false;   // the cooperative kernel's MANY variant (more shadow-casting lights than one batch)
  bool tileOrder = true;        // centre-first tile order (QA_NO_TILE_ORDER=1 turns it off)
  uint32_t *dOrder = nullptr;   // tile launch order of the last region shape
  uint64_t orderKey = 0;
  int syncSamples = -1;  // -1: decide per scene (SelectKernel), 0/1 forced by QA_SYNC
  uint32_t stackDepth = 32;
  size_t ldsBytes = 0;
  // photon / caustics maps (qa_photon.hip); valid until the next scene upload or qa_photon_maps_clear
  bool photonReady = false;
  KernelFn kernelPm = nullptr, kernelPmStats = nullptr;
  // the megakernel with cooperative mesh walks (qa_kernel_cs.h): global-memory scenes without area lights
  KernelFn kernelCs = nullptr;
  bool csFits = false;          // the scene fits qa_integrate_cs's limits (20-bit scene-wide node / triangle indices, <= 256 nodes, ...)
  size_t ldsBytesCs = 0;
  const uint4 *csNodesDev = nullptr, *csTrisDev = nullptr, *csLeafBoxDev = nullptr;   // scene allocations (freed with the scene)
  const CsInst *csInstDev = nullptr;
  const CsCull *csCullDev = nullptr;
  float csCullS1 = 0, csCullS2 = 0, csCullK3 = 0, csCullK4 = 0;
  bool csCullVariant = false;    // the cooperative kernel chosen tests the nodes' bounds first (SelectKernel)
  bool csCullOk = false;         // the widening constants are finite (otherwise every instance is visited)
  int blocksPerCUCs = 2;
  int blocksPerCUPm = 2;
  uint32_t stackDepthPm = 0;   // LDS stack entries per lane when the kd-tree gather runs on it
  size_t ldsBytesPm = 0;
  void *dPhotons[2] = {nullptr, nullptr};      // qa_photon records (what qa_photon_maps_download returns)
  void *dPmTables[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};   // DPhotonMap node / dir / power
  void *dHeap = nullptr;
  qa_photon_params photonParams{};
  uint64_t photonEmitted[2] = {0, 0}, photonEmissions[2] = {0, 0};
  std::vector<qa_photon> hostPhotons[2];   // balanced, [0] unused
  WfHost wf;
  std::string kernelName;       // the integrator the next frame is planned to run on (SetKernelName)
  std::string launchedName;     // what the last qa_render_* call really launched (empty before the first)
  // qa_set_option
  bool optCoop = true;          // "coop": cooperative mesh walks (qa_kernel_cs.h) where the scene allows them
  bool optCsCull = true;        // "cs_cull": instance culling in the cooperative kernel's sweeps (0: every instance is visited; A/B tests)
  uint32_t optWalkZeroTerms = 0; // "walk_zero_terms": tests - also walk the shadow rays of lights whose term is zero whatever they find
  uint32_t optCsForceExact = 0; // "cs_force_exact": tests of the exact walks (bit 0 closest-hit, bit 1 shadow queries)
  uint32_t optCsPool = 0;       // "cs_pool_limit": upper bound for the walks' pool capacity (tests force the overflow path)
  bool optVerbose = false;      // "verbose": tree / launch-shape report on stderr at upload
};

void FreePhotonMaps(qa_ctx *c);  // qa_photon.hip
// qa_wf.hip
void FreeStaged(qa_ctx *c);
void SelectStaged(qa_ctx *c);
bool StagedTakes(const qa_ctx *c, uint32_t flags, int spp_max, int max_bounce, size_t slots);
int RenderStaged(qa_ctx *c, const DScene &ds, const RenderParams &rp, hipStream_t s, DCounters *frameCounters);

inline void FreeScene(qa_ctx *c)
{
  FreePhotonMaps(c);
  for (void *p : c->sceneAllocs) (void) hipFree(p);
  c->sceneAllocs.clear();
  if (c->dBlob) (void) hipFree(c->dBlob);
  c->dBlob = nullptr;
  c->haveScene = false;
}

template <class T>
inline int DeviceCopy(qa_ctx *c, const std::vector<T> &v, const T **out)
{
  *out = nullptr;
  if (v.empty()) return QA_OK;
  void *p = nullptr;
  HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
  c->sceneAllocs.push_back(p);
  HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<const T *>(p);
  return QA_OK;
}

