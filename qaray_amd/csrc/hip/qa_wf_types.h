// qa_wf_types.h — plain data shared by the staged integrator's kernels (qa_wf.h) and its host driver
// (qa_wf.hip, qa_ctx.h): slot-state columns, queues and counters in HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qa {

#define QA_WF_MAX_LIGHTS 4      /* non-ambient lights a staged scene may have */
#define QA_WF_SLOT_BITS 24      /* slots per batch (16.7 M pixels) */
#define QA_WF_SLOT_MASK 0xFFFFFFu
#define QA_WF_NOHIT 0xFFFFFFFFu
#define QA_WF_EXACT 0x40000000u  /* redoFlag: the key is wf_redo's exact answer */

// info word of a slot
#define WF_PH_SAMPLE 0u         /* needs a camera ray */
#define WF_PH_TRACED 1u         /* a closest-hit query is in flight / has come back */
#define WF_PH_LIGHTS 2u         /* the path has ended, the shadow rays of its last hit are in flight */
#define WF_PH_DONE 3u
#define WF_INFO_SIDX(i) ((i) & 0xFFFFu)
#define WF_INFO_BOUNCE(i) (((i) >> 16) & 0xFu)
#define WF_INFO_FROMDIFF(i) (((i) >> 20) & 1u)
#define WF_INFO_PRIMARY(i) (((i) >> 21) & 1u)
#define WF_INFO_PHASE(i) (((i) >> 22) & 3u)
#define WF_INFO_PEND(i) (((i) >> 24) & 0xFu)
struct WfCounters {           // one per iteration (ring), zeroed by the host
  uint32_t nClosest, nShadow; // rays queued by wf_logic
  uint32_t nJobs;             // jobs queued by wf_cull
  uint32_t jobHead;           // next job wf_trace hands out
  uint32_t nRedo;             // rays queued for the exact repeat
  uint32_t active;            // slots not yet DONE after wf_logic
  uint32_t rayHead;           // next ray wf_cull hands out
  uint32_t pad;
};

struct WfStats {              // accumulated since the last reset (diagnostics / roofline geometry bytes)
  unsigned long long jobs, nodeSteps, leafSteps, triTests, redo, suspended, laneSlots, waveRounds;
};

struct WfBuf {
  uint32_t n;                 // slots of THIS group (multiple of 64: whole 8x8 tiles)
  uint32_t groupIndex, groupCount;   // the frame's tiles are dealt round-robin to groupCount groups: local tile t = frame tile t * groupCount + groupIndex
  uint32_t numLights;         // non-ambient lights
  int32_t lightIdx[QA_WF_MAX_LIGHTS];
  // ---- per slot (SoA) ----
  float4 *P;                  // ray origin | rng state
  float4 *D;                  // ray direction | info word
  float4 *T;                  // throughput | absorbMtl
  float4 *L;                  // radiance of the running sample
  float4 *mean, *cstd;        // SuperSamplerHalton's running mean / variance
  float4 *Tp;                 // throughput at the hit whose lights are pending
  float4 *SH;                 // [numLights][n] shadow ray direction | t_max
  float4 *C;                  // [numLights][n] unshadowed contribution of light j at the pending hit
  unsigned long long *key;    // closest result: distance bits << 32 | instance << 24 | triangle ; all ones = miss
  uint32_t *vis;              // bit j: light j visible from the pending hit
  // ---- queues ----
  uint32_t *rayq;             // [n] closest queries: slot ; [n .. n + numLights*n) shadow queries: slot | light << 24
  float4 *jobA, *jobB;        // jobs: (local origin, limit) , (local direction, bits: slot | type << 24 | instance << 27)
  uint32_t jobCap;
  uint32_t *redoq;            // rays to repeat exactly: slot | type << 24 (type 0 = closest, j + 1 = shadow ray j)
  uint32_t *out;              // [n] jobs of the slot's rays that have not finished yet
  uint32_t *redoFlag;         // [n] bit t: the answer of ray type t needs the exact repeat
  // suspended jobs (step budget exhausted), double-buffered by pass parity
  float4 *contA[2], *contB[2];
  uint4 *contC[2];            // best triangle, current node word, stack entries, -
  uint32_t *contStack[2];     // [contCap][stackDepth]
  uint32_t *contCount;        // [2]
  uint32_t contCap;
  uint32_t stackDepth;        // LDS stack entries per lane of wf_redo (reference tree depth)
  uint32_t traceStack;        // ... of wf_trace (4-wide tree; a full stack sends the ray to wf_redo)
  uint32_t gateOpen;          // this pass may start new samples (set per pass by the host)
  uint32_t reserve;           // jobs a wave of wf_trace reserves with one atomic (QA_WF_RESERVE)
  uint32_t refillAt;          // lanes of a wave that must be out of work before it commits / refills (QA_WF_REFILL)
  uint32_t debug;             // QA_WF_DEBUG bits: 1 = skip the order check
  WfStats *stats;
};

}  // namespace qa
