// Device helpers and launch structures shared by the kernels of xm_capi.hip (lane-per-read passes) and xm_sched_kernel.hip (the gapped pass as a wave
// scheduler).  Each translation unit gets its own copy of the device code (no device linking); the structures are what the host passes.
#pragma once
#include "xm_worker.h"
#include "xm_kernel_args.h"

namespace xm {

// Light pass -> gapped pass hand-over (xm_worker.h, SavedRead).  mode 1 (light pass): a read's persistent arena is a region of the pool
// below, the lane's arena holds the temporaries only; a read that stops in front of the gapped chain keeps its region (regionOf[q]) and the
// lane takes a fresh one.  mode 2 (gapped pass): a read with a saved region continues from it on whatever lane picks it up; the lane's
// arena = [one region for reads without saved state | temporaries].  mode 0: plain runRead in the lane's arena.
struct HandOver {
  int mode;                        // (3: a pass of the light pass's shape over reads the gapped pass handed back, every one with a saved region)
  int handBack;                    // mode 2: a resumed read stops with XM_ST_NEED_LIGHT when its candidate is done
  int lightLevel;                  // mode 3: Caps::heavyAllowed of the light pass
  int seedScale;                   // scale the regions are sized for (the light pass's)
  uint8_t* regions;
  unsigned long long regionBytes;
  long long nRegions;
  int32_t* regionOf;               // per read: region that holds its SavedRead, -1 = none
  unsigned long long* cursor;      // next unused region
};


// A lane's scratch in the scheduler kernel: [region for a read that comes without saved state | chain temporaries | search arrays | memo]
struct SchedLayout { unsigned long long tmpBytes, searchBytes, memoBytes; };
// Large search sets of a launch (batches of long reads): buffers of bufBytes each, in eight groups of nPerGroup - a buffer is only ever used by
// workgroups of ONE XCD (group = HW_REG_XCC_ID of the workgroup), because memory written through one XCD's L2 and reused through another's inside a
// launch is not coherent (its write-backs can land on the new user's data).  owner[i]: 0 free, 1 taken.  n = 0: no pool.
struct BigSetPool { uint8_t* base; unsigned long long bufBytes; int32_t* owner; int32_t nPerGroup, pad; };
struct SchedLaunch {
  int grid, block;
  IndexView ix; Params params; BatchView batch;
  const int64_t* todo; long long nTodo;
  int scale, lanesPerWave, quantum, gate;
  uint8_t* arenas; unsigned long long arenaBytes;
  SchedLayout lay;
  OutView out;
  unsigned long long* nextItem; DevCounters* counters;
  PNode* waveNodes;
  HandOver ho;
  SearchPool searchPool;
  BigSetPool bigSets;
};
int xmSchedLaunch(const SchedLaunch& a, void* stream);  // xm_sched_kernel.hip; returns hipError_t as int
int xmSchedProfile(unsigned long long* out16, int reset);  // XM_PROFILE builds: the scheduler kernel's phase timers

#if defined(__HIPCC__)
__device__ __forceinline__ void addCounters(DevCounters* g, const DevCounters& l) {
  atomicAdd(&g->reads, l.reads); atomicAdd(&g->headerProbes, l.headerProbes); atomicAdd(&g->bucketFetches, l.bucketFetches);
  atomicAdd(&g->hitsFetched, l.hitsFetched); atomicAdd(&g->candidatesExtended, l.candidatesExtended); atomicAdd(&g->pathAlignerCalls, l.pathAlignerCalls);
  atomicAdd(&g->pathAlignerNodes, l.pathAlignerNodes); atomicAdd(&g->quickAccepts, l.quickAccepts); atomicAdd(&g->alignmentsOut, l.alignmentsOut);
  atomicAdd(&g->refWindowBytes, l.refWindowBytes); atomicAdd(&g->readBytes, l.readBytes);
  for (int i = 0; i < 16; i++) if (l.t[i]) atomicAdd(&g->t[i], l.t[i]);
}

// what a lane does with a read it is done with (both kernels): the result into the arenas, or the status the host's pass logic acts on
__device__ __forceinline__ void publishRead(const OutView& out, int64_t q, const ReadResult& rr, const ReadCtx& cx, DevCounters& local) {
  int32_t st = cx.status;
  if (st == XM_OK) {
    int64_t ni, nd;
    resultSize(rr, ni, nd);
    unsigned long long io = atomicAdd(&out.cursor[0], (unsigned long long)ni);
    unsigned long long dofs = atomicAdd(&out.cursor[1], (unsigned long long)nd);
    if (io + (unsigned long long)ni > out.intCap || dofs + (unsigned long long)nd > out.dblCap) {
      st = XM_ST_OUT_OVERFLOW;
    } else {
      OutWriter w;
      w.ints = out.ints + io; w.dbls = out.dbls + dofs; w.ni = 0; w.nd = 0;
      resultWrite(rr, w, &local);
      out.intOff[q] = (int64_t)io; out.dblOff[q] = (int64_t)dofs; out.intLen[q] = (int32_t)ni; out.dblLen[q] = (int32_t)nd;
    }
  }
  if (st == XM_ST_NEED_HEAVY) {  // bits 8..15: cost hint (penalty x 8, capped) for the order of the gapped pass
    float h = cx.heavyHint * 8.0f;
    int hi = h > 255.0f ? 255 : (h > 0.0f ? (int)h : 0);
    st |= hi << 8;
  }
  out.status[q] = st;
}

#endif

}  // namespace xm
