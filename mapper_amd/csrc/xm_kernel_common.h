// Device helpers and launch structures of the kernels of xm_capi.hip (lane-per-read passes); the structures are what the host passes.
#pragma once
#include "xm_worker.h"
#include "xm_kernel_args.h"

namespace xm {

// Light pass -> gapped pass hand-over (xm_worker.h, SavedRead).  mode 1 (light pass): a read's persistent arena is a region of the pool
// below, the lane's arena holds the temporaries only; a read that stops in front of the gapped chain keeps its region (regionOf[q]) and the
// lane takes a fresh one.  mode 2 (gapped pass): a read with a saved region continues from it on whatever lane picks it up; the lane's
// arena = [one region for reads without saved state | temporaries].  mode 0: plain runRead in the lane's arena.
struct HandOver {
  int mode;
  int seedScale;                   // scale the regions are sized for (the light pass's)
  uint8_t* regions;
  unsigned long long regionBytes;
  long long nRegions;
  int32_t* regionOf;               // per read: region that holds its SavedRead, -1 = none
  unsigned long long* cursor;      // next unused region
};


// After every pass the reads it could not finish are in the work lists of the passes still to come; only the list sizes travel to the host.
struct PassCtl {
  unsigned long long nHeavy, nHeavyLate, nScale[2], nOut[2];
  unsigned long long errQuery;  // smallest query index whose status is an error (~0 = none)
  unsigned long long nConf[2];  // reads that wait for a value of the confidence table (XM_ST_NEED_CONF)
};
// The lists a pass files its unfinished reads into, as the lanes publish them (a read's status decides; no kernel behind the pass, whose few microseconds
// of work waited ~12 ms on average for a wave slot behind the other contexts' persistent launches: 12 % of the traced GPU time of round 4's headline run)
struct PassLists {
  int64_t* heavy; int64_t* heavyLate;  // light pass: reads for the gapped pass - the ones whose straight alignment cost hintThreshold / 8 penalty units or more first
  int64_t* scale; int64_t* out; int64_t* conf;  // the halves [ts] / [to] / [tc] of the double-buffered lists (the pass may be reading the other halves)
  int hintThreshold, ts, to, tc;
  PassCtl* ctl;
};

#if defined(__HIPCC__)
__device__ __forceinline__ void addCounters(DevCounters* g, const DevCounters& l) {
  atomicAdd(&g->reads, l.reads); atomicAdd(&g->headerProbes, l.headerProbes); atomicAdd(&g->bucketFetches, l.bucketFetches);
  atomicAdd(&g->hitsFetched, l.hitsFetched); atomicAdd(&g->candidatesExtended, l.candidatesExtended); atomicAdd(&g->pathAlignerCalls, l.pathAlignerCalls);
  atomicAdd(&g->pathAlignerNodes, l.pathAlignerNodes); atomicAdd(&g->quickAccepts, l.quickAccepts); atomicAdd(&g->alignmentsOut, l.alignmentsOut);
  atomicAdd(&g->refWindowBytes, l.refWindowBytes); atomicAdd(&g->readBytes, l.readBytes);
  if (l.boundChecks | l.boundPieceChecks) { atomicAdd(&g->boundChecks, l.boundChecks); atomicAdd(&g->boundRejects, l.boundRejects); atomicAdd(&g->boundCells, l.boundCells); atomicAdd(&g->boundPieceChecks, l.boundPieceChecks); atomicAdd(&g->boundPieceRejects, l.boundPieceRejects); }
  for (int i = 0; i < 16; i++) if (l.t[i]) atomicAdd(&g->t[i], l.t[i]);
}

// what a lane does with a read it is done with: the result into the arenas, or the status the host's pass logic acts on
// -> the status the read ends this pass with (XM_ST_OUT_OVERFLOW is only known here: the caller takes the read's work out of its counters then, as for every read a later pass runs again)
__device__ __forceinline__ int32_t publishRead(const OutView& out, int64_t q, const ReadResult& rr, const ReadCtx& cx, DevCounters& local, const PassLists& L) {
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
  out.status[q] = st;
  if (st == XM_OK) return st;
  if (st == XM_ST_NEED_HEAVY) {
    // reads whose straight alignment was bad enough for an indel go to the front of the gapped pass: they are the long ones, and a
    // launch ends with its longest wave (longest-processing-time-first).  Cost hint: penalty x 8, capped
    const float h = cx.heavyHint * 8.0f;
    const int hi = h > 255.0f ? 255 : (h > 0.0f ? (int)h : 0);
    if (hi >= L.hintThreshold) L.heavy[atomicAdd(&L.ctl->nHeavy, 1ull)] = q;
    else L.heavyLate[atomicAdd(&L.ctl->nHeavyLate, 1ull)] = q;
  } else if (st == XM_ST_OVERFLOW) L.scale[atomicAdd(&L.ctl->nScale[L.ts], 1ull)] = q;
  else if (st == XM_ST_OUT_OVERFLOW) L.out[atomicAdd(&L.ctl->nOut[L.to], 1ull)] = q;
  else if (st == XM_ST_NEED_CONF) L.conf[atomicAdd(&L.ctl->nConf[L.tc], 1ull)] = q;
  else atomicMin(&L.ctl->errQuery, (unsigned long long)q);
  return st;
}

#endif

}  // namespace xm
