// xmapper-hip: the wave-per-read kernels (xm_wave.h): one wavefront aligns one read, its state lives in the wave's share of LDS.
//   config 0: light tier, single-end batches   (seed, vote, ungapped alignment, accept; 4 waves per workgroup)
//   config 1: light tier, batches with pairs
//   config 2: heavy tier: everything the light tier hands on (reads that need the gapped chain HashBlock_Aligner -> BlockAligner ->
//             PathAligner of xm_extend.h, reads that outgrow the light capacities); one wave per workgroup, scratch arena per wave in HBM
// A read the wave form does not take leaves with XM_ST_WAVE_FALLBACK and is aligned by the lane-per-read kernel of xm_capi.hip.
#define XM_NOINL_LINKAGE inline  // the out-of-line functions of the shared headers are defined (strongly) by xm_capi.hip
#define XM_PAL_WAVES 1           // PathAligner's LDS slot: one per wave of a workgroup; the heavy tier runs one wave per workgroup
#define XM_WAVE_UNIFORM 1        // all lanes of a wave work on the same read (pathAlign: no turns at the slot)
#include <hip/hip_runtime.h>
#ifndef WV_SE_MINWAVES
#define WV_SE_MINWAVES 4  // waves per SIMD the single-end light kernel is compiled for (register budget 512 / that)
#endif
#include "xm_wave.h"
#include "xm_kernel_args.h"

namespace xm {

namespace {

__device__ __forceinline__ void waveAddCounters(DevCounters* g, const DevCounters& l) {
  atomicAdd(&g->reads, l.reads); atomicAdd(&g->headerProbes, l.headerProbes); atomicAdd(&g->bucketFetches, l.bucketFetches);
  atomicAdd(&g->hitsFetched, l.hitsFetched); atomicAdd(&g->candidatesExtended, l.candidatesExtended); atomicAdd(&g->pathAlignerCalls, l.pathAlignerCalls);
  atomicAdd(&g->pathAlignerNodes, l.pathAlignerNodes); atomicAdd(&g->quickAccepts, l.quickAccepts); atomicAdd(&g->alignmentsOut, l.alignmentsOut);
  atomicAdd(&g->refWindowBytes, l.refWindowBytes); atomicAdd(&g->readBytes, l.readBytes);
  for (int i = 0; i < 16; i++) if (l.t[i]) atomicAdd(&g->t[i], l.t[i]);
}
__device__ __forceinline__ void saveCounters(const DevCounters& l, unsigned long long* b) {
  b[0] = l.reads; b[1] = l.headerProbes; b[2] = l.bucketFetches; b[3] = l.hitsFetched; b[4] = l.candidatesExtended; b[5] = l.pathAlignerCalls; b[6] = l.pathAlignerNodes;
  b[7] = l.quickAccepts; b[8] = l.alignmentsOut; b[9] = l.refWindowBytes; b[10] = l.readBytes;
}
__device__ __forceinline__ void restoreCounters(DevCounters& l, const unsigned long long* b) {
  l.reads = b[0]; l.headerProbes = b[1]; l.bucketFetches = b[2]; l.hitsFetched = b[3]; l.candidatesExtended = b[4]; l.pathAlignerCalls = b[5]; l.pathAlignerNodes = b[6];
  l.quickAccepts = b[7]; l.alignmentsOut = b[8]; l.refWindowBytes = b[9]; l.readBytes = b[10];
}
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}

template <class CFG, int TIER, int WAVES, int MINWAVES>
__global__ void __launch_bounds__(WAVES * 64, MINWAVES) xm_wave_kernel(IndexView ix, Params params, BatchView batch, const int64_t* todo, long long nTodo, OutView out,
                                                                       unsigned long long* nextItem, DevCounters* counters, uint8_t* arenas, unsigned long long arenaBytes,
                                                                       int chainScale, PNode* waveNodes, int itemsPerFetch) {
  typedef WaveLdsT<CFG> LDS;
  __shared__ LDS lds[WAVES];
  if (TIER == 1) { xmSetWaveNodes(waveNodes); xmSetPairMode(0); }
  xmWaveLoadTables(ix);
  xmLoadMergeRule();  // (ends with a barrier)
  const int waveInBlock = (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63u);
  XM_LDSP(LDS)* L = (XM_LDSP(LDS)*)&lds[waveInBlock];
  const unsigned long long waveIndex = (unsigned long long)blockIdx.x * WAVES + (unsigned)waveInBlock;
  DevCounters local;
  memset(&local, 0, sizeof(local));
  Caps caps = makeCaps(TIER == 1 ? chainScale : 1);
  caps.heavyAllowed = 2; caps.deferPath = 0;
  Arena tmp;
  tmp.init(arenas ? arenas + waveIndex * arenaBytes : nullptr, arenas ? (size_t)arenaBytes : 0);
  WEnv e;
  e.ix = ix; e.params = params; e.params.StartingInsertionStartFree = 0; e.dc = &local; e.tier = TIER; e.caps = &caps; e.tmp = &tmp;
  while (true) {
    unsigned long long first = 0;
    if (lane == 0) first = atomicAdd(nextItem, (unsigned long long)itemsPerFetch);
    first = uni64(first);
    if ((long long)first >= nTodo) break;
    for (int k = 0; k < itemsPerFetch; k++) {
      const long long item = (long long)first + k;
      if (item >= nTodo) break;
      const int64_t q = todo ? todo[item] : (int64_t)item;
      ReadIn in;
      in.nMates = batch.mateCount[q];
      for (int m = 0; m < 2; m++) {
        in.mate[m] = batch.codes + batch.mateOffset[q * 2 + m];
        in.mateLen[m] = m < in.nMates ? batch.mateLength[q * 2 + m] : 0;
        e.mateBase[m] = in.mate[m];
      }
      // single-end Query: expectedInnerDistance 0, deviation 1 (spacing penalty is always 0, T/SamWriter_Test.java:26)
      in.expectedInner = in.nMates > 1 ? batch.expectedInner[q] : 0.0;
      in.deviation = in.nMates > 1 ? batch.deviation[q] : 1.0;
      unsigned long long before[11];
      saveCounters(local, before);
      WResult rr;
      wAlignRead(L, e, in, rr);
      int32_t st = wvUni(L->status);
      if (st == XM_OK) {
        int64_t ni, nd;
        wResultSize(L, rr, ni, nd);
        unsigned long long io = 0, dofs = 0;
        if (lane == 0) { io = atomicAdd(&out.cursor[0], (unsigned long long)ni); dofs = atomicAdd(&out.cursor[1], (unsigned long long)nd); }
        io = uni64(io); dofs = uni64(dofs);
        if (io + (unsigned long long)ni > out.intCap || dofs + (unsigned long long)nd > out.dblCap) {
          st = XM_ST_WAVE_FALLBACK;  // result arena full: the lane-per-read passes grow it and align the read
        } else if (lane == 0) {
          WV_TIMER(e, WT_WRITE);
          wResultWrite(L, rr, out.ints + io, out.dbls + dofs, &local);
          out.intOff[q] = (int64_t)io; out.dblOff[q] = (int64_t)dofs; out.intLen[q] = (int32_t)ni; out.dblLen[q] = (int32_t)nd;
        }
      }
      if (st != XM_OK) restoreCounters(local, before);  // work of a read that another pass aligns is counted there
      if (lane == 0) out.status[q] = st;
    }
  }
  if (lane == 0) waveAddCounters(counters, local);
}

template <class CFG, int TIER, int WAVES, int MINWAVES>
hipError_t launchOne(const WaveLaunch& a, hipStream_t s) {
  hipLaunchKernelGGL((xm_wave_kernel<CFG, TIER, WAVES, MINWAVES>), dim3(a.grid), dim3(WAVES * 64), 0, s, a.ix, a.params, a.batch, a.todo, a.nTodo, a.out, a.nextItem, a.counters,
                     a.arenas, a.arenaBytes, a.chainScale, (PNode*)a.waveNodes, a.itemsPerFetch);
  return hipGetLastError();
}

}  // namespace

void xmWaveGeometry(int config, int* wavesPerBlock, int* ldsBytesPerBlock, int* wavesPerSimd) {
  const int mergeRule = (int)sizeof(MergeRuleTable);
  if (config == 0) { *wavesPerBlock = 4; *wavesPerSimd = WV_SE_MINWAVES; *ldsBytesPerBlock = 4 * (int)sizeof(WaveLdsT<WCfgLightSE>) + mergeRule; }
  else if (config == 1) { *wavesPerBlock = 2; *wavesPerSimd = 3; *ldsBytesPerBlock = 2 * (int)sizeof(WaveLdsT<WCfgLightPE>) + mergeRule; }
  else { *wavesPerBlock = 1; *wavesPerSimd = 1; *ldsBytesPerBlock = (int)sizeof(WaveLdsT<WCfgHeavy>) + XM_PAL_SLOT_BYTES + mergeRule + 16; }
}

int xmWaveLaunch(const WaveLaunch& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.config == 0) return (int)launchOne<WCfgLightSE, 0, 4, WV_SE_MINWAVES>(a, s);
  if (a.config == 1) return (int)launchOne<WCfgLightPE, 0, 2, 3>(a, s);
  return (int)launchOne<WCfgHeavy, 1, 1, 1>(a, s);
}

}  // namespace xm
