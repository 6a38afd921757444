// xmapper-hip: the wave-per-read kernels (xm_wave.h): one wavefront aligns one read, its state lives in the wave's share of LDS.
//   config 0 / 1: light tier, single-end batches / batches with pairs (seed, vote, ungapped alignment, accept)
//   config 2 / 3: chain tier: + the gapped chain of xm_wave_chain.h (HashBlock_Aligner, BlockAligner, the StraightAligners) up to PathAligner's search
//   config 4:     the chain tier with the largest capacities
// A tier hands a read it cannot finish to the next one (XM_ST_WAVE_GAPPED).  PathAligner's searches are requests in the reads' memos
// (XM_ST_WAVE_SEARCH): xm_wave_search_kernel runs them, one wavefront per search, and the chain tier runs those reads again.
// A read the wave form does not take leaves with XM_ST_WAVE_FALLBACK and is aligned by the lane-per-read kernel of xm_capi.hip.
#define XM_NOINL_LINKAGE inline  // the out-of-line functions of the shared headers are defined (strongly) by xm_capi.hip
#include <hip/hip_runtime.h>
#ifndef WV_CONFIG
#error "compile with -DWV_CONFIG=0..4 (one wave-per-read kernel configuration), 5 (search kernel) or 100 (geometry, dispatch)"
#endif
#ifndef WV_SE_MINWAVES
#define WV_SE_MINWAVES 4  // waves per SIMD the single-end light kernel is compiled for (register budget 512 / that)
#endif
#include "xm_wave.h"
#include "xm_kernel_args.h"

namespace xm {

namespace {

#if WV_CONFIG <= 4
__device__ __forceinline__ void waveAddCounters(DevCounters* g, const DevCounters& l) {
  atomicAdd(&g->reads, l.reads); atomicAdd(&g->headerProbes, l.headerProbes); atomicAdd(&g->bucketFetches, l.bucketFetches);
  atomicAdd(&g->hitsFetched, l.hitsFetched); atomicAdd(&g->candidatesExtended, l.candidatesExtended); atomicAdd(&g->pathAlignerCalls, l.pathAlignerCalls);
  atomicAdd(&g->pathAlignerNodes, l.pathAlignerNodes); atomicAdd(&g->quickAccepts, l.quickAccepts); atomicAdd(&g->alignmentsOut, l.alignmentsOut);
  atomicAdd(&g->refWindowBytes, l.refWindowBytes); atomicAdd(&g->readBytes, l.readBytes);
  for (int i = 0; i < 16; i++) if (l.t[i]) atomicAdd(&g->t[i], l.t[i]);
}
#endif
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

#if WV_CONFIG <= 4
template <class CFG, int TIER, int WAVES, int MINWAVES>
__global__ void __launch_bounds__(WAVES * 64, MINWAVES) xm_wave_kernel(IndexView ix, Params params, BatchView batch, const int64_t* todo, long long nTodo, OutView out,
                                                                       unsigned long long* nextItem, DevCounters* counters, WMemo* memoBase, const int32_t* slotOf, WSNode* waveNodes, int itemsPerFetch) {
  typedef WaveLdsT<CFG> LDS;
  __shared__ LDS lds[WAVES];
  xmWaveLoadTables(ix);
  xmLoadMergeRule();  // (ends with a barrier)
  const int waveInBlock = (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63u);
  XM_LDSP(LDS)* L = (XM_LDSP(LDS)*)&lds[waveInBlock];
  DevCounters local;
  memset(&local, 0, sizeof(local));
  WEnv e;
  e.ix = ix; e.params = params; e.params.StartingInsertionStartFree = 0; e.dc = &local; e.tier = TIER; e.caps = nullptr; e.tmp = nullptr; e.memo = nullptr;
  e.searchNodes = TIER >= 1 && waveNodes ? waveNodes + ((unsigned long long)blockIdx.x * WAVES + (unsigned)waveInBlock) * WSearchLdsInline::kNodes : nullptr;
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
      if (TIER >= 1) e.memo = memoBase + slotOf[q];
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
                     a.memoBase, a.slotOf, (WSNode*)a.waveNodes, a.itemsPerFetch);
  return hipGetLastError();
}

#endif
#if WV_CONFIG == 5
// One wavefront per waiting search (wPathSearch, xm_wave_search.h): lookup structures in the wave's LDS, node payloads in the wave's buffer
// in HBM; the result goes into the read's memo.
#ifndef WV_SEARCH_WAVES
#define WV_SEARCH_WAVES 1  // waves per workgroup (the search kernel only sees the searches that outgrow the chain tiers' inline capacities)
#endif
#ifndef WV_SEARCH_MINWAVES
#define WV_SEARCH_MINWAVES 1
#endif
#ifndef WV_SEARCH_LDS
#define WV_SEARCH_LDS WSearchLdsKernel  // (experiments: WSearchLdsInline = the capacities of the chain tiers' inline searches)
#endif
__global__ void __launch_bounds__(WV_SEARCH_WAVES * 64, WV_SEARCH_MINWAVES) xm_wave_search_kernel(IndexView ix, Params params, BatchView batch, const int64_t* list, long long n, WMemo* memoBase, const int32_t* slotOf,
                                                                                  unsigned long long* nextItem, WSNode* waveNodes, DevCounters* counters) {
  __shared__ WV_SEARCH_LDS lds[WV_SEARCH_WAVES];
  DevCounters local;
  memset(&local, 0, sizeof(local));
  const int waveInBlock = (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63u);
  const unsigned long long waveIndex = (unsigned long long)blockIdx.x * WV_SEARCH_WAVES + (unsigned)waveInBlock;
  XM_LDSP(WV_SEARCH_LDS)* S = (XM_LDSP(WV_SEARCH_LDS)*)&lds[waveInBlock];
  WSNode* nodes = waveNodes + waveIndex * WV_SEARCH_LDS::kNodes;
  params.StartingInsertionStartFree = 0;
  while (true) {
    unsigned long long item = 0;
    if (lane == 0) item = atomicAdd(nextItem, 1ull);
    item = uni64(item);
    if ((long long)item >= n) break;
    const int64_t q = list[item];
    WMemo* M = memoBase + slotOf[q];
    const int mi = M->req.seqAId >> 1;
    wRunSearch(S, nodes, ix, params, batch.codes + batch.mateOffset[q * 2 + mi], batch.mateLength[q * 2 + mi], M, &local);
  }
  if (lane == 0) for (int i = 0; i < 16; i++) if (local.t[i]) atomicAdd(&counters->t[i], local.t[i]);
}
#endif
#if WV_CONFIG == 100
__global__ void __launch_bounds__(256) xm_wave_memo_init_kernel(WMemo* memoBase, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { memoBase[i].count = 0; memoBase[i].pending = 0; }
}

// Test entry (xm_test_local_align, modes 2 and 3): ONE wave-cooperative PathAligner search over a query and a reference text given as they are
// (the reference's PathAligner_Test cases), with the search kernel's capacities (BIG) or the capacities of the chain tiers' inline searches.
template <class SL>
__global__ void __launch_bounds__(64) xm_test_wave_search_kernel(TestSearch t) {
  __shared__ SL lds;
  XM_LDSP(SL)* S = (XM_LDSP(SL)*)&lds;
  WSearchReq req;
  req.seqAId = 0; req.contig = 0; req.qsStart = 0; req.qsEnd = t.queryLength; req.rsStart = 0; req.rsEnd = t.referenceLength;
  req.predictedBestOffset = t.predictedBestOffset; req.confident = t.confident; req.startingInsertionStartFree = 0; req.pad = 0;
  req.maxIns = t.maxIns; req.maxDel = t.maxDel; req.maxErrorRate = t.params.MaxErrorRate;
  WSearchResult res;
  wPathSearch(S, (WSNode*)t.nodes, t.ix, t.params, t.query, t.queryLength, req, res, nullptr);
  if ((threadIdx.x & 63u) == 0) {
    t.outInts[0] = res.ok; t.outInts[1] = res.nb; t.outInts[2] = res.status; t.outInts[3] = res.nodesPut;
    for (int i = 0; i < res.nb && i < t.blockCap; i++) { t.outInts[4 + 4 * i] = res.blocks[i].startA; t.outInts[5 + 4 * i] = res.blocks[i].startB; t.outInts[6 + 4 * i] = res.blocks[i].lenA; t.outInts[7 + 4 * i] = res.blocks[i].lenB; }
    t.outDbls[0] = res.totalPenalty; t.outDbls[1] = res.alignedPenalty;
  }
}
#endif
}  // namespace

#if WV_CONFIG == 0
int xmWaveLaunch0(const WaveLaunch& a, void* s) { return (int)launchOne<WCfgLightSE, 0, 4, WV_SE_MINWAVES>(a, (hipStream_t)s); }
#elif WV_CONFIG == 1
int xmWaveLaunch1(const WaveLaunch& a, void* s) { return (int)launchOne<WCfgLightPE, 0, 2, 3>(a, (hipStream_t)s); }
#elif WV_CONFIG == 2
int xmWaveLaunch2(const WaveLaunch& a, void* s) { return (int)launchOne<WCfgMidSE, 1, 1, 2>(a, (hipStream_t)s); }
#elif WV_CONFIG == 3
int xmWaveLaunch3(const WaveLaunch& a, void* s) { return (int)launchOne<WCfgMidPE, 1, 1, 2>(a, (hipStream_t)s); }
#elif WV_CONFIG == 4
int xmWaveLaunch4(const WaveLaunch& a, void* s) { return (int)launchOne<WCfgHeavy, 1, 1, 1>(a, (hipStream_t)s); }
#elif WV_CONFIG == 5
int xmSearchLaunch(const SearchLaunch& a, void* stream) {
  hipLaunchKernelGGL(xm_wave_search_kernel, dim3(a.grid), dim3(WV_SEARCH_WAVES * 64), 0, (hipStream_t)stream, a.ix, a.params, a.batch, a.list, a.n, a.memoBase, a.slotOf, a.nextItem,
                     (WSNode*)a.waveNodes, a.counters);
  return (int)hipGetLastError();
}
void xmSearchGeometry(int* wavesPerBlock, int* ldsBytesPerBlock, int* wavesPerSimd, int* memoBytes, int* nodeBytesPerWave) {
  *wavesPerBlock = WV_SEARCH_WAVES; *wavesPerSimd = WV_SEARCH_MINWAVES; *ldsBytesPerBlock = WV_SEARCH_WAVES * (int)sizeof(WV_SEARCH_LDS);
  *memoBytes = (int)sizeof(WMemo); *nodeBytesPerWave = WV_SEARCH_LDS::kNodes * (int)sizeof(WSNode);
}
#elif WV_CONFIG == 100
int xmWaveLaunch0(const WaveLaunch&, void*); int xmWaveLaunch1(const WaveLaunch&, void*); int xmWaveLaunch2(const WaveLaunch&, void*);
int xmWaveLaunch3(const WaveLaunch&, void*); int xmWaveLaunch4(const WaveLaunch&, void*);
int xmWaveInlineNodeBytes() { return WSearchLdsInline::kNodes * (int)sizeof(WSNode); }
void xmWaveGeometry(int config, int* wavesPerBlock, int* ldsBytesPerBlock, int* wavesPerSimd) {
  const int shared = (int)sizeof(MergeRuleTable) + WV_TABLECACHE * (int)sizeof(Table);
  if (config == 0) { *wavesPerBlock = 4; *wavesPerSimd = WV_SE_MINWAVES; *ldsBytesPerBlock = 4 * (int)sizeof(WaveLdsT<WCfgLightSE>) + shared; }
  else if (config == 1) { *wavesPerBlock = 2; *wavesPerSimd = 3; *ldsBytesPerBlock = 2 * (int)sizeof(WaveLdsT<WCfgLightPE>) + shared; }
  else if (config == 2) { *wavesPerBlock = 1; *wavesPerSimd = 2; *ldsBytesPerBlock = (int)sizeof(WaveLdsT<WCfgMidSE>) + shared; }
  else if (config == 3) { *wavesPerBlock = 1; *wavesPerSimd = 2; *ldsBytesPerBlock = (int)sizeof(WaveLdsT<WCfgMidPE>) + shared; }
  else { *wavesPerBlock = 1; *wavesPerSimd = 1; *ldsBytesPerBlock = (int)sizeof(WaveLdsT<WCfgHeavy>) + shared; }
}
int xmWaveLaunch(const WaveLaunch& a, void* stream) {
  if (a.config == 0) return xmWaveLaunch0(a, stream);
  if (a.config == 1) return xmWaveLaunch1(a, stream);
  if (a.config == 2) return xmWaveLaunch2(a, stream);
  if (a.config == 3) return xmWaveLaunch3(a, stream);
  return xmWaveLaunch4(a, stream);
}
int xmTestWaveSearchNodeBytes(int big) { return (big ? WSearchLdsKernel::kNodes : WSearchLdsInline::kNodes) * (int)sizeof(WSNode); }
int xmTestWaveSearchLaunch(const TestSearch& t, void* stream) {
  if (t.big) hipLaunchKernelGGL((xm_test_wave_search_kernel<WSearchLdsKernel>), dim3(1), dim3(64), 0, (hipStream_t)stream, t);
  else hipLaunchKernelGGL((xm_test_wave_search_kernel<WSearchLdsInline>), dim3(1), dim3(64), 0, (hipStream_t)stream, t);
  return (int)hipGetLastError();
}
int xmMemoInitLaunch(WMemo* memoBase, long long n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(xm_wave_memo_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, memoBase, n);
  return (int)hipGetLastError();
}
#endif

}  // namespace xm
