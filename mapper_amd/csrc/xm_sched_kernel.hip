// xmapper-hip: the gapped pass as a wave-level scheduler (xm_sched.h) - its own translation unit, so that the lane-per-read kernels of xm_capi.hip and
// this kernel each get their own copies of the out-of-line device functions (the register allocation of those functions follows their callers).
#define XM_NOINL_LINKAGE inline  // the out-of-line functions of the shared headers are defined (strongly) by xm_capi.hip
#include <hip/hip_runtime.h>
#include "xm_sched.h"
#include "xm_kernel_common.h"

#ifndef XM_WAVES_PER_SIMD
#define XM_WAVES_PER_SIMD 4
#endif

namespace xm {

namespace {

// XM_PROFILE builds: where the waves of the scheduler kernel spend their time (shader-clock ticks, counted by the lowest active lane of a wave):
// [0] whole loop, [1] chain phases that start a read, [2] chain phases that replay, [3] search phases, [4] of them in the big buffer, [5] loop
// iterations, [6] searches, [7] sum over search phases of (ticks x lanes searching), [8] same for chain phases (ticks x lanes with chain work), [9] publish
#if defined(XM_PROFILE)
__device__ unsigned long long xm_sched_prof[16];
#define SP_TIC(var) unsigned long long var = clock64()
#define SP_LEADER() ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1)
#define SP_ADD(slot, v) do { if (SP_LEADER()) atomicAdd(&xm_sched_prof[slot], (unsigned long long)(v)); } while (0)
#else
#define SP_TIC(var) do { } while (0)
#define SP_ADD(slot, v) do { } while (0)
#endif

// a large search set of the workgroup's XCD, or -1 (none free right now: the search stays parked and asks again in the next round)
__device__ __forceinline__ int bigSetAcquire(const BigSetPool& p, uint32_t seed) {
  if (p.nPerGroup <= 0) return -1;
  const int group = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7;  // HW_REG_XCC_ID, bits 3:0
  const int base = group * p.nPerGroup;
  for (int i = 0; i < 16; i++) {
    const int j = base + (int)((seed + (uint32_t)i * 2654435761u) % (uint32_t)p.nPerGroup);
    if (atomicCAS(&p.owner[j], 0, 1) == 0) return j;
  }
  return -1;
}
__device__ __forceinline__ void bigSetRelease(const BigSetPool& p, int idx) {
  __threadfence();
  atomicExch(&p.owner[idx], 0);
}

// The gapped pass as a wave-level scheduler (xm_sched.h): every lane holds one read; the lanes of a wave advance their chains together (chain
// phase) until each read is finished or parked at a PathAligner search, then the parked lanes run their searches together (search phase), and so
// on; a lane whose read is finished takes the next one of the list.  A lane's scratch: [region for a read that comes without saved state |
// chain temporaries | search arrays | memo].
__global__ void __launch_bounds__(256, XM_WAVES_PER_SIMD) xm_sched_kernel(IndexView ix, Params params, BatchView batch, const int64_t* todo, long long nTodo, int scale, int lanesPerWave, int quantum, int gate,
                                                       uint8_t* arenas, unsigned long long arenaBytes, SchedLayout lay, OutView out, unsigned long long* nextItem, DevCounters* counters,
                                                       PNode* waveNodes, HandOver ho, SearchPool searchPool, BigSetPool bigSets) {
  xmSetWaveNodes(waveNodes);
  xmSetPairMode(0);
  xmSetSearchPool(searchPool);
  xmLoadMergeRule();  // (every thread of the block: it ends with a barrier)
  const int laneInWave = (int)(threadIdx.x & 63u);
  if (laneInWave >= lanesPerWave) return;
  const unsigned long long lane = ((unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (unsigned)lanesPerWave + (unsigned)laneInWave;
  uint8_t* const arena = arenas + lane * arenaBytes;
  uint8_t* const tmp = arena + ho.regionBytes;
  uint8_t* const searchArena = tmp + lay.tmpBytes;
  MemoHdr* const memo = (MemoHdr*)(searchArena + lay.searchBytes);
  DevCounters local;
  memset(&local, 0, sizeof(local));
  DevCounters before = local;
  ReadCtx cx;
  ReadResult rr;
  uint8_t* curSearch = searchArena;  // the arrays of the lane's search: its own small set, or a large set of the launch's pool
  int bigIdx = -1;                   // the pool buffer the lane holds (-1: none), wantBig: its search outgrew the small set and waits for one
  bool wantBig = false;
  int state = 0;  // 0: no read, 1: chain work, 2: parked at a search
  bool fresh = false, drained = false, dealt = true;
  const long long nWaves = (long long)gridDim.x * (blockDim.x >> 6), waveIndex = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int64_t q = 0;
  SP_TIC(tLoop);
  while (true) {
    SP_ADD(5, 1);
    if (state == 0 && !drained) {
      // The list starts with the reads that look expensive (the ones with an indel: xm_heavy_list_kernel).  The first read of every lane is dealt out
      // lane-major - items 0..waves-1 to lane 0 of every wave, the next `waves` items to lane 1, ... - so that every wave gets the same number of
      // them; after that the lanes draw from the counter, which the host has set behind the dealt items.
      unsigned long long item;
      const long long mine = (long long)laneInWave * (long long)nWaves + waveIndex;
      if (dealt) { dealt = false; item = mine < nTodo ? (unsigned long long)mine : atomicAdd(nextItem, 1ull); }
      else item = atomicAdd(nextItem, 1ull);
      if ((long long)item >= nTodo) drained = true;
      else { q = todo ? todo[item] : (int64_t)item; state = 1; fresh = true; }
    }
    if (state == 0) break;  // (the list is drained: lanes leave one by one, the wave's remaining reads go on)
    // ---- chain phase.  A chain phase lasts as long as its slowest lane whatever the number of lanes in it (the chains diverge), so the lanes
    // that are ready for one wait until there are `gate` of them, as long as the wave has searches to run meanwhile.
    const int nReady = __popcll(__ballot(state == 1)), nSearching = __popcll(__ballot(state == 2));
    const bool runChain = nReady >= gate || nSearching == 0;
    if (state == 1 && runChain) {
      SP_TIC(tChain);
#if defined(XM_PROFILE)
      const bool wasFresh = fresh;
      const int nChain = __popcll(__ballot(1));
#endif
      if (fresh) {
        fresh = false;
        before = local;
        memoInitInLane(memo, (int)lay.memoBytes);
        const int32_t rg = ho.regionOf[q];
        if (rg >= 0) {
          runReadResumed(cx, savedReadOf(ho.regions + (unsigned long long)rg * ho.regionBytes, (size_t)ho.regionBytes), &ix, scale, tmp, (size_t)lay.tmpBytes, &local, rr, memo, true, 2, 0);
        } else {
          ReadIn in;
          in.nMates = batch.mateCount[q];
          for (int m = 0; m < 2; m++) {
            in.mate[m] = batch.codes + batch.mateOffset[q * 2 + m];
            in.mateLen[m] = m < in.nMates ? batch.mateLength[q * 2 + m] : 0;
          }
          in.expectedInner = in.nMates > 1 ? batch.expectedInner[q] : 0.0;
          in.deviation = in.nMates > 1 ? batch.deviation[q] : 1.0;
          runReadRetaining(cx, &ix, params, in, ho.seedScale, arena, (size_t)ho.regionBytes, tmp, (size_t)lay.tmpBytes, &local, rr, 2, scale, memo, true);
        }
      } else {
        schedReplay(cx, rr);
      }
#if defined(XM_PROFILE)
      { const unsigned long long dt = clock64() - tChain; if (wasFresh) SP_ADD(1, dt); else SP_ADD(2, dt); SP_ADD(8, dt * (unsigned long long)nChain); }
#endif
      if (schedParked(cx)) {
        state = 2;
        curSearch = searchArena; wantBig = false;
        schedSearchBegin(memo, searchArena, cx.caps);
      } else {
        if (cx.status != XM_OK) local = before;  // work of a read that is rerun by a later pass is counted there
        publishRead(out, q, rr, cx, local);
        state = 0;
      }
    }
    // ---- search phase
    if (state == 2) {
      SP_TIC(tSearch);
#if defined(XM_PROFILE)
      const int nSearch = __popcll(__ballot(1));
      SP_ADD(6, nSearch);
#endif
      // (every search of the wave gets `quantum` explored entries per round: a search of thousands of entries does not keep the lanes whose
      // searches took a hundred from their chains)
      if (wantBig) {
        bigIdx = bigSetAcquire(bigSets, (uint32_t)lane * 40503u + (uint32_t)q);
        if (bigIdx >= 0) { wantBig = false; curSearch = bigSets.base + (unsigned long long)bigIdx * bigSets.bufBytes; schedSearchRestartBig(memo, curSearch, cx.caps); }
      }
      int how = wantBig ? 0 : schedSearchRun(memo, curSearch, quantum, &local);
      if (how == 2 && bigSets.nPerGroup > 0) { wantBig = true; how = 0; }  // (a large set: asked for at the start of the next round)
      if (how == 3) { schedLogResult(memo, false, nullptr, 0, XM_ST_OVERFLOW); how = 1; }
      if (how != 0 && bigIdx >= 0) { bigSetRelease(bigSets, bigIdx); bigIdx = -1; curSearch = searchArena; }
      const bool over = how != 0;
      const bool done = how != 2;
#if defined(XM_PROFILE)
      { const unsigned long long dt = clock64() - tSearch; SP_ADD(3, dt); SP_ADD(7, dt * (unsigned long long)nSearch); }
      {  // entries explored in this round: the most by one lane (= the round's length) and by all lanes together
        const int mySteps = ((const WSearch*)curSearch)->lastSteps;
        unsigned long long m = __ballot(1);
        int mx = 0; long long sum = 0;
        while (m) { const int l = __ffsll((long long)m) - 1; const int v = __shfl(mySteps, l); mx = v > mx ? v : mx; sum += v; m &= m - 1; }
        SP_ADD(10, mx); SP_ADD(11, sum);
      }
      SP_TIC(tBig);
#endif
      unsigned long long waiting = __ballot(done ? 0 : 1);
      while (waiting) {
        const int leader = __ffsll((long long)waiting) - 1;
        if (!done && laneInWave == leader) {
          Arena wb;
          if (xmWaveSearchBuffer(wb)) schedSearchBig(memo, wb, cx.caps, &local);
          else schedLogResult(memo, false, nullptr, 0, XM_ST_OVERFLOW);  // (no buffer: the read runs again in a pass with more scratch)
        }
        waiting &= waiting - 1;
      }
#if defined(XM_PROFILE)
      SP_ADD(4, clock64() - tBig);
#endif
      if (over) state = 1;
    }
  }
#if defined(XM_PROFILE)
  SP_ADD(0, clock64() - tLoop);
#endif
  addCounters(counters, local);
}


}  // namespace

// XM_PROFILE builds: the scheduler kernel's phase timers since the last call (reset != 0: cleared); 0 in product builds
int xmSchedProfile(unsigned long long* out16, int reset) {
#if defined(XM_PROFILE)
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(xm_sched_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  unsigned long long ws[8];
  if (hipMemcpyFromSymbol(ws, HIP_SYMBOL(xm_ws_prof), sizeof(ws)) != hipSuccess) return 1;
  out16[9] = ws[0]; out16[12] = ws[1]; out16[13] = ws[2]; out16[14] = ws[3]; out16[15] = ws[4]; out16[4] = ws[5];  // (the search's own timers: poll, list entry, lookups, arithmetic, puts, state)
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(xm_sched_prof), z, sizeof(z)) != hipSuccess) return 1; if (hipMemcpyToSymbol(HIP_SYMBOL(xm_ws_prof), z, sizeof(ws)) != hipSuccess) return 1; }
  return 0;
#else
  for (int i = 0; i < 16; i++) out16[i] = 0;
  (void)reset;
  return 0;
#endif
}

int xmSchedLaunch(const SchedLaunch& a, void* stream) {
  hipLaunchKernelGGL(xm_sched_kernel, dim3(a.grid), dim3(a.block), 0, (hipStream_t)stream, a.ix, a.params, a.batch, a.todo, a.nTodo, a.scale, a.lanesPerWave, a.quantum, a.gate, a.arenas, a.arenaBytes, a.lay, a.out,
                     a.nextItem, a.counters, a.waveNodes, a.ho, a.searchPool, a.bigSets);
  return (int)hipGetLastError();
}

}  // namespace xm
