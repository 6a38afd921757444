// libxmapper_hip.so: kernels, HBM residency and the C ABI of include/xmapper_hip.h.
//
// Kernel design (round 1): one read per lane, persistent lanes.  The per-read algorithm is a long, data-dependent
// state machine (adaptive pyramid walk -> bucket probes -> votes -> ungapped check -> best-first gapped search), so a
// lane pulls reads from a global counter until the batch is drained; each lane owns a scratch arena in HBM (sized by
// `scale`), the index is read-only in HBM (bucket header = 2 adjacent u32, positions contiguous per bucket, reference
// one 4-bit code per byte).  Results go to a bump-allocated result arena and are put into query order on the host.
// Reads whose fixed-capacity scratch overflowed are rerun by a second launch with a 4x larger scale (never on the CPU).
#include "../../include/xmapper_hip.h"
#include "xm_worker.h"
#include "xm_wsearch.h"
#include "xm_index_host.h"
#include "xm_kernel_args.h"
#include "xm_kernel_common.h"
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <unordered_set>
#include <mutex>
#include <shared_mutex>
#include <memory>
#include <atomic>
#include <array>
#include <type_traits>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <cstdio>

namespace xm { bool deviceHashLengths(HostIndex& h, int minLen, int maxLen, int device); }  // xm_index_device.hip
using namespace xm;

namespace {

thread_local std::string g_error;
int fail(const std::string& msg) { g_error = msg; return 1; }

#define HIP_CHECK(expr)                                                                                     \
  do {                                                                                                      \
    hipError_t _e = (expr);                                                                                 \
    if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e));      \
  } while (0)

static long long envInt(const char* name, long long dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoll(v) : dflt;
}

// experiment knobs are validated: a value outside [lo, hi] (or not a power of two where the capacities need one) is an error, not a silent corruption
static long long envKnob(const char* name, long long dflt, long long lo, long long hi, bool pow2 = false) {
  const long long v = envInt(name, dflt);
  if (v < lo || v > hi || (pow2 && (v & (v - 1)) != 0))
    throw std::runtime_error(std::string(name) + "=" + std::to_string(v) + " is not valid: expected " + (pow2 ? "a power of two in " : "a value in ") + std::to_string(lo) + ".." + std::to_string(hi));
  return v;
}

#ifndef XM_WAVES_PER_SIMD
#define XM_WAVES_PER_SIMD 4  // 128 registers per lane: the path is latency-bound, four waves per SIMD hide more of it than the spills cost
#endif

#ifdef XM_READ_TIMES
// diagnostic builds (-DXM_READ_TIMES, XM_READ_TIMES_FILE=path): shader-clock ticks the last pass spent on every read, written to the file
__device__ unsigned long long* xm_read_times = nullptr;
#endif
// One lane aligns one read at a time (AlignerWorker.align, M/AlignerWorker.java:256-484) and loops until the batch is drained.
__global__ void __launch_bounds__(256, XM_WAVES_PER_SIMD) xm_align_kernel(IndexView ix, Params params, BatchView batch, const int64_t* todo, long long nTodo, int scale, int heavyAllowed, int lanesPerWave,
                                                       uint8_t* arenas, unsigned long long arenaBytes, OutView out, unsigned long long* nextItem, DevCounters* counters,
                                                       long long taperUnit, long long firstStride, PNode* waveNodes, HandOver ho, int pairLanes,
                                                       SearchPool searchPool, PassLists lists, int boundFilter) {
  // lanesPerWave < 64 (gapped pass): the extension chain diverges so much that a wave runs its reads nearly one after another, so
  // spreading them over more, partly filled waves shortens the critical path; the idle lanes own no scratch arena
  xmSetWaveNodes(waveNodes);
  xmSetPairMode(pairLanes);
  xmSetSearchPool(searchPool);
  xmSetBoundFilter(boundFilter);  // gapped passes of long reads: the rejection filter in front of PathAligner's searches (xm_bound.h)
  xmLoadMergeRule();  // (every thread of the block: it ends with a barrier)
  // pairLanes (gapped pass, lanesPerWave <= 32): a read is run by 2^pairLanes adjacent lanes doing the same work (xm_extend.h, xmSetPairMode: 1 = two lanes,
  // 3 = eight, passes of long reads); `laneInWave` below is the read's slot in the wave, `second` marks the lanes that leave atomics and result writes to the first
  const int physLane = (int)(threadIdx.x & 63u);
  const int groupMask = (1 << pairLanes) - 1;
  const int laneInWave = physLane >> pairLanes;
  const bool second = (physLane & groupMask) != 0;
  if (laneInWave >= lanesPerWave) return;
  unsigned long long lane = ((unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (unsigned)lanesPerWave + (unsigned)laneInWave;
  uint8_t* arena = arenas + lane * arenaBytes;
  long long myRegion = (long long)lane;  // (mode 1) the pool's first regions are the lanes' initial ones, the cursor starts behind them
  DevCounters local;
  memset(&local, 0, sizeof(local));
  ReadCtx cx;
  // Gapped pass: the list starts with the reads that look expensive.  The first read of every lane is dealt out lane-major (items
  // 0..waves-1 to lane 0 of every wave, the next `waves` items to lane 1, ...), so that every wave gets the same number of them and
  // they all start at once; after that the lanes draw from the counter, which the host has set behind the dealt items.
  bool dealt = firstStride > 0;
  const long long waveIndex = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  while (true) {
    unsigned long long item;
    const long long mine = (long long)laneInWave * firstStride + waveIndex;
    if (dealt && mine < nTodo) {
      dealt = false;
      item = (unsigned long long)mine;
    } else {
      dealt = false;
      {
        // End of the work list (gapped pass): the lanes of a wave run their reads mostly one after the other, so when the list runs dry
        // every wave would still hold lanesPerWave unfinished reads and the launch would end with that long serial tail.  The higher
        // lanes therefore stop taking reads early; the last reads are spread one per wave.
        // (pair mode: the read's first lane decides for both - two separate loads of the counter could differ, and a lane that left alone
        // would leave its partner exchanging values with an inactive lane)
        int leave = 0;
        if (taperUnit > 0 && laneInWave > 0 && !second) {
          long long remaining = nTodo - (long long)__hip_atomic_load(nextItem, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          leave = remaining < (long long)laneInWave * taperUnit;
        }
        if (pairLanes) leave = __shfl(leave, physLane & ~groupMask);
        if (leave) break;
      }
      item = 0;
      if (!second) item = atomicAdd(nextItem, 1ull);
      if (pairLanes) item = (unsigned long long)__shfl((long long)item, physLane & ~groupMask);
      if ((long long)item >= nTodo) break;
    }
    int64_t q = todo ? todo[item] : (int64_t)item;
    ReadIn in;
    in.nMates = batch.mateCount[q];
    for (int m = 0; m < 2; m++) {
      in.mate[m] = batch.codes + batch.mateOffset[q * 2 + m];
      in.mateLen[m] = m < in.nMates ? batch.mateLength[q * 2 + m] : 0;
    }
    // single-end Query: expectedInnerDistance 0, deviation 1 (spacing penalty is always 0, T/SamWriter_Test.java:26)
    in.expectedInner = in.nMates > 1 ? batch.expectedInner[q] : 0.0;
    in.deviation = in.nMates > 1 ? batch.deviation[q] : 1.0;
    ReadResult rr;
#ifdef XM_READ_TIMES
    const unsigned long long readT0 = clock64();
#endif
    DevCounters before = local;
    if (ho.mode == 1) {
      uint8_t* region = ho.regions + (unsigned long long)myRegion * ho.regionBytes;
      runReadRetaining(cx, &ix, params, in, scale, region, (size_t)ho.regionBytes, arena, (size_t)arenaBytes, &local, rr, heavyAllowed);
      if (cx.status == XM_ST_NEED_HEAVY && savedReadOf(region, (size_t)ho.regionBytes)->valid) {
        // the read keeps this region; the lane needs a fresh one only if it will take another read (the list counter only grows, so a
        // lane that sees the list drained here finds it drained at its next fetch and leaves)
        const bool drained = (long long)__hip_atomic_load(nextItem, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nTodo;
        if (drained) {
          ho.regionOf[q] = (int32_t)myRegion;
        } else {
          long long fresh = (long long)atomicAdd(ho.cursor, 1ull);
          if (fresh < ho.nRegions) { ho.regionOf[q] = (int32_t)myRegion; myRegion = fresh; }  // (pool used up: the read is seeded again by the gapped pass)
        }
      }
    } else if (ho.mode == 2) {
      const int32_t rg = ho.regionOf[q];
      uint8_t* tmp = arena + ho.regionBytes;
      const size_t tmpBytes = (size_t)(arenaBytes - ho.regionBytes);
      if (rg >= 0) runReadResumed(cx, savedReadOf(ho.regions + (unsigned long long)rg * ho.regionBytes, (size_t)ho.regionBytes), &ix, scale, tmp, tmpBytes, &local, rr);
      else runReadRetaining(cx, &ix, params, in, ho.seedScale, arena, (size_t)ho.regionBytes, tmp, tmpBytes, &local, rr, 2, scale);
    } else {
      runRead(cx, &ix, params, in, scale, arena, (size_t)arenaBytes, &local, rr, heavyAllowed);
    }
    XM_PAIR_CHECK(0, cx.status);
    XM_PAIR_CHECK(1, ((long long)rr.nComponents << 40) ^ ((long long)rr.single[0] << 20) ^ (long long)rr.empty[0] ^ ((long long)local.pathAlignerNodes << 4));
    int32_t st = cx.status;
#ifdef XM_READ_TIMES
    if (xm_read_times && !second) xm_read_times[q] = clock64() - readT0;
#endif
    if (st != XM_OK) local = before;  // work of a read that is rerun by a later pass is counted there
    if (second) continue;             // (pair mode: the first lane of the read publishes)
    // (a read that found the result arena full is run again too: round 6 - its work used to be counted twice, PathAligner calls and nodes of the first call on a fresh context)
    if (publishRead(out, q, rr, cx, local, lists) == XM_ST_OUT_OVERFLOW) local = before;
  }
  if (!second) addCounters(counters, local);
}

// Test entry (xm_test_local_align): the reference's component-level known-answer tests (PathAligner_Test.java:10-39: PathAligner alone;
// HashBlockAligner_Test.java:10-48: HashBlock_Aligner -> StraightAligner -> PathAligner_Runner) over two given texts, run by the code the align
// kernel runs.  chain 0: one search, in the wave's LDS slot (mode 0), in HBM mode (mode 1) or in the lane-private form (mode 4); chain 1: hashBlockAlign with the searches
// slot-first as in the kernel (mode 0), all in HBM mode (mode 1) or all in the lane-private form of xm_wsearch.h (mode 4).  One lane works; out: found, nb, status, nodes, blocks; penalties.
__global__ void __launch_bounds__(256, XM_WAVES_PER_SIMD) xm_test_local_kernel(int chain, int mode, Params params, const uint8_t* query, int queryLength, const uint8_t* reference, int referenceLength,
                                                            double maxIns, double maxDel, int scale, uint8_t* arena, unsigned long long arenaBytes, PNode* waveNodes, int blockCap,
                                                            int32_t* outInts, double* outDbls) {
  xmSetWaveNodes(waveNodes);
  xmSetPairMode(0);
  xmSetSearchPool(SearchPool{nullptr, 0, 0, 0});
  xmSetBoundFilter(mode >= 8 ? 1 : 0);  // (mode + 8: the search behind the rejection filter of xm_bound.h)
  mode &= 7;
  xmLoadMergeRule();  // (every thread of the block: it ends with a barrier)
  if (threadIdx.x != 0) return;
  DevCounters local;
  memset(&local, 0, sizeof(local));
  Caps caps = makeCaps(scale);
  caps.searchInHbmOnly = mode == 1 ? 1 : (mode == 4 ? 2 : 0);
  Arena tmp;
  tmp.init(arena, (size_t)arenaBytes);
  int32_t status = XM_OK;
  float hint = 0;
  ExtEnv e;
  e.caps = &caps; e.dc = &local; e.status = &status; e.tmp = &tmp;
  e.query.base = query; e.query.len = queryLength; e.query.rc = 0; e.query.id = 0;
  e.reference.base = reference; e.reference.len = referenceLength; e.reference.rc = 0; e.reference.id = 0;
  e.contig = 0;
  e.heavyHint = &hint;
  Matcher* slots = arenaArray<Matcher>(tmp, 3);
  for (int i = 0; i < 3; i++) {
    slots[i].present = arenaArray<uint8_t>(tmp, caps.maxSections);
    slots[i].tables = arenaArray<int16_t>(tmp, caps.matcherEntries);
    slots[i].tableCap = caps.matcherEntries;
    slots[i].maxSections = caps.maxSections;
    slots[i].nSections = 0;
    slots[i].presentMask = 0;
    slots[i].sectionLength = 0;
  }
  e.slotA = &slots[0]; e.slotB = &slots[1]; e.slotT = &slots[2];
  SeqAl out;
  out.blocks = arenaArray<ABlock>(tmp, caps.maxBlocks);
  out.nb = 0; out.contig = 0; out.referenceReversed = 0; out.seqAId = 0; out.totalPenalty = 0; out.alignedPenalty = 0;
  bool found = false;
  if (tmp.overflow) status = XM_ST_OVERFLOW;
  else {
    const Section qs{0, queryLength}, rs{0, referenceLength};
    Analysis an;  // AlignmentAnalysis as the tests construct it: nothing known about the offset, the two extension limits given
    an.matcher = nullptr; an.predictedBestOffset = 0; an.lastCheckedOffset = 0; an.confidentAboutBestOffset = false;
    an.maxInsertionExtensionPenalty = maxIns; an.maxDeletionExtensionPenalty = maxDel;
    if (chain == 0) found = pathAlign(e, qs, rs, params, an, out);
    else found = hashBlockAlign(e, qs, rs, params, an, out, e.slotB, NextStraight3());
  }
  outInts[0] = found && status == XM_OK ? 1 : 0; outInts[1] = found ? out.nb : 0; outInts[2] = status; outInts[3] = (int32_t)local.pathAlignerNodes;
  outInts[4 + 4 * blockCap] = (int32_t)local.boundChecks; outInts[5 + 4 * blockCap] = (int32_t)local.boundRejects; outInts[6 + 4 * blockCap] = (int32_t)local.boundCells;  // (behind the blocks)
  if (found) {
    for (int i = 0; i < out.nb && i < blockCap; i++) { outInts[4 + 4 * i] = out.blocks[i].startA; outInts[5 + 4 * i] = out.blocks[i].startB; outInts[6 + 4 * i] = out.blocks[i].lenA; outInts[7 + 4 * i] = out.blocks[i].lenB; }
    outDbls[0] = out.totalPenalty; outDbls[1] = out.alignedPenalty;
  }
}

// Test entry (xm_test_bound): the rejection filter of xm_bound.h alone, on one problem - a section of a query against a window of a reference - as a lane of a
// long-read chain runs it (lane 0 of a wave, its region of the wave's slot; pair = 1: lanes 0 and 1 together, 3: lanes 0 .. 7).  out: taken, rejected, cells.
__global__ void __launch_bounds__(256, XM_WAVES_PER_SIMD) xm_test_bound_kernel(Params params, const uint8_t* query, int queryLength, int queryRc, int startA, int endA, const uint8_t* reference, int referenceLength,
                                                            int startB, int endB, int predictedBestOffset, int pair, uint8_t* arena, unsigned long long arenaBytes, int64_t* out) {
  xmSetWaveNodes(nullptr);
  xmSetPairMode(pair);
  xmSetSearchPool(SearchPool{nullptr, 0, 0, 0});
  xmSetBoundFilter(3);
  xmLoadMergeRule();  // (every thread of the block: it ends with a barrier)
  if (threadIdx.x >= (1u << pair)) return;
  BoundProblem bp;
  bp.qBase = query; bp.qLen = queryLength; bp.qRc = queryRc != 0; bp.rBase = reference; bp.referenceLen = referenceLength;
  bp.startA = startA; bp.endA = endA; bp.startB = startB; bp.endB = endB; bp.predictedBestOffset = predictedBestOffset;
  bp.mutation = params.MutationPenalty; bp.insStart = params.InsertionStart_Penalty; bp.insExt = params.InsertionExtension_Penalty; bp.delStart = params.DeletionStart_Penalty;
  bp.delExt = params.DeletionExtension_Penalty; bp.maxErrorRate = params.MaxErrorRate; bp.ambiguity = params.AmbiguityPenalty;
  bp.budget = (endA - startA) * params.MaxErrorRate; bp.piece = 0;
  bool taken = false;
  unsigned long long cells = 0;
  Arena tmp;
  tmp.init(arena, (size_t)arenaBytes);  // (the two lanes of a pair keep the same band in the same memory, as they do in the passes: same values twice)
  const bool rejected = boundRejects(bp, pair, tmp, taken, cells);
  if (threadIdx.x == 0) { out[0] = taken ? 1 : 0; out[1] = rejected ? 1 : 0; out[2] = (int64_t)cells; }
}

// ---------------------------------------------------------------- pile-up of the alignments on the reference
// What MatchDatabase.addAlignments / groupByPosition feed the mutation and VCF writers with (M/Mapper.java:700-708,758-785; the classes are
// un-vendored, the behaviour is pinned by T/MatchDatabase_Test.java and T/MutationsWriter_Test.java): per forward reference position the depth
// and the counts of differing query bases, plus one event per insertion / deletion block.  One lane per query walks its result stream in HBM.
// Counts are integers in units of 1 / XM_PILEUP_UNIT of a read base (a query with n alignments adds 1/n per alignment, the two mates of a pair
// add 1/2 each where they overlap: T/MatchDatabase_Test.java:38-69), so sums do not depend on the order of the atomic adds.
struct PileupView {
  unsigned long long* depth;     // [totalForwardSize]
  unsigned long long* alt;       // [4][totalForwardSize]: query base A, C, G, T where it differs from an unambiguous reference base
  long long total;
  long long* events;             // 7 per event: contig, position, type (1 insertion, 2 deletion), length, query, mate | reversed << 1, startA; weight in [7]
  unsigned long long eventCap;
  unsigned long long* eventCount;
  long long queryBase;           // index of the batch's first query among all queries added so far
  unsigned long long* mid;       // [totalForwardSize] depth from query bases that are not near a query end (null: no query-end fraction set)
  double endFraction;            // MatchDatabase(queryEndFraction), --distinguish-query-ends (Mapper.java:76,351-353,700)
};
// a query base "near the end of the query": within endFraction of the query's length of either end  [inferred: the rule lives in the un-vendored
// MatchDatabase; pinned by MutationsWriter_Test.java:114-134 only for fraction 0.5 = every base]
__device__ __forceinline__ bool xmNearQueryEnd(int k, int readLen, double f) { return (double)k < f * readLen || (double)k >= readLen - f * readLen; }
__global__ void __launch_bounds__(256) xm_pileup_kernel(IndexView ix, BatchView batch, const int32_t* ints, const int64_t* intOff, PileupView pv) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= batch.nq) return;
  const int32_t* p = ints + intOff[q];
  const int numComponents = *p++;
  for (int c = 0; c < numComponents; c++) {
    const int numAlignments = *p++;
    if (numAlignments < 1) continue;
    for (int a = 0; a < numAlignments; a++) {
      // a query's alignments share one read's worth of weight exactly: UNIT / n each, the first UNIT mod n of them one unit more (n above 16 need not
      // divide UNIT; the sums over a query then still come out whole)
      const unsigned long long w = XM_PILEUP_UNIT / (unsigned long long)numAlignments + ((unsigned long long)a < XM_PILEUP_UNIT % (unsigned long long)numAlignments ? 1ull : 0ull);
      p++;  // innerDistance
      const int numSequences = *p++;
      // the reference interval of each sequence alignment first (mates of a pair share the depth where they overlap)
      int contigOf[2] = {-1, -1};
      long long lo[2] = {0, 0}, hi[2] = {0, 0};
      {
        const int32_t* t = p;
        for (int sq = 0; sq < numSequences && sq < 2; sq++) {
          contigOf[sq] = t[0];
          const int nb = t[2];
          t += 3;
          if (nb > 0) { lo[sq] = t[1]; hi[sq] = t[4 * (nb - 1) + 1] + t[4 * (nb - 1) + 3]; }
          t += 4 * nb;
        }
      }
      long long ovLo = 0, ovHi = 0;
      if (numSequences == 2 && contigOf[0] == contigOf[1]) { ovLo = lo[0] > lo[1] ? lo[0] : lo[1]; ovHi = hi[0] < hi[1] ? hi[0] : hi[1]; }
      for (int sq = 0; sq < numSequences; sq++) {
        const int contig = *p++;
        const int reversed = *p++;
        const int nb = *p++;
        const int mate = numComponents > 1 ? c : sq;
        const uint8_t* read = batch.codes + batch.mateOffset[q * 2 + mate];
        const int readLen = batch.mateLength[q * 2 + mate];
        const long long base = ix.contigStart[contig];
        for (int b = 0; b < nb; b++, p += 4) {
          const int startA = p[0], startB = p[1], lenA = p[2], lenB = p[3];
          if (lenA == lenB) {
            for (int i = 0; i < lenA; i++) {
              const long long pos = startB + i;
              const unsigned long long wi = (pos >= ovLo && pos < ovHi) ? (sq == 0 ? w / 2 : w - w / 2) : w;
              const uint8_t r = ix.refCodes[base + pos];
              const int k = startA + i;
              if (pv.mid && !xmNearQueryEnd(k, readLen, pv.endFraction)) atomicAdd(&pv.mid[base + pos], wi);
              const uint8_t qb = reversed ? bpComplement(read[readLen - 1 - k]) : read[k];
              atomicAdd(&pv.depth[base + pos], wi);
              if (!bpIsAmbiguous(r) && !bpIsAmbiguous(qb) && qb != r) atomicAdd(&pv.alt[(long long)encodedCharToInt(qb) * pv.total + base + pos], wi);
            }
          } else {
            if (lenA == 0) for (int i = 0; i < lenB; i++) {  // a deletion: the read spans these reference bases
              const long long pos = startB + i;
              const unsigned long long wi = (pos >= ovLo && pos < ovHi) ? (sq == 0 ? w / 2 : w - w / 2) : w;
              atomicAdd(&pv.depth[base + pos], wi);
              if (pv.mid && !xmNearQueryEnd(startA, readLen, pv.endFraction)) atomicAdd(&pv.mid[base + pos], wi);  // (the gap sits in front of query base startA)
            }
            const unsigned long long at = atomicAdd(pv.eventCount, 1ull);
            if (at < pv.eventCap) {
              long long* e = pv.events + at * 8;
              e[0] = contig; e[1] = startB; e[2] = lenA > 0 ? 1 : 2; e[3] = lenA > 0 ? lenA : lenB; e[4] = pv.queryBase + q; e[5] = mate | (reversed << 1) | ((pv.mid && xmNearQueryEnd(startA, readLen, pv.endFraction)) ? 4 : 0); e[6] = startA;
              e[7] = (long long)((startB >= ovLo && startB < ovHi) ? (sq == 0 ? w / 2 : w - w / 2) : w);
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- pass bookkeeping on the device (PassCtl, PassLists: xm_kernel_common.h)
// after a pass of the wave-per-read form (xm_wave_kernel.hip): reads for the next tier, reads with a waiting search request, reads left
// to the lane-per-read passes
struct WaveCtl { unsigned long long nNext, nSearch, nFallback, errQuery; };
__global__ void __launch_bounds__(256) xm_wave_classify_kernel(const int64_t* todo, long long nTodo, const int32_t* status, int64_t* listNext, int32_t* slotOfOut, int64_t* listSearch,
                                                               int64_t* listFallback, WaveCtl* ctl) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nTodo) return;
  const int64_t q = todo ? todo[i] : (int64_t)i;
  const int32_t st = status[q] & 0xFF;
  if (st == XM_OK) return;
  if (st == 9 /* XM_ST_WAVE_GAPPED */ && listNext) {
    const unsigned long long pos = atomicAdd(&ctl->nNext, 1ull);
    listNext[pos] = q;
    if (slotOfOut) slotOfOut[q] = (int32_t)pos;  // the read's memo (it keeps it through the later tiers)
  } else if (st == 10 /* XM_ST_WAVE_SEARCH */ && listSearch) listSearch[atomicAdd(&ctl->nSearch, 1ull)] = q;
  else if (st == 8 /* XM_ST_WAVE_FALLBACK */ || st == 9 || st == 10) listFallback[atomicAdd(&ctl->nFallback, 1ull)] = q;
  else atomicMin(&ctl->errQuery, (unsigned long long)q);
}

// Exclusive prefix sums of the per-query stream lengths (query order), three small kernels: block totals, scan of the totals,
// final offsets.  4096 queries per block.
constexpr int XM_SCAN_PER_THREAD = 16;
constexpr int XM_SCAN_PER_BLOCK = 256 * XM_SCAN_PER_THREAD;

__device__ __forceinline__ void blockReduce2(long long& a, long long& b, long long* shA, long long* shB) {
  shA[threadIdx.x] = a; shB[threadIdx.x] = b;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) { shA[threadIdx.x] += shA[threadIdx.x + d]; shB[threadIdx.x] += shB[threadIdx.x + d]; }
    __syncthreads();
  }
  a = shA[0]; b = shB[0];
}

__global__ void __launch_bounds__(256) xm_scan_totals_kernel(long long nq, const int32_t* intLen, const int32_t* dblLen, long long* blockI, long long* blockD) {
  __shared__ long long shA[256], shB[256];
  long long base = (long long)blockIdx.x * XM_SCAN_PER_BLOCK + (long long)threadIdx.x * XM_SCAN_PER_THREAD;
  long long a = 0, b = 0;
  for (int k = 0; k < XM_SCAN_PER_THREAD; k++) if (base + k < nq) { a += intLen[base + k]; b += dblLen[base + k]; }
  blockReduce2(a, b, shA, shB);
  if (threadIdx.x == 0) { blockI[blockIdx.x] = a; blockD[blockIdx.x] = b; }
}

__global__ void xm_scan_blocks_kernel(long long nBlocks, long long nq, long long* blockI, long long* blockD, int64_t* finalIntOff, int64_t* finalDblOff) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long long a = 0, b = 0;
  for (long long i = 0; i < nBlocks; i++) { long long x = blockI[i], y = blockD[i]; blockI[i] = a; blockD[i] = b; a += x; b += y; }
  finalIntOff[nq] = a; finalDblOff[nq] = b;
}

__global__ void __launch_bounds__(256) xm_scan_final_kernel(long long nq, const int32_t* intLen, const int32_t* dblLen, const long long* blockI, const long long* blockD,
                                                            int64_t* finalIntOff, int64_t* finalDblOff) {
  __shared__ long long shA[256], shB[256];
  long long base = (long long)blockIdx.x * XM_SCAN_PER_BLOCK + (long long)threadIdx.x * XM_SCAN_PER_THREAD;
  long long a = 0, b = 0;
  for (int k = 0; k < XM_SCAN_PER_THREAD; k++) if (base + k < nq) { a += intLen[base + k]; b += dblLen[base + k]; }
  shA[threadIdx.x] = a; shB[threadIdx.x] = b;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {  // inclusive Hillis-Steele scan of the 256 thread totals
    long long x = 0, y = 0;
    if ((int)threadIdx.x >= d) { x = shA[threadIdx.x - d]; y = shB[threadIdx.x - d]; }
    __syncthreads();
    shA[threadIdx.x] += x; shB[threadIdx.x] += y;
    __syncthreads();
  }
  long long offA = blockI[blockIdx.x] + shA[threadIdx.x] - a, offB = blockD[blockIdx.x] + shB[threadIdx.x] - b;
  for (int k = 0; k < XM_SCAN_PER_THREAD; k++) if (base + k < nq) {
    finalIntOff[base + k] = offA; finalDblOff[base + k] = offB;
    offA += intLen[base + k]; offB += dblLen[base + k];
  }
}

// canonical streams: every query's slice copied from where its pass left it to its place in query order
__global__ void __launch_bounds__(256) xm_gather_kernel(long long nq, const int64_t* srcIntOff, const int64_t* srcDblOff, const int32_t* intLen, const int32_t* dblLen,
                                                        const int64_t* finalIntOff, const int64_t* finalDblOff, const int32_t* srcInts, const double* srcDbls,
                                                        int32_t* dstInts, double* dstDbls) {
  long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const int32_t* si = srcInts + srcIntOff[q]; int32_t* di = dstInts + finalIntOff[q];
  const double* sd = srcDbls + srcDblOff[q]; double* dd = dstDbls + finalDblOff[q];
  int ni = intLen[q], nd = dblLen[q];
  for (int k = 0; k < ni; k++) di[k] = si[k];
  for (int k = 0; k < nd; k++) dd[k] = sd[k];
}

// Bucket lines of one table from its CSR form (IndexView::lines32 / lines64): one lane per bucket.
template <typename W, typename P>
__global__ void __launch_bounds__(256) xm_lines_kernel(Table t, const uint32_t* bucketOff, const P* positions, W* lines) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= t.capacity) return;
  const uint32_t* off = bucketOff + t.offBase + k;
  W line[8];
  xmFillLine(line, off[0], off[1], positions + t.posBase);
  W* dst = lines + (t.offBase + k) * 8;
  for (int j = 0; j < 8; j++) dst[j] = line[j];
}

// Where a probe's positions go: the 64 probes i = 64 c ... 64 c + 63 (the lanes of one wavefront) write theirs one behind the other, in probe order, from
// out_positions[64 c max_per_probe] on - whole lines leave for HBM, and only as many bytes as there are positions (rows of max_per_probe slots cost the
// memory 56 bytes a probe at 7 slots, of which a genome's buckets fill 12; a row per position index, written only where a bucket has that many, is holes
// in every line, and a partly written line costs a read besides the write).  The reader finds probe i's positions behind those of the probes before it
// in its chunk: offsets are the running sum of min(max(counts, 0), max_per_probe) over the chunk.  Returns this lane's first slot.
__device__ __forceinline__ long long xmProbeSlot(long long i, int m, int maxPerProbe) {
  // exclusive prefix sum of m (0 ... 7: three bits) over the lanes of the wave
  const unsigned long long b0 = __ballot(m & 1), b1 = __ballot(m & 2), b2 = __ballot(m & 4), b3 = __ballot(m & 8);
  const unsigned long long below = (1ull << (threadIdx.x & 63u)) - 1ull;
  const int before = __popcll(b0 & below) + 2 * __popcll(b1 & below) + 4 * __popcll(b2 & below) + 8 * __popcll(b3 & below);
  return (i & ~63ll) * (long long)maxPerProbe + before;
}

// PackedMap.getNumMatchesLowerBound + PackedMap.get for a batch of (used length, key): one lane per probe.
__global__ void __launch_bounds__(256) xm_seed_probe_kernel(IndexView ix, long long n, const int32_t* usedLength, const int32_t* keys, int maxPerProbe,
                                                            int32_t* counts, int64_t* outPositions) {
  // (the probe through the CSR arrays - two adjacent offsets, then the positions: what an index without bucket lines offers)
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  const int used = live ? usedLength[i] : -1;
  const bool ok = live && used >= 0 && used <= ix.maxHashedLength;
  int count = -2;
  int64_t first = 0;
  if (ok) {
    const Table* t = &ix.tables[used];
    const uint32_t k = packedKey(t, keys[i]);
    const uint32_t* off = ix.bucketOff + t->offBase + k;
    const uint32_t o0 = off[0], o1 = off[1];
    count = (o0 & XM_OVERFULL) ? -1 : (int)((o1 & ~XM_OVERFULL) - (o0 & ~XM_OVERFULL));
    if (count > t->maxCount) count = -1;
    first = t->posBase + (int64_t)(o0 & ~XM_OVERFULL);
  }
  const int m = (count > 0 && maxPerProbe > 0) ? (count < maxPerProbe ? count : maxPerProbe) : 0;
  const long long slot = xmProbeSlot(i, m, maxPerProbe);
  if (!live) return;
  counts[i] = count;
  for (int j = 0; j < m; j++) outPositions[slot + j] = ix.posIs64 ? (int64_t)ix.positions64[first + j] : (int64_t)ix.positions32[first + j];
}

// The same bulk probe over bucket lines with several probes in flight per lane (round 5; it replaces the group-of-lanes form of round 2, which ran at half of this
// GPU's random-sector rate): a probe's chain - (length, key) -> table descriptor -> key mod capacity -> line - is short but dependent, so what the rate
// needs is many chains at a time.  A lane takes XM_PROBES_PER_LANE probes a whole launch apart (coalesced reads of the inputs and writes of the counts), takes
// the table descriptors from LDS (the block copies them there once: no trip to memory between the inputs and the line), and has asked for all its lines
// before it looks at the first.  64-bit lines: the whole 64-byte line as four 16-byte loads; 32-bit lines: two.  Header only (maxPerProbe == 0): the first
// 16 bytes.  Buckets with more than XM_LINE_SLOTS positions (1.4 % of a genome-like index) read the CSR arrays behind that.
constexpr int XM_PROBES_PER_LANE = 4;
constexpr int XM_PROBE_LDS_TABLES = 512;
template <bool W64>
__global__ void __launch_bounds__(256) xm_seed_probe_lines_kernel(IndexView ix, long long n, const int32_t* usedLength, const int32_t* keys, int maxPerProbe,
                                                                  int32_t* counts, int64_t* outPositions) {
  __shared__ Table sTables[XM_PROBE_LDS_TABLES];
  const int nTables = ix.maxHashedLength + 1;
  const bool inLds = nTables <= XM_PROBE_LDS_TABLES;
  if (inLds) {
    for (int t = (int)threadIdx.x; t < nTables; t += (int)blockDim.x) sTables[t] = ix.tables[t];
    __syncthreads();
  }
  const long long lanes = (long long)gridDim.x * blockDim.x;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int used[XM_PROBES_PER_LANE], key[XM_PROBES_PER_LANE];
#pragma unroll
  for (int p = 0; p < XM_PROBES_PER_LANE; p++) {
    const long long i = tid + (long long)p * lanes;
    used[p] = i < n ? usedLength[i] : -1;
    key[p] = i < n ? keys[i] : 0;
  }
  Table tb[XM_PROBES_PER_LANE];
  uint32_t k[XM_PROBES_PER_LANE];
  typedef typename std::conditional<W64, ulonglong2, uint4>::type Vec;   // 16 bytes of a line
  constexpr int NV = W64 ? 4 : 2;
  Vec v[XM_PROBES_PER_LANE][NV];
#pragma unroll
  for (int p = 0; p < XM_PROBES_PER_LANE; p++) {
    const bool ok = used[p] >= 0 && used[p] <= ix.maxHashedLength;
    tb[p] = inLds ? sTables[ok ? used[p] : 0] : ix.tables[ok ? used[p] : 0];
    k[p] = packedKey(&tb[p], key[p]);
    const Vec* lp = W64 ? (const Vec*)(ix.lines64 + (tb[p].offBase + k[p]) * 8) : (const Vec*)(ix.lines32 + (tb[p].offBase + k[p]) * 8);
    if (ok) v[p][0] = lp[0];
  }
  // the rest of a line only where its positions are wanted: the first 16 bytes hold the count and three positions (one with 64-bit positions), and a
  // request costs the memory pipeline the same whether it brings 16 bytes of a new sector or the next 16 of the one before
  int cnt[XM_PROBES_PER_LANE];
#pragma unroll
  for (int p = 0; p < XM_PROBES_PER_LANE; p++) {
    const bool ok = used[p] >= 0 && used[p] <= ix.maxHashedLength;
    const uint32_t h = W64 ? (uint32_t)((const unsigned long long*)&v[p][0])[0] : ((const uint32_t*)&v[p][0])[0];
    int count = (h & XM_OVERFULL) ? -1 : (int)h;
    if (ok && count > tb[p].maxCount) count = -1;
    cnt[p] = ok ? count : -2;
    const int want = (maxPerProbe > 0 && count > 0 && count <= XM_LINE_SLOTS) ? (count < maxPerProbe ? count : maxPerProbe) : 0;   // positions to take from the line
    const Vec* lp = W64 ? (const Vec*)(ix.lines64 + (tb[p].offBase + k[p]) * 8) : (const Vec*)(ix.lines32 + (tb[p].offBase + k[p]) * 8);
#pragma unroll
    for (int q = 1; q < NV; q++) if (ok && 1 + want > q * (W64 ? 2 : 4)) v[p][q] = lp[q];
  }
#pragma unroll
  for (int p = 0; p < XM_PROBES_PER_LANE; p++) {
    const long long i = tid + (long long)p * lanes;
    const int count = cnt[p];
    const int m = (i < n && count > 0 && maxPerProbe > 0) ? (count < maxPerProbe ? count : maxPerProbe) : 0;
    const long long slot = xmProbeSlot(i, m, maxPerProbe);  // (every lane of the wave: the lanes' probes of one p are 64 consecutive ones)
    if (i >= n) continue;
    counts[i] = count;
    if (m == 0) continue;
    if (count <= XM_LINE_SLOTS) {
#pragma unroll
      for (int j = 0; j < XM_LINE_SLOTS; j++) {
        if (j < m) outPositions[slot + j] = W64 ? (int64_t)((const unsigned long long*)&v[p][0])[1 + j] : (int64_t)((const uint32_t*)&v[p][0])[1 + j];
      }
      continue;
    }
    const int64_t first = tb[p].posBase + (int64_t)(ix.bucketOff[tb[p].offBase + k[p]] & ~XM_OVERFULL);
    for (int j = 0; j < m; j++) outPositions[slot + j] = ix.posIs64 ? (int64_t)ix.positions64[first + j] : (int64_t)ix.positions32[first + j];
  }
}

// Measurement helper: one random 64-byte sector per access out of a table far larger than the caches, 16 bytes of it read.
// The sectors/s this reaches is the ceiling a hash-probe kernel (one 8-byte bucket header per probe) can be held against.
__global__ void __launch_bounds__(256) xm_random_gather_kernel(const uint4* table, unsigned long long nSectors, long long nAccesses, int perThread, unsigned long long seed,
                                                               unsigned int* sink) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int k = 0; k < perThread; k++) {
    long long a = t * perThread + k;
    if (a >= nAccesses) break;
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(a + 1);  // SplitMix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    uint4 v = table[(z % nSectors) * 4];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;  // keeps the loads alive
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  void ensure(size_t count) {
    if (count <= n && p) return;
    if (p) (void)hipFree(p);
    p = nullptr;
    n = count ? count : 1;
    HIP_CHECK(hipMalloc((void**)&p, n * sizeof(T)));
  }
  // like ensure, but an allocation the GPU has no room for returns false (the buffer is then empty) instead of throwing
  bool tryEnsure(size_t count) {
    if (count <= n && p) return true;
    if (p) (void)hipFree(p);
    p = nullptr;
    const size_t want = count ? count : 1;
    n = 0;
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) { (void)hipGetLastError(); p = nullptr; return false; }
    HIP_CHECK(e);
    n = want;
    return true;
  }
  // grow to `count`, keeping the first `keep` elements
  void growKeep(size_t count, size_t keep, hipStream_t s) {
    if (count <= n && p) return;
    T* np = nullptr;
    HIP_CHECK(hipMalloc((void**)&np, count * sizeof(T)));
    if (p && keep) HIP_CHECK(hipMemcpyAsync(np, p, (keep < n ? keep : n) * sizeof(T), hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (p) (void)hipFree(p);
    p = np; n = count;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  void swapWith(DevBuf& o) { std::swap(p, o.p); std::swap(n, o.n); }
  ~DevBuf() { release(); }
};

// Result streams live in pinned host memory (the final device-to-host copy is then one DMA per stream); the buffers are recycled
// through a process-wide pool because pinning is far more expensive than the copy itself.
struct PinnedPool {
  struct Buf { void* p; size_t bytes; };
  std::mutex mu;
  std::vector<Buf> idle;
  size_t idleBytes = 0;
  std::atomic<size_t> allocatedBytes{0}, highWater{0};  // pinned host memory this process holds through the pool (in use + idle), and the most it ever held
  void account(long long delta) {
    const size_t now = (size_t)((long long)allocatedBytes.fetch_add((size_t)delta) + delta);
    size_t hw = highWater.load();
    while (now > hw && !highWater.compare_exchange_weak(hw, now)) {}
  }
  void* get(size_t bytes, size_t* got) {
    if (bytes < 64) bytes = 64;
    {
      std::lock_guard<std::mutex> lock(mu);
      int best = -1;
      for (int i = 0; i < (int)idle.size(); i++)
        if (idle[i].bytes >= bytes && idle[i].bytes <= bytes * 2 + 4096 && (best < 0 || idle[i].bytes < idle[best].bytes)) best = i;
      if (best >= 0) {
        Buf b = idle[best];
        idle.erase(idle.begin() + best);
        idleBytes -= b.bytes;
        *got = b.bytes;
        return b.p;
      }
    }
    void* p = nullptr;
    size_t want = bytes + bytes / 8;  // headroom so that the next, slightly larger batch reuses it
    HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocPortable));
    account((long long)want);
    *got = want;
    return p;
  }
  void put(void* p, size_t bytes) {
    if (!p) return;
    std::vector<Buf> drop;
    {
      std::lock_guard<std::mutex> lock(mu);
      idle.push_back(Buf{p, bytes});
      idleBytes += bytes;
      while (idleBytes > (4ull << 30) && !idle.empty()) {  // oldest first
        drop.push_back(idle.front());
        idleBytes -= idle.front().bytes;
        idle.erase(idle.begin());
      }
    }
    for (auto& b : drop) { (void)hipHostFree(b.p); account(-(long long)b.bytes); }
  }
};
static PinnedPool* g_pinned = new PinnedPool();  // never destroyed: the HIP runtime may be gone before static destructors run

// xm_result plus what xm_result_free needs to know about its buffers
struct ResultBox {
  xm_result pub;
  size_t bytesInts, bytesDbls, bytesIntOff, bytesDblOff;
};

}  // namespace

// The tables of one reference, hashed once and shared by every context and replica of the index: the reference shares one HashBlock_Database
// between all AlignerWorkers of a run through per-thread views (HashBlock_Database.java:129-133, Mapper.java:1026-1040, Api.java:78).
struct HostShare {
  HostIndex host;
  std::mutex mu;                     // growth of the host tables (Readable_HashBlock_Database.getContainingMap, :108-113), save
  std::atomic<int> hashedLength{0};  // copy of host.maxHashedLength readable without mu
};

// The same tables in one GPU's HBM: one per (index, device), shared by every context of the index on that GPU.
struct DeviceTables {
  std::shared_ptr<HostShare> hs;
  int device = 0;
  std::shared_mutex rw;      // align / probe calls of any number of contexts hold it shared while their kernels read the tables; (re)upload holds it exclusive
  std::mutex allocMu;        // contexts of one GPU size and allocate their scratch one after the other (they all look at the same free memory)
  std::atomic<int> uploadedLength{-1};   // host.maxHashedLength the device tables hold (written under hs->mu + rw, read without them by ensureTablesFor's first test)
  std::atomic<int> contexts{0};  // handles that share these tables (contexts of this GPU): a context sizes its launches for its share of the wave slots
  DevBuf<int64_t> dContigStart, dSeqCumStart, dDupKeyStart;
  DevBuf<int32_t> dContigLen, dDupKeys;
  DevBuf<uint8_t> dRefCodes;
  DevBuf<Table> dTables;
  DevBuf<uint32_t> dBucketOff, dPositions32;
  DevBuf<uint64_t> dPositions64;
  DevBuf<int32_t> dBaLogStep;
  DevBuf<uint32_t> dLines32;  // bucket lines (IndexView::lines32 / lines64), built on the device from the CSR tables by upload()
  DevBuf<uint64_t> dLines64;
  bool posIs64 = false;
  IndexView view;
  int numCUs = 0;
  hipStream_t stream = nullptr;

  // Host tables -> HBM (caller holds hs->mu and rw exclusively).  With `peer` (a replica on another GPU): the tables are copied from the peer's HBM
  // instead (hipMemcpyPeer: over xGMI), not sent over PCIe a second time.
  void upload(const DeviceTables* peer = nullptr) {
    const HostIndex& host = hs->host;
    HIP_CHECK(hipSetDevice(device));
    if (!stream) HIP_CHECK(hipStreamCreate(&stream));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    numCUs = prop.multiProcessorCount;
    auto up = [&](auto& buf, const auto& vec, const auto& peerBuf) {
      buf.ensure(vec.size());
      if (vec.empty()) return;
      if (peer) HIP_CHECK(hipMemcpyPeer(buf.p, device, peerBuf.p, peer->device, vec.size() * sizeof(vec[0])));
      else HIP_CHECK(hipMemcpy(buf.p, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice));
    };
    const DeviceTables& src = peer ? *peer : *this;
    up(dContigStart, host.contigStart, src.dContigStart); up(dContigLen, host.contigLen, src.dContigLen); up(dSeqCumStart, host.seqCumStart, src.dSeqCumStart);
    up(dRefCodes, host.refCodes, src.dRefCodes); up(dTables, host.tables, src.dTables); up(dBucketOff, host.bucketOff, src.dBucketOff);
    up(dDupKeyStart, host.dupKeyStart, src.dDupKeyStart); up(dDupKeys, host.dupKeys, src.dDupKeys);
    // (XM_FORCE_POS64=1: test hook, the 64-bit position arrays of references beyond 2^32 encoded positions on a small reference)
    posIs64 = peer ? peer->posIs64 : (host.seqCumStart.back() > 0xFFFFFFFFll || envInt("XM_FORCE_POS64", 0) != 0);
    if (posIs64) {
      up(dPositions64, host.positions, src.dPositions64);
    } else if (peer) {
      dPositions32.ensure(host.positions.size());
      if (!host.positions.empty()) HIP_CHECK(hipMemcpyPeer(dPositions32.p, device, peer->dPositions32.p, peer->device, host.positions.size() * sizeof(uint32_t)));
    } else {
      std::vector<uint32_t> p32(host.positions.size());
      for (size_t i = 0; i < p32.size(); i++) p32[i] = (uint32_t)host.positions[i];
      up(dPositions32, p32, dPositions32);
    }
    // bucket lines: 32 bytes (64 with 64-bit positions) per bucket, when they fit beside the index (XM_INDEX_LINES=0: CSR probes only)
    view.lines32 = nullptr; view.lines64 = nullptr;
    dLines32.release(); dLines64.release();
    if (envInt("XM_INDEX_LINES", 1) != 0 && !host.bucketOff.empty()) {
      const size_t words = host.bucketOff.size() * 8;
      size_t freeB = 0, totalB = 0;
      const size_t need = words * (posIs64 ? 8 : 4);
      if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && need < freeB / 2) {
        if (posIs64) dLines64.ensure(words); else dLines32.ensure(words);
        for (const Table& t : host.tables) {
          if (t.capacity < 1) continue;
          const unsigned grid = (unsigned)(((long long)t.capacity + 255) / 256);
          if (posIs64) hipLaunchKernelGGL((xm_lines_kernel<uint64_t, uint64_t>), dim3(grid), dim3(256), 0, stream, t, (const uint32_t*)dBucketOff.p, (const uint64_t*)dPositions64.p, dLines64.p);
          else hipLaunchKernelGGL((xm_lines_kernel<uint32_t, uint32_t>), dim3(grid), dim3(256), 0, stream, t, (const uint32_t*)dBucketOff.p, (const uint32_t*)dPositions32.p, dLines32.p);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(stream));
        view.lines32 = dLines32.p; view.lines64 = dLines64.p;
      }
    }
    view.numContigs = host.numContigs(); view.minInterestingSize = host.minInterestingSize; view.maxHashedLength = host.maxHashedLength;
    view.enableGapmers = host.enableGapmers; view.posIs64 = posIs64 ? 1 : 0; view.dupWindow = host.dupWindow; view.dupGranularity = host.dupGranularity();
    view.totalForwardAndReverseSize = host.totalForwardSize * 2;
    view.contigStart = dContigStart.p; view.contigLen = dContigLen.p; view.seqCumStart = dSeqCumStart.p; view.refCodes = dRefCodes.p;
    view.tables = dTables.p; view.bucketOff = dBucketOff.p; view.positions32 = dPositions32.p; view.positions64 = dPositions64.p;
    view.dupKeyStart = dDupKeyStart.p; view.dupKeys = dDupKeys.p;
    view.conf = nullptr; view.confMask = 0; view.confMiss = nullptr;  // (per call: alignResidentLocked)
    {
      int32_t steps[24];
      blockAlignerLogSteps(steps, 24);
      dBaLogStep.ensure(24);
      HIP_CHECK(hipMemcpy(dBaLogStep.p, steps, sizeof(steps), hipMemcpyHostToDevice));
      view.baLogStep = dBaLogStep.p;
    }
    uploadedLength = host.maxHashedLength;
    hs->hashedLength.store(host.maxHashedLength);
  }
  ~DeviceTables() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

// An xm_index handle is a CONTEXT of an index: what one host thread needs to align batches on one GPU - a stream, batch buffers, scratch and a
// result pool of its own - over tables it shares with every other context of the same index (HostShare: all of them; DeviceTables: those on
// its GPU).  xm_index_build / xm_index_load make the first context; xm_context_new and xm_index_replicate add contexts.
struct xm_index {
  std::shared_ptr<HostShare> hs;
  std::shared_ptr<DeviceTables> dt;  // null with host_only
  bool hostOnly = false;
  int device = 0;
  long long scratchBytes = 0;        // xm_context_set_scratch: upper limit of this context's scratch (0: XM_SCRATCH_GIB / the default)
  std::mutex mu;                     // calls on one context serialise; contexts run side by side
  HostIndex& host() { return hs->host; }
  const HostIndex& host() const { return hs->host; }
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // per-call scratch kept across calls
  DevBuf<uint8_t> dArenas, dCodes;
  DevBuf<int32_t> dMateCount, dMateLength, dStatus, dIntLen, dDblLen, dOutInts;
  DevBuf<int64_t> dMateOffset, dIntOff, dDblOff, dTodo;
  DevBuf<double> dExpected, dDeviation, dOutDbls;
  DevBuf<unsigned long long> dCursors;  // [0],[1] result cursors, [2] next item
  DevBuf<int64_t> dListHeavy, dListHeavyLate, dListScale[2], dListOut[2], dFinalIntOff, dFinalDblOff;
  DevBuf<int32_t> dRegionOf;
  DevBuf<PNode> dWaveNodes;  // per wave: node payloads of its LDS-mode search
  DevBuf<uint8_t> dSearchPool;  // one buffer per wave for the arrays of HBM-mode searches (SearchPool, xm_extend.h)
  // wave-per-read passes
  DevBuf<int64_t> dListWaveHeavy, dListWaveNext, dListWaveSearch[2], dListFallback;
  DevBuf<uint8_t> dWaveMemo;
  DevBuf<int32_t> dWaveSlotOf;
  DevBuf<WaveCtl> dWaveCtl;
  DevBuf<uint8_t> dWaveArenas;
  DevBuf<PNode> dWaveNodes2;
  bool residentAnyPaired = false, stagedAnyPaired = false;
  DevBuf<PassCtl> dCtl;
  DevBuf<long long> dBlockI, dBlockD;
  DevBuf<int32_t> dFinalInts;
  DevBuf<double> dFinalDbls;
  DevBuf<DevCounters> dCounters;
  // confidence table (IndexView::conf): host copy, device copy, the settings it was computed for, miss list, reads that wait for it
  std::vector<ConfEntry> confHost;
  size_t confCount = 0;
  double confSig[4] = {0, 0, 0, 0};   // Max_PenaltySpan, MutationPenalty, granularity, total size
  bool confDirty = true;
  DevBuf<ConfEntry> dConf;
  DevBuf<uint8_t> dConfMiss;
  DevBuf<int64_t> dListConf[2];
  std::vector<int32_t> residentLens, stagedLens;  // distinct total query lengths of the batch (the table is seeded for them)
  int64_t residentNq = -1;   // batch kept in HBM by xm_batch_upload
  int64_t residentGen = 0, lastAlignedGen = -1;  // which resident batch the streams of the last align call belong to
  int64_t lastAlignedNq = -1;  // queries whose result streams (dFinalInts / dFinalDbls / dFinalIntOff) are still in HBM from the last align call (xm_pileup_add_last)
  int residentMaxLen = 0;    // longest mate of that batch
  double residentH2dMs = 0;
  // second set of batch buffers: xm_batch_stage copies the next batch on its own stream while xm_align_resident works on the resident one
  std::mutex stageMu;
  hipStream_t copyStream = nullptr;
  hipEvent_t cev0 = nullptr, cev1 = nullptr;
  DevBuf<uint8_t> sCodes;
  DevBuf<int32_t> sMateCount, sMateLength;
  DevBuf<int64_t> sMateOffset;
  DevBuf<double> sExpected, sDeviation;
  int64_t stagedNq = -1;
  int stagedMaxLen = 0;
  double stagedH2dMs = 0;

  static constexpr size_t kConfMissCap = 1 << 16;
  std::unordered_set<int32_t> confSeeded;  // query lengths whose whole-substitution sums are in the table
  double confSeedRate = -1;               // (... for this MaxErrorRate)
  bool confInsert(double penalty, int32_t qlen, const Params& p) {  // -> false: already there
    uint64_t bits;
    memcpy(&bits, &penalty, 8);
    if ((confCount + 1) * 2 > confHost.size()) {  // grow (and rehash) at half load
      std::vector<ConfEntry> old;
      old.swap(confHost);
      confHost.assign(old.empty() ? (size_t)1 << 14 : old.size() * 2, ConfEntry{0, 0, 0, 0.0});
      for (const ConfEntry& e : old) if (e.used) { uint32_t h = confHash(e.penaltyBits, e.queryLength) & (uint32_t)(confHost.size() - 1); while (confHost[h].used) h = (h + 1) & (uint32_t)(confHost.size() - 1); confHost[h] = e; }
    }
    const uint32_t mask = (uint32_t)(confHost.size() - 1);
    uint32_t h = confHash(bits, qlen) & mask;
    while (confHost[h].used) {
      if (confHost[h].penaltyBits == bits && confHost[h].queryLength == qlen) return false;
      h = (h + 1) & mask;
    }
    const HostIndex& hst = hs->host;
    confHost[h] = ConfEntry{bits, qlen, 1, confidenceLengthOnHost(penalty, qlen, p.Max_PenaltySpan, p.MutationPenalty, hst.dupGranularity(), hst.totalForwardSize * 2)};
    confCount++;
    confDirty = true;
    return true;
  }
  // the table for this call: reset when the settings it depends on changed; seeded with what the batch's reads will ask for in the common case
  // (an alignment without indels costs a whole number of substitutions: the sums 0, m, m + m, ... up to the allowed penalty), whatever else
  // comes up (ambiguity and unaligned penalties, spacing penalties of pairs, other sums) is added after the pass that missed it
  void confPrepare(const Params& p, hipStream_t s) {
    const HostIndex& hst = hs->host;
    const double sig[4] = {p.Max_PenaltySpan, p.MutationPenalty, hst.dupGranularity(), (double)(hst.totalForwardSize * 2)};
    if (memcmp(sig, confSig, sizeof(sig)) != 0) { memcpy(confSig, sig, sizeof(sig)); confHost.clear(); confCount = 0; confDirty = true; confSeeded.clear(); }
    if (confSeedRate != p.MaxErrorRate) { confSeedRate = p.MaxErrorRate; confSeeded.clear(); }
    // a length is seeded once; a call seeds a bounded number of entries (fixed-length batches: a few dozen; a batch of unsplit long reads has thousands
    // of distinct lengths with thousands of sums each, of which the reads ask for a few: what is not seeded comes in through the miss path)
    long long budget = envKnob("XM_CONF_SEED", 1 << 18, 0, 1 << 26);  // (0: nothing seeded, every key through the miss path - the tests run that)
    for (int32_t len : residentLens) {
      if (confSeeded.count(len)) continue;
      const double limit = (double)len * p.MaxErrorRate + p.Max_PenaltySpan + p.MutationPenalty;
      const double steps = p.MutationPenalty > 0 ? std::min(4096.0, std::floor(limit / p.MutationPenalty) + 2) : 1;
      if (steps > (double)budget) continue;
      budget -= (long long)steps;
      double pen = 0;
      for (int j = 0; j < (int)steps && pen <= limit; j++) { confInsert(pen, len, p); pen += p.MutationPenalty; }
      confSeeded.insert(len);
    }
    confUpload(s);
    if (!dConfMiss.p) {
      dConfMiss.ensure(sizeof(ConfMiss) + kConfMissCap * sizeof(ConfMissKey));
      ConfMiss hdr;
      memset(&hdr, 0, sizeof(hdr));
      hdr.cap = kConfMissCap;
      HIP_CHECK(hipMemcpyAsync(dConfMiss.p, &hdr, offsetof(ConfMiss, keys), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));
    }
  }
  void confUpload(hipStream_t s) {
    if (!confDirty) return;
    if (confHost.empty()) confHost.assign((size_t)1 << 14, ConfEntry{0, 0, 0, 0.0});
    HIP_CHECK(hipStreamSynchronize(s));   // (no kernel of this context reads the old copy any more)
    dConf.ensure(confHost.size());
    HIP_CHECK(hipMemcpy(dConf.p, confHost.data(), confHost.size() * sizeof(ConfEntry), hipMemcpyHostToDevice));
    confDirty = false;
  }
  // after a pass with XM_ST_NEED_CONF reads: the keys they left, evaluated and added; -> number of new entries
  size_t confAbsorbMisses(const Params& p, hipStream_t s) {
    ConfMiss hdr;
    HIP_CHECK(hipMemcpyAsync(&hdr, dConfMiss.p, offsetof(ConfMiss, keys), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    const size_t n = (size_t)std::min<unsigned long long>(hdr.n, hdr.cap);
    std::vector<ConfMissKey> keys(n);
    if (n) HIP_CHECK(hipMemcpy(keys.data(), dConfMiss.p + offsetof(ConfMiss, keys), n * sizeof(ConfMissKey), hipMemcpyDeviceToHost));
    const unsigned long long zero = 0;
    HIP_CHECK(hipMemcpy(dConfMiss.p, &zero, sizeof(zero), hipMemcpyHostToDevice));
    size_t added = 0;
    for (const ConfMissKey& k : keys) { double pen; memcpy(&pen, &k.penaltyBits, 8); if (confInsert(pen, k.queryLength, p)) added++; }
    confUpload(s);
    return added;
  }

  void initContext() {  // stream and events of this context (the device tables exist)
    dt->contexts.fetch_add(1);
    HIP_CHECK(hipSetDevice(device));
    if (!stream) { HIP_CHECK(hipStreamCreate(&stream)); HIP_CHECK(hipEventCreate(&ev0)); HIP_CHECK(hipEventCreate(&ev1)); }
  }
  // Readable_HashBlock_Database.getContainingMap's growth (:108-113) for a batch whose longest mate is maxLen, then the device tables brought up to
  // the host's: any context may grow the shared tables; every GPU's copy follows before its next launch.  Called without rw held.
  void ensureTablesFor(int maxLen) {
    if (maxLen > hs->hashedLength.load()) {
      std::lock_guard<std::mutex> lock(hs->mu);
      if (maxLen > hs->host.maxHashedLength) { hs->host.ensureLength(maxLen); hs->hashedLength.store(hs->host.maxHashedLength); }
    }
    if (dt && dt->uploadedLength < hs->hashedLength.load()) {
      std::lock_guard<std::mutex> lock(hs->mu);
      std::unique_lock<std::shared_mutex> wr(dt->rw);  // (waits for the launches of every context of this GPU)
      if (dt->uploadedLength < hs->host.maxHashedLength) dt->upload();
    }
  }
  ~xm_index() {
    if (dt) dt->contexts.fetch_sub(1);
    if (hostOnly) return;
    (void)hipSetDevice(device);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (cev0) (void)hipEventDestroy(cev0);
    if (cev1) (void)hipEventDestroy(cev1);
    if (stream) (void)hipStreamDestroy(stream);
    if (copyStream) (void)hipStreamDestroy(copyStream);
  }  // (every DevBuf member releases its memory itself; the shared tables go with their last context)
};

struct xm_pileup {
  xm_index* index = nullptr;            // the context whose batches are added (xm_pileup_add_last needs it alive; read / events / free do not)
  std::shared_ptr<HostShare> hs;
  int device = 0;
  DevBuf<unsigned long long> dDepth, dAlt, dEventCount, dMid;
  double endFraction = 0;
  DevBuf<long long> dEvents;
  long long total = 0, queriesAdded = 0;
  std::vector<long long> events;  // (host) 8 per event, in the order of the calls
};

extern "C" {

const char* xm_last_error(void) { return g_error.c_str(); }
#ifndef XM_BUILD_STAMP
#define XM_BUILD_STAMP "unstamped"
#endif
const char* xm_build_stamp(void) { return XM_BUILD_STAMP; }
int32_t xm_abi_version(void) { return 2; }
int64_t xm_pinned_host_bytes(int64_t* high_water) {
  if (high_water) *high_water = (int64_t)g_pinned->highWater.load();
  return (int64_t)g_pinned->allocatedBytes.load();
}

int xm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int xm_index_build(const xm_ref* ref, const xm_build_opts* optsIn, xm_index** out) {
  if (!ref || !out) return fail("xm_index_build: null argument");
  xm_build_opts o;
  memset(&o, 0, sizeof(o));
  o.enable_gapmers = 1; o.dup_window = 1000; o.dup_min_copies = 2; o.device = -1;
  if (optsIn) o = *optsIn;
  xm_index* idx = nullptr;
  try {
    idx = new xm_index();
    idx->hs = std::make_shared<HostShare>();
    idx->host().setReference(ref->num_contigs, ref->names, ref->codes, ref->lengths);
    idx->hostOnly = o.host_only != 0;
    if (!idx->hostOnly) {
      int n = 0;
      if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
        throw std::runtime_error("no HIP device available: libxmapper_hip.so has no CPU path (pass host_only=1 only to inspect the index)");
      int dev = o.device;
      if (dev < 0) HIP_CHECK(hipGetDevice(&dev));
      idx->device = dev;
      idx->host().deviceHasher = &deviceHashLengths;  // the tables are hashed on this GPU (references without ambiguity codes)
      idx->host().deviceForBuild = dev;
    }
    idx->host().build(o.enable_gapmers, o.min_interesting_size, o.max_hashed_length, o.dup_window, o.dup_min_copies, o.dup_min_length, o.dup_max_length);
    idx->hs->hashedLength.store(idx->host().maxHashedLength);
    if (!idx->hostOnly) {
      idx->dt = std::make_shared<DeviceTables>();
      idx->dt->hs = idx->hs; idx->dt->device = idx->device;
      idx->dt->upload();
      idx->initContext();
    }
    *out = idx;
    return 0;
  } catch (std::exception& e) {
    delete idx;
    return fail(std::string("xm_index_build: ") + e.what());
  }
}

int xm_index_replicate(xm_index* src, int32_t device, xm_index** out) {
  if (!src || !out) return fail("xm_index_replicate: null argument");
  if (src->hostOnly) return fail("xm_index_replicate: index was built with host_only=1");
  xm_index* idx = nullptr;
  try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) throw std::runtime_error("no HIP device available");
    if (device < 0 || device >= n) throw std::runtime_error("device " + std::to_string(device) + " does not exist (" + std::to_string(n) + " devices)");
    idx = new xm_index();
    idx->hs = src->hs;          // the host tables are shared, never copied
    idx->hostOnly = false;
    idx->device = device;
    idx->scratchBytes = src->scratchBytes;
    if (device == src->device) {
      idx->dt = src->dt;        // a context on the same GPU reads the same tables in HBM
    } else {
      int can = 0;
      HIP_CHECK(hipDeviceCanAccessPeer(&can, device, src->device));
      if (can) { HIP_CHECK(hipSetDevice(device)); hipError_t e = hipDeviceEnablePeerAccess(src->device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e); (void)hipGetLastError(); }
      std::lock_guard<std::mutex> lock(src->hs->mu);            // (not while the tables grow)
      std::shared_lock<std::shared_mutex> rd(src->dt->rw);      // (nor while the source's copy is brought up to them)
      idx->dt = std::make_shared<DeviceTables>();
      idx->dt->hs = idx->hs; idx->dt->device = device;
      std::unique_lock<std::shared_mutex> wr(idx->dt->rw);
      if (src->dt->uploadedLength == src->hs->host.maxHashedLength) idx->dt->upload(src->dt.get());  // HBM to HBM (xGMI)
      else idx->dt->upload();
      HIP_CHECK(hipDeviceSynchronize());
    }
    idx->initContext();
    *out = idx;
    return 0;
  } catch (std::exception& e) {
    delete idx;
    return fail(std::string("xm_index_replicate: ") + e.what());
  }
}

int xm_context_new(xm_index* index, xm_index** out) {
  if (!index || !out) return fail("xm_context_new: null argument");
  return xm_index_replicate(index, index->device, out);
}

int xm_context_set_scratch(xm_index* idx, int64_t bytes) {
  if (!idx) return fail("xm_context_set_scratch: null argument");
  if (bytes < 0) return fail("xm_context_set_scratch: negative size");
  std::lock_guard<std::mutex> lock(idx->mu);
  idx->scratchBytes = bytes;
  if (bytes > 0 && idx->dArenas.n > (size_t)bytes && !idx->hostOnly) {  // what the context holds beyond its new limit goes back to the GPU now
    (void)hipSetDevice(idx->device);
    idx->dArenas.release();
  }
  return 0;
}

int xm_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes) {
  try {
    HIP_CHECK(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_device_memory: ") + e.what()); }
}

int xm_index_save(xm_index* idx, const char* path) {
  if (!idx || !path) return fail("xm_index_save: null argument");
  try {
    std::lock_guard<std::mutex> lock(idx->hs->mu);
    idx->host().save(path);
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_index_save: ") + e.what()); }
}

int xm_index_load(const char* path, const xm_ref* ref, const xm_build_opts* optsIn, xm_index** out) {
  if (!path || !out) return fail("xm_index_load: null argument");
  xm_build_opts o;
  memset(&o, 0, sizeof(o));
  o.enable_gapmers = 1; o.dup_window = 1000; o.dup_min_copies = 2; o.device = -1;
  if (optsIn) o = *optsIn;
  xm_index* idx = nullptr;
  try {
    idx = new xm_index();
    idx->hs = std::make_shared<HostShare>();
    idx->host().load(path);
    if (ref) {  // the file must answer exactly this build request (the reference's cache keys, M/HashBlock_Database.java:106-114)
      HostIndex want;
      want.setReference(ref->num_contigs, ref->names, ref->codes, ref->lengths);
      if (!idx->host().matchesRequest(want, o.enable_gapmers, o.min_interesting_size, o.dup_window, o.dup_min_copies, o.dup_min_length, o.dup_max_length))
        throw std::runtime_error("the file was built from another reference or with other settings");
    }
    idx->hostOnly = o.host_only != 0;
    if (!idx->hostOnly) {
      int n = 0;
      if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
        throw std::runtime_error("no HIP device available: libxmapper_hip.so has no CPU path (pass host_only=1 only to inspect the index)");
      int dev = o.device;
      if (dev < 0) HIP_CHECK(hipGetDevice(&dev));
      idx->device = dev;
      idx->host().deviceHasher = &deviceHashLengths;
      idx->host().deviceForBuild = dev;
    }
    if (o.max_hashed_length > idx->host().maxHashedLength) idx->host().ensureLength(o.max_hashed_length);
    idx->hs->hashedLength.store(idx->host().maxHashedLength);
    if (!idx->hostOnly) {
      idx->dt = std::make_shared<DeviceTables>();
      idx->dt->hs = idx->hs; idx->dt->device = idx->device;
      idx->dt->upload();
      idx->initContext();
    }
    *out = idx;
    return 0;
  } catch (std::exception& e) {
    delete idx;
    return fail(std::string("xm_index_load: ") + e.what());
  }
}

int xm_index_ensure_length(xm_index* idx, int32_t length) {
  if (!idx) return fail("null index");
  try {
    idx->ensureTablesFor(length);
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_index_ensure_length: ") + e.what()); }
}

void xm_index_free(xm_index* idx) { delete idx; }

int xm_index_get_info(const xm_index* idx, xm_index_info_t* info) {
  if (!idx || !info) return fail("null argument");
  std::lock_guard<std::mutex> hostLock(idx->hs->mu);  // (another context's batch may be growing the shared host tables)
  const HostIndex& h = idx->host();
  info->num_contigs = h.numContigs(); info->min_interesting_size = h.minInterestingSize; info->max_hashed_length = h.maxHashedLength;
  info->enable_gapmers = h.enableGapmers; info->dup_window = h.dupWindow; info->position_bytes = (idx->hostOnly ? h.seqCumStart.back() > 0xFFFFFFFFll : idx->dt->posIs64) ? 8 : 4;
  info->total_forward_size = h.totalForwardSize;
  info->num_positions = (int64_t)h.positions.size();
  info->index_bytes = (int64_t)(h.bucketOff.size() * 4 + h.positions.size() * (size_t)info->position_bytes + h.refCodes.size() + h.dupKeys.size() * 4);
  info->dup_granularity = h.dupGranularity();
  info->built_on_device = h.builtOnDevice ? 1 : 0;
  info->bucket_line_bytes = idx->hostOnly ? 0 : (idx->dt->view.lines64 ? 64 : (idx->dt->view.lines32 ? 32 : 0));
  info->hash_seconds = h.hashSeconds; info->duplication_seconds = h.dupSeconds;
  return 0;
}

int xm_index_table_info(const xm_index* idx, int32_t L, int32_t* capacity, int32_t* maxCount, int64_t* numStored, int64_t* numOverfull) {
  if (!idx) return fail("null index");
  std::lock_guard<std::mutex> hostLock(idx->hs->mu);  // (another context's batch may be growing the shared host tables)
  const HostIndex& h = idx->host();
  if (L < 0 || L > h.maxHashedLength) return fail("length not hashed");
  const Table& t = h.tables[(size_t)L];
  *capacity = t.capacity; *maxCount = t.maxCount;
  int64_t o = 0;
  for (int k = 0; k < t.capacity; k++) if (h.bucketOff[(size_t)(t.offBase + k)] & XM_OVERFULL) o++;
  *numOverfull = o;
  *numStored = (int64_t)(h.bucketOff[(size_t)(t.offBase + t.capacity)] & ~XM_OVERFULL);
  return 0;
}

int xm_index_table_shape(const xm_index* idx, int32_t L, int32_t* capacity, int32_t* maxCount) {
  if (!idx || !capacity || !maxCount) return fail("null argument");
  std::lock_guard<std::mutex> hostLock(idx->hs->mu);  // (another context's batch may be growing the shared host tables)
  const HostIndex& h = idx->host();
  if (L < 0 || L > h.maxHashedLength) return fail("length not hashed");
  *capacity = h.tables[(size_t)L].capacity; *maxCount = h.tables[(size_t)L].maxCount;
  return 0;
}

// How many buckets the hashed tables have, how many hold at least one position, and how many are overfull (more than max(L^2, 5) entries: dropped, HashBlock_Database.java:569-577)
int xm_index_bucket_stats(const xm_index* idx, int64_t* buckets, int64_t* occupied, int64_t* overfull) {
  if (!idx || !buckets || !occupied || !overfull) return fail("null argument");
  std::lock_guard<std::mutex> hostLock(idx->hs->mu);
  const HostIndex& h = idx->host();
  int64_t nb = 0, no = 0, nf = 0;
  for (int L = 0; L <= h.maxHashedLength; L++) {
    const Table& t = h.tables[(size_t)L];
    if (t.capacity <= 1) continue;  // (the placeholder maps of lengths that are not hashed)
    nb += t.capacity;
    for (int k = 0; k < t.capacity; k++) {
      const uint32_t a = h.bucketOff[(size_t)t.offBase + (size_t)k], b = h.bucketOff[(size_t)t.offBase + (size_t)k + 1];
      if (a & XM_OVERFULL) nf++;
      else if ((b & ~XM_OVERFULL) > (a & ~XM_OVERFULL)) no++;
    }
  }
  *buckets = nb; *occupied = no; *overfull = nf;
  return 0;
}

int xm_index_table_dump(const xm_index* idx, int32_t L, int32_t* counts, int64_t* positionsOut) {
  if (!idx) return fail("null index");
  std::lock_guard<std::mutex> hostLock(idx->hs->mu);  // (another context's batch may be growing the shared host tables)
  const HostIndex& h = idx->host();
  if (L < 0 || L > h.maxHashedLength) return fail("length not hashed");
  const Table& t = h.tables[(size_t)L];
  int64_t w = 0;
  for (int k = 0; k < t.capacity; k++) {
    uint32_t o0 = h.bucketOff[(size_t)(t.offBase + k)], o1 = h.bucketOff[(size_t)(t.offBase + k + 1)];
    if (o0 & XM_OVERFULL) { counts[k] = -1; continue; }
    int c = (int)((o1 & ~XM_OVERFULL) - (o0 & ~XM_OVERFULL));
    counts[k] = c;
    for (int j = 0; j < c; j++) positionsOut[w++] = (int64_t)h.positions[(size_t)(t.posBase + (o0 & ~XM_OVERFULL) + j)];
  }
  return 0;
}

int64_t xm_index_dup_keys(const xm_index* idx, int32_t contig, int32_t* out, int64_t cap) {
  if (!idx || contig < 0 || contig >= idx->host().numContigs()) return -1;
  std::lock_guard<std::mutex> hostLock(idx->hs->mu);  // (another context's batch may be growing the shared host tables)
  const HostIndex& h = idx->host();
  int64_t a = h.dupKeyStart[(size_t)contig], b = h.dupKeyStart[(size_t)contig + 1];
  for (int64_t i = a; i < b && i - a < cap; i++) out[i - a] = h.dupKeys[(size_t)i];
  return b - a;
}

void xm_result_free(xm_result* r) {
  if (!r) return;
  ResultBox* box = (ResultBox*)r;  // pub is the first member
  g_pinned->put(r->ints, box->bytesInts); g_pinned->put(r->dbls, box->bytesDbls);
  g_pinned->put(r->int_off, box->bytesIntOff); g_pinned->put(r->dbl_off, box->bytesDblOff);
  free(box);
}

// validation + Readable_HashBlock_Database growth + host-to-device copy of one batch; the batch stays resident in HBM
static int validateBatch(const xm_query_batch* b, bool* anyPaired = nullptr, std::vector<int32_t>* totalLengths = nullptr) {  // -> longest mate
  const int64_t nq = b->num_queries;
  int maxLen = 1;
  if (anyPaired) *anyPaired = false;
  std::vector<uint8_t> seen(totalLengths ? 60001 : 0, 0);   // total query lengths that occur (Query.getLength(): what the confidence table is keyed by)
  for (int64_t q = 0; q < nq; q++) {
    if (b->mate_count[q] < 1 || b->mate_count[q] > 2) throw std::runtime_error("mate_count must be 1 or 2");
    if (anyPaired && b->mate_count[q] == 2) *anyPaired = true;
    int total = 0;
    for (int m = 0; m < b->mate_count[q]; m++) {
      int32_t len = b->mate_length[q * 2 + m];
      if (len < 1 || len > 30000) throw std::runtime_error("mate length out of range (1..30000; longer reads are split by the caller as --split-queries-past-size does)");
      if (b->mate_offset[q * 2 + m] < 0 || b->mate_offset[q * 2 + m] + len > b->codes_length) throw std::runtime_error("mate outside of codes");
      if (len > maxLen) maxLen = len;
      total += len;
      if (totalLengths) seen[(size_t)len] = 1;
    }
    if (totalLengths) seen[(size_t)total] = 1;
  }
  if (totalLengths) {
    totalLengths->clear();
    for (size_t i = 0; i < seen.size(); i++) if (seen[i]) totalLengths->push_back((int32_t)i);
  }
  return maxLen;
}

static void uploadBatchLocked(xm_index* idx, const xm_query_batch* b) {
  const int64_t nq = b->num_queries;
  bool anyPaired = false;
  const int maxLen = validateBatch(b, &anyPaired, &idx->residentLens);
  idx->residentAnyPaired = anyPaired;
  idx->ensureTablesFor(maxLen);  // Readable_HashBlock_Database.getContainingMap growth, done before the launch
  HIP_CHECK(hipSetDevice(idx->device));
  hipStream_t s = idx->stream;
  idx->residentNq = -1;
  if (nq > 0) {
    hipEvent_t e0 = idx->ev0, e1 = idx->ev1;
    HIP_CHECK(hipEventRecord(e0, s));
    idx->dMateCount.ensure((size_t)nq); idx->dMateOffset.ensure((size_t)nq * 2); idx->dMateLength.ensure((size_t)nq * 2);
    idx->dCodes.ensure((size_t)b->codes_length); idx->dExpected.ensure((size_t)nq); idx->dDeviation.ensure((size_t)nq);
    HIP_CHECK(hipMemcpyAsync(idx->dMateCount.p, b->mate_count, sizeof(int32_t) * (size_t)nq, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(idx->dMateOffset.p, b->mate_offset, sizeof(int64_t) * (size_t)nq * 2, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(idx->dMateLength.p, b->mate_length, sizeof(int32_t) * (size_t)nq * 2, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(idx->dCodes.p, b->codes, (size_t)b->codes_length, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(idx->dExpected.p, b->expected_inner, sizeof(double) * (size_t)nq, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(idx->dDeviation.p, b->deviation, sizeof(double) * (size_t)nq, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipEventRecord(e1, s));
    HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    idx->residentH2dMs = ms;
  }
  idx->residentNq = nq;
  idx->residentGen++;
  idx->residentMaxLen = maxLen;
}

static int alignResidentLocked(xm_index* idx, const xm_params* p, xm_result** out);

int xm_batch_upload(xm_index* idx, const xm_query_batch* b) {
  if (!idx || !b) return fail("xm_batch_upload: null argument");
  if (idx->hostOnly) return fail("xm_batch_upload: index was built with host_only=1");
  try {
    std::lock_guard<std::mutex> lock(idx->mu);
    uploadBatchLocked(idx, b);
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_batch_upload: ") + e.what()); }
}

int xm_batch_stage(xm_index* idx, const xm_query_batch* b) {
  if (!idx || !b) return fail("xm_batch_stage: null argument");
  if (idx->hostOnly) return fail("xm_batch_stage: index was built with host_only=1");
  try {
    std::lock_guard<std::mutex> stageLock(idx->stageMu);
    const int64_t nq = b->num_queries;
    bool anyPaired = false;
    const int maxLen = validateBatch(b, &anyPaired, &idx->stagedLens);
    idx->stagedAnyPaired = anyPaired;
    idx->ensureTablesFor(maxLen);  // (tables that grow wait for the launches that read them: DeviceTables::rw)
    HIP_CHECK(hipSetDevice(idx->device));
    if (!idx->copyStream) { HIP_CHECK(hipStreamCreateWithFlags(&idx->copyStream, hipStreamNonBlocking)); HIP_CHECK(hipEventCreate(&idx->cev0)); HIP_CHECK(hipEventCreate(&idx->cev1)); }
    hipStream_t s = idx->copyStream;
    idx->stagedNq = -1;
    idx->stagedH2dMs = 0;
    if (nq > 0) {
      HIP_CHECK(hipEventRecord(idx->cev0, s));
      idx->sMateCount.ensure((size_t)nq); idx->sMateOffset.ensure((size_t)nq * 2); idx->sMateLength.ensure((size_t)nq * 2);
      idx->sCodes.ensure((size_t)b->codes_length); idx->sExpected.ensure((size_t)nq); idx->sDeviation.ensure((size_t)nq);
      HIP_CHECK(hipMemcpyAsync(idx->sMateCount.p, b->mate_count, sizeof(int32_t) * (size_t)nq, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(idx->sMateOffset.p, b->mate_offset, sizeof(int64_t) * (size_t)nq * 2, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(idx->sMateLength.p, b->mate_length, sizeof(int32_t) * (size_t)nq * 2, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(idx->sCodes.p, b->codes, (size_t)b->codes_length, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(idx->sExpected.p, b->expected_inner, sizeof(double) * (size_t)nq, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(idx->sDeviation.p, b->deviation, sizeof(double) * (size_t)nq, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipEventRecord(idx->cev1, s));
      HIP_CHECK(hipStreamSynchronize(s));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, idx->cev0, idx->cev1));
      idx->stagedH2dMs = ms;
    }
    idx->stagedNq = nq;
    idx->stagedMaxLen = maxLen;
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_batch_stage: ") + e.what()); }
}

int xm_batch_commit(xm_index* idx) {
  if (!idx) return fail("xm_batch_commit: null argument");
  try {
    std::lock_guard<std::mutex> stageLock(idx->stageMu);
    if (idx->stagedNq < 0) return fail("xm_batch_commit: no staged batch (call xm_batch_stage first)");
    std::lock_guard<std::mutex> lock(idx->mu);  // (waits for a running xm_align_resident)
    idx->dMateCount.swapWith(idx->sMateCount); idx->dMateOffset.swapWith(idx->sMateOffset); idx->dMateLength.swapWith(idx->sMateLength);
    idx->dCodes.swapWith(idx->sCodes); idx->dExpected.swapWith(idx->sExpected); idx->dDeviation.swapWith(idx->sDeviation);
    idx->residentGen++;
    idx->residentNq = idx->stagedNq; idx->residentMaxLen = idx->stagedMaxLen; idx->residentH2dMs = idx->stagedH2dMs; idx->residentAnyPaired = idx->stagedAnyPaired;
    idx->residentLens.swap(idx->stagedLens);
    idx->stagedNq = -1;
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_batch_commit: ") + e.what()); }
}

int xm_align_resident(xm_index* idx, const xm_params* p, xm_result** out) {
  if (!idx || !p || !out) return fail("xm_align_resident: null argument");
  try {
    std::lock_guard<std::mutex> lock(idx->mu);
    if (idx->residentNq < 0) throw std::runtime_error("no batch is resident (call xm_batch_upload first)");
    return alignResidentLocked(idx, p, out);
  } catch (std::exception& e) { return fail(std::string("xm_align_resident: ") + e.what()); }
}

int xm_align_batch(xm_index* idx, const xm_params* p, const xm_query_batch* b, xm_result** out) {
  if (!idx || !p || !b || !out) return fail("xm_align_batch: null argument");
  if (idx->hostOnly) return fail("xm_align_batch: index was built with host_only=1; this library aligns on the GPU only");
  try {
    std::lock_guard<std::mutex> lock(idx->mu);
    uploadBatchLocked(idx, b);
    int rc = alignResidentLocked(idx, p, out);
    if (rc == 0) (*out)->h2d_ms = idx->residentH2dMs;
    return rc;
  } catch (std::exception& e) {
    return fail(std::string("xm_align_batch: ") + e.what());
  }
}

static int alignResidentLocked(xm_index* idx, const xm_params* p, xm_result** out) {
  ResultBox* box = (ResultBox*)calloc(1, sizeof(ResultBox));
  xm_result* res = &box->pub;
  try {
    const int64_t nq = idx->residentNq;
    HIP_CHECK(hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    // the shared tables stay as they are while this call's kernels read them (another context that grows them waits; so does this one's next growth)
    std::shared_lock<std::shared_mutex> tablesInUse(idx->dt->rw);
    IndexView view = idx->dt->view;
    const int numCUs = idx->dt->numCUs;
    res->num_queries = nq;
    res->int_off = (int64_t*)g_pinned->get(sizeof(int64_t) * (size_t)(nq + 1), &box->bytesIntOff);
    res->dbl_off = (int64_t*)g_pinned->get(sizeof(int64_t) * (size_t)(nq + 1), &box->bytesDblOff);
    if (nq == 0) {
      res->ints = (int32_t*)g_pinned->get(4, &box->bytesInts); res->dbls = (double*)g_pinned->get(8, &box->bytesDbls);
      res->int_off[0] = res->dbl_off[0] = 0;
      *out = res;
      return 0;
    }
    hipEvent_t e0 = idx->ev0, e1 = idx->ev1;
    float ms = 0;

    BatchView bv{nq, idx->dMateCount.p, idx->dMateOffset.p, idx->dMateLength.p, idx->dCodes.p, idx->dExpected.p, idx->dDeviation.p};
    Params params;
    params.MutationPenalty = p->MutationPenalty; params.InsertionStart_Penalty = p->InsertionStart_Penalty; params.InsertionExtension_Penalty = p->InsertionExtension_Penalty;
    params.DeletionStart_Penalty = p->DeletionStart_Penalty; params.DeletionExtension_Penalty = p->DeletionExtension_Penalty; params.MaxErrorRate = p->MaxErrorRate;
    params.UnalignedPenalty = p->UnalignedPenalty; params.AmbiguityPenalty = p->AmbiguityPenalty; params.Max_PenaltySpan = p->Max_PenaltySpan;
    params.MaxNumMatches = p->MaxNumMatches; params.StartingInsertionStartFree = 0;

    idx->confPrepare(params, s);
    view.conf = idx->dConf.p; view.confMask = (uint32_t)(idx->confHost.size() - 1); view.confMiss = (ConfMiss*)idx->dConfMiss.p;
    idx->dListConf[0].ensure((size_t)nq); idx->dListConf[1].ensure((size_t)nq);
    idx->dStatus.ensure((size_t)nq); idx->dIntOff.ensure((size_t)nq); idx->dDblOff.ensure((size_t)nq); idx->dIntLen.ensure((size_t)nq); idx->dDblLen.ensure((size_t)nq);
    idx->dCursors.ensure(4); idx->dCounters.ensure(1); idx->dCtl.ensure(1);
    idx->dListHeavy.ensure((size_t)nq); idx->dListHeavyLate.ensure((size_t)nq);
    HIP_CHECK(hipMemsetAsync(idx->dCounters.p, 0, sizeof(DevCounters), s));
    HIP_CHECK(hipMemsetAsync(idx->dCursors.p, 0, sizeof(unsigned long long) * 4, s));
    PassCtl ctl0;
    memset(&ctl0, 0, sizeof(ctl0));
    ctl0.errQuery = ~0ull;
    HIP_CHECK(hipMemcpyAsync(idx->dCtl.p, &ctl0, sizeof(ctl0), hipMemcpyHostToDevice, s));

    // Passes, all on the GPU:
    //  (1) light pass over every read at scale 1: reads that reach the gapped extension chain stop with XM_ST_NEED_HEAVY
    //      instead of serialising their wave;
    //  (2) gapped pass over exactly those reads at scale 4, continued from the state the light pass saved (HandOver);
    //  (3) reads whose scratch overflowed are rerun with 16x, 64x, ... the scratch.
    // The work lists are built on the GPU by the lanes themselves (PassLists); every pass appends to the same result arenas.
#ifdef XM_READ_TIMES
    DevBuf<unsigned long long> dReadTimes;
    const char* readTimesFile = getenv("XM_READ_TIMES_FILE");
    if (readTimesFile && *readTimesFile) {
      dReadTimes.ensure((size_t)nq);
      HIP_CHECK(hipMemset(dReadTimes.p, 0, sizeof(unsigned long long) * (size_t)nq));
      unsigned long long* ptr = dReadTimes.p;
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(xm_read_times), &ptr, sizeof(ptr)));
    }
#endif
    const int64_t* todo = nullptr;  // device list of the current pass; null on the first pass = all reads
    long long nTodo = nq;
    unsigned long long pendingHeavy = 0, pendingScale = 0;
    int ts = 0, to = 0, tc = 0;  // which of the two scale / out / confidence lists receives new entries
    unsigned long long pendingConf = 0;
    int confRounds = 0;
    // the scratch capacities are sized for ~150-300 bp mates at scale 1; batches of longer reads start at a larger scale instead of
    // sending every read through a pass that can only overflow
    int scale = idx->residentMaxLen <= 320 ? 1 : (idx->residentMaxLen <= 1280 ? 4 : 16), overflowScale = scale;
    const int gappedScale = scale < 4 ? (int)envKnob("XM_GAPPED_SCALE", 4, 1, 64, true) : scale * (int)envKnob("XM_GAPPED_FACTOR", 4, 1, 64, true);
    bool heavy = false;
    unsigned long long intCap = (unsigned long long)nq * 40 + 4096, dblCap = (unsigned long long)nq * 12 + 4096;
    idx->dOutInts.ensure((size_t)intCap); idx->dOutDbls.ensure((size_t)dblCap);
    intCap = idx->dOutInts.n; dblCap = idx->dOutDbls.n;
    unsigned long long cursors[4] = {0, 0, 0, 0};
    double kernelMs = 0;
    int launches = 0;
    int64_t rerun = 0;
    const size_t arenaUnit = (size_t)envKnob("XM_ARENA_KB", 288, 64, 16384) * 1024;  // scratch of a lane at scale 1 (experiment knob: the capacities do not follow it, a smaller arena only overflows earlier)
    // scratch limit of this context: xm_context_set_scratch, else XM_SCRATCH_GIB (experiment knob), else 200 GiB
    const unsigned long long scratchWanted = idx->scratchBytes > 0 ? (unsigned long long)idx->scratchBytes : (unsigned long long)envKnob("XM_SCRATCH_GIB", 200, 1, 280) << 30;
    int scratchShift = 0;  // halved after an allocation the GPU had no room for (another process, or contexts that were given more than there is)
    // wave slots a launch is sized for (in waves per SIMD): alone on the GPU a context fills it (4 are resident at 128 registers; the light pass asks
    // for twice that, the second half starts as the first drains).  Contexts that share the GPU (xm_context_new) must leave each other room: a
    // persistent launch that holds every slot keeps the next context's launch waiting until its own tail, and the contexts then run one after the
    // other instead of side by side.  Together the contexts of a GPU ask for 12 waves per SIMD worth of light lanes and 6 of gapped lanes: two contexts 6 / 3
    // each (round 3), three 4 / 2 - 14.1-14.3 M reads/s against 13.1-13.2 with two, once the runtime has hardware queues for three contexts' streams
    // (GPU_MAX_HW_QUEUES, mapper_amd/_capi.py); with 6 / 3 each three contexts measured 12.3-12.7, four with 3 / 1 13.5 (profiles/r04/NOTES.md 15)
    // (the contexts that EXIST on the GPU, not the ones aligning at the moment: sizing by activity was tried in round 5 and made the headline bimodal - a context that
    // finds itself alone launches for the whole GPU, the runtime then gives its queue scratch memory for a whole GPU's waves (5.8 KB per lane), and in about half the
    // runs the other contexts' launches then ran one after the other for the rest of the process, 4.6 M reads/s instead of 15.  A process that keeps contexts it does
    // not use should close them.)
    const int gpuContexts = idx->dt->contexts.load();
    const bool sharedGpu = gpuContexts > 1;
    // Batches of long reads (gapped pass beyond scale 4: every read goes through the chain, and its searches - thousands of nodes each, all in HBM mode -
    // are most of its time): the lanes of a wave run their searches one after the other, so 8 reads per wave on twice as many waves instead of 32
    // (1 kb queries: 382 ms -> 265-280 ms per 150 k; 4 to 8 reads per wave and 8 to 16 waves per SIMD worth of lanes measure the same, profiles/r03/NOTES.md 13)
    const bool longReads = gappedScale > 4;
    // (long reads: every lane of the light pass holds a region of 288 KiB, and every read goes on to the gapped pass, whose lanes are 6.7 MB each:
    // two waves per SIMD worth of light lanes leave the scratch to those)
    const long long lightWaves = envKnob("XM_LIGHT_WAVES", longReads ? 2 : (sharedGpu ? std::max(2, 12 / gpuContexts) : 8), 1, 16), fullWaves = envKnob("XM_FULL_WAVES", longReads ? 8 : (sharedGpu ? std::max(1, 6 / gpuContexts) : 4), 1, 16);
    const long long fullLpw = envKnob("XM_FULL_LPW", longReads ? 8 : 32, 1, 64), lightLpw = envKnob("XM_LIGHT_LPW", 64, 1, 64);
    const long long lightLevel = envKnob("XM_LIGHT_LEVEL", 0, 0, 2);  // what the light pass still does itself (Caps::heavyAllowed)
    // straight-alignment penalty x 8 from which a read is put first in the gapped pass and dealt out evenly (0: no order).  Batches of single reads of up to 320
    // bases: 8 penalty units - the reads with an indel (they mismatch on one whole side of it), whose searches are the long ones of the pass: started first
    // they do not end it (gapped pass 70.0 / 70.5 -> 65.2 / 64.8 ms per 1 M reads, same box; with 4 units 75 ms; pairs 146 -> 152-154 ms: not for them)
    const long long heavyHintThreshold = envKnob("XM_HEAVY_HINT", (idx->residentAnyPaired || longReads) ? 0 : 64, 0, 1 << 20);
    const long long taperWaves = envKnob("XM_TAPER_PCT", 100, 0, 1000);  // lane l of a gapped-pass wave stops taking reads when fewer than l * waves * pct/100 are left
    auto scratchBudget = [&]() -> unsigned long long {  // scratch: up to the limit, never more than 3/4 of what is free now (+ what this context already holds)
      unsigned long long want = scratchWanted >> scratchShift;
      size_t freeB = 0, totalB = 0;
      if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
        const unsigned long long avail = (unsigned long long)(freeB + idx->dArenas.n) / 4 * 3;
        if (want > avail) want = avail;
      }
      return want < (64ull << 20) ? (64ull << 20) : want;
    };
    // contexts of one GPU size and allocate their scratch one after the other: they all look at the same free memory
    auto allocScratch = [&](size_t bytes) -> bool {
      if (idx->dArenas.tryEnsure(bytes)) return true;
      if (++scratchShift > 8) throw std::runtime_error("no room in HBM for the scratch of even a few lanes (" + std::to_string(bytes >> 20) + " MiB asked)");
      return false;
    };
    // light pass -> gapped pass hand-over (HandOver, SavedRead): the reads the light pass stops in front of the gapped chain keep their seeding
    // state in HBM and the gapped pass continues from it.  Scratch layout while saved regions are alive: [region pool | lane arenas].
    const bool pairMode = envInt("XM_PAIR_LANES", 1) != 0;
    const bool groupLanes = envInt("XM_GROUP_LANES", 1) != 0;
    // (batches of long reads only: where reads align, the filter costs what it saves - 2 % of the search nodes of configs[1], 16 % of a repeat-rich reference's
    // sit in searches it rejects, and it would look at every search: profiles/r06/NOTES.md 1.  XM_BOUND_FILTER=0: off, for comparison)
    const bool boundFilterOn = envInt("XM_BOUND_FILTER", 1) != 0 && longReads;
    bool boundFilterUsed = false;
    // temporaries of a gapped-pass lane (reads that resume from a saved region): 7/12 of the arena of that scale by default (experiment knob: percent of it)
    // HBM-mode searches take their arrays from a pool of the launch (SearchPool) in batches of short reads (gapped pass at scale <= 4): a lane's
    // temporaries then hold the chain's structures only (matchers 148 KB + piece lists 23 KB + small change at scale 4; default 30 % of 7/12 of
    // the arena = 201 KB).  Batches of long reads run every search in HBM mode: no pool, whole temporaries.
    const bool searchPoolOn = envInt("XM_SEARCH_POOL", 1) != 0 && gappedScale <= 4;
    const long long gappedTmpPct = envKnob("XM_GAPPED_TMP_PCT", searchPoolOn ? 20 : 100, 5, 100);  // (134 KB: matchers 74 KB, piece lists 23 KB, the rest small change)
    // (+ the node arrays of a long-read chain: applyChainCaps)
    auto gappedTmpBytes = [&](size_t arena) -> size_t { return ((size_t)((arena - arenaPersistBytes(arena)) * (size_t)gappedTmpPct / 100) & ~(size_t)15) + chainExtraTmpBytes(gappedScale); };
    // light pass: a lane's temporaries hold the three matchers alignMatch sets aside (37 KB at scale 1; the chain that would fill them does not
    // run there) and the joined text of overlapping mates; a read's region holds its seeding state: 49 KB single-end, 99 KB paired at scale 1
    // (ambiguity codes add up to 18 KB per mate: a pair with them in both mates overflows its region - and the region of the same size a gapped-pass lane seeds reads
    // without saved state in - so it is filed for the pass behind the gapped pass and run from its start at four times the gapped pass's scale, one read per wave.
    // Correct (the ambiguity fuzz equals the oracle) and late: FASTQ pairs with N tails in both mates pay a latency-bound extra pass.  Known, not fixed: knowing it
    // at upload would mean reading every base of the batch on the host.)
    const size_t lightTmpUnit = (size_t)envKnob("XM_LIGHT_TMP_KB", 48, 16, 16384) * 1024;
    const size_t regionPersistUnit = (size_t)envKnob("XM_REGION_KB", idx->residentAnyPaired ? 120 : 72, 32, 16384) * 1024;
    const bool handOver = envInt("XM_HANDOVER", 1) != 0;
    bool orderedList = false;   // the next launch's list is the gapped pass's ordered one (expensive-looking reads first): only that list is dealt out lane-major
    int hoMode = handOver ? 1 : 0;   // mode of the next launch
    const int seedScale = scale;
    const size_t regionBytes = ((regionPersistUnit * (size_t)seedScale) & ~(size_t)15) + ((sizeof(SavedRead) + 15) & ~(size_t)15);
    long long nRegions = 0;
    size_t regionsTotal = 0;         // bytes at the start of the scratch that hold saved reads (0: none alive)
    if (handOver) {
      idx->dRegionOf.ensure((size_t)nq);
      HIP_CHECK(hipMemsetAsync(idx->dRegionOf.p, 0xFF, sizeof(int32_t) * (size_t)nq, s));
    }
    // ---- passes 0 (XM_WAVE=1; off by default: measured slower than the lane-per-read passes on MI355X this round, profiles/r02/NOTES.md):
    // the wave-per-read form (xm_wave_kernel.hip).  Light tier over every read (seed, vote, ungapped alignment, accept); chain
    // tier over the reads that need the gapped chain (or more LDS): a read that meets a PathAligner search leaves the request in its memo, the
    // search kernel runs all waiting searches (one wavefront each), and those reads run again with the results, until none waits; then the
    // same with the largest capacities for the reads that outgrew the chain tier's.  What the wave form does not take (ambiguity codes in the
    // read or its reference window, mates longer than 256 bases, overlapping mates, a structure that outgrows LDS) goes through the
    // lane-per-read passes below.
    if (envInt("XM_WAVE", 0) != 0 && idx->residentMaxLen <= 256) {
      const bool tracePasses = envInt("XM_TRACE_PASSES", 0) != 0;
      idx->dListWaveHeavy.ensure((size_t)nq); idx->dListWaveNext.ensure((size_t)nq); idx->dListFallback.ensure((size_t)nq); idx->dWaveCtl.ensure(1);
      idx->dWaveSlotOf.ensure((size_t)nq);
      WaveCtl wctl{0, 0, 0, ~0ull};
      HIP_CHECK(hipMemcpyAsync(idx->dWaveCtl.p, &wctl, sizeof(wctl), hipMemcpyHostToDevice, s));
      OutView ov{idx->dOutInts.p, idx->dOutDbls.p, intCap, dblCap, idx->dCursors.p, idx->dStatus.p, idx->dIntOff.p, idx->dDblOff.p, idx->dIntLen.p, idx->dDblLen.p};
      const int lastTier = (int)envKnob("XM_WAVE_TIERS", 3, 1, 3) - 1;  // (experiment knob: 1 = light tier only, 2 = light + chain tier)
      int sWaves = 4, sLds = 1, sPerSimd = 4, memoBytes = 1, nodesPerWave = 1;
      xmSearchGeometry(&sWaves, &sLds, &sPerSimd, &memoBytes, &nodesPerWave);
      unsigned long long fallbackSoFar = 0;
      // one launch of a tier over `list` (null = all reads) + classification; returns the counts of the lists it filled
      auto launchTier = [&](int tier, const int64_t* list, long long n, int64_t* listNext, int32_t* slotOfOut, int64_t* listSearch) {
        WaveLaunch wl;
        wl.config = tier == 0 ? (idx->residentAnyPaired ? 1 : 0) : (tier == 1 ? (idx->residentAnyPaired ? 3 : 2) : 4);
        int wavesPerBlock = 1, ldsPerBlock = 1, wavesPerSimd = 1;
        xmWaveGeometry(wl.config, &wavesPerBlock, &ldsPerBlock, &wavesPerSimd);
        long long blocksPerCU = std::min<long long>((160 * 1024) / ldsPerBlock, (long long)(wavesPerSimd * 4) / wavesPerBlock);
        if (blocksPerCU < 1) blocksPerCU = 1;
        wl.itemsPerFetch = (int)envKnob(tier == 0 ? "XM_WAVE_FETCH" : "XM_WAVE_CHAIN_FETCH", tier == 0 ? 8 : 1, 1, 1024);
        long long blocks = std::min<long long>((long long)numCUs * blocksPerCU, (n + (long long)wavesPerBlock * wl.itemsPerFetch - 1) / ((long long)wavesPerBlock * wl.itemsPerFetch));
        if (blocks < 1) blocks = 1;
        wl.grid = (int)blocks; wl.block = wavesPerBlock * 64;
        wl.ix = view; wl.params = params; wl.batch = bv; wl.todo = list; wl.nTodo = n; wl.out = ov; wl.nextItem = idx->dCursors.p + 2; wl.counters = idx->dCounters.p;
        wl.memoBase = (WMemo*)idx->dWaveMemo.p; wl.slotOf = idx->dWaveSlotOf.p;
        wl.waveNodes = nullptr;
        if (tier >= 1 && envInt("XM_WAVE_INLINE_SEARCH", 1) != 0) {  // (0: every search through the memo and the search kernel)
          idx->dWaveNodes2.ensure(((size_t)blocks * wavesPerBlock * (size_t)xmWaveInlineNodeBytes() + sizeof(PNode) - 1) / sizeof(PNode));
          wl.waveNodes = idx->dWaveNodes2.p;
        }
        HIP_CHECK(hipMemsetAsync(idx->dCursors.p + 2, 0, sizeof(unsigned long long), s));
        HIP_CHECK(hipMemsetAsync(idx->dWaveCtl.p, 0, 2 * sizeof(unsigned long long), s));  // nNext, nSearch
        HIP_CHECK(hipEventRecord(e0, s));
        const int rc = xmWaveLaunch(wl, (void*)s);
        if (rc != 0) throw std::runtime_error(std::string("wave kernel launch: ") + hipGetErrorString((hipError_t)rc));
        HIP_CHECK(hipEventRecord(e1, s));
        hipLaunchKernelGGL(xm_wave_classify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, list, n, idx->dStatus.p, listNext, slotOfOut, listSearch, idx->dListFallback.p, idx->dWaveCtl.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(&wctl, idx->dWaveCtl.p, sizeof(wctl), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(cursors, idx->dCursors.p, sizeof(cursors), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        kernelMs += ms;
        res->counters[tier == 0 ? 12 : 13] += (int64_t)(ms * 1000.0);  // kernel microseconds: light tier / chain tiers
        launches++;
        if (tracePasses) fprintf(stderr, "[xm] wave tier %d config %d: reads %lld, %d x %d threads: %.3f ms -> next tier %llu, searches %llu, lane-per-read %llu (so far)\n", tier, wl.config, n, wl.grid,
                                 wl.block, ms, listNext ? wctl.nNext : 0ull, wctl.nSearch, wctl.nFallback);
        if (wctl.errQuery != ~0ull) {
          int32_t code = 0;
          HIP_CHECK(hipMemcpy(&code, idx->dStatus.p + wctl.errQuery, sizeof(code), hipMemcpyDeviceToHost));
          throw std::runtime_error("Failed to align query " + std::to_string(wctl.errQuery) + ": the reference implementation would have thrown here (status " + std::to_string(code & 0xFF) + ")");
        }
        fallbackSoFar = wctl.nFallback;
      };
      auto launchSearches = [&](const int64_t* list, long long n) {
        SearchLaunch sl;
        long long blocks = std::min<long long>((long long)numCUs * std::min<long long>((160 * 1024) / sLds, (long long)(sPerSimd * 4) / sWaves), (n + sWaves - 1) / sWaves);
        if (blocks < 1) blocks = 1;
        sl.grid = (int)blocks; sl.block = sWaves * 64;
        sl.ix = view; sl.params = params; sl.batch = bv; sl.list = list; sl.n = n; sl.memoBase = (WMemo*)idx->dWaveMemo.p; sl.slotOf = idx->dWaveSlotOf.p; sl.nextItem = idx->dCursors.p + 2;
        idx->dWaveArenas.ensure((size_t)blocks * sWaves * (size_t)nodesPerWave);  // (node payloads: bytes per wave)
        sl.waveNodes = idx->dWaveArenas.p; sl.counters = idx->dCounters.p;
        HIP_CHECK(hipMemsetAsync(idx->dCursors.p + 2, 0, sizeof(unsigned long long), s));
        HIP_CHECK(hipEventRecord(e0, s));
        const int rc = xmSearchLaunch(sl, (void*)s);
        if (rc != 0) throw std::runtime_error(std::string("search kernel launch: ") + hipGetErrorString((hipError_t)rc));
        HIP_CHECK(hipEventRecord(e1, s));
        HIP_CHECK(hipStreamSynchronize(s));
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        kernelMs += ms;
        res->counters[14] += (int64_t)(ms * 1000.0);  // search kernel microseconds
        launches++;
        if (tracePasses) fprintf(stderr, "[xm] search kernel: %lld searches, %d x %d threads: %.3f ms\n", n, sl.grid, sl.block, ms);
      };
      // light tier
      launchTier(0, nullptr, nq, lastTier >= 1 ? idx->dListWaveHeavy.p : (int64_t*)nullptr, idx->dWaveSlotOf.p, nullptr);
      long long nChain = lastTier >= 1 ? (long long)wctl.nNext : 0;
      if (nChain > 0) {
        idx->dWaveMemo.ensure((size_t)nChain * (size_t)memoBytes);
        if (xmMemoInitLaunch((WMemo*)idx->dWaveMemo.p, nChain, (void*)s) != 0) throw std::runtime_error("memo init launch failed");
        idx->dListWaveSearch[0].ensure((size_t)nChain); idx->dListWaveSearch[1].ensure((size_t)nChain);
        long long nBig = 0;  // reads for the chain tier with the largest capacities (dListWaveNext, filled behind what is already there)
        for (int tier = 1; tier <= 2 && tier <= lastTier; tier++) {
          const int64_t* list = tier == 1 ? idx->dListWaveHeavy.p : idx->dListWaveNext.p;
          long long n = tier == 1 ? nChain : nBig;
          int which = 0, rounds = 0;
          while (n > 0) {
            // (tier 1 appends its hand-overs to dListWaveNext behind those of its earlier rounds)
            launchTier(tier, list, n, tier == 1 && lastTier >= 2 ? idx->dListWaveNext.p + nBig : (int64_t*)nullptr, nullptr, idx->dListWaveSearch[which].p);
            if (tier == 1 && lastTier >= 2) nBig += (long long)wctl.nNext;
            const long long nSearch = (long long)wctl.nSearch;
            if (nSearch == 0) break;
            if (++rounds > 4 * 16) throw std::runtime_error("internal error: search rounds do not end");
            launchSearches(idx->dListWaveSearch[which].p, nSearch);
            list = idx->dListWaveSearch[which].p; n = nSearch;
            which ^= 1;
          }
        }
      }
      todo = idx->dListFallback.p;
      nTodo = (long long)fallbackSoFar;
      HIP_CHECK(hipMemsetAsync(idx->dCursors.p + 2, 0, sizeof(unsigned long long), s));
    }
    while (nTodo > 0) {
      std::unique_lock<std::mutex> sizing(idx->dt->allocMu);
      size_t arenaBytes = arenaUnit * (size_t)scale;  // bytes of scratch a lane owns in this launch
      if (hoMode == 1) arenaBytes = lightTmpUnit * (size_t)scale;                   // temporaries only (+ one region of the pool per lane / the read's own region)
      if (hoMode == 2) arenaBytes = regionBytes + gappedTmpBytes(arenaBytes);       // a region for reads without saved state + temporaries
      // launch shape (measured on MI355X, profiles/r01/NOTES.md): 8 waves per SIMD worth of lanes in the light pass; the gapped chain
      // diverges inside each wave, so it runs 32 reads per wave on 4 waves per SIMD.  The XM_* variables are experiment knobs.
      // a pass over few reads spreads them over all the wave slots of the GPU (the time of a launch is its longest wave)
      const long long waveSlots = (long long)numCUs * 4 * (heavy ? fullWaves : lightWaves);
      int lpw = (int)(heavy ? fullLpw : lightLpw);  // active lanes per wave
      if (heavy) lpw = (int)std::max(1ll, std::min((long long)lpw, (nTodo + waveSlots - 1) / waveSlots));
      long long lanes = waveSlots * lpw;
      const unsigned long long budget = scratchBudget();
      if (hoMode == 1) lanes = std::min(lanes, (long long)(budget / (arenaBytes + regionBytes)));
      else if (regionsTotal > 0) lanes = std::min(lanes, (long long)((idx->dArenas.n - regionsTotal) / arenaBytes));  // (sized below, before the pool was filled)
      else lanes = std::min(lanes, (long long)(budget / arenaBytes));
      if (lanes > nTodo) lanes = nTodo;
      // long reads, scratch for fewer lanes than asked for: fewer reads per wave before fewer waves than the GPU holds at a time (4 per SIMD) - a wave's
      // reads wait for each other's searches, an empty wave slot does nothing
      // (contexts that share the GPU share its wave slots)
      const long long slotsHeld = (long long)numCUs * 16 / std::max(1, gpuContexts);
      if (heavy && longReads && lpw > 1 && lanes / lpw < slotsHeld) lpw = (int)std::max(1ll, lanes / slotsHeld);
      long long nWaves = (lanes + lpw - 1) / lpw;
      if (nWaves < 1) nWaves = 1;
      if (hoMode != 1 && regionsTotal > 0) {  // the scratch cannot grow now: whole waves (and whole blocks of four) that fit behind the pool
        const long long cap = (long long)((idx->dArenas.n - regionsTotal) / arenaBytes);
        if (cap < 1) throw std::runtime_error("the scratch behind the saved reads is smaller than one lane's arena (XM_SCRATCH_GIB / XM_ARENA_KB too small for this batch)");
        if (lpw > cap) lpw = (int)cap;
        long long w = cap / lpw;
        if (w >= 4) w &= ~3ll;
        if ((nWaves >= 4 ? ((nWaves + 3) & ~3ll) : nWaves) > w) nWaves = w;
      }
      int block = nWaves < 4 ? (int)nWaves * 64 : 256;
      int grid = (int)((nWaves * 64 + block - 1) / block);
      lanes = (long long)grid * (block / 64) * lpw;
      if (hoMode == 1) {
        // pool: one region per lane + one per read that may stop (at most 40 % of the scratch; reads beyond that are seeded again by the
        // gapped pass).  The scratch is sized here for the gapped pass as well: it must not move while saved regions are alive.
        // (a lane takes a fresh region only before it fetches another read, and only nTodo - lanes reads are fetched by lanes that already had one)
        // (+ some slack: lanes that see a few reads left all take a region, but only some of them get a read)
        long long extra = nTodo > lanes ? (long long)nTodo - lanes + std::min(lanes, 4096ll) : 0;
        // (long reads: a fifth - their seeding is 4 % of their time, and a gapped-pass lane of theirs is 6.7 MB: the scratch is worth more as lanes)
        extra = std::min(extra, (long long)(budget * (longReads ? 1 : 2) / 5 / regionBytes) - lanes);
        extra = std::min(extra, ((long long)budget - lanes * (long long)(arenaBytes + regionBytes)) / (long long)regionBytes);
        if (extra < 0) extra = 0;
        nRegions = lanes + extra;
        regionsTotal = (size_t)nRegions * regionBytes;
        const size_t gappedArena = arenaUnit * (size_t)gappedScale, gappedLane = regionBytes + gappedTmpBytes(gappedArena);
        long long gappedLanes = std::min((long long)nq, (long long)numCUs * 4 * fullWaves * fullLpw);
        gappedLanes = std::min(gappedLanes, std::max(1ll, ((long long)budget - (long long)regionsTotal) / (long long)gappedLane));
        size_t behind = std::max((size_t)lanes * arenaBytes, (size_t)gappedLanes * gappedLane);
        behind = std::max(behind, gappedArena);  // (a rerun after a full result arena runs plain, at least one lane of it)
        if (!allocScratch(regionsTotal + behind + 1024)) { regionsTotal = 0; nRegions = 0; continue; }  // (sized again with half the budget)
        const unsigned long long firstFree = (unsigned long long)lanes;
        HIP_CHECK(hipMemcpyAsync(idx->dCursors.p + 3, &firstFree, sizeof(unsigned long long), hipMemcpyHostToDevice, s));
      } else if (regionsTotal > 0) {
        if (regionsTotal + (size_t)lanes * arenaBytes > idx->dArenas.n) throw std::runtime_error("internal error: scratch layout (hand-over)");
      } else {
        if (!allocScratch((size_t)lanes * arenaBytes)) continue;
      }
      sizing.unlock();
      // two lanes per read (xm_extend.h, xmSetPairMode); eight in the passes that run the rejection filter (8 reads per wave at most): its recurrence
      // spreads a column's cells over them (XM_GROUP_LANES=0: two there as well)
      int pairLanes = (heavy && lpw <= 32 && pairMode) ? 1 : 0;
      // the rejection filter in front of PathAligner's searches (xm_bound.h): the gapped passes of batches of long reads - their searches do not use the wave's
      // LDS slot, which the filter cuts into one region per read of the wave (8); reads that do not align spend 83 % of their search nodes in searches it proves null
      const int boundFilter = (heavy && boundFilterOn && lpw <= XM_BOUND_REGIONS && scale >= XM_HBM_ONLY_FROM) ? 1 : 0;
      if (boundFilter) boundFilterUsed = true;
      if (boundFilter && pairLanes && groupLanes) pairLanes = 3;
      const int boundFilterArg = boundFilter ? (1 | (envInt("XM_GROUP_SWEEP", 1) != 0 ? 2 : 0)) : 0;
      uint8_t* laneArenas = idx->dArenas.p + regionsTotal;
      HandOver ho{hoMode, seedScale, idx->dArenas.p, (unsigned long long)regionBytes, nRegions, idx->dRegionOf.p, idx->dCursors.p + 3};
      const int launchedMode = hoMode;
      idx->dWaveNodes.ensure((size_t)grid * (block / 64) * XM_PAL_NODES);
      SearchPool pool{nullptr, 0, 0, 0};
      if (searchPoolOn && heavy && scale == gappedScale) {
        pool.bufBytes = searchPoolBytes(makeCaps(scale));
        pool.n = (int32_t)((long long)grid * (block / 64));  // one per wave of the launch
        idx->dSearchPool.ensure((size_t)pool.n * pool.bufBytes);
        pool.base = idx->dSearchPool.p;
      }
      idx->dListScale[ts].ensure((size_t)nq); idx->dListOut[to].ensure((size_t)nq);
      // gapped pass with an ordered list: the first read of every lane is dealt out (kernel), the counter starts behind those items
      const long long firstStride = (heavy && orderedList && heavyHintThreshold > 0 && scale == gappedScale) ? (long long)grid * (block / 64) : 0;
      const unsigned long long firstItem = (unsigned long long)std::min((long long)nTodo, firstStride * lpw);
      HIP_CHECK(hipMemcpyAsync(idx->dCursors.p + 2, &firstItem, sizeof(unsigned long long), hipMemcpyHostToDevice, s));
      OutView ov{idx->dOutInts.p, idx->dOutDbls.p, intCap, dblCap, idx->dCursors.p, idx->dStatus.p, idx->dIntOff.p, idx->dDblOff.p, idx->dIntLen.p, idx->dDblLen.p};
      // the lanes file the reads they could not finish into the work lists of the passes to come as they publish them (PassLists; no kernel behind the pass)
      PassLists lists{idx->dListHeavy.p, idx->dListHeavyLate.p, idx->dListScale[ts].p, idx->dListOut[to].p, idx->dListConf[tc].p, (int)heavyHintThreshold, ts, to, tc, idx->dCtl.p};
      HIP_CHECK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(xm_align_kernel, dim3(grid), dim3(block), 0, s, view, params, bv, todo, nTodo, scale, heavy ? 2 : (int)lightLevel, lpw,
                         laneArenas, (unsigned long long)arenaBytes, ov, idx->dCursors.p + 2, idx->dCounters.p,
                         heavy ? (long long)((double)nWaves * taperWaves / 100.0) : 0ll, firstStride, idx->dWaveNodes.p, ho, pairLanes, pool, lists, boundFilterArg);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipEventRecord(e1, s));
      PassCtl ctl;
      HIP_CHECK(hipMemcpyAsync(&ctl, idx->dCtl.p, sizeof(ctl), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipMemcpyAsync(cursors, idx->dCursors.p, sizeof(cursors), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      kernelMs += ms;
      res->counters[!heavy ? 12 : 15] += (int64_t)(ms * 1000.0);  // kernel microseconds: light pass / gapped pass and reruns
      launches++;
      orderedList = false;
      hoMode = 0;                                // (the gapped pass below switches to 2; reruns run plain)
      if (launchedMode == 2) regionsTotal = 0;   // the saved reads have all been consumed
#ifdef XM_LIGHT_ONLY
      fprintf(stderr, "[xm] light-only experiment build: pass %d %.3f ms\n", launches, ms);
      break;  // (experiment build, scripts/gpu_light_only.sh: only the first pass is meaningful)
#endif
      const bool tracePasses = envInt("XM_TRACE_PASSES", 0) != 0;
      if (tracePasses) fprintf(stderr, "[xm] pass %d: %s reads %lld scale %d lpw %d waves %lld: %.3f ms -> heavy %llu scale %llu out %llu\n", launches, !heavy ? "light" : "gapped",
                               nTodo, scale, lpw, nWaves, ms, ctl.nHeavy + ctl.nHeavyLate, ctl.nScale[ts], ctl.nOut[to]);
#ifdef XM_PROFILE
      if (tracePasses && heavy) {  // reads of a wave that stood at a PathAligner call together, this pass
        unsigned long long a[16] = {0}, z[16] = {0};
        HIP_CHECK(hipMemcpyFromSymbol(a, HIP_SYMBOL(xm_arrive_prof), sizeof(a)));
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(xm_arrive_prof), z, sizeof(z)));
        fprintf(stderr, "[xm] pass %d: pair checks (status, result, search problem, search outcome): %llu %llu %llu %llu\n", launches, a[4], a[5], a[6], a[7]);
        fprintf(stderr, "[xm] pass %d: PathAligner arrivals %llu with %llu reads (%.2f per arrival); arrivals of four reads or more: %llu with %llu reads\n", launches, a[0], a[1], a[0] ? (double)a[1] / (double)a[0] : 0.0, a[2], a[3]);
      }
#endif
      if (ctl.errQuery != ~0ull) {
        int32_t code = 0;
        HIP_CHECK(hipMemcpy(&code, idx->dStatus.p + ctl.errQuery, sizeof(code), hipMemcpyDeviceToHost));
        code &= 0xFF;
        std::string q = std::to_string(ctl.errQuery);
        if (code == XM_ST_NEED_GROW) throw std::runtime_error("Failed to align query " + q + ": gapmer longer than the hashed lengths");
        throw std::runtime_error("Failed to align query " + q + ": the reference implementation would have thrown here (status " + std::to_string(code) + ")");
      }
      pendingHeavy = ctl.nHeavy + ctl.nHeavyLate;
      pendingScale = ctl.nScale[ts];
      pendingConf = ctl.nConf[tc];   // (accumulates over the passes until the list is run)
      if (ctl.nOut[to] > 0) {  // result arena too small: rerun those reads with the same settings and room to spare
        todo = idx->dListOut[to].p; nTodo = (long long)ctl.nOut[to];
        to ^= 1;
        HIP_CHECK(hipMemsetAsync(&idx->dCtl.p->nOut[to], 0, sizeof(unsigned long long), s));
        unsigned long long keepI = std::min(cursors[0], intCap), keepD = std::min(cursors[1], dblCap);
        unsigned long long newI = std::max(intCap * 4 + 65536, cursors[0] * 2), newD = std::max(dblCap * 4 + 65536, cursors[1] * 2);
        idx->dOutInts.growKeep((size_t)newI, (size_t)keepI, s); idx->dOutDbls.growKeep((size_t)newD, (size_t)keepD, s);
        intCap = idx->dOutInts.n; dblCap = idx->dOutDbls.n;
        rerun += nTodo;
        continue;
      }
      if (pendingHeavy > 0) {
        // the gapped pass runs at scale 4 straight away: far fewer lanes are needed than in the light pass, and most reads whose
        // gapped search outgrows the scale-1 scratch then finish here instead of costing one more (latency-bound) pass
        if (ctl.nHeavyLate > 0) {  // one list: the expensive-looking reads first, the others behind them
          HIP_CHECK(hipMemcpyAsync(idx->dListHeavy.p + ctl.nHeavy, idx->dListHeavyLate.p, sizeof(int64_t) * (size_t)ctl.nHeavyLate, hipMemcpyDeviceToDevice, s));
        }
        todo = idx->dListHeavy.p; nTodo = (long long)pendingHeavy;
        orderedList = true;
        HIP_CHECK(hipMemsetAsync(&idx->dCtl.p->nHeavy, 0, 2 * sizeof(unsigned long long), s));  // nHeavy, nHeavyLate (a gapped pass never adds to these lists)
        scale = gappedScale;
        if (overflowScale < gappedScale) overflowScale = gappedScale;
        heavy = true;
        if (regionsTotal > 0) hoMode = 2;
        if (envInt("XM_PROF_GAPPED_ONLY", 0) != 0)  // XM_PROFILE builds: the in-kernel timers of the gapped pass alone
          HIP_CHECK(hipMemsetAsync((char*)idx->dCounters.p + offsetof(DevCounters, t), 0, sizeof(((DevCounters*)nullptr)->t), s));
        continue;
      }
      if (pendingScale == 0 && pendingConf > 0) {
        // reads that met a (penalty, length) the confidence table did not hold: the host evaluates the keys they left (its libm, the oracle's)
        // and they run again, start to finish, in a pass of their own
        if (++confRounds > 1024) throw std::runtime_error("internal error: the confidence table does not converge");
        idx->confAbsorbMisses(params, s);
        view.conf = idx->dConf.p; view.confMask = (uint32_t)(idx->confHost.size() - 1);
        todo = idx->dListConf[tc].p; nTodo = (long long)pendingConf;
        tc ^= 1;
        HIP_CHECK(hipMemsetAsync(&idx->dCtl.p->nConf[tc], 0, sizeof(unsigned long long), s));
        rerun += nTodo;
        regionsTotal = 0;
        if (scale < gappedScale) scale = gappedScale;
        if (overflowScale < scale) overflowScale = scale;
        heavy = true;
        continue;
      }
      if (pendingScale == 0) break;
      todo = idx->dListScale[ts].p; nTodo = (long long)pendingScale;
      ts ^= 1;
      HIP_CHECK(hipMemsetAsync(&idx->dCtl.p->nScale[ts], 0, sizeof(unsigned long long), s));
      rerun += nTodo;
      regionsTotal = 0;  // (no gapped pass ran: whatever the light pass saved is not wanted any more)
      overflowScale *= 4;
      scale = overflowScale;
      heavy = true;
      if (scale > 4096) throw std::runtime_error("Failed to align: scratch scale limit reached (query needs more than 4096x the default scratch)");
    }
    // ---- canonical streams in query order: offsets by prefix sum, slices gathered on the device, one copy per stream to the host
    HIP_CHECK(hipEventRecord(e0, s));
    const long long nBlocks = (nq + XM_SCAN_PER_BLOCK - 1) / XM_SCAN_PER_BLOCK;
    idx->dBlockI.ensure((size_t)nBlocks); idx->dBlockD.ensure((size_t)nBlocks);
#ifdef XM_READ_TIMES
    if (dReadTimes.p) {
      std::vector<unsigned long long> t((size_t)nq);
      HIP_CHECK(hipMemcpy(t.data(), dReadTimes.p, sizeof(unsigned long long) * (size_t)nq, hipMemcpyDeviceToHost));
      unsigned long long* none = nullptr;
      HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(xm_read_times), &none, sizeof(none)));
      if (FILE* f = fopen(readTimesFile, "wb")) { fwrite(t.data(), sizeof(unsigned long long), t.size(), f); fclose(f); }
      dReadTimes.release();
    }
#endif
    idx->dFinalIntOff.ensure((size_t)nq + 1); idx->dFinalDblOff.ensure((size_t)nq + 1);
    const size_t usedI = (size_t)std::min(cursors[0], intCap), usedD = (size_t)std::min(cursors[1], dblCap);  // upper bounds of the totals
    idx->dFinalInts.ensure(usedI); idx->dFinalDbls.ensure(usedD);
    hipLaunchKernelGGL(xm_scan_totals_kernel, dim3((unsigned)nBlocks), dim3(256), 0, s, (long long)nq, idx->dIntLen.p, idx->dDblLen.p, idx->dBlockI.p, idx->dBlockD.p);
    hipLaunchKernelGGL(xm_scan_blocks_kernel, dim3(1), dim3(64), 0, s, nBlocks, (long long)nq, idx->dBlockI.p, idx->dBlockD.p, idx->dFinalIntOff.p, idx->dFinalDblOff.p);
    hipLaunchKernelGGL(xm_scan_final_kernel, dim3((unsigned)nBlocks), dim3(256), 0, s, (long long)nq, idx->dIntLen.p, idx->dDblLen.p, idx->dBlockI.p, idx->dBlockD.p,
                       idx->dFinalIntOff.p, idx->dFinalDblOff.p);
    hipLaunchKernelGGL(xm_gather_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, (long long)nq, idx->dIntOff.p, idx->dDblOff.p, idx->dIntLen.p, idx->dDblLen.p,
                       idx->dFinalIntOff.p, idx->dFinalDblOff.p, idx->dOutInts.p, idx->dOutDbls.p, idx->dFinalInts.p, idx->dFinalDbls.p);
    HIP_CHECK(hipGetLastError());
    res->ints = (int32_t*)g_pinned->get(sizeof(int32_t) * (usedI ? usedI : 1), &box->bytesInts);
    res->dbls = (double*)g_pinned->get(sizeof(double) * (usedD ? usedD : 1), &box->bytesDbls);
    idx->lastAlignedNq = nq;
    idx->lastAlignedGen = idx->residentGen;
    HIP_CHECK(hipMemcpyAsync(res->int_off, idx->dFinalIntOff.p, sizeof(int64_t) * (size_t)(nq + 1), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(res->dbl_off, idx->dFinalDblOff.p, sizeof(int64_t) * (size_t)(nq + 1), hipMemcpyDeviceToHost, s));
    if (usedI) HIP_CHECK(hipMemcpyAsync(res->ints, idx->dFinalInts.p, sizeof(int32_t) * usedI, hipMemcpyDeviceToHost, s));
    if (usedD) HIP_CHECK(hipMemcpyAsync(res->dbls, idx->dFinalDbls.p, sizeof(double) * usedD, hipMemcpyDeviceToHost, s));
    DevCounters dc;
    HIP_CHECK(hipMemcpyAsync(&dc, idx->dCounters.p, sizeof(dc), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipEventRecord(e1, s));
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    res->d2h_ms = ms;
    res->num_ints = res->int_off[nq]; res->num_dbls = res->dbl_off[nq];
    res->counters[0] = (int64_t)dc.reads; res->counters[1] = (int64_t)dc.headerProbes; res->counters[2] = (int64_t)dc.bucketFetches; res->counters[3] = (int64_t)dc.hitsFetched;
    res->counters[4] = (int64_t)dc.candidatesExtended; res->counters[5] = (int64_t)dc.pathAlignerCalls; res->counters[6] = (int64_t)dc.pathAlignerNodes;
    res->counters[7] = (int64_t)dc.quickAccepts; res->counters[8] = (int64_t)dc.alignmentsOut; res->counters[9] = (int64_t)dc.refWindowBytes; res->counters[10] = (int64_t)dc.readBytes;
    res->counters[11] = rerun;
    res->extra[0] = (int64_t)dc.boundChecks; res->extra[1] = (int64_t)dc.boundRejects; res->extra[2] = (int64_t)dc.boundCells; res->extra[3] = boundFilterUsed ? 1 : 0; res->extra[4] = (int64_t)dc.boundPieceChecks; res->extra[5] = (int64_t)dc.boundPieceRejects;
    for (int i = 0; i < 16; i++) res->prof[i] = (int64_t)dc.t[i];
    res->kernel_ms = kernelMs;
    res->kernel_launches = launches;
    *out = res;
    return 0;
  } catch (...) {
    xm_result_free(res);
    throw;
  }
}

int xm_seed_probe_packed(xm_index* idx, int64_t n, const int32_t* usedLength, const int32_t* keys, int32_t maxPerProbe, int32_t* counts, int64_t* outPositions, double* kernelMs) {
  if (!idx || idx->hostOnly) return fail("xm_seed_probe_packed: needs a device-resident index");
  if (maxPerProbe < 0 || maxPerProbe > 15) return fail("xm_seed_probe_packed: max_per_probe must be 0 ... 15");
  try {
    std::lock_guard<std::mutex> lock(idx->mu);
    HIP_CHECK(hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    DevBuf<int32_t> dUsed, dKeys, dCounts;
    DevBuf<int64_t> dPos;
    struct Release { DevBuf<int32_t>&a, &b, &c; DevBuf<int64_t>& d; ~Release() { a.release(); b.release(); c.release(); d.release(); } } releaseAll{dUsed, dKeys, dCounts, dPos};
    dUsed.ensure((size_t)n); dKeys.ensure((size_t)n); dCounts.ensure((size_t)n); dPos.ensure((size_t)n * (size_t)(maxPerProbe > 0 ? maxPerProbe : 1));
    HIP_CHECK(hipMemcpyAsync(dUsed.p, usedLength, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(dKeys.p, keys, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s));
    int block = 256;
    int grid = (int)((n + block - 1) / block);
    HIP_CHECK(hipEventRecord(idx->ev0, s));
    std::shared_lock<std::shared_mutex> tablesInUse(idx->dt->rw);
    IndexView view = idx->dt->view;
    if (envInt("XM_PROBE_NO_LINES", 0) != 0) { view.lines32 = nullptr; view.lines64 = nullptr; }  // measurement: the CSR probe (two dependent accesses) on the same index
    const unsigned batched = (unsigned)((n + (long long)block * XM_PROBES_PER_LANE - 1) / ((long long)block * XM_PROBES_PER_LANE));
    if (n > 0 && view.lines64) hipLaunchKernelGGL((xm_seed_probe_lines_kernel<true>), dim3(batched), dim3(block), 0, s, view, (long long)n, dUsed.p, dKeys.p, (int)maxPerProbe, dCounts.p, dPos.p);
    else if (n > 0 && view.lines32) hipLaunchKernelGGL((xm_seed_probe_lines_kernel<false>), dim3(batched), dim3(block), 0, s, view, (long long)n, dUsed.p, dKeys.p, (int)maxPerProbe, dCounts.p, dPos.p);
    else if (n > 0) hipLaunchKernelGGL(xm_seed_probe_kernel, dim3(grid), dim3(block), 0, s, view, (long long)n, dUsed.p, dKeys.p, (int)maxPerProbe, dCounts.p, dPos.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(idx->ev1, s));
    HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, idx->ev0, idx->ev1));
    if (kernelMs) *kernelMs = ms;
    HIP_CHECK(hipMemcpy(counts, dCounts.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    if (outPositions && maxPerProbe > 0) HIP_CHECK(hipMemcpy(outPositions, dPos.p, sizeof(int64_t) * (size_t)n * (size_t)maxPerProbe, hipMemcpyDeviceToHost));
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_seed_probe_packed: ") + e.what()); }
}

int xm_measure_random_gather(int device, int64_t table_bytes, int64_t accesses, double* kernel_ms) {
  try {
    HIP_CHECK(hipSetDevice(device));
    if (table_bytes < 4096 || accesses < 1) return fail("xm_measure_random_gather: bad arguments");
    DevBuf<uint4> table;
    DevBuf<unsigned int> sink;
    struct Release { DevBuf<uint4>& a; DevBuf<unsigned int>& b; ~Release() { a.release(); b.release(); } } releaseAll{table, sink};
    const size_t nSectors = (size_t)table_bytes / 64;
    table.ensure(nSectors * 4);
    sink.ensure(1);
    HIP_CHECK(hipMemset(table.p, 0, nSectors * 64));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    const int perThread = 4;
    const long long threads = (accesses + perThread - 1) / perThread;
    float best = 0;
    for (int rep = 0; rep < 3; rep++) {  // first repetition warms up
      HIP_CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(xm_random_gather_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, table.p, (unsigned long long)nSectors, (long long)accesses, perThread,
                         0x5EED0000ull + rep, sink.p);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipEventRecord(e1, 0));
      HIP_CHECK(hipEventSynchronize(e1));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && (best == 0 || ms < best)) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (kernel_ms) *kernel_ms = best;
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_measure_random_gather: ") + e.what()); }
}

int xm_pileup_new(xm_index* idx, xm_pileup** out) {
  if (!idx || !out) return fail("xm_pileup_new: null argument");
  if (idx->hostOnly) return fail("xm_pileup_new: index was built with host_only=1");
  xm_pileup* p = nullptr;
  try {
    std::lock_guard<std::mutex> lock(idx->mu);
    HIP_CHECK(hipSetDevice(idx->device));
    p = new xm_pileup();
    p->index = idx;
    p->hs = idx->hs;
    p->device = idx->device;
    p->total = idx->host().totalForwardSize;
    p->dDepth.ensure((size_t)p->total); p->dAlt.ensure((size_t)p->total * 4); p->dEventCount.ensure(1);
    HIP_CHECK(hipMemset(p->dDepth.p, 0, sizeof(unsigned long long) * (size_t)p->total));
    HIP_CHECK(hipMemset(p->dAlt.p, 0, sizeof(unsigned long long) * (size_t)p->total * 4));
    *out = p;
    return 0;
  } catch (std::exception& e) {
    if (p) { p->dDepth.release(); p->dAlt.release(); p->dEventCount.release(); delete p; }
    return fail(std::string("xm_pileup_new: ") + e.what());
  }
}

int xm_pileup_set_query_ends(xm_pileup* p, double fraction) {
  if (!p) return fail("xm_pileup_set_query_ends: null argument");
  if (!(fraction >= 0 && fraction < 1)) return fail("--distinguish-query-ends must be >= 0 and < 1");  // Mapper.java:424-425
  if (p->queriesAdded > 0) return fail("xm_pileup_set_query_ends: alignments were already added");
  try {
    HIP_CHECK(hipSetDevice(p->device));
    p->endFraction = fraction;
    if (fraction > 0) {
      p->dMid.ensure((size_t)p->total);
      HIP_CHECK(hipMemset(p->dMid.p, 0, sizeof(unsigned long long) * (size_t)p->total));
    }
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_pileup_set_query_ends: ") + e.what()); }
}

int xm_pileup_read_middle(xm_pileup* p, int32_t contig, int64_t first, int64_t n, uint64_t* depth) {
  if (!p || !p->hs || !depth) return fail("xm_pileup_read_middle: null argument");
  try {
    const HostIndex& host = p->hs->host;
    if (contig < 0 || contig >= host.numContigs() || first < 0 || n < 0 || first + n > host.contigLen[(size_t)contig]) throw std::runtime_error("range outside of the contig");
    HIP_CHECK(hipSetDevice(p->device));
    HIP_CHECK(hipDeviceSynchronize());
    const size_t at = (size_t)host.contigStart[(size_t)contig] + (size_t)first;
    // (no query-end fraction: every base is a middle base)
    if (n) HIP_CHECK(hipMemcpy(depth, (p->endFraction > 0 ? p->dMid.p : p->dDepth.p) + at, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost));
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_pileup_read_middle: ") + e.what()); }
}

int xm_pileup_add_last(xm_pileup* p, int64_t* num_events) {
  if (!p || !p->index) return fail("xm_pileup_add_last: null argument");
  xm_index* idx = p->index;
  try {
    std::lock_guard<std::mutex> lock(idx->mu);
    if (idx->lastAlignedNq < 0 || idx->lastAlignedNq != idx->residentNq || idx->lastAlignedGen != idx->residentGen)
      throw std::runtime_error("the batch of the last align call is no longer resident (call xm_pileup_add_last after xm_align_batch / xm_align_resident, before the next batch is uploaded or committed)");
    HIP_CHECK(hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    const int64_t nq = idx->lastAlignedNq;
    if (nq > 0) {
      const unsigned long long cap = (unsigned long long)idx->dFinalInts.n / 4 + 1;  // (an event is a block: at least four ints of the stream)
      p->dEvents.ensure((size_t)cap * 8);
      HIP_CHECK(hipMemsetAsync(p->dEventCount.p, 0, sizeof(unsigned long long), s));
      BatchView bv{nq, idx->dMateCount.p, idx->dMateOffset.p, idx->dMateLength.p, idx->dCodes.p, idx->dExpected.p, idx->dDeviation.p};
      PileupView pv{p->dDepth.p, p->dAlt.p, p->total, p->dEvents.p, cap, p->dEventCount.p, p->queriesAdded, p->endFraction > 0 ? p->dMid.p : nullptr, p->endFraction};
      std::shared_lock<std::shared_mutex> tablesInUse(idx->dt->rw);
      hipLaunchKernelGGL(xm_pileup_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, idx->dt->view, bv, (const int32_t*)idx->dFinalInts.p, (const int64_t*)idx->dFinalIntOff.p, pv);
      HIP_CHECK(hipGetLastError());
      unsigned long long n = 0;
      HIP_CHECK(hipMemcpyAsync(&n, p->dEventCount.p, sizeof(n), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      if (n > cap) throw std::runtime_error("internal error: more indel events than blocks");
      const size_t at = p->events.size();
      p->events.resize(at + (size_t)n * 8);
      if (n) HIP_CHECK(hipMemcpy(p->events.data() + at, p->dEvents.p, sizeof(long long) * (size_t)n * 8, hipMemcpyDeviceToHost));
      // the order of the atomic appends is not fixed: events of a call are put in (query, position) order
      std::vector<std::array<long long, 8>> ev((size_t)n);
      for (size_t i = 0; i < (size_t)n; i++) for (int k = 0; k < 8; k++) ev[i][(size_t)k] = p->events[at + i * 8 + (size_t)k];
      std::sort(ev.begin(), ev.end(), [](const std::array<long long, 8>& a, const std::array<long long, 8>& b) {
        if (a[4] != b[4]) return a[4] < b[4];
        if (a[0] != b[0]) return a[0] < b[0];
        if (a[1] != b[1]) return a[1] < b[1];
        if (a[5] != b[5]) return a[5] < b[5];
        return a[6] < b[6];
      });
      for (size_t i = 0; i < (size_t)n; i++) for (int k = 0; k < 8; k++) p->events[at + i * 8 + (size_t)k] = ev[i][(size_t)k];
    }
    p->queriesAdded += nq;
    if (num_events) *num_events = (int64_t)(p->events.size() / 8);
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_pileup_add_last: ") + e.what()); }
}

int xm_pileup_read(xm_pileup* p, int32_t contig, int64_t first, int64_t n, uint64_t* depth, uint64_t* alt) {
  if (!p || !p->hs || !depth || !alt) return fail("xm_pileup_read: null argument");
  try {
    const HostIndex& host = p->hs->host;  // (the pile-up shares the reference with its index: it outlives the context it was made from)
    if (contig < 0 || contig >= host.numContigs() || first < 0 || n < 0 || first + n > host.contigLen[(size_t)contig]) throw std::runtime_error("range outside of the contig");
    HIP_CHECK(hipSetDevice(p->device));
    HIP_CHECK(hipDeviceSynchronize());  // (adds of a context's stream that may still be running)
    const size_t at = (size_t)host.contigStart[(size_t)contig] + (size_t)first;
    if (n) {
      HIP_CHECK(hipMemcpy(depth, p->dDepth.p + at, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost));
      for (int b = 0; b < 4; b++) HIP_CHECK(hipMemcpy(alt + (size_t)b * (size_t)n, p->dAlt.p + (size_t)b * (size_t)p->total + at, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost));
    }
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_pileup_read: ") + e.what()); }
}

int64_t xm_pileup_events(xm_pileup* p, int64_t first, int64_t n, int64_t* out) {
  if (!p || (n > 0 && !out)) return -1;
  const int64_t have = (int64_t)(p->events.size() / 8);
  if (first < 0 || first > have) return -1;
  const int64_t m = std::min<int64_t>(n, have - first);
  if (m > 0) memcpy(out, p->events.data() + (size_t)first * 8, sizeof(int64_t) * (size_t)m * 8);
  return m;
}

void xm_pileup_free(xm_pileup* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  p->dDepth.release(); p->dAlt.release(); p->dEventCount.release(); p->dEvents.release(); p->dMid.release();
  delete p;
}

// Test-only: what the rejection filter did in this thread's last xm_test_local_align call (searches taken, searches rejected, cells computed).
static thread_local int64_t g_testBound[3] = {0, 0, 0};
void xm_test_bound_counters(int64_t* out3) { for (int i = 0; i < 3; i++) out3[i] = g_testBound[i]; }

// Test-only entry (tests/test_gpu_bound.py): the rejection filter alone on one problem (xm_test_bound_kernel).  out3: taken, rejected, cells computed.
int xm_test_bound(int32_t device, const xm_params* p, const uint8_t* query, int32_t query_length, int32_t query_rc, int32_t start_a, int32_t end_a, const uint8_t* reference, int32_t reference_length,
                  int32_t start_b, int32_t end_b, int32_t predicted_best_offset, int32_t pair, int64_t* out3) {
  if (!p || !query || !reference || !out3) return fail("xm_test_bound: null argument");
  if (query_length < 1 || reference_length < 1 || start_a < 0 || end_a > query_length || start_a > end_a || start_b < 0 || end_b > reference_length || start_b > end_b) return fail("xm_test_bound: bad sections");
  try {
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    Params params;
    memset(&params, 0, sizeof(params));
    params.MutationPenalty = p->MutationPenalty; params.InsertionStart_Penalty = p->InsertionStart_Penalty; params.InsertionExtension_Penalty = p->InsertionExtension_Penalty;
    params.DeletionStart_Penalty = p->DeletionStart_Penalty; params.DeletionExtension_Penalty = p->DeletionExtension_Penalty; params.MaxErrorRate = p->MaxErrorRate;
    params.UnalignedPenalty = p->UnalignedPenalty; params.AmbiguityPenalty = p->AmbiguityPenalty; params.Max_PenaltySpan = p->Max_PenaltySpan;
    params.MaxNumMatches = p->MaxNumMatches; params.StartingInsertionStartFree = 0;
    DevBuf<uint8_t> dq, dr, dArena;
    DevBuf<int64_t> dOut;
    struct Release { DevBuf<uint8_t>&a, &b, &d; DevBuf<int64_t>& c; ~Release() { a.release(); b.release(); c.release(); d.release(); } } releaseAll{dq, dr, dArena, dOut};
    dq.ensure((size_t)query_length); dr.ensure((size_t)reference_length); dOut.ensure(4); dArena.ensure(64 * 1024);
    HIP_CHECK(hipMemcpy(dq.p, query, (size_t)query_length, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dr.p, reference, (size_t)reference_length, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(dOut.p, 0, sizeof(int64_t) * 4));
    hipLaunchKernelGGL(xm_test_bound_kernel, dim3(1), dim3(256), 0, 0, params, (const uint8_t*)dq.p, (int)query_length, (int)query_rc, (int)start_a, (int)end_a, (const uint8_t*)dr.p, (int)reference_length,
                       (int)start_b, (int)end_b, (int)predicted_best_offset, (int)(pair == 3 ? 3 : (pair ? 1 : 0)), dArena.p, (unsigned long long)(64 * 1024), dOut.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out3, dOut.p, sizeof(int64_t) * 3, hipMemcpyDeviceToHost));
    return 0;
  } catch (std::exception& e) { return fail(std::string("xm_test_bound: ") + e.what()); }
}

// Test-only entry (tests/test_gpu_kat.py): see xm_test_local_kernel above and xm_test_wave_search_kernel (xm_wave_kernel.hip).
int xm_test_local_align(int32_t device, int32_t chain, int32_t mode, const xm_params* p, const uint8_t* query, int32_t query_length, const uint8_t* reference, int32_t reference_length,
                        double max_ins_ext, double max_del_ext, int32_t block_cap, int32_t* blocks, int32_t* num_blocks, double* penalties, int64_t* nodes_put) {
  if (!p || !query || !reference || !blocks || !num_blocks || !penalties) { fail("xm_test_local_align: null argument"); return -1; }
  const bool withBound = mode >= 8;  // mode + 8 (modes 0, 1, 4): the search behind the rejection filter of xm_bound.h; xm_test_bound_counters() says what it did
  if (withBound) mode -= 8;
  if (chain < 0 || chain > 1 || mode < 0 || mode > 4 || (withBound && (mode == 2 || mode == 3)) || (chain == 1 && (mode == 2 || mode == 3)) || query_length < 1 || reference_length < 1 || query_length > 30000 || reference_length > 100000 || block_cap < 1)
  { fail("xm_test_local_align: bad arguments (chain 0: modes 0 LDS slot, 1 HBM, 2 wave search with the search kernel's capacities, 3 with the inline capacities, 4 lane-private form; chain 1: modes 0, 1, 4)"); return -1; }
  try {
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
    Params params;
    memset(&params, 0, sizeof(params));
    params.MutationPenalty = p->MutationPenalty; params.InsertionStart_Penalty = p->InsertionStart_Penalty; params.InsertionExtension_Penalty = p->InsertionExtension_Penalty;
    params.DeletionStart_Penalty = p->DeletionStart_Penalty; params.DeletionExtension_Penalty = p->DeletionExtension_Penalty; params.MaxErrorRate = p->MaxErrorRate;
    params.UnalignedPenalty = p->UnalignedPenalty; params.AmbiguityPenalty = p->AmbiguityPenalty; params.Max_PenaltySpan = p->Max_PenaltySpan;
    params.MaxNumMatches = p->MaxNumMatches; params.StartingInsertionStartFree = 0;
    const int cap = block_cap < 256 ? block_cap : 256;
    DevBuf<uint8_t> dq, dr, arena, nodes;
    DevBuf<int32_t> dInts;
    DevBuf<double> dDbls;
    DevBuf<int64_t> dStart;
    DevBuf<int32_t> dLen;
    struct Release {  // (the buffers of this call are released on every way out)
      DevBuf<uint8_t>&a, &b, &c, &d; DevBuf<int32_t>&e; DevBuf<double>& f; DevBuf<int64_t>& g; DevBuf<int32_t>& h;
      ~Release() { a.release(); b.release(); c.release(); d.release(); e.release(); f.release(); g.release(); h.release(); }
    } releaseAll{dq, dr, arena, nodes, dInts, dDbls, dStart, dLen};
    dq.ensure((size_t)query_length); dr.ensure((size_t)reference_length); dInts.ensure((size_t)8 + 4 * (size_t)cap); dDbls.ensure(2);
    HIP_CHECK(hipMemcpy(dq.p, query, (size_t)query_length, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dr.p, reference, (size_t)reference_length, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(dInts.p, 0, sizeof(int32_t) * (8 + 4 * (size_t)cap)));
    HIP_CHECK(hipMemset(dDbls.p, 0, sizeof(double) * 2));
    if (mode == 2 || mode == 3) {
      TestSearch t;
      memset(&t, 0, sizeof(t));
      t.big = mode == 2 ? 1 : 0;
      const int64_t start0 = 0;
      const int32_t len0 = reference_length;
      dStart.ensure(1); dLen.ensure(1);
      HIP_CHECK(hipMemcpy(dStart.p, &start0, 8, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(dLen.p, &len0, 4, hipMemcpyHostToDevice));
      t.ix.numContigs = 1; t.ix.refCodes = dr.p; t.ix.contigStart = dStart.p; t.ix.contigLen = dLen.p;
      t.params = params; t.query = dq.p; t.queryLength = query_length; t.referenceLength = reference_length; t.predictedBestOffset = 0; t.confident = 0; t.blockCap = cap;
      t.maxIns = max_ins_ext; t.maxDel = max_del_ext;
      nodes.ensure((size_t)xmTestWaveSearchNodeBytes(t.big));
      t.nodes = nodes.p; t.outInts = dInts.p; t.outDbls = dDbls.p;
      const int rc = xmTestWaveSearchLaunch(t, 0);
      if (rc != 0) throw std::runtime_error(std::string("test search launch: ") + hipGetErrorString((hipError_t)rc));
    } else {
      const int scale = 4;
      const size_t arenaBytes = (size_t)288 * 1024 * scale;
      arena.ensure(arenaBytes);
      nodes.ensure((size_t)XM_PAL_NODES * 4 * sizeof(PNode));
      hipLaunchKernelGGL(xm_test_local_kernel, dim3(1), dim3(256), 0, 0, (int)chain, (int)mode + (withBound ? 8 : 0), params, (const uint8_t*)dq.p, (int)query_length, (const uint8_t*)dr.p, (int)reference_length,
                         max_ins_ext, max_del_ext, scale, arena.p, (unsigned long long)arenaBytes, (PNode*)nodes.p, cap, dInts.p, dDbls.p);
      HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<int32_t> ints((size_t)8 + 4 * (size_t)cap);
    double dbls[2];
    HIP_CHECK(hipMemcpy(ints.data(), dInts.p, sizeof(int32_t) * ints.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; i++) g_testBound[i] = ints[(size_t)4 + 4 * (size_t)cap + (size_t)i];
    HIP_CHECK(hipMemcpy(dbls, dDbls.p, sizeof(dbls), hipMemcpyDeviceToHost));
    if (nodes_put) *nodes_put = ints[3];
    const int ok = (mode == 2 || mode == 3) ? ints[0] : (ints[2] != XM_OK ? -1 : ints[0]);
    if (ok < 0) { fail("xm_test_local_align: the search failed with status " + std::to_string(ints[2])); return -1; }
    if (ok == 0) { *num_blocks = 0; return 1; }
    if (ints[1] > cap) { fail("xm_test_local_align: more blocks than block_cap"); return -1; }
    *num_blocks = ints[1];
    memcpy(blocks, ints.data() + 4, sizeof(int32_t) * 4 * (size_t)ints[1]);
    penalties[0] = dbls[0]; penalties[1] = dbls[1];
    return 0;
  } catch (std::exception& e) { fail(std::string("xm_test_local_align: ") + e.what()); return -1; }
}

}  // extern "C"
