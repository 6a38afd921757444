// xmapper-hip device core: shared definitions.
//
// The per-read seed-and-extend path of X-Mapper (reference: /root/reference/src/main/java/mapper/, cited as M/)
// written for gfx950.  All code in csrc/xm_*.h is plain C-style C++ with fixed-capacity arrays carved out of a
// per-read scratch arena in HBM, no recursion, no host library calls; fp64 in the reference's evaluation order
// (build with -ffp-contract=off).  The same sources compile for the host (XM_HD empty) ONLY for the CPU-side
// simulation harness under tests/hostsim, which exists so the kernel logic can be checked where there is no GPU;
// libxmapper_hip.so never contains or calls a host build of this code.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <stdlib.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define XM_HD __host__ __device__
#if defined(__HIP_DEVICE_COMPILE__)
#define XM_GLOBAL(T) T __attribute__((address_space(1)))  // pointer known to address HBM: global_load/store instead of flat
#define XM_LDS(T) T __attribute__((address_space(3)))     // pointer known to address the local data share: ds_read/ds_write instead of flat
#else
#define XM_GLOBAL(T) T
#define XM_LDS(T) T
#endif
#define XM_INL __host__ __device__ __forceinline__
#ifndef XM_NOINL_LINKAGE
#define XM_NOINL_LINKAGE  // (a second translation unit of the library that includes these headers makes the out-of-line functions inline)
#endif
#define XM_NOINL XM_NOINL_LINKAGE __host__ __device__ __noinline__
#define XM_NOINL_DECL XM_NOINL_LINKAGE __host__ __device__
#else
#define XM_HD
#define XM_GLOBAL(T) T
#define XM_LDS(T) T
#define XM_INL inline
#define XM_NOINL
#define XM_NOINL_DECL
#endif

namespace xm {

// ---------------------------------------------------------------- status codes (per read)
enum : int32_t {
  XM_OK = 0,
  XM_ST_OVERFLOW = 1,      // a fixed-capacity scratch structure overflowed: rerun this read with a larger scale
  XM_ST_OUT_OVERFLOW = 2,  // the result arena overflowed: rerun with a larger result arena
                           // (3 is not in use: reads may hold any number of ambiguous bases)
  XM_ST_NEED_GROW = 4,     // a gapmer uses more bases than the largest hashed length (host must grow the index)
  XM_ST_INTERNAL = 5,      // the reference would have thrown (e.g. TreeMap.subMap fromKey > toKey): whole batch fails
  XM_ST_NEED_HEAVY = 6,    // light pass only: the read needs the gapped extension chain; the full pass reruns it
                           // (7 and 12 are not in use)
  XM_ST_NEED_CONF = 11,    // quicklyConfidentInBestAlignment needs a value of the confidence table the host has not put there yet (ConfView): the
                           // read left its key in the miss list; the host evaluates it and the read runs again  (8-10: the wave form's, xm_wave.h)
};

// ---------------------------------------------------------------- Java arithmetic
XM_INL int32_t jadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
XM_INL int32_t jmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
XM_INL int32_t jabs(int32_t v) { return v == INT32_MIN ? v : (v < 0 ? -v : v); }
XM_INL int32_t j2i(double d) {  // (int)double: truncation, saturation, NaN -> 0
  if (d != d) return 0;
  if (d >= 2147483647.0) return INT32_MAX;
  if (d <= -2147483648.0) return INT32_MIN;
  return (int32_t)d;
}
XM_INL double dmin(double a, double b) { return a < b ? a : b; }  // Math.min on non-NaN operands
XM_INL double dmax(double a, double b) { return a > b ? a : b; }
XM_INL double jmaxd(double a, double b) {  // Math.max incl. NaN propagation
  if (a != a || b != b) return NAN;
  return a > b ? a : b;
}
XM_INL int imin(int a, int b) { return a < b ? a : b; }
XM_INL int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
XM_INL int imax(int a, int b) { return a > b ? a : b; }
XM_INL int iabs(int a) { return a < 0 ? -a : a; }
XM_INL double jnextUp(double d) {  // Math.nextUp
  if (d != d || d == INFINITY) return d;
  if (d == 0.0) { uint64_t b = 1; double r; __builtin_memcpy(&r, &b, 8); return r; }
  uint64_t b;
  __builtin_memcpy(&b, &d, 8);
  if (d > 0) b += 1; else b -= 1;
  double r;
  __builtin_memcpy(&r, &b, 8);
  return r;
}

// ---------------------------------------------------------------- Basepairs (4-bit IUPAC mask A=1 C=2 G=4 T=8)
XM_INL uint8_t bpComplement(uint8_t b) { return (uint8_t)(((b & 1) << 3) | ((b & 2) << 1) | ((b & 4) >> 1) | ((b & 8) >> 3)); }
XM_INL bool bpCanMatch(uint8_t a, uint8_t b) { return (a & b) != 0; }
XM_INL int bpPop(uint8_t b) { return __builtin_popcount(b & 15); }
XM_INL bool bpIsAmbiguous(uint8_t b) { return bpPop(b) != 1; }
XM_INL bool bpIsFullyAmbiguous(uint8_t b) { return (b & 15) == 15; }
XM_INL double bpFalseNegativeRate(uint8_t b) { int n = bpPop(b); return n <= 1 ? 0.0 : (double)(n - 1) / 3.0; }

// A sequence view: forward bytes in memory, optionally read as its reverse complement.
// `rc` doubles as Sequence.getComplementedFrom() != null.
struct SeqView {
  const uint8_t* base;
  int32_t len;
  uint8_t rc;
  uint8_t id;  // identity (Java object identity): queries: mate*2+rc, 4 = joined; references: unused
  // (reads, reference and joined mates all live in HBM: the loads are global_load, which the compiler can keep several of in flight)
  XM_INL uint8_t at(int i) const { XM_GLOBAL(const uint8_t)* const g = (XM_GLOBAL(const uint8_t)*)base; return rc ? bpComplement(g[len - 1 - i]) : g[i]; }
};

// Eight consecutive bases of a view, starting at view position i, as one 64-bit word (base k in byte k).  The caller guarantees
// 0 <= i and i + 8 <= len.  A reverse-complement view reads its bytes backwards and complements them (a 4-bit code's complement is its
// bit reversal).
XM_INL uint64_t seqWord8(const SeqView& s, int i) {
  uint64_t x;
  if (!s.rc) {
    __builtin_memcpy(&x, (const void*)(s.base + i), 8);
    return x;
  }
  __builtin_memcpy(&x, (const void*)(s.base + (s.len - 8 - i)), 8);
  x = __builtin_bswap64(x);
  return ((x & 0x0101010101010101ull) << 3) | ((x & 0x0202020202020202ull) << 1) | ((x & 0x0404040404040404ull) >> 1) | ((x & 0x0808080808080808ull) >> 3);
}
// How many positions k in [0, limit) can match (Basepairs.canMatch: the two codes share a base) before the first one that cannot:
// a.at(ai + k) against b.at(bi + k).  Eight positions per pair of loads while both views have eight left.
XM_INL int seqMatchRun(const SeqView& a, int ai, const SeqView& b, int bi, int limit) {
  int k = 0;
  while (k + 8 <= limit && ai + k + 8 <= a.len && bi + k + 8 <= b.len) {
    const uint64_t t = seqWord8(a, ai + k) & seqWord8(b, bi + k);   // per byte: the bases both codes allow (codes are < 16)
    const uint64_t zero = ~(t + 0x7F7F7F7F7F7F7F7Full) & 0x8080808080808080ull;  // bit 7 of byte j set iff byte j of t is 0 (bytes are < 16: no carry between bytes)
    if (zero) return k + (__builtin_ctzll(zero) >> 3);
    k += 8;
  }
  while (k < limit && bpCanMatch(a.at(ai + k), b.at(bi + k))) k++;
  return k;
}
// the same going down: a.at(ai - k) against b.at(bi - k), k in [0, limit)
XM_INL int seqMatchRunBack(const SeqView& a, int ai, const SeqView& b, int bi, int limit) {
  int k = 0;
  while (k + 8 <= limit && ai - k - 7 >= 0 && bi - k - 7 >= 0) {
    const uint64_t t = seqWord8(a, ai - k - 7) & seqWord8(b, bi - k - 7);  // byte j holds position (.. - 7 + j): the nearest position is byte 7
    const uint64_t zero = ~(t + 0x7F7F7F7F7F7F7F7Full) & 0x8080808080808080ull;
    if (zero) return k + (__builtin_clzll(zero) >> 3);
    k += 8;
  }
  while (k < limit && bpCanMatch(a.at(ai - k), b.at(bi - k))) k++;
  return k;
}

// Per byte of eight 4-bit codes: bit 7 set where the code is NOT exactly one of A C G T (Basepairs.isAmbiguous: its bit count is not 1)
XM_INL uint64_t seqAmbiguousBytes(uint64_t x) {
  x &= 0x0F0F0F0F0F0F0F0Full;
  const uint64_t pairs = (x & 0x0505050505050505ull) + ((x >> 1) & 0x0505050505050505ull);       // bit counts of the two 2-bit halves
  const uint64_t pop = (pairs & 0x0303030303030303ull) + ((pairs >> 2) & 0x0303030303030303ull);  // per byte: 0..4
  const uint64_t bad = pop ^ 0x0101010101010101ull;                                                // per byte: 0 iff exactly one bit
  return (bad + 0x7F7F7F7F7F7F7F7Full) & 0x8080808080808080ull;                                    // (bytes are < 8: no carry between bytes)
}
// Per byte: bit 7 set where the byte of t (all bytes < 128) is zero
XM_INL uint64_t seqZeroBytes(uint64_t t) { return ~(t + 0x7F7F7F7F7F7F7F7Full) & 0x8080808080808080ull; }
// how many of s.at(start) .. s.at(end - 1) are ambiguous codes; eight per load
XM_INL int seqCountAmbiguous(const SeqView& s, int start, int end) {
  int n = 0, i = start;
  for (; i + 8 <= end && i + 8 <= s.len && i >= 0; i += 8) n += __builtin_popcountll(seqAmbiguousBytes(seqWord8(s, i)));
  for (; i < end; i++) if (__builtin_popcount(s.at(i) & 15) != 1) n++;
  return n;
}

// ---------------------------------------------------------------- AlignmentParameters (M/AlignmentParameters.java:8-35)
struct Params {
  double MutationPenalty, InsertionStart_Penalty, InsertionExtension_Penalty, DeletionStart_Penalty, DeletionExtension_Penalty,
      MaxErrorRate, UnalignedPenalty, AmbiguityPenalty, Max_PenaltySpan;
  int32_t MaxNumMatches;
  int32_t StartingInsertionStartFree;
  XM_INL double getStartingInsertionStartPenalty() const { return StartingInsertionStartFree ? 0.0 : InsertionStart_Penalty; }  // :36-40
  XM_INL double getMinPossibleNonzeroPenalty() const {                                                                            // :42-47
    double r = MutationPenalty;
    r = dmin(r, getStartingInsertionStartPenalty() + InsertionStart_Penalty);
    r = dmin(r, DeletionStart_Penalty + DeletionExtension_Penalty);
    return r;
  }
  XM_INL double getPenalty(uint8_t q, uint8_t r) const {  // :156-180
    if (!bpCanMatch(r, q)) return MutationPenalty;
    return AmbiguityPenalty * bpFalseNegativeRate((uint8_t)(q | r));
  }
};

// ---------------------------------------------------------------- index in HBM
struct Table {          // one PackedMap (M/PackedMap.java) as CSR
  int32_t capacity;     // number of buckets (key mod capacity)
  int32_t maxCount;     // maxInterestingCountPerKey
  int64_t offBase;      // first entry of this table in bucketOff (capacity+1 entries)
  int64_t posBase;      // first entry of this table in positions
};
static const uint32_t XM_OVERFULL = 0x80000000u;  // bit 31 of bucketOff[k]: bucket k is overfull

struct IndexView {
  int32_t numContigs, minInterestingSize, maxHashedLength, enableGapmers, posIs64;
  int32_t dupWindow;
  double dupGranularity;
  int64_t totalForwardAndReverseSize;
  const int64_t* contigStart;   // [numContigs] first base of the forward contig in refCodes
  const int32_t* contigLen;     // [numContigs]
  const int64_t* seqCumStart;   // [2*numContigs+1] encoded-position base of fwd0,rev0,fwd1,rev1,...
  const uint8_t* refCodes;      // forward contigs, one 4-bit code per byte
  const Table* tables;          // [maxHashedLength+1]
  const uint32_t* bucketOff;
  const uint32_t* positions32;
  const uint64_t* positions64;
  const int64_t* dupKeyStart;   // [numContigs+1]
  const int32_t* dupKeys;       // sorted duplication starts per forward contig (M/Readable_DuplicationDetector.java)
  // Bucket lines (null: not built).  One line of eight words per bucket, at the bucket's index in bucketOff: word 0 = the bucket's count
  // (bit 31: overfull), words 1..7 = its first positions.  A probe (PackedMap.getNumMatchesLowerBound, PackedMap.get) is then ONE access of
  // 32 bytes (64 with 64-bit positions) that returns the header AND the hits; only buckets with more than XM_LINE_SLOTS positions (1.4 % of
  // the buckets of a genome-like reference) still go through bucketOff + positions.  Built on the device from the CSR tables (xm_lines_kernel).
  const uint32_t* lines32;
  const uint64_t* lines64;
  // ---- per call (set on the launch's copy of the view, not shared between contexts)
  // Transcendentals in output-affecting decisions are evaluated on the HOST, by the libm the oracle uses (SURVEY.md section 7 hard part 4):
  // AlignerWorker.quicklyConfidentInBestAlignment's pow / log term (AlignerWorker.java:532-549) comes from a table keyed by (penalty, query
  // length) that the host fills (xm_confidence.h); BlockAligner's (int)Math.log(refLen / Math.log(4.0)) (BlockAligner.java:48) is a step function
  // of refLen whose steps the host computes.
  const struct ConfEntry* conf;   // open addressing, confMask + 1 slots
  uint32_t confMask;
  struct ConfMiss* confMiss;      // keys the table did not hold
  const int32_t* baLogStep;       // [24] baLogStep[k] = smallest refLen >= 1 with (int)log(refLen / log(4.0)) >= k  (k = 1..23; INT32_MAX beyond)
};
struct ConfEntry { uint64_t penaltyBits; int32_t queryLength; int32_t used; double totalLengthForHighConfidence; };
struct ConfMissKey { uint64_t penaltyBits; int32_t queryLength; int32_t pad; };
struct ConfMiss { unsigned long long n; unsigned long long cap; ConfMissKey keys[1]; };
XM_INL uint32_t confHash(uint64_t penaltyBits, int32_t queryLength) {
  uint64_t x = (penaltyBits ^ (penaltyBits >> 31)) * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)queryLength * 0xC2B2AE3D27D4EB4Full;
  return (uint32_t)(x >> 37);
}
// -> true and the value, or false (not in the table)
XM_INL bool confLookup(const ConfEntry* table, uint32_t mask, double penalty, int32_t queryLength, double& value) {
  if (!table) return false;
  uint64_t bits;
  __builtin_memcpy(&bits, &penalty, 8);
  uint32_t h = confHash(bits, queryLength) & mask;
  while (true) {
    const ConfEntry e = table[h];
    if (!e.used) return false;
    if (e.penaltyBits == bits && e.queryLength == queryLength) { value = e.totalLengthForHighConfidence; return true; }
    h = (h + 1) & mask;
  }
}
// BlockAligner.java:48 (sic): (int)Math.log(refLen / Math.log(4.0)) + 1, from the host's steps
XM_INL int32_t baNumBasesToEncodeReferencePosition(const int32_t* step, int32_t refLen) {
  if (refLen == 0) return INT32_MIN + 1;  // log(0) = -Infinity, (int) saturates
  if (refLen < 0) return 1;               // log of a negative number is NaN, (int)NaN = 0
  int k = 0;
  while (k + 1 < 24 && refLen >= step[k + 1]) k++;
  return k + 1;
}
constexpr int XM_LINE_SLOTS = 7;
constexpr int64_t XM_LINE_FLAG = 1ll << 62;  // an index "into positions" with this bit set addresses a word of the lines array instead
// the line of bucket j (= t.offBase + k) from the CSR form; same code on the device (xm_lines_kernel) and in the host simulation
template <typename W, typename P>
XM_INL void xmFillLine(W* line, uint32_t o0, uint32_t o1, const P* tablePositions) {
  const uint32_t first = o0 & ~XM_OVERFULL, count = (o1 & ~XM_OVERFULL) - first;
  line[0] = (W)(count | (o0 & XM_OVERFULL));
  for (uint32_t j = 0; j < (uint32_t)XM_LINE_SLOTS; j++) line[1 + j] = ((o0 & XM_OVERFULL) == 0 && j < count) ? (W)tablePositions[first + j] : (W)0;
}

XM_INL SeqView refView(const IndexView& ix, int contig, bool rc) {
  SeqView v;
  v.base = ix.refCodes + ix.contigStart[contig];
  v.len = ix.contigLen[contig];
  v.rc = rc ? 1 : 0;
  v.id = 0;
  return v;
}

// ---------------------------------------------------------------- capacities (scale s = 1, 4, 16, ...)
struct Caps {
  int32_t scale;
  int32_t heavyAllowed;  // how far into the gapped chain a read may go before it stops with XM_ST_NEED_HEAVY: 0 = not at all (stops
                         // before HashBlock_Aligner), 1 = up to BlockAligner (the hash-block analysis runs, the piece-wise alignment does not), 2 = all
  int32_t searchInHbmOnly;  // test entry only (xm_test_local_align): every PathAligner search in HBM mode, the LDS slot is not tried
  int32_t maxLevels, maxPyramidBlocks, maxHistory, maxCounters, maxPending, maxQM, maxGoodAlignments, maxBlocks,
      maxNodes, nodeHash, gridCap, maxBuckets, bucketHash, matcherEntries, maxSections, maxPieces, maxCountMap, maxJoined;
};
XM_INL Caps makeCaps(int scale) {
  Caps c;
  c.scale = scale;
  c.heavyAllowed = 2;
  c.searchInHbmOnly = 0;
  c.maxLevels = 48 * scale;
  c.maxPyramidBlocks = 1536 * scale;
  c.maxHistory = 192 * scale;
  c.maxCounters = 96 * scale;
  c.maxPending = 128 * scale;  // (blocks put aside until the path is exhausted: 1 kb reads that align nowhere need 257-320 at scale 4)
  c.maxQM = 64 * scale;
  c.maxGoodAlignments = 32 * scale;  // (reads in repeats: a tandem repeat offers a read dozens of places; room for 16 sent ~50 reads per million of the repeat-rich workload
                                     // into the pass behind the gapped pass, which lasts as long as its slowest read: profiles/r04/NOTES.md.  The seeding state of a pair
                                     // with ambiguity codes and the two aligners of getUnpairedAlignments must still fit a region: qmaInit's pool, the sub-aligners' half)
  c.maxBlocks = 16 * scale;
  c.maxNodes = 1536 * scale;
  c.nodeHash = 4096 * scale;  // power of two >= 2 * maxNodes
  c.gridCap = 10000 * scale;  // cells of PathAligner's dense (x,y) grid; larger problems use the hash
  c.maxBuckets = 512 * scale;
  c.bucketHash = 2048 * scale;  // power of two >= 4 * maxBuckets
  c.matcherEntries = 6144 * scale;
  c.maxSections = 40 * scale;
  c.maxPieces = 16 * scale;
  c.maxCountMap = 32 * scale;
  c.maxJoined = 512 * scale;
  return c;
}

// ---------------------------------------------------------------- bump arena
#if defined(XM_ARENA_TRACE) && !defined(__HIPCC__)
void xm_arena_trace(const void* base, size_t offset, size_t bytes);  // host simulation only (tests/hostsim): which structure lies where in an arena
#endif
struct Arena {
  uint8_t* base;
  size_t size, used;
  bool overflow;
  XM_INL void init(void* p, size_t n) { base = (uint8_t*)p; size = n; used = 0; overflow = false; }
  XM_INL void* alloc(size_t bytes) {
    size_t a = (used + 15) & ~(size_t)15;
    if (a + bytes > size) { overflow = true; return (void*)base; }  // caller checks `overflow` before use
    used = a + bytes;
#if defined(XM_ARENA_TRACE) && !defined(__HIPCC__)
    xm_arena_trace(base, a, bytes);
#endif
#ifdef XM_ARENA_POISON  // diagnostic builds: whatever is allocated starts as garbage (nothing may read arena memory it has not written)
    for (size_t i_ = 0; i_ < bytes; i_++) base[a + i_] = 0xA5;
#endif
    return base + a;
  }
};
template <typename T>
XM_INL T* arenaArray(Arena& a, size_t n) { return (T*)a.alloc(sizeof(T) * (n ? n : 1)); }

// ---------------------------------------------------------------- counters for the roofline accounting
struct DevCounters {
  unsigned long long reads, headerProbes, bucketFetches, hitsFetched, candidatesExtended, pathAlignerCalls, pathAlignerNodes,
      quickAccepts, blocksOut, alignmentsOut, refWindowBytes, readBytes;
  unsigned long long boundChecks, boundRejects, boundCells, boundPieceChecks, boundPieceRejects;  // (the last two: pieces the filter examined / proved unalignable before their chain ran)  // the rejection filter in front of PathAligner (xm_bound.h): searches it took, searches it proved null, cells it computed
  unsigned long long t[16];  // XM_PROFILE builds only: shader-clock ticks per phase, summed over lanes
};

#if defined(XM_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define XM_TIC(var) unsigned long long var = clock64()
#if XM_PROFILE == 2  // wave time: only the lowest active lane of the wave counts (per-lane sum / wave time = mean active lanes)
#define XM_TOC(dc, slot, var) do { unsigned long long m_ = __ballot(1); if ((dc) && (int)__lane_id() == __ffsll((long long)m_) - 1) (dc)->t[slot] += clock64() - var; } while (0)
#else
#define XM_TOC(dc, slot, var) do { if (dc) (dc)->t[slot] += clock64() - var; } while (0)
#endif
#else
#define XM_TIC(var) do { } while (0)
#define XM_TOC(dc, slot, var) do { } while (0)
#endif
enum { T_TOTAL = 0, T_PYRAMID = 1, T_WALK = 2, T_HITS = 3, T_STRAIGHT = 4, T_ANALYZE = 5, T_PATH = 6, T_PATH_INIT = 7, T_BLOCK = 8, T_MATCHER_INDEX = 9, T_CONFIDENT = 10, T_OUTER = 11, T_BOUND = 12 };

}  // namespace xm
