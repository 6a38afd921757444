// xmapper-hip device core: an exact rejection filter in front of PathAligner's best-first search (round 6).
//
// PathAligner.align (M/PathAligner.java:55-293) returns null when the smallest key of its queue passes maxInterestingPenalty + 1e-6 (:169) - having
// explored every node below that budget.  For reads that do not align (configs[4]: 10 kb reads with 5 % substitutions + 5 % indel events, cut into 1 kb
// queries, whose pieces BlockAligner hands to the search one by one) that is the worst case of the search and nearly all of its work: 85 % of all search
// nodes of that workload sit in searches that return null (measured with the oracle: profiles/r06/NOTES.md 1).  Any sound LOWER BOUND on the penalty of
// every node the search can hold at x == goalX decides such a search without running it:
//
//   a node's three penalties (:573-719) are sums of move prices along ONE path from a start node (:120-150) - a base pair getPenalty(a, b), a new
//   insertion left.penalty + InsertionStart + InsertionExtension (or "disallowed" = 1e6), an extended one left.insertX + InsertionExtension, the same for
//   deletions - so none is below the value the plain affine-gap recurrence (every move allowed at its regular price, no heuristic, no band) gives its cell;
//   the search ends with an answer only through a node at goalX that it takes from a bucket whose key is <= maxInterestingPenalty + 1e-6 (:169,180), and a
//   node's key is never below its penalty (estimateOverallPenalty :475-521 adds to it; putNode :458-461 only raises it).
//
// Hence: recurrence's minimum over the cells of column goalX > maxInterestingPenalty + 1e-6  =>  align() returns null.  The filter computes that minimum on an
// integer grid (prices rounded DOWN to 1/60 penalty unit: a lower bound of the lower bound; Mapper's default prices are multiples of it) and rejects when
// it exceeds floor((max + 1e-6 + 1e-7) * 60) - the 1e-7 covers the rounding of the reference's double sums, which it does not reproduce.  Whatever the
// filter does not take (a problem outside its limits, a search it cannot reject) goes to the search as before, so the result streams cannot change;
// the oracle carries the same recurrence as an OBSERVER that never acts on it (oracle/xmo_extend.h PathAligner::boundRejects), asserts on every search of every test
// that a rejected search did return null, and counts the nodes the reference spent in rejected searches - which is what keeps DevCounters::pathAlignerNodes
// comparable: device nodes + oracle's nodes in rejected searches == oracle nodes, device rejects == oracle rejects (tests/test_gpu_bound.py, bench.py).
//
// Form: a band of diagonals d = y' - x' in search coordinates (x' counts query bases consumed, y' reference bases of the window, both in the direction the
// search runs: chooseSearchReverse :17-53) - a path whose penalty stays within the budget cannot leave the start diagonals 0 .. m - n by more bases than the
// budget buys insertions or deletions.  Column by column; a column's cells with a value within the budget form an interval [lo, hi] of band slots, the next
// column is computed over [lo - 1, hi + (what a deletion run reaches)] only, and an empty interval ends the filter: no path reaches goalX within the budget.
// A slot is one 32-bit word of LDS (H in the low half, the insertion state E in the high half), the window's bases are copied into LDS once: the wave's 10 KB
// search slot is cut into 8 regions of 1260 bytes (200 band slots + 460 bases), one per read of the wave (the passes that run the filter hold 8 reads per
// wave at most; a search that takes its turn at the slot never runs while a lane is in here - a wave executes one path at a time).
// Lanes: the passes that run the filter give a read EIGHT adjacent lanes (xm_extend.h, xmSetPairMode 3).  A column's cells are computed eight at a time; the one
// dependency between them, the deletion state that runs up the column, is a prefix minimum taken across the lanes (boundSweep).  One lane, or the two of a
// pair, compute every cell each (the host simulation of the tests as well).
// Two places call it: pathAlign (every search) and BlockAligner's alignPiece (boundPieceApplies: the bound over a piece's whole inner chain - a piece it proves
// unalignable skips its hash-block analyses, straight alignments and searches).
#pragma once
#include "xm_defs.h"
#include <type_traits>

namespace xm {

constexpr int XM_BOUND_SCALE = 60;        // grid: 1/60 penalty unit
constexpr int XM_BOUND_KMAX = 200;        // band slots of a region in LDS (+ one slot that stays "beyond the budget" behind the band)
constexpr int XM_BOUND_MMAX = 456;        // bases of the reference window in a region in LDS
constexpr int XM_BOUND_REGION = (XM_BOUND_KMAX + 1) * 4 + XM_BOUND_MMAX;  // 1260 bytes = 315 words (odd: the regions of a wave start in different LDS banks)
constexpr int XM_BOUND_REGIONS = 8;       // per wave
constexpr int XM_BOUND_KMAX_WIDE = 2048;  // problems that do not fit a region (a window far longer than the query: a third of a percent of the searches of configs[4],
constexpr int XM_BOUND_MMAX_WIDE = 4096;  // a seventh of their nodes): the same recurrence in the lane's temporaries in HBM
constexpr int XM_BOUND_INF = 0xFFFF;      // a value beyond the budget (budgets stay below 60000 units)

// The problem as the filter sees it (PaProblem of xm_extend.h without the analysis fields it does not use)
struct BoundProblem {
  const uint8_t* qBase; int32_t qLen; bool qRc; const uint8_t* rBase; int32_t referenceLen;
  int32_t startA, endA, startB, endB, predictedBestOffset;
  double mutation, insStart, insExt, delStart, delExt, maxErrorRate, ambiguity;
  double budget;   // the penalty the answer must stay within (a search: its maxInterestingPenalty = section length x MaxErrorRate, :60; a piece: BlockAligner's maxPenalty)
  int32_t piece;   // 1: the bound over a piece's whole inner chain (boundPieceProblem below): both ends of the window free, the piece's first and last base left out
};

struct BoundPrices { int32_t mut, isie, ie, dsde, de, amb1, amb2, amb3, thr; };

// prices and budget on the grid (the oracle's observer evaluates the same expressions: IEEE double products and floors)
XM_INL bool boundPrices(const BoundProblem& b, BoundPrices& c) {
  const double s = (double)XM_BOUND_SCALE;
  const double t = floor((b.budget + 0.000001 + 0.0000001) * s);
  if (!(t >= 0 && t < 60000.0)) return false;
  c.thr = (int32_t)t;
  c.mut = (int32_t)floor(b.mutation * s);
  c.isie = (int32_t)floor((b.insStart + b.insExt) * s);
  c.ie = (int32_t)floor(b.insExt * s);
  c.dsde = (int32_t)floor((b.delStart + b.delExt) * s);
  c.de = (int32_t)floor(b.delExt * s);
  // AmbiguityPenalty * getMutationFalseNegativeRate(union) (M/AlignmentParameters.java:156-180): a union of 2, 3, 4 bases
  c.amb1 = (int32_t)floor(b.ambiguity * (1.0 / 3.0) * s); c.amb2 = (int32_t)floor(b.ambiguity * (2.0 / 3.0) * s); c.amb3 = (int32_t)floor(b.ambiguity * (3.0 / 3.0) * s);
  // (prices a path can collect without end must be positive, and none may be negative: the recurrence's values only grow along a path)
  if (c.mut < 0 || c.isie < 1 || c.ie < 1 || c.dsde < 1 || c.de < 1 || c.amb1 < 0 || c.mut > 30000 || c.isie > 30000 || c.dsde > 30000 || c.amb3 > 30000) return false;
  return true;
}

// Geometry of the band: false = the filter does not take the problem (the search runs).  dlo: diagonal of slot 0; K slots.
XM_INL bool boundBand(int n, int m, bool mayExtend, const BoundPrices& c, int& dlo, int& K, bool freeStart = false) {
  // not taken: windows at a contig end in the search's direction (start nodes with unaligned moves at 0.1 per base: :141-150,592-594)
  if (mayExtend || n < 1 || m < 1 || m > XM_BOUND_MMAX_WIDE) return false;
  // bases all insertions of a path within the budget can hold (freeStart: a path may begin in the middle of an insertion, which then costs its extensions only)
  const int maxIns = freeStart ? c.thr / c.ie + 1 : (c.thr < c.isie ? 0 : (c.thr - c.isie) / c.ie + 1);
  const int maxDel = c.thr < c.dsde ? 0 : (c.thr - c.dsde) / c.de + 1;
  // start nodes: (0, y') for y' = 0 .. m - n (:120-131), or - a window shorter than the query - (x', 0) for x' = 0 .. n - m (:132-139): diagonals 0 .. m - n or n - m .. 0
  // (freeStart - the bound over a piece's whole chain: a start node at every row of column 0)
  const int d0 = freeStart ? 0 : (m >= n ? 0 : -(n - m)), d1 = freeStart ? m : (m >= n ? m - n : 0);
  dlo = d0 - maxIns;
  if (dlo < -n) dlo = -n;
  int dhi = d1 + maxDel;
  if (dhi > m) dhi = m;
  K = dhi - dlo + 1;
  return K <= XM_BOUND_KMAX_WIDE;
}

#if defined(__HIP_DEVICE_COMPILE__)
// (included by xm_extend.h behind palSlot(): the wave's search slot in LDS)
#if defined(XM_WAVE_UNIFORM) || defined(XM_BOUND_OFF)
// (XM_BOUND_OFF: experiment builds without the filter's code; the wave-per-read kernels, xm_wave_kernel.hip: their searches are wave-cooperative and use the wave's slot; no filter there)
XM_INL void xmSetBoundFilter(int) {}
XM_INL bool xmBoundFilter() { return false; }
XM_INL bool xmBoundCooperative() { return false; }
#else
__shared__ int xm_bound_filter;
XM_INL void xmSetBoundFilter(int on) { if (threadIdx.x == 0) xm_bound_filter = on; }  // (before the block's first barrier)
XM_INL bool xmBoundFilter() { return xm_bound_filter != 0; }
XM_INL bool xmBoundCooperative() { return (xm_bound_filter & 2) != 0; }  // (experiment switch: XM_GROUP_SWEEP=0 - eight lanes per read, every lane computing every cell)
#endif
// the region of the read this lane runs: reads sit in lanes 0 .. 7 of a wave (2^groupShift lanes per read: 0 .. 15, 0 .. 63); null = the lane has none
XM_INL uint8_t* boundRegion(int groupShift) {
  const int r = (int)__lane_id() >> groupShift;
  if (r >= XM_BOUND_REGIONS) return nullptr;
  return palSlot() + r * XM_BOUND_REGION;
}
XM_INL int boundLaneOfEight() { return (int)__lane_id() & 7; }
#else
XM_INL int boundLaneOfEight() { return 0; }
XM_INL bool& xmBoundFilterHost() { static thread_local bool on = false; return on; }
XM_INL void xmSetBoundFilter(int on) { xmBoundFilterHost() = on != 0; }
XM_INL bool xmBoundFilter() { return xmBoundFilterHost(); }
XM_INL bool xmBoundCooperative() { return false; }
XM_INL uint8_t* boundRegion(int) { static thread_local uint32_t region[(XM_BOUND_REGION + 3) / 4]; return (uint8_t*)region; }  // host simulation (tests only)
#endif

// The lanes that run one read (xmSetPairMode: 2^GS adjacent lanes, all on the same instructions with the same values) as the recurrence uses them: lane g of the
// group takes every 2^GS-th cell of a column.  GS = 0 (one lane, the two lanes of a pair, the host simulation): every lane computes every cell.
template <int GS>
struct BoundGroup {
  static constexpr int G = 1 << GS;
#if defined(__HIP_DEVICE_COMPILE__)
  int g, base;
  XM_INL BoundGroup() : g(GS ? ((int)__lane_id() & (G - 1)) : 0), base(GS ? ((int)__lane_id() & ~(G - 1)) : (int)__lane_id()) {}
  // min of v over the lanes of the group before this one (`none` for the first).  Eight lanes: data-parallel primitives inside the 16-lane row (row_shr:n takes
  // the value n lanes down; what crosses into the group from its neighbour is masked out) - no trip through the LDS crossbar, no wait
  // (`old` = what a lane without a source in its row keeps.  The result is pinned to a register of its own: ROCm 7.2's compiler folded `g >= 1 ? dpp(v) : none`
  // into one conditional move and lost the shift - scripts/dpp_check/dpp_check.hip shows it, profiles/r06/NOTES.md)
  template <int CTRL> static XM_INL int dpp(int old, int v) { int r = __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, 0xF, false); asm volatile("" : "+v"(r)); return r; }
  XM_INL int exclusiveMin(int v, int none) const {
    if constexpr (GS == 0) return none;
    else if constexpr (GS == 3) {
      int t = dpp<0x111>(none, v);
      int ex = g >= 1 ? t : none;
      t = dpp<0x111>(none, ex); ex = imin(ex, g >= 1 ? t : none);
      t = dpp<0x112>(none, ex); ex = imin(ex, g >= 2 ? t : none);
      t = dpp<0x114>(none, ex); ex = imin(ex, g >= 4 ? t : none);
      return ex;
    } else {
      int ex = __shfl_up(v, 1, G);
      if (g == 0) ex = none;
#pragma unroll
      for (int d = 1; d < G; d <<= 1) { const int t = __shfl_up(ex, d, G); if (g >= d) ex = imin(ex, t); }
      return ex;
    }
  }
  // min of v over all lanes of the group, in every lane (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror: the other half of the eight)
  XM_INL int allMin(int v) const {
    if constexpr (GS == 0) return v;
    else if constexpr (GS == 3) {
      v = imin(v, dpp<0xB1>(v, v));
      v = imin(v, dpp<0x4E>(v, v));
      v = imin(v, dpp<0x141>(v, v));
      return v;
    } else {
#pragma unroll
      for (int d = 1; d < G; d <<= 1) v = imin(v, __shfl_xor(v, d, G));
      return v;
    }
  }
  XM_INL uint32_t liveMask(bool live) const { return (uint32_t)(__ballot(live ? 1 : 0) >> base) & ((1u << G) - 1u); }
  // what the lanes of the group wrote is read by the others from here on (one wave: its memory operations complete in order; this keeps the compiler from moving them)
  XM_INL void sync() const { if constexpr (GS != 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } }
#else
  static constexpr int g = 0;
  XM_INL int exclusiveMin(int, int none) const { return none; }
  XM_INL int allMin(int v) const { return v; }
  XM_INL uint32_t liveMask(bool live) const { return live ? 1u : 0u; }
  XM_INL void sync() const {}
#endif
};

// The recurrence over a band held in `W` (K + 1 words: slot K stays "beyond the budget") and the window's bases in `TB`, both in LDS (a region of the wave's slot)
// or both in the lane's temporaries in HBM (WIDE).  true = no cell of column n stays within the budget.
// The deletion state along a column is the one dependency between its cells: F(k) = min(F(k-1) + de, H(k-1) + ds + de).  Written out it is a prefix minimum,
//   F(k) = min over j < k of ( H0(j) - j de ) + ds + de + (k - 1) de,   H0 = the cell's value without its deletion state (a run that starts from a cell which itself
// came out of a run costs a second start: never the minimum, ds >= 0) - so a group of lanes computes H0 of 2^GS cells at once, takes the prefix minimum
// across its lanes, and carries the minimum of the cells done so far up the column.
template <bool WIDE, int GS, typename CharA>
XM_INL bool boundSweep(typename std::conditional<WIDE, XM_GLOBAL(uint32_t)*, XM_LDS(uint32_t)*>::type const W, typename std::conditional<WIDE, XM_GLOBAL(const uint8_t)*, XM_LDS(const uint8_t)*>::type const TB,
                       const BoundPrices& c, int n, int m, int dlo, int K, bool searchReverse, CharA charA, unsigned long long& cells, bool freeStart = false) {
  constexpr int G = 1 << GS;
  const BoundGroup<GS> grp;
  const int g = grp.g;
  // (prices as scalars: a struct the compiler keeps in private memory would cost a trip to it per use)
  const int thr = c.thr, mut = c.mut, isie = c.isie, ie = c.ie, dsde = c.dsde, de = c.de;
  const unsigned long long ambPacked = ((unsigned long long)(unsigned)c.amb1 << 16) | ((unsigned long long)(unsigned)c.amb2 << 32) | ((unsigned long long)(unsigned)c.amb3 << 48);
  const uint32_t INFW = ((uint32_t)XM_BOUND_INF << 16) | (uint32_t)XM_BOUND_INF;
  const int BIG = 1 << 28;                            // "no cell yet" in the prefix minimum (prices x slots stay below 2^27)
  auto sub = [&](uint32_t a, uint32_t b) -> int {   // (selects, no branch: the loop body stays straight-line code)
    const int amb = (int)((ambPacked >> (16 * (__builtin_popcount((a | b) & 15u) - 1))) & 0xFFFFu);   // 0 for two equal unambiguous bases
    return (a & b) == 0 ? mut : amb;                                                                   // !Basepairs.canMatch -> a mutation
  };
  // column 0: the start nodes (0, y') for y' = 0 .. m - n at penalty 0, insertion state "disallowed" (:120-131 with startingInsertionStartPenalty disallowed;
  // a window shorter than the query: (0, 0) alone, and one start node (x', 0) at the foot of every column up to n - m, :132-139)
  // (freeStart: every row of column 0 is a start node, and a path may arrive there in the middle of an insertion: insertion state 0 as well)
  int lo = -dlo, hi = -dlo + (freeStart ? m : (m >= n ? m - n : 0));
  for (int k = lo + g; k <= hi; k += G) W[k] = freeStart ? 0u : (uint32_t)XM_BOUND_INF << 16;
  grp.sync();
  unsigned long long done = 0;
  uint8_t aNext = charA(searchReverse ? n - 1 : 0);
  for (int x = 1; x <= n; x++) {
    const uint8_t a = aNext;
    if (x < n) aNext = charA(searchReverse ? n - 1 - x : x);
    // slot k of this column is the cell (x, y' = x + dlo + k); cells exist for 1 <= y' <= m
    const int kGeom = 1 - x - dlo;
    const bool foot = !freeStart && x <= n - m;                   // a start node (x, 0) below the column's first cell
    const int kEnd = imin(K - 1, m - x - dlo);
    int newLo = 0x7FFFFFFF, newHi = -1;
    const int tb0 = x + dlo - 1;                                 // base of the window under slot k: TB[tb0 + k]
    if (!foot) {
      // the slots up to the previous column's interval [lo, hi]: slot k + 1 of that column is read as it stands (k + 1 >= lo here, and slot hi + 1 is made
      // "beyond the budget" first), so the loop has no test in it and its loads do not wait for each other
      // (values are kept below 2^16 by a min, not reset to "beyond the budget": a value above the budget stays above it along every path, and only values within
      // it count as alive)
      const int kStart = imax(imax(lo - 1, kGeom), 0);
      const int kMain = imin(kEnd, hi);
      W[hi + 1] = INFW;
      grp.sync();
      const uint32_t a32 = a;
      int carry = BIG;                                           // min of H0(j) - j de over the cells of this column so far
      for (int kb = kStart; kb <= kMain; kb += G) {
        const int k = kb + g;
        const bool in = k <= kMain;
        const int kk = in ? k : kMain;
        uint32_t cur = W[kk];                                    // slot k of the previous column = cell (x - 1, y' - 1)
        if (kk < lo) cur = INFW;
        const uint32_t nxt = W[kk + 1];                          // cell (x - 1, y')
        const int s = sub(a32, TB[tb0 + kk]);
        const int hD = (int)(cur & 0xFFFFu), hL = (int)(nxt & 0xFFFFu), eL = (int)(nxt >> 16);
        const int e = imin(imin(eL + ie, hL + isie), XM_BOUND_INF);
        const int h0 = imin(hD + s, e);
        const int v = in ? h0 - k * de : BIG;
        const int ex = grp.exclusiveMin(v, BIG);
        const int f = imin(carry, ex) + dsde + (k - 1) * de;
        const int h = imin(imin(h0, f), XM_BOUND_INF);
        if (in) W[k] = (uint32_t)h | ((uint32_t)e << 16);
        const uint32_t lm = grp.liveMask(in && h <= thr);
        if (lm) { newLo = imin(newLo, kb + __builtin_ctz(lm)); newHi = kb + 31 - __builtin_clz(lm); }
        carry = imin(carry, grp.allMin(v));
      }
      done += (unsigned long long)imax(kMain - kStart + 1, 0);
      // above the interval only a deletion run arrives: cell k holds carry + ds + de + (k - 1) de, while that stays within the budget
      for (int kb = kMain + 1; kb <= kEnd; kb += G) {
        const int k = kb + g;
        const int f = carry + dsde + (k - 1) * de;
        const bool live = k <= kEnd && f <= thr;
        if (live) W[k] = (uint32_t)f | ((uint32_t)XM_BOUND_INF << 16);
        const uint32_t lm = grp.liveMask(live);
        if (!lm) break;
        newLo = imin(newLo, kb); newHi = kb + 31 - __builtin_clz(lm);
        done += (unsigned long long)__builtin_popcount(lm);
        if (lm != (1u << G) - 1u) break;
      }
    } else {
      // (rare: a window shorter than the query.  Every lane computes every cell)
      int k = kGeom;
      int f = XM_BOUND_INF, hBelow = 0;                          // deletion state entering the cell, H of the cell below (slot k - 1 of this column)
      W[kGeom - 1] = (uint32_t)XM_BOUND_INF << 16;               // (slot kGeom - 1 >= 0: dlo <= -(n - m))  H = 0, insertion state "disallowed"
      newLo = kGeom - 1; newHi = kGeom - 1;
      uint32_t cur = (k >= lo && k <= hi) ? W[k] : INFW;
      for (; k <= kEnd; k++) {
        const uint32_t nxt = (k + 1 >= lo && k + 1 <= hi) ? W[k + 1] : INFW;  // slot k + 1 of the previous column = cell (x - 1, y')
        f = imin(f + de, hBelow + dsde);
        if (k > hi && f > thr) break;                            // above the previous column's interval only a deletion run arrives
        const int hD = (int)(cur & 0xFFFFu), hL = (int)(nxt & 0xFFFFu), eL = (int)(nxt >> 16);
        int e = imin(eL + ie, hL + isie);
        int h = imin(imin(hD + sub(a, TB[tb0 + k]), e), f);
        if (e > thr) e = XM_BOUND_INF;
        if (f > thr) f = XM_BOUND_INF;
        if (h > thr) h = XM_BOUND_INF; else { if (k < newLo) newLo = k; newHi = k; }
        W[k] = (uint32_t)h | ((uint32_t)e << 16);
        hBelow = h;
        cur = nxt;
        done++;
      }
    }
    grp.sync();
    if (newHi < 0) { cells = done; return true; }
    lo = newLo; hi = newHi;
  }
  cells = done;
  return false;
}

// The bound over a PIECE's whole inner chain (BlockAligner.alignPiece :215-249 -> StraightAligner -> HashBlock_Aligner -> StraightAligner -> PathAligner_Runner): when it holds,
// alignPiece returns null, and the chain below it - its hash-block analyses above all - need not run.  Why: every way the chain has of answering is a path through the same grid.
//   * A search (PathAligner) is handed the piece and a window INSIDE this one (HashBlock_Aligner only narrows, :64-80), in either direction; its nodes' penalties are bounded from
//     below by the recurrence over the larger window with a start node at every row (fewer start nodes, a smaller grid: only higher values).  A window one base shorter than the
//     piece (an offset of maxPossibleOffset + 1, which the matcher's inclusive upper limit allows, :128-131) gives the search one free query base (:132-139): the piece's first
//     and last base are left out of the recurrence, which only lowers it.
//   * A straight alignment (StraightAligner.straightAlignment :73-94) at an offset inside [minPossibleOffset, maxPossibleOffset + 1] covers all of the piece but possibly its last
//     base, one diagonal of the grid; StraightAligner returns it only if its aligned penalty is within the budget (:28-70: "<= 0", "<= maxInterestingPenalty").
//   * The budgets down the chain only shrink (StraightAligner :62-64, MaxErrorRate = min(rate, MaxErrorRate)).
//   * A path may arrive at the first kept column in the middle of an insertion or of a deletion run: the start nodes carry insertion state 0, and sit at every row.
// Conditions (the caller checks them; the oracle's observer the same ones): the window touches neither end of the contig (no unaligned moves, :141-150,592-594); it is at least as
// long as the piece; the offset the chain starts from lies in [minPossibleOffset, maxPossibleOffset + 1] (a straight alignment is not cut by the window); and the analyses of the
// chain would not share the caller's matcher (HashBlock_Aligner.java:113-121: its section length is more than 1.5 x the piece's lookup uncertainty, or there is none) - a shared
// matcher's sections are indexed, or marked null, in the order lookups reach them (:203-215), and skipping lookups would change what later analyses see.
XM_INL bool boundPieceApplies(int qStart, int qEnd, int wStart, int wEnd, int referenceLen, int parentOffset, bool parentHasMatcher, int parentSectionLength) {
  const int n = qEnd - qStart, m = wEnd - wStart;
  if (n < 8 || m < n || wStart <= 0 || wEnd >= referenceLen) return false;
  const int minOff = wStart - qStart, maxOff = wEnd - qEnd, u = maxOff - minOff;
  if (parentOffset < minOff || parentOffset > maxOff + 1) return false;
  if (parentHasMatcher && !(parentSectionLength > u + u / 2)) return false;
  return true;
}

// true = the search of this problem returns null (proved); false = not decided.  taken / cells: whether the filter took the problem, and the cells it computed.
// tmp: the lane's temporaries (a problem too wide for a region of the wave's slot keeps its band there, for the length of this call)
// (out of line: inlined into pathAlign - i.e. into innerChain - the same code made gapped passes with several reads per wave end in a memory fault, with the
// filter switched on or off, while a build with this function out of line, or without it, ran the same batches; profiles/r06/NOTES.md 2)
XM_NOINL bool boundRejects(const BoundProblem& bp, int groupShift, Arena& tmp, bool& taken, unsigned long long& cells) {
  taken = false;
  cells = 0;
  BoundPrices c;
  if (!boundPrices(bp, c)) return false;
  const bool piece = bp.piece != 0;
  // (a piece: the bound must hold for every search and every straight alignment of the piece's chain - any window inside this one, either direction, the piece's first or
  // last base possibly left out: forward coordinates, the bases 1 .. n - 2 of the piece, a start node at every row; boundPieceProblem has the conditions)
  const int n = bp.endA - bp.startA - (piece ? 2 : 0), m = bp.endB - bp.startB;
  XM_GLOBAL(const uint8_t)* const qg = (XM_GLOBAL(const uint8_t)*)bp.qBase;
  XM_GLOBAL(const uint8_t)* const rg = (XM_GLOBAL(const uint8_t)*)bp.rBase;
  const int startA = bp.startA + (piece ? 1 : 0), startB = bp.startB, qLen = bp.qLen;
  const bool qRc = bp.qRc;
  auto charA = [=](int i) -> uint8_t { const int k = startA + i; return qRc ? bpComplement(qg[qLen - 1 - k]) : qg[k]; };
  auto charB = [=](int j) -> uint8_t { return rg[startB + j]; };
  // chooseSearchReverse :17-53 (the search evaluates it again; it decides which end of the window the start nodes lie at)
  bool searchReverse = !piece;
  if (!piece) {
    const int diagonal = bp.startB - (bp.startA + bp.predictedBestOffset);  // :81
    const int s = imax(bp.startA, bp.startB - bp.predictedBestOffset), t = imin(bp.endA, bp.endB - bp.predictedBestOffset);
    int sumMis = 0, numMis = 0, sumMatch = 0, numMatch = 0;
    for (int i = 0; i < t - s; i++) {
      const int j = i - diagonal;
      if (j >= 0 && j < m) { if ((charA(i) & charB(j)) == 0) { sumMis += i; numMis++; } else { sumMatch += i; numMatch++; } }
    }
    if (numMis > 1 && numMatch > 1) searchReverse = (sumMis / numMis) > (sumMatch / numMatch);
  }
  const bool mayExtend = piece ? (bp.startB == 0 || bp.endB == bp.referenceLen) : (searchReverse ? bp.startB == 0 : bp.endB == bp.referenceLen);  // :87-93
  int dlo, K;
  if (!boundBand(n, m, mayExtend, c, dlo, K, piece)) return false;
  const bool wide = K > XM_BOUND_KMAX || m > XM_BOUND_MMAX;
  uint8_t* region = nullptr;
  const size_t mark = tmp.used;
  if (wide) {
    if (tmp.overflow) return false;
    region = (uint8_t*)tmp.alloc((size_t)(K + 1) * 4 + (size_t)m);
    if (tmp.overflow) { tmp.overflow = false; tmp.used = mark; return false; }  // (no room: the search runs as it would have)
  } else {
    region = boundRegion(groupShift);
    if (!region) return false;
  }
  taken = true;
  bool rejected;
  const bool coop = groupShift == 3 && xmBoundCooperative();
  if (wide) {
    XM_GLOBAL(uint32_t)* const W = (XM_GLOBAL(uint32_t)*)region;
    XM_GLOBAL(uint8_t)* const TB = (XM_GLOBAL(uint8_t)*)(region + (size_t)(K + 1) * 4);
    if (coop) {
      for (int j = boundLaneOfEight(); j < m; j += 8) TB[j] = charB(searchReverse ? m - 1 - j : j);
      rejected = boundSweep<true, 3>(W, (XM_GLOBAL(const uint8_t)*)TB, c, n, m, dlo, K, searchReverse, charA, cells, piece);
    } else {
      for (int j = 0; j < m; j++) TB[j] = charB(searchReverse ? m - 1 - j : j);
      rejected = boundSweep<true, 0>(W, (XM_GLOBAL(const uint8_t)*)TB, c, n, m, dlo, K, searchReverse, charA, cells, piece);
    }
    tmp.used = mark;
  } else {
    XM_LDS(uint32_t)* const W = (XM_LDS(uint32_t)*)region;
    XM_LDS(uint8_t)* const TB = (XM_LDS(uint8_t)*)(region + (XM_BOUND_KMAX + 1) * 4);
    if (coop) {
      // the window in search order, a base per lane and round
      for (int j = boundLaneOfEight(); j < m; j += 8) TB[j] = charB(searchReverse ? m - 1 - j : j);
      rejected = boundSweep<false, 3>(W, (XM_LDS(const uint8_t)*)TB, c, n, m, dlo, K, searchReverse, charA, cells, piece);
    } else {
      // the window in search order (eight loads in flight per round)
      for (int j0 = 0; j0 < m; j0 += 8) {
        uint8_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const int j = imin(j0 + k, m - 1); v[k] = charB(searchReverse ? m - 1 - j : j); }
#pragma unroll
        for (int k = 0; k < 8; k++) if (j0 + k < m) TB[j0 + k] = v[k];
      }
      rejected = boundSweep<false, 0>(W, (XM_LDS(const uint8_t)*)TB, c, n, m, dlo, K, searchReverse, charA, cells, piece);
    }
  }
  return rejected;
}

}  // namespace xm
