// xmapper-hip device core, wave-per-read form: ONE WAVEFRONT ALIGNS ONE READ.
//
// The read, its HashBlock pyramid, the vote counters, the staged hit lists, the candidate lists and the accepted alignments
// live in the wave's share of the CU's local data share (WaveLds); the index and the reference are read from HBM.  Control flow is
// wave-uniform (every lane follows the same path on the same values); the data-parallel steps are spread over the 64 lanes:
//   - pyramid levels are built lazily, one window of <= 64 block positions per round, one lane per block (M/HashBlock_ParentRow.java:69-127);
//   - a bucket's positions are decoded and flank-voted one hit per lane (M/Counting_HashBlockPath.java:98-153) and staged in LDS, the
//     order-dependent counter updates then consume the staged list in bucket order;
//   - the ungapped alignment compares 64 bases per round and adds the penalties in index order (M/StraightAligner.java:73-94).
// What the reference does per read is unchanged: this file restates M/HashBlockPath.java, M/Counting_HashBlockPath.java,
// M/HashBlockMatch_Counter.java, M/HashBlockPaths_Counter.java, M/QueryMatch_Aligner.java, M/StraightAligner.java and
// M/AlignerWorker.java:306-644 over the LDS-resident structures; the gapped chain (HashBlock_Aligner -> BlockAligner -> PathAligner)
// is entered through xm_extend.h.  Reads this form does not take (ambiguity codes in the read, mates longer than WV_MAXLEN, overlapping
// mates, a structure that outgrows its LDS capacity) leave with XM_ST_WAVE_FALLBACK and are aligned by the lane-per-read kernel.
//
// Source form: "uniform" code is written once and executed by every lane; lane-parallel steps are WV_PAR ... WV_ENDPAR regions in
// which `wl` is the lane.  On the GPU a region is straight-line code of the wave (wl = lane id); in the host simulation of the
// test tier (tests/hostsim) a region is a loop over the 64 lanes and uniform code runs once.  Values a lane keeps from one region
// to the next are declared with WV_VAR.  Cross-lane operations (ballot, broadcast) sit between regions.
#pragma once
#include "xm_worker.h"

namespace xm {

constexpr int32_t XM_ST_WAVE_FALLBACK = 8;  // the wave form does not take this read: the lane-per-read passes align it
constexpr int32_t XM_ST_WAVE_SEARCH = 10;   // chain tier: a PathAligner search request is waiting in the read's memo (search kernel, then the read runs again)
constexpr int32_t XM_ST_WAVE_GAPPED = 9;    // light tier: the read needs the heavy tier of the wave form (gapped chain, or a structure outgrew the light capacities)

#if defined(__HIP_DEVICE_COMPILE__)
#define XM_LDSP(T) T __attribute__((address_space(3)))
#define WV_VAR(T, name) T name
#define WV(name) name
#define WV_PAR for (int wl = (int)__lane_id(), wv_once_ = 1; wv_once_; wv_once_ = 0) {
#define WV_ENDPAR } __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#define WV_LANE0 if ((int)__lane_id() == 0)
XM_INL unsigned long long wvBallot(bool p) { return __ballot(p ? 1 : 0); }
#define WV_BALLOT(name) wvBallot((name) != 0)
XM_INL int wvBcastI(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
#define WV_BCAST_I(name, lane) wvBcastI((int)(name), (lane))
XM_INL int wvUni(int v) { return __builtin_amdgcn_readfirstlane(v); }
XM_INL void wvFence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
#else
#define XM_LDSP(T) T
#define WV_VAR(T, name) T name[64]
#define WV(name) name[wl]
#define WV_PAR for (int wl = 0; wl < 64; wl++) {
#define WV_ENDPAR }
#define WV_LANE0
template <typename T>
XM_INL unsigned long long wvBallotHost(const T* v) { unsigned long long m = 0; for (int i = 0; i < 64; i++) if (v[i] != 0) m |= 1ull << i; return m; }
#define WV_BALLOT(name) wvBallotHost(name)
#define WV_BCAST_I(name, lane) ((int)(name)[lane])
XM_INL int wvUni(int v) { return v; }
XM_INL void wvFence() {}
#endif

// ---------------------------------------------------------------- capacities of one wave's LDS share
constexpr int WV_MAXWIN = 4;                 // windows of 64 block positions per mate
constexpr int WV_MAXLEN = WV_MAXWIN * 64;    // longest mate the wave form takes
constexpr int WV_MAXLEVELS = 12;             // stored pyramid levels 1 .. WV_MAXLEVELS-1
constexpr int WV_MAXBLOCKS = 8;              // AlignedBlocks per SequenceAlignment
constexpr int WV_MAXHITS = 64;               // staged hits per round
// A configuration fixes the capacities of one kernel instance.  The light tiers are sized for what nearly every read needs
// (measured high-water marks: profiles/r02/NOTES.md); a read that outgrows one leaves with XM_ST_WAVE_FALLBACK.
struct WCfgLightSE { static constexpr bool kLast = false; static constexpr int kTier = 0, kMates = 1, kChunks = 6, kCounters = 8, kHistory = 12, kPending = 24, kQM = 4, kGood = 3, kPool = 12, kMRef = 1, kMQ = 1, kCountMap = 1, kPieces = 1, kPiecePool = 1; };
struct WCfgLightPE { static constexpr bool kLast = false; static constexpr int kTier = 0, kMates = 2, kChunks = 12, kCounters = 8, kHistory = 12, kPending = 24, kQM = 4, kGood = 3, kPool = 16, kMRef = 1, kMQ = 1, kCountMap = 1, kPieces = 1, kPiecePool = 1; };
// tier 1: + the gapped chain (PathAligner's searches go through the read's memo to the search kernel); WCfgHeavy: the same with the largest capacities
struct WCfgMidSE { static constexpr bool kLast = false; static constexpr int kTier = 1, kMates = 1, kChunks = 24, kCounters = 8, kHistory = 24, kPending = 48, kQM = 4, kGood = 3, kPool = 16, kMRef = 384, kMQ = WV_MAXLEN, kCountMap = 32, kPieces = 16, kPiecePool = 48; };
struct WCfgMidPE { static constexpr bool kLast = false; static constexpr int kTier = 1, kMates = 2, kChunks = 28, kCounters = 8, kHistory = 24, kPending = 48, kQM = 4, kGood = 3, kPool = 24, kMRef = 384, kMQ = WV_MAXLEN, kCountMap = 32, kPieces = 16, kPiecePool = 48; };
struct WCfgHeavy { static constexpr bool kLast = true; static constexpr int kTier = 1, kMates = 2, kChunks = 40, kCounters = 24, kHistory = 128, kPending = 96, kQM = 12, kGood = 6, kPool = 40, kMRef = 384, kMQ = WV_MAXLEN, kCountMap = 32, kPieces = 16, kPiecePool = 64; };

struct WCounter {  // M/HashBlockMatch_Counter.java + its SequenceMatch
  int32_t offset, contig, numMatches, numDistinctMismatches, lastMismatchedPosition, lastMatchedBlockId, historyProcessedIndex, priority;
  int16_t next, prev;
  uint8_t seqAId, mapSel, good, pad;
};
struct WHist { int16_t start, end; int32_t id; };             // what a counter needs of a history block
struct WQBlock { int32_t start, len, used, fwd, rev, id, flags; };
struct WHit { int32_t offset, contig; uint8_t seqAId, pass; uint16_t pad; };
template <int N>
struct WListT { int32_t id, n; int8_t items[N]; };
struct WSeqMatch { int32_t offset, contig, seqAId; };
struct WQMatch { int32_t n, priority, hint; WSeqMatch c[2]; };
struct WPathState {  // M/HashBlockPath.java
  int32_t batchIndex, curExists;
  int32_t curStart, curLen, curFwd, curRev, curFlags, curGapDir, curExtraGap;
  int32_t gapComputed, gapStatus;
  WQBlock gap;
  int32_t havePrev1, havePrev2, prevFwd1, prevFwd2;
};
template <class CFG>
struct WMateT {  // one Counting_HashBlockPath with its HashBlockPath and pyramid
  int32_t len;
  int32_t queryId, rcId;                 // sequence identity of the path's query / its reverse complement (mate*2 + rc)
  uint8_t codes[WV_MAXLEN];              // the path's query (mate 2: already reverse-complemented, M/AlignerWorker.java:317-318)
  uint16_t frontier[WV_MAXLEVELS];       // level k is known for block starts < frontier[k]
  uint8_t chunkOf[WV_MAXLEVELS][WV_MAXWIN];
  unsigned long long exists[WV_MAXLEVELS][WV_MAXWIN];
  WPathState path;
  WCounter counters[CFG::kCounters]; int32_t nCounters;
  int8_t good[CFG::kCounters]; int32_t nGood;
  int32_t foundGood, done;
  WHist history[CFG::kHistory]; int32_t nHistory;
  WQBlock pending[CFG::kPending]; int32_t pendHead, pendTail;
  int32_t numBlocksMatchingAnywhere, maxNonoverlappingBlockVisited, numNonoverlappingBlocksVisited, minNumDistinctMismatches;
  int32_t maxIndelLengthToConsider, nextBlockId;
  WListT<CFG::kCounters> hp, best, all;
};
struct WSeqAl {  // SequenceAlignment (blocks: WAligner::pool[firstBlock ..])
  int32_t nb, contig, firstBlock;
  uint8_t referenceReversed, seqAId; uint16_t pad;
  double totalPenalty, alignedPenalty;
};
struct WAl { int32_t nSeq, innerDistance; WSeqAl seq[2]; double spacingPenalty, overlapMultiplier, duplicationBonus, totalPenalty; };
template <class CFG>
struct WAlignerT {  // QueryMatch_Aligner
  double maxErrorRate, bestPenalty;        // parameters.MaxErrorRate is the one parameter align() changes (:39-48)
  int32_t nMates, queryLength, nGood, poolUsed, nBest, pad;
  WAl good[CFG::kGood];
  ABlock pool[CFG::kPool];
  int8_t bestIdx[CFG::kGood];
};

template <class CFG>
struct WMatcherT {  // HashBlock_Matcher: the k-mer codes of the reference window and of the query instead of per-section tables (xm_wave_chain.h)
  int32_t referenceStart, referenceLength, blockLength, sectionLength, maxSectionIndex, nSections;
  unsigned long long presentLo, presentHi;   // locations[i] != null
  int16_t rCode[CFG::kMRef];
  int16_t qCode[CFG::kMQ];
};
struct WPiece { int32_t nb, firstBlock, referenceReversed, pad; double totalPenalty, alignedPenalty; };

template <class CFG>
struct WaveLdsT {
  static constexpr bool kLast = CFG::kLast;
  static constexpr int kTier = CFG::kTier, kMates = CFG::kMates, kChunks = CFG::kChunks, kCounters = CFG::kCounters, kHistory = CFG::kHistory, kPending = CFG::kPending, kQM = CFG::kQM,
                       kGood = CFG::kGood, kPool = CFG::kPool, kMRef = CFG::kMRef, kMQ = CFG::kMQ, kCountMap = CFG::kCountMap, kPieces = CFG::kPieces, kPiecePool = CFG::kPiecePool;
  int32_t status, nMates, listIdCounter, nChunksUsed;
  int32_t why, tier;  // where a read left the wave form (diagnostics); tier of the kernel
  int32_t searchCursor, padSearch;  // how many pathAlign calls this run of the read has made
  int32_t mateLen[2];
  double expectedInner, deviation;
  WMateT<CFG> m[CFG::kMates];
  // pyramid blocks, 64 block positions per chunk, one array per field (a lane per position: conflict-free)
  int32_t chunkFwd[CFG::kChunks][64], chunkRev[CFG::kChunks][64];
  uint32_t chunkMeta[CFG::kChunks][64];  // bits 0-9 length, 10-17 extraGapmerLength, 18-21 merge flags, 22-23 gapDirection + 1
  WHit hits[WV_MAXHITS];
  // HashBlockPaths_Counter
  int32_t maxOffsetBetweenComponents, foundNonemptyResult, havePrevious, prevListId[2];
  WQMatch assembled[CFG::kQM]; int32_t nAssembled;
  WQMatch filtered[CFG::kQM]; int32_t nFiltered;
  WAlignerT<CFG> al[CFG::kMates];
  ABlock candBlocks[2][WV_MAXBLOCKS];
  // the gapped chain (xm_wave_chain.h; tiers 1 and 2)
  WMatcherT<CFG> mt[3];                       // HashBlock_Matchers: slot A (outer HashBlock_Aligner), B (inner), T (temporaries)
  int32_t cmKeys[CFG::kCountMap], cmVals[CFG::kCountMap], cmN, padCm;  // CountMap of the running analyzePenalty
  WPiece pieces[2][CFG::kPieces];             // BlockAligner: the two piece lists
  ABlock piecePool[2][CFG::kPiecePool];
  ABlock scratchBlocks[WV_MAXBLOCKS];
};

// ---------------------------------------------------------------- search requests and results of a read (HBM), see wPathAlign in xm_wave_chain.h
constexpr int WV_MEMO_MAX = 16;  // searches of one read the memo holds (a read that needs more is left to the lane-per-read passes)
struct WSearchResult { int32_t ok, nb, status, nodesPut; double totalPenalty, alignedPenalty; ABlock blocks[WV_MAXBLOCKS]; };  // ok: 1 alignment, 0 null, -1 failed (status)
struct WSearchReq { int32_t seqAId, contig, qsStart, qsEnd, rsStart, rsEnd, predictedBestOffset, confident, startingInsertionStartFree, pad; double maxIns, maxDel, maxErrorRate; };
struct WMemo { int32_t count, pending, pad0, pad1; WSearchReq req; WSearchResult res[WV_MEMO_MAX]; };

struct WEnv {  // what a wave carries in registers
  IndexView ix;                 // by value: with the wave functions inlined these stay in scalar registers (kernel arguments)
  Params params;
  DevCounters* dc;
  const uint8_t* mateBase[2];   // the mates as given in the batch (HBM), for the gapped chain
  int32_t tier;                 // 0 light (no gapped chain), 1 heavy
  const Caps* caps;             // heavy tier: scratch capacities of the gapped chain
  Arena* tmp;                   // (unused by the wave tiers)
  WMemo* memo;                  // chain tiers: the read's memo (searches that outgrow the inline capacities)
  struct WSNode* searchNodes;   // chain tiers: the wave's node payload buffer in HBM (WSearchLdsInline::kNodes entries)
};

#define WL_T XM_LDSP(LDS)*
// wave functions are inlined into their kernel: a call sends the callee-saved registers through scratch memory, and that per read
#define WV_FN XM_INL
// XM_WAVE_PROFILE builds: shader-clock ticks of a scope, summed into DevCounters::t[slot] (every lane keeps its own copy; lane 0's is published)
#if defined(XM_WAVE_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
struct WTimer {
  DevCounters* dc; int slot; unsigned long long t0;
  XM_INL WTimer(DevCounters* d, int s) : dc(d), slot(s), t0(clock64()) {}
  XM_INL ~WTimer() { if (dc) dc->t[slot] += clock64() - t0; }
};
#define WV_TIMER(e, slot) WTimer wtimer_##slot((e).dc, slot)
#else
#define WV_TIMER(e, slot) do { } while (0)
#endif
enum { WT_TOTAL = 0, WT_WALK = 2, WT_STEP = 3, WT_UNGAPPED = 4, WT_HITS = 5, WT_CHAIN = 6, WT_CONFIDENT = 10, WT_ALIGNMATCH = 11, WT_MATEINIT = 12, WT_WRITE = 13, WT_OPTIMISTIC = 14 };
// a structure outgrew its capacity: the light tier hands the read to the heavy tier, the heavy tier to the lane-per-read kernel
template <class LDS>
XM_INL int32_t wOverflowStatus(WL_T L) { return LDS::kLast ? XM_ST_WAVE_FALLBACK : XM_ST_WAVE_GAPPED; }

// ---------------------------------------------------------------- sequences
template <class LDS>
XM_INL uint8_t wSeqAt(WL_T L, int seqAId, int i) {  // Sequence.encodedCharAt for query identity seqAId (mate*2 + rc)
  const int mi = seqAId >> 1;
  if (seqAId == L->m[mi].queryId) return L->m[mi].codes[i];
  return bpComplement(L->m[mi].codes[L->m[mi].len - 1 - i]);
}
XM_INL uint8_t wRefAt(const IndexView& ix, int contig, bool rc, int i) {  // reference contig or its reverse complement
  XM_GLOBAL(const uint8_t)* const g = (XM_GLOBAL(const uint8_t)*)(ix.refCodes + ix.contigStart[contig]);
  return rc ? bpComplement(g[ix.contigLen[contig] - 1 - i]) : g[i];
}

// ---------------------------------------------------------------- pyramid: lazy, windowed, one lane per block
struct WBlock { int32_t start, len, fwd, rev, flags, gapDir, extraGap; };
XM_INL uint32_t wPackMeta(const WBlock& b) { return (uint32_t)b.len | ((uint32_t)b.extraGap << 10) | ((uint32_t)b.flags << 18) | ((uint32_t)(b.gapDir + 1) << 22); }
XM_INL void wUnpackMeta(uint32_t v, WBlock& b) { b.len = (int)(v & 1023u); b.extraGap = (int)((v >> 10) & 255u); b.flags = (int)((v >> 18) & 15u); b.gapDir = (int)((v >> 22) & 3u) - 1; }

template <class LDS>
XM_INL WBlock wBlockAt(WL_T L, int mi, int level, int pos) {  // the block of `level` starting at pos (caller knows it exists)
  WBlock b;
  b.start = pos;
  if (level == 0) {
    const PBlock p = level0Block(L->m[mi].codes[pos], pos);
    b.len = 1; b.fwd = p.fwd; b.rev = p.rev; b.flags = p.flags; b.gapDir = 0; b.extraGap = 0;
    return b;
  }
  const int ch = L->m[mi].chunkOf[level][pos >> 6];
  b.fwd = L->chunkFwd[ch][pos & 63];
  b.rev = L->chunkRev[ch][pos & 63];
  wUnpackMeta(L->chunkMeta[ch][pos & 63], b);
  return b;
}
// first block start > pos among the known part of `level` that is <= lim, or -1
template <class LDS>
XM_INL int wNextExisting(WL_T L, int mi, int level, int pos, int lim) {
  if (level == 0) return pos + 1 <= lim ? pos + 1 : -1;
  int p = pos + 1;
  while (p <= lim) {
    const int w = p >> 6;
    unsigned long long word = L->m[mi].exists[level][w] >> (p & 63);
    if (word) { const int q = p + __builtin_ctzll(word); return q <= lim ? q : -1; }
    p = (w + 1) << 6;
  }
  return -1;
}

// M/HashBlock.java:20-44,192-259 over WBlocks
XM_INL WBlock wMergeBlocks(const WBlock& Lb, const WBlock& Rb) {
  WBlock b;
  b.start = Lb.start;
  b.len = Rb.start + Rb.len - Lb.start;
  b.fwd = mergeHash(Lb.len, Lb.fwd, Rb.len, Rb.fwd);
  b.rev = mergeHash(Rb.len, Rb.rev, Lb.len, Lb.rev);
  const int anchor = (Lb.fwd != Rb.rev) ? ((Lb.fwd > Rb.rev) ? 2 : 1) : 0;
  const int fr = (b.fwd < b.rev) ? 0 : ((b.fwd == b.rev) ? 1 : 2);
  const int lc = (Lb.len < Rb.len) ? 0 : ((Lb.len == Rb.len) ? 1 : 2);
  const int lBits = (Lb.flags >> 2) & 3, rBits = (Rb.flags >> 2) & 3;
  const int aBits = (anchor == 2) ? rBits : lBits, oBits = (anchor == 2) ? lBits : rBits;
  const uint8_t rule = mergeRuleLookup((((anchor * 3 + fr) * 4 + aBits) * 4 + oBits) * 3 + lc);
  b.flags = rule & 15;
  b.gapDir = (int)(rule >> 4) - 1;
  b.extraGap = (Lb.len + Rb.len - b.len) / 4;
  return b;
}

// Make level `level` known for block starts < upTo (clipped to the mate's length).  The block of level k at p is
// merge(block of level k-1 at p, the next block of level k-1 after p) when those two touch or overlap and one of them asks
// for the merge (M/HashBlock_ParentRow.java:69-127,200-208); a block of level k-1 is at most 2^(k-1) long, which bounds how
// much of the level below a window needs.
template <class LDS>
WV_FN void wPyrEnsure(WL_T L, int mi, int level, int upTo) {
  const int len = L->m[mi].len;
  if (level <= 0) return;
  if (level >= WV_MAXLEVELS) { { L->status = wOverflowStatus(L); L->why = 1; } return; }
  // how far each level below must be known: a block of level k at p reaches into level k-1 as far as p + 2^(k-1)
  // (need of level k = upTo + 2^k + 2^(k+1) + ... + 2^(level-1), clipped to the mate)
  for (int k = 1; k <= level; k++) {
    const int needK = imin(len, upTo + (1 << level) - (1 << k));
    while ((int)L->m[mi].frontier[k] < needK) {
      const int a = L->m[mi].frontier[k];
      const int w = a >> 6;
      const int b = imin(needK, imin((w + 1) << 6, len));  // one window at a time
      if (L->m[mi].chunkOf[k][w] == 0xFF) {
        if (L->nChunksUsed >= LDS::kChunks) { { L->status = wOverflowStatus(L); L->why = 2; } return; }
        L->m[mi].chunkOf[k][w] = (uint8_t)L->nChunksUsed;
        L->nChunksUsed = L->nChunksUsed + 1;
      }
      wvFence();
      const int ch = L->m[mi].chunkOf[k][w];
      WV_VAR(int, made);
      WV_PAR
        WV(made) = 0;
        const int p = (w << 6) + wl;
        if (p < a || p >= b) continue;
        if (k > 1 && !((L->m[mi].exists[k - 1][p >> 6] >> (p & 63)) & 1ull)) continue;
        const WBlock B = wBlockAt(L, mi, k - 1, p);
        const int lim = imin(B.start + B.len, len - 1);
        const int q = wNextExisting(L, mi, k - 1, p, lim);
        if (q < 0) continue;
        const WBlock R = wBlockAt(L, mi, k - 1, q);
        if (!((B.flags & F_RMR) || (R.flags & F_RML))) continue;
        const WBlock nb = wMergeBlocks(B, R);
        if (nb.len > 1023 || nb.extraGap > 255 || nb.extraGap < 0) continue;  // (cannot happen for mates <= WV_MAXLEN)
        L->chunkFwd[ch][wl] = nb.fwd;
        L->chunkRev[ch][wl] = nb.rev;
        L->chunkMeta[ch][wl] = wPackMeta(nb);
        WV(made) = 1;
      WV_ENDPAR
      const unsigned long long mask = WV_BALLOT(made);
      L->m[mi].exists[k][w] = L->m[mi].exists[k][w] | mask;
      L->m[mi].frontier[k] = (uint16_t)b;
      wvFence();
    }
  }
}

// HashBlock_Row.get(index): the block of `level` that starts exactly at pos
template <class LDS>
XM_INL bool wPyrGet(WL_T L, int mi, int level, int pos, WBlock& out) {
  if (pos < 0 || pos >= L->m[mi].len) return false;
  if (level == 0) { out = wBlockAt(L, mi, 0, pos); return true; }
  if (level >= WV_MAXLEVELS) { { L->status = wOverflowStatus(L); L->why = 3; } return false; }
  if ((int)L->m[mi].frontier[level] <= pos) { wPyrEnsure(L, mi, level, pos + 9); if (L->status) return false; }
  if (!((L->m[mi].exists[level][pos >> 6] >> (pos & 63)) & 1ull)) return false;
  out = wBlockAt(L, mi, level, pos);
  return true;
}
// HashBlock_Row.getAfter(position): the first block of `level` that starts after pos
template <class LDS>
XM_INL bool wPyrGetAfter(WL_T L, int mi, int level, int pos, WBlock& out) {
  const int len = L->m[mi].len;
  if (level == 0) {
    if (pos + 1 >= len) return false;
    out = wBlockAt(L, mi, 0, pos + 1 < 0 ? 0 : pos + 1);
    return true;
  }
  if (level >= WV_MAXLEVELS) { { L->status = wOverflowStatus(L); L->why = 4; } return false; }
  int from = pos;
  while (true) {
    const int f = L->m[mi].frontier[level];
    if (f > from + 1) {
      const int q = wNextExisting(L, mi, level, from, f - 1);
      if (q >= 0) { out = wBlockAt(L, mi, level, q); return true; }
      from = f - 1;
    }
    if (f >= len) return false;
    wPyrEnsure(L, mi, level, imax(f, from + 1) + 40);
    if (L->status) return false;
  }
}

// ---------------------------------------------------------------- gapmer (M/HashBlock.java:67-150) over the LDS copy of the mate
template <class LDS>
XM_INL int wWithGapAndExtension(WL_T L, int mi, const WBlock& b, WQBlock& out) {
  int targetExtraLength = b.len;
  const int32_t mx = b.fwd > b.rev ? b.fwd : b.rev;
  targetExtraLength += jabs(mx) % 3;
  targetExtraLength += b.extraGap;
  const int gapLength = b.len / 2;
  const int extensionLength = targetExtraLength - gapLength;
  out.flags = 0;
  if (b.gapDir == 0) {
    out.start = b.start; out.len = b.len; out.used = b.len; out.fwd = b.fwd; out.rev = b.rev; out.flags = b.flags;
    return 1;
  }
  int32_t extensionHash = 0;
  if (b.gapDir < 0) {
    const int extensionEnd = b.start - gapLength;
    const int extensionStart = extensionEnd - extensionLength;
    if (extensionStart < 0) return 0;
    for (int i = extensionEnd - 1; i >= extensionStart; i--) {
      extensionHash = jmul(extensionHash, 7654337);
      extensionHash = jadd(extensionHash, gapmerCharCode(L->m[mi].codes[i]));
    }
    out.start = extensionStart;
    out.len = extensionLength + gapLength + b.len;
  } else {
    const int extensionStart = b.start + b.len + gapLength;
    const int extensionEnd = extensionStart + extensionLength;
    if (extensionEnd > L->m[mi].len) return 0;
    for (int i = extensionStart; i < extensionEnd; i++) {
      extensionHash = jmul(extensionHash, 7654337);
      extensionHash = jadd(extensionHash, gapmerCharCode(bpComplement(L->m[mi].codes[i])));
    }
    out.start = b.start;
    out.len = b.len + gapLength + extensionLength;
  }
  out.fwd = jadd(b.fwd, extensionHash);
  out.rev = jadd(b.rev, extensionHash);
  out.used = b.len + extensionLength;
  return 2;
}

XM_INL bool wqPrimaryPolarity(const WQBlock& b) {  // M/HashBlock.java:329-334
  const bool rml = (b.flags & F_RML) != 0, rmr = (b.flags & F_RMR) != 0;
  if (rml != rmr) return rml;
  return b.fwd >= b.rev;
}
XM_INL int32_t wqLookupKey(const WQBlock& b) { return wqPrimaryPolarity(b) ? b.fwd : b.rev; }

// The table descriptors (one per gapmer length) are read on every probe: the kernel keeps a copy of the first WV_TABLECACHE in LDS.
constexpr int WV_TABLECACHE = 160;
#if defined(__HIP_DEVICE_COMPILE__)
__shared__ Table xm_wave_tables[WV_TABLECACHE];
XM_INL void xmWaveLoadTables(const IndexView& ix) {  // every thread of the block, before the block's first barrier
  const int n = imin(ix.maxHashedLength + 1, WV_TABLECACHE);
  for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) {
    xm_wave_tables[i].capacity = ix.tables[i].capacity; xm_wave_tables[i].maxCount = ix.tables[i].maxCount;
    xm_wave_tables[i].offBase = ix.tables[i].offBase; xm_wave_tables[i].posBase = ix.tables[i].posBase;
  }
}
XM_INL Table wTable(const IndexView& ix, int used) {
  Table t;
  if (used < WV_TABLECACHE) { t.capacity = xm_wave_tables[used].capacity; t.maxCount = xm_wave_tables[used].maxCount; t.offBase = xm_wave_tables[used].offBase; t.posBase = xm_wave_tables[used].posBase; }
  else t = ix.tables[used];
  return t;
}
#else
XM_INL void xmWaveLoadTables(const IndexView&) {}
XM_INL Table wTable(const IndexView& ix, int used) { return ix.tables[used]; }
#endif

// M/Readable_HashBlock_Database.java:72-80 + M/PackedMap.java:228-236: one 8-byte header probe
template <class LDS>
XM_INL int wNumMatchesLowerBound(WL_T L, const WEnv& e, const WQBlock& b) {
  const IndexView& ix = e.ix;
  if (b.used < ix.minInterestingSize) return INT32_MAX;
  if (b.used > ix.maxHashedLength) { L->status = XM_ST_NEED_GROW; return INT32_MAX; }
  const Table t = wTable(ix, b.used);
  const uint32_t k = packedKey(&t, wqLookupKey(b));
  if (e.dc) e.dc->headerProbes++;
  if (ix.lines32 || ix.lines64) {  // bucket lines (IndexView): the header is word 0 of the bucket's line
    const uint32_t h = ix.lines64 ? (uint32_t)ix.lines64[(t.offBase + k) * 8] : ix.lines32[(t.offBase + k) * 8];
    return (h & XM_OVERFULL) ? INT32_MAX : (int)h;
  }
  const uint32_t* off = ix.bucketOff + t.offBase + k;
  const uint32_t o0 = off[0], o1 = off[1];
  if (o0 & XM_OVERFULL) return INT32_MAX;
  return (int)((o1 & ~XM_OVERFULL) - (o0 & ~XM_OVERFULL));
}
XM_INL int wDbMaxNumMatchesAllowed(const IndexView& ix, const WQBlock& b) {  // :82-90 (the table of a hashed length always exists here)
  if (b.used < ix.minInterestingSize) return -1;
  if (b.used > ix.maxHashedLength) return 0;
  return wTable(ix, b.used).maxCount;
}

// ---------------------------------------------------------------- HashBlockPath (M/HashBlockPath.java)
template <class LDS>
XM_INL void wPathSetCur(WL_T L, int mi, const WBlock& b) {
  XM_LDSP(WPathState)* p = &L->m[mi].path;
  p->curStart = b.start; p->curLen = b.len; p->curFwd = b.fwd; p->curRev = b.rev; p->curFlags = b.flags; p->curGapDir = b.gapDir; p->curExtraGap = b.extraGap;
  p->gapComputed = 0;
}
template <class LDS>
XM_INL WBlock wPathCur(WL_T L, int mi) {
  XM_LDSP(WPathState)* p = &L->m[mi].path;
  WBlock b;
  b.start = p->curStart; b.len = p->curLen; b.fwd = p->curFwd; b.rev = p->curRev; b.flags = p->curFlags; b.gapDir = p->curGapDir; b.extraGap = p->curExtraGap;
  return b;
}
template <class LDS>
XM_INL void wPathMoveRight(WL_T L, int mi) {  // :125-128
  WBlock nb;
  const bool ok = wPyrGetAfter(L, mi, L->m[mi].path.batchIndex, L->m[mi].path.curStart, nb);
  L->m[mi].path.curExists = ok ? 1 : 0;
  if (ok) wPathSetCur(L, mi, nb);
  L->m[mi].path.gapComputed = 0;
}
template <class LDS>
XM_INL void wPathMoveDown(WL_T L, int mi) {  // :99-108
  L->m[mi].path.batchIndex = L->m[mi].path.batchIndex - 1;
  wPathMoveRight(L, mi);
}
template <class LDS>
XM_INL void wPathMoveUpOrRight(WL_T L, int mi) {  // :111-122
  WBlock up;
  if (wPyrGet(L, mi, L->m[mi].path.batchIndex + 1, L->m[mi].path.curStart, up)) {
    L->m[mi].path.batchIndex = L->m[mi].path.batchIndex + 1;
    wPathSetCur(L, mi, up);
  } else {
    if (L->status) return;
    wPathMoveRight(L, mi);
  }
}
template <class LDS>
XM_INL bool wPathWithGap(WL_T L, const WEnv& e, int mi, WQBlock& out) {  // :197-203
  XM_LDSP(WPathState)* p = &L->m[mi].path;
  if (!e.ix.enableGapmers) {
    out.start = p->curStart; out.len = p->curLen; out.used = p->curLen; out.fwd = p->curFwd; out.rev = p->curRev; out.flags = p->curFlags; out.id = -1;
    return true;
  }
  if (!p->gapComputed) {
    WQBlock g;
    g.start = g.len = g.used = g.fwd = g.rev = g.flags = 0;
    const int st = wWithGapAndExtension(L, mi, wPathCur(L, mi), g);
    p->gapStatus = st;
    p->gap.start = g.start; p->gap.len = g.len; p->gap.used = g.used; p->gap.fwd = g.fwd; p->gap.rev = g.rev; p->gap.flags = g.flags; p->gap.id = -1;
    p->gapComputed = 1;
  }
  if (p->gapStatus == 0) return false;
  out.start = p->gap.start; out.len = p->gap.len; out.used = p->gap.used; out.fwd = p->gap.fwd; out.rev = p->gap.rev; out.flags = p->gap.flags; out.id = -1;
  return true;
}
template <class LDS>
XM_INL int wPathMaxNumMatchesAllowed(WL_T L, const WEnv& e, int mi, const WQBlock& b) {  // :205-219
  if (b.len >= L->m[mi].len / 6) return wDbMaxNumMatchesAllowed(e.ix, b);
  if (b.flags & F_RMR) return 5;
  return b.used + 1;
}
// advanceToNextPosition :143-195 (mates with ambiguity codes never come here, so there are no multi blocks to skip)
template <class LDS>
WV_FN bool wPathAdvance(WL_T L, const WEnv& e, int mi) {
  const int singleLen = L->m[mi].path.curLen;
  const bool gapmers = e.ix.enableGapmers != 0;
  if (maxGapmerNumBasepairsUsed(singleLen) < e.ix.minInterestingSize && gapmers) {
    wPathMoveUpOrRight(L, mi);
  } else {
    WQBlock ext;
    if (wPathWithGap(L, e, mi, ext)) {
      const int numMatches = wNumMatchesLowerBound(L, e, ext);
      if (L->status) return false;
      if (numMatches < 6) {
        if (L->m[mi].path.batchIndex > 0) wPathMoveDown(L, mi); else wPathMoveRight(L, mi);
      } else {
        if (numMatches > wPathMaxNumMatchesAllowed(L, e, mi, ext)) wPathMoveUpOrRight(L, mi);
        else wPathMoveRight(L, mi);
      }
    } else {
      const int typical = singleLen * 3 / 2;
      if (typical <= e.ix.minInterestingSize && gapmers) wPathMoveUpOrRight(L, mi);
      else { if (L->m[mi].path.batchIndex > 0) wPathMoveDown(L, mi); else wPathMoveRight(L, mi); }
    }
  }
  return L->m[mi].path.curExists && L->status == 0;
}
// getNextInterestingBlock :27-50 (+ getNextBlockWithGoodNumberOfMatches :68-96, recentlySeen :52-65)
template <class LDS>
WV_FN bool wPathNextInterestingBlock(WL_T L, const WEnv& e, int mi, WQBlock& out) {
  WV_TIMER(e, WT_WALK);
  if (!L->m[mi].path.curExists) return false;
  while (true) {
    if (!wPathAdvance(L, e, mi)) return false;
    WQBlock ext;
    if (!wPathWithGap(L, e, mi, ext)) continue;
    const int n = wNumMatchesLowerBound(L, e, ext);
    if (L->status) return false;
    if (!(n <= wPathMaxNumMatchesAllowed(L, e, mi, ext))) continue;
    XM_LDSP(WPathState)* p = &L->m[mi].path;
    bool seen = false;
    if (p->havePrev1 && ext.fwd == p->prevFwd1) seen = true;
    else if (p->havePrev2 && ext.fwd == p->prevFwd2) seen = true;
    p->havePrev2 = p->havePrev1;
    p->prevFwd2 = p->prevFwd1;
    p->havePrev1 = 1;
    p->prevFwd1 = ext.fwd;
    if (seen) continue;
    ext.id = L->m[mi].nextBlockId;
    L->m[mi].nextBlockId = ext.id + 1;
    out = ext;
    return true;
  }
}

// ---------------------------------------------------------------- Counting_HashBlockPath: counters (M/HashBlockMatch_Counter.java)
template <class LDS>
XM_INL void wCounterUpdate(WL_T L, const WEnv& e, int mi, int ci) {  // :41-46,74-88
  XM_LDSP(WCounter)* k = &L->m[mi].counters[ci];
  const int nHistory = L->m[mi].nHistory;
  int idx = k->historyProcessedIndex;
  if (idx >= nHistory) return;
  int nd = k->numDistinctMismatches, lastPos = k->lastMismatchedPosition;
  const int lastId = k->lastMatchedBlockId, off = k->offset, refLen = e.ix.contigLen[k->contig];
  for (; idx < nHistory; idx++) {
    const int bStart = L->m[mi].history[idx].start, bEnd = L->m[mi].history[idx].end, bId = L->m[mi].history[idx].id;
    if (bId != lastId && bStart >= lastPos && off + bEnd <= refLen) { nd++; lastPos = bEnd; }
  }
  k->numDistinctMismatches = nd; k->lastMismatchedPosition = lastPos; k->historyProcessedIndex = idx;
}
template <class LDS>
XM_INL int wCounterNumDistinctMismatches(WL_T L, const WEnv& e, int mi, int ci) { wCounterUpdate(L, e, mi, ci); return L->m[mi].counters[ci].numDistinctMismatches; }
template <class LDS>
XM_INL void wDeclareGood(WL_T L, const WEnv& e, int mi, int ci) {  // M/Counting_HashBlockPath.java:280-285
  if (L->m[mi].counters[ci].good) return;
  const int n = L->m[mi].nGood;
  if (n >= LDS::kCounters) { { L->status = wOverflowStatus(L); L->why = 5; } return; }
  L->m[mi].good[n] = (int8_t)ci;
  L->m[mi].nGood = n + 1;
  L->m[mi].counters[ci].good = 1;
  L->m[mi].counters[ci].priority = wCounterNumDistinctMismatches(L, e, mi, ci);  // setGood
}
template <class LDS>
XM_INL void wCompAddMatch(WL_T L, const WEnv& e, int mi, int ci, const WQBlock& qb, int queryBlockNumMatches, const WSeqMatch& fm) {  // :254-277
  XM_LDSP(WCounter)* k = &L->m[mi].counters[ci];
  const int nm = k->numMatches + 1;
  k->numMatches = nm;
  k->lastMatchedBlockId = qb.id;
  wCounterUpdate(L, e, mi, ci);
  if (nm <= 1) {
    if (nm == 1) {
      L->m[mi].foundGood = 1;
      wDeclareGood(L, e, mi, ci);
    } else if (queryBlockNumMatches <= qb.len) {
      const int distanceFromStart = fm.offset;
      const int distanceFromEnd = e.ix.contigLen[fm.contig] - (fm.offset + L->mateLen[fm.seqAId >> 1]);
      if (imin(distanceFromStart, distanceFromEnd) < 0) wDeclareGood(L, e, mi, ci);
    }
  }
}
template <class LDS>
WV_FN void wCompUpdateMatches(WL_T L, const WEnv& e, int mi, const WSeqMatch& m, const WQBlock& qb, int queryBlockNumMatches) {  // :193-252
  const uint8_t mapSel = (m.seqAId & 1) ? 0 : 1;  // (sic) reversed matches are filed under "forward", M/Counting_HashBlockPath.java:197-200
  int cur = -1, lower = -1, higher = -1, lowerOff = 0, higherOff = 0;
  const int nC = L->m[mi].nCounters;
  for (int i = 0; i < nC; i++) {
    XM_LDSP(const WCounter)* k = &L->m[mi].counters[i];
    if (k->mapSel != mapSel || k->contig != m.contig) continue;
    const int ko = k->offset;
    if (ko == m.offset) { cur = i; break; }
    if (ko < m.offset) { if (lower < 0 || ko > lowerOff) { lower = i; lowerOff = ko; } }
    else { if (higher < 0 || ko < higherOff) { higher = i; higherOff = ko; } }
  }
  if (cur < 0) {
    if (nC >= LDS::kCounters) { { L->status = wOverflowStatus(L); L->why = 6; } return; }
    cur = nC;
    L->m[mi].nCounters = nC + 1;
    XM_LDSP(WCounter)* k = &L->m[mi].counters[cur];
    k->offset = m.offset; k->contig = m.contig; k->seqAId = (uint8_t)m.seqAId; k->mapSel = mapSel; k->good = 0;
    k->numMatches = 0;
    k->numDistinctMismatches = L->m[mi].numNonoverlappingBlocksVisited;
    k->lastMismatchedPosition = qb.start;
    k->lastMatchedBlockId = -2;
    k->historyProcessedIndex = L->m[mi].nHistory - 1;
    k->priority = 0;
    k->next = -1; k->prev = -1;
    const int maxIndel = L->m[mi].maxIndelLengthToConsider;
    if (lower >= 0 && iabs(lowerOff - m.offset) <= maxIndel) { k->prev = (int16_t)lower; L->m[mi].counters[lower].next = (int16_t)cur; }
    if (higher >= 0 && iabs(higherOff - m.offset) <= maxIndel) { k->next = (int16_t)higher; L->m[mi].counters[higher].prev = (int16_t)cur; }
  }
  const int prev = L->m[mi].counters[cur].prev;
  if (prev >= 0) wCompAddMatch(L, e, mi, prev, qb, queryBlockNumMatches, m);
  const int next = L->m[mi].counters[cur].next;
  if (next >= 0) wCompAddMatch(L, e, mi, next, qb, queryBlockNumMatches, m);
  bool updateThisOne = true;
  if ((prev >= 0 && L->m[mi].counters[prev].good) || (next >= 0 && L->m[mi].counters[next].good)) {
    if (!L->m[mi].counters[cur].good) updateThisOne = false;
  }
  if (updateThisOne) wCompAddMatch(L, e, mi, cur, qb, queryBlockNumMatches, m);
}
// counters of one map in (contig, offset) order: the next one after (lastContig, lastOffset), or -1
template <class LDS>
XM_INL int wNextCounterInOrder(WL_T L, int mi, int mapSel, int lastContig, int lastOffset, bool first) {
  int best = -1, bc = 0, bo = 0;
  const int nC = L->m[mi].nCounters;
  for (int i = 0; i < nC; i++) {
    XM_LDSP(const WCounter)* k = &L->m[mi].counters[i];
    if (k->mapSel != mapSel) continue;
    const int kc = k->contig, ko = k->offset;
    if (!first && (kc < lastContig || (kc == lastContig && ko <= lastOffset))) continue;
    if (best < 0 || kc < bc || (kc == bc && ko < bo)) { best = i; bc = kc; bo = ko; }
  }
  return best;
}
template <class LDS>
WV_FN void wTryEnsureGoodMatchCounter(WL_T L, const WEnv& e, int mi) {  // :291-308
  if (!L->m[mi].foundGood && L->m[mi].nCounters <= L->m[mi].len) {
    for (int mapSel = 0; mapSel < 2; mapSel++) {
      int lc = 0, lo = 0;
      bool first = true;
      while (true) {
        const int i = wNextCounterInOrder(L, mi, mapSel, lc, lo, first);
        if (i < 0) break;
        first = false; lc = L->m[mi].counters[i].contig; lo = L->m[mi].counters[i].offset;
        wDeclareGood(L, e, mi, i);
      }
    }
    L->m[mi].foundGood = 1;
  }
}

// getNextInterestingBlock :344-368
template <class LDS>
WV_FN bool wCompNextInterestingBlock(WL_T L, const WEnv& e, int mi, WQBlock& out) {
  L->m[mi].all.id = 0;  // previousAllPositions = null
  while (true) {
    WQBlock b;
    if (!wPathNextInterestingBlock(L, e, mi, b)) {
      if (L->status) return false;
      const int h = L->m[mi].pendHead;
      if (h >= L->m[mi].pendTail) return false;
      XM_LDSP(const WQBlock)* p = &L->m[mi].pending[h];
      out.start = p->start; out.len = p->len; out.used = p->used; out.fwd = p->fwd; out.rev = p->rev; out.id = p->id; out.flags = p->flags;
      L->m[mi].pendHead = h + 1;
      return true;
    }
    if (b.start < L->m[mi].maxNonoverlappingBlockVisited) {
      const int t = L->m[mi].pendTail;
      if (t >= LDS::kPending) { { L->status = wOverflowStatus(L); L->why = 7; } return false; }
      XM_LDSP(WQBlock)* p = &L->m[mi].pending[t];
      p->start = b.start; p->len = b.len; p->used = b.used; p->fwd = b.fwd; p->rev = b.rev; p->id = b.id; p->flags = b.flags;
      L->m[mi].pendTail = t + 1;
      continue;
    }
    out = b;
    return true;
  }
}

// step() :40-179.  The bucket's positions are decoded and flank-voted one per lane and staged in LDS (WaveLds::hits); the
// counters then take the staged hits in bucket order.
template <class LDS>
WV_FN bool wCompStep(WL_T L, const WEnv& e, int mi) {
  WV_TIMER(e, WT_STEP);
  if (L->m[mi].done) return false;
  const IndexView& ix = e.ix;
  WQBlock qb;
  int64_t first = 0;
  bool invert = false;
  int nHits;
  while (true) {  // getNextInterestingMatch :371-384 / Readable_HashBlock_Database.matchBlock :22-38
    if (!wCompNextInterestingBlock(L, e, mi, qb)) {
      if (L->status) return false;
      L->m[mi].done = 1;
      if (L->m[mi].numBlocksMatchingAnywhere < 1) wTryEnsureGoodMatchCounter(L, e, mi);
      return false;
    }
    if (qb.used < ix.minInterestingSize) continue;
    if (qb.used > ix.maxHashedLength) { L->status = XM_ST_NEED_GROW; return false; }
    const Table t = wTable(ix, qb.used);
    const uint32_t k = packedKey(&t, wqLookupKey(qb));
    if (e.dc) { e.dc->headerProbes++; e.dc->bucketFetches++; }
    if (ix.lines32 || ix.lines64) {
      const int64_t line = (t.offBase + k) * 8;
      const uint32_t h = ix.lines64 ? (uint32_t)ix.lines64[line] : ix.lines32[line];
      if (h & XM_OVERFULL) continue;
      nHits = (int)h;
      if (nHits > t.maxCount) continue;
      if (nHits <= XM_LINE_SLOTS) first = XM_LINE_FLAG | (line + 1);
      else first = t.posBase + (int64_t)(ix.bucketOff[t.offBase + k] & ~XM_OVERFULL);
    } else {
      const uint32_t* off = ix.bucketOff + t.offBase + k;
      const uint32_t o0 = off[0], o1 = off[1];
      if (o0 & XM_OVERFULL) continue;
      nHits = (int)((o1 & ~XM_OVERFULL) - (o0 & ~XM_OVERFULL));
      if (nHits > t.maxCount) continue;
      first = t.posBase + (int64_t)(o0 & ~XM_OVERFULL);
    }
    invert = !wqPrimaryPolarity(qb);
    if (e.dc) e.dc->hitsFetched += (unsigned long long)nHits;
    break;
  }
  {
    const int n = L->m[mi].nHistory;
    if (n >= LDS::kHistory) { { L->status = wOverflowStatus(L); L->why = 8; } return false; }
    L->m[mi].history[n].start = (int16_t)qb.start; L->m[mi].history[n].end = (int16_t)(qb.start + qb.len); L->m[mi].history[n].id = qb.id;
    L->m[mi].nHistory = n + 1;
  }
  const int qLen = L->m[mi].len;
  const int queryId = L->m[mi].queryId, rcId = L->m[mi].rcId;
  WV_TIMER(e, WT_HITS);
  for (int h0 = 0; h0 < nHits; h0 += WV_MAXHITS) {
    const int nRound = imin(WV_MAXHITS, nHits - h0);
    wvFence();
    WV_PAR
      if (wl >= nRound) continue;
      const int64_t enc = xmPositionAt(ix, first + h0 + wl);
      RefPos rp = decodePosition(ix, enc);
      const int refLen = ix.contigLen[rp.contig];
      if (invert) { rp.start = refLen - rp.start - qb.len; rp.rc ^= 1; }  // Readable_HashBlock_Database.reverseComplement :55-59
      int numMismatchedItems = 0, numMatchedItems = 0;
      for (int distance = 1; distance < 20; distance++) {  // :98-148
        int checkOffset = -distance;
        int queryIndex = qb.start + checkOffset;
        if (queryIndex >= 0 && queryIndex < qLen) {
          const int referenceIndex = rp.start + checkOffset;
          if (referenceIndex >= 0 && referenceIndex < refLen) {
            if (!bpCanMatch(L->m[mi].codes[queryIndex], wRefAt(ix, rp.contig, rp.rc != 0, referenceIndex))) numMismatchedItems++; else numMatchedItems++;
          }
        }
        checkOffset = qb.len - 1 + distance;
        queryIndex = qb.start + checkOffset;
        if (queryIndex >= 0 && queryIndex < qLen) {
          const int referenceIndex = rp.start + checkOffset;
          if (referenceIndex >= 0 && referenceIndex < refLen) {
            if (!bpCanMatch(L->m[mi].codes[queryIndex], wRefAt(ix, rp.contig, rp.rc != 0, referenceIndex))) numMismatchedItems++; else numMatchedItems++;
          }
        }
        if (numMatchedItems < numMismatchedItems) break;
        if (numMatchedItems >= numMismatchedItems + qb.used) break;
      }
      XM_LDSP(WHit)* hit = &L->hits[wl];
      hit->pass = numMismatchedItems > numMatchedItems ? 0 : 1;
      hit->contig = rp.contig;
      if (rp.rc) {  // :155-161
        const int reverseQueryBlockStart = qLen - (qb.start + qb.len);
        const int reverseReferenceBlockStart = refLen - (rp.start + qb.len);
        hit->offset = reverseReferenceBlockStart - reverseQueryBlockStart;
        hit->seqAId = (uint8_t)rcId;
      } else {
        hit->offset = rp.start - qb.start;
        hit->seqAId = (uint8_t)queryId;
      }
    WV_ENDPAR
    for (int h = 0; h < nRound; h++) {
      if (!L->hits[h].pass) continue;
      WSeqMatch fm;
      fm.offset = L->hits[h].offset; fm.contig = L->hits[h].contig; fm.seqAId = L->hits[h].seqAId;
      wCompUpdateMatches(L, e, mi, fm, qb, nHits);
      if (L->status) return false;
    }
  }
  if (qb.start >= L->m[mi].maxNonoverlappingBlockVisited) {
    L->m[mi].maxNonoverlappingBlockVisited = qb.start + qb.len;
    L->m[mi].numNonoverlappingBlocksVisited = L->m[mi].numNonoverlappingBlocksVisited + 1;
  }
  L->m[mi].numBlocksMatchingAnywhere = L->m[mi].numBlocksMatchingAnywhere + 1;
  L->m[mi].minNumDistinctMismatches = -1;
  return true;
}

// lists of counters (the reference hands out List objects and compares their identities, M/HashBlockPaths_Counter.java:116-133)
enum { WLIST_HP = 0, WLIST_BEST = 1, WLIST_ALL = 2 };
template <class LDS>
XM_INL auto wList(WL_T L, int mi, int which) { return which == WLIST_HP ? &L->m[mi].hp : (which == WLIST_BEST ? &L->m[mi].best : &L->m[mi].all); }

template <class LDS>
WV_FN int wCompFindGoodPositionsHavingPriorityUpTo(WL_T L, const WEnv& e, int mi, int priority) {  // :406-433 -> WLIST_HP
  while (true) {
    if (L->m[mi].numNonoverlappingBlocksVisited >= jadd(priority, 1)) break;
    if (!wCompStep(L, e, mi)) break;
  }
  if (L->status) return WLIST_HP;
  if (L->m[mi].hp.id != 0 && L->m[mi].hp.n == L->m[mi].nGood) return WLIST_HP;
  int n = 0;
  const int nGood = L->m[mi].nGood;
  for (int i = 0; i < nGood; i++) {
    const int ci = L->m[mi].good[i];
    if (L->m[mi].counters[ci].priority <= priority) L->m[mi].hp.items[n++] = (int8_t)ci;
  }
  L->m[mi].hp.n = n;
  L->listIdCounter = L->listIdCounter + 1;
  L->m[mi].hp.id = L->listIdCounter;
  return WLIST_HP;
}
template <class LDS>
WV_FN int wCompGetAllPositions(WL_T L, const WEnv& e, int mi) {  // :435-451 -> WLIST_ALL
  if (L->m[mi].all.id == 0) {
    int n = 0;
    for (int mapSel = 0; mapSel < 2; mapSel++) {
      int lc = 0, lo = 0;
      bool first = true;
      while (true) {
        const int i = wNextCounterInOrder(L, mi, mapSel, lc, lo, first);
        if (i < 0) break;
        first = false; lc = L->m[mi].counters[i].contig; lo = L->m[mi].counters[i].offset;
        L->m[mi].all.items[n++] = (int8_t)i;
      }
    }
    L->m[mi].all.n = n;
    L->listIdCounter = L->listIdCounter + 1;
    L->m[mi].all.id = L->listIdCounter;
  }
  return WLIST_ALL;
}
template <class LDS>
WV_FN int wCompGetBestMatches(WL_T L, const WEnv& e, int mi) {  // :471-493 (+ getNumGoodDistinctMismatches :457-469) -> WLIST_BEST
  L->listIdCounter = L->listIdCounter + 1;
  L->m[mi].best.id = L->listIdCounter;
  L->m[mi].best.n = 0;
  if (L->m[mi].numBlocksMatchingAnywhere < 1) return WLIST_BEST;
  const int nGood = L->m[mi].nGood;
  if (L->m[mi].minNumDistinctMismatches < 0) {
    int mn = L->m[mi].numNonoverlappingBlocksVisited - 1;
    for (int i = 0; i < nGood; i++) {
      const int cnt = wCounterNumDistinctMismatches(L, e, mi, L->m[mi].good[i]);
      if (mn >= cnt) mn = cnt;
    }
    L->m[mi].minNumDistinctMismatches = mn;
  }
  const int mn = L->m[mi].minNumDistinctMismatches;
  int n = 0;
  for (int i = 0; i < nGood; i++) {
    const int ci = L->m[mi].good[i];
    const int cnt = wCounterNumDistinctMismatches(L, e, mi, ci);
    if (cnt <= mn) L->m[mi].best.items[n++] = (int8_t)ci;
  }
  L->m[mi].best.n = n;
  return WLIST_BEST;
}

// ---------------------------------------------------------------- HashBlockPaths_Counter (M/HashBlockPaths_Counter.java)
template <class LDS>
XM_INL WSeqMatch wCounterMatch(WL_T L, int mi, int ci) {
  WSeqMatch m;
  m.offset = L->m[mi].counters[ci].offset; m.contig = L->m[mi].counters[ci].contig; m.seqAId = L->m[mi].counters[ci].seqAId;
  return m;
}
XM_INL int wSmStartB(const WSeqMatch& m) { return imax(0, m.offset); }
template <class LDS>
XM_INL int wSmEndB(WL_T L, const WEnv& e, const WSeqMatch& m) { return imin(m.offset + L->mateLen[m.seqAId >> 1], e.ix.contigLen[m.contig]); }
XM_INL bool wSmEquals(const WSeqMatch& a, const WSeqMatch& b) { return a.offset == b.offset && a.seqAId == b.seqAId && a.contig == b.contig; }

template <class LDS>
XM_INL void wStoreQMatch(XM_LDSP(WQMatch)* d, const WQMatch& q) {
  d->n = q.n; d->priority = q.priority; d->hint = q.hint;
  for (int i = 0; i < 2; i++) { d->c[i].offset = q.c[i].offset; d->c[i].contig = q.c[i].contig; d->c[i].seqAId = q.c[i].seqAId; }
}
template <class LDS>
XM_INL WQMatch wLoadQMatch(XM_LDSP(const WQMatch)* s) {
  WQMatch q;
  q.n = s->n; q.priority = s->priority; q.hint = s->hint;
  for (int i = 0; i < 2; i++) { q.c[i].offset = s->c[i].offset; q.c[i].contig = s->c[i].contig; q.c[i].seqAId = s->c[i].seqAId; }
  return q;
}

template <class LDS>
WV_FN void wPcMatchWithoutCache(WL_T L, const WEnv& e, const int* which) {  // :136-247 + assembleQueryMatches :249-265
  L->nAssembled = 0;
  if (L->nMates == 1) {
    auto l0 = wList(L, 0, which[0]);
    const int n = l0->n;
    for (int i = 0; i < n; i++) {
      if (L->nAssembled >= LDS::kQM) { { L->status = wOverflowStatus(L); L->why = 9; } return; }
      const int ci = l0->items[i];
      WQMatch q;
      q.n = 1; q.priority = L->m[0].counters[ci].priority; q.hint = 0; q.c[0] = wCounterMatch(L, 0, ci); q.c[1] = q.c[0];
      wStoreQMatch<LDS>(&L->assembled[L->nAssembled], q);
      L->nAssembled = L->nAssembled + 1;
    }
    return;
  }
  if constexpr (LDS::kMates < 2) { L->status = XM_ST_INTERNAL; return; } else {
  auto list0 = wList(L, 0, which[0]); auto list1 = wList(L, 1, which[1]);
  decltype(list0) lists[2] = {list0, list1};
  const bool lastComponentIsLargest = lists[0]->n <= lists[1]->n;
  const int firstComp = lastComponentIsLargest ? 0 : 1;
  const int secondComp = 1 - firstComp;
  const int nSecond = lists[secondComp]->n, nFirst = lists[firstComp]->n;
  for (int j = 0; j < nSecond; j++) {
    const int ci = lists[secondComp]->items[j];
    XM_LDSP(const WCounter)* k = &L->m[secondComp].counters[ci];
    const int querySequenceLength = L->mateLen[k->seqAId >> 1];
    const int maxReverseOffset = querySequenceLength / 2;
    const bool sequenceMatchReversed = (k->seqAId & 1) != 0;
    const bool queryMatchReversed = (sequenceMatchReversed == (secondComp % 2 == 0));
    const int offset = k->offset, kContig = k->contig;
    int searchStart, searchEnd;
    const bool otherSequenceExpectEarlier = (queryMatchReversed == lastComponentIsLargest);
    if (otherSequenceExpectEarlier) { searchStart = offset - maxReverseOffset; searchEnd = jadd(offset, L->maxOffsetBetweenComponents); }
    else { searchStart = offset - L->maxOffsetBetweenComponents; searchEnd = offset + maxReverseOffset; }
    if (searchStart > searchEnd) { L->status = XM_ST_INTERNAL; return; }  // TreeMap.subMap would throw
    // entries of the first component filed under the same (direction, contig) with offset in [searchStart, searchEnd], ascending
    int8_t nearby[LDS::kCounters];
    int nn = 0;
    for (int i = 0; i < nFirst; i++) {
      const int fi = lists[firstComp]->items[i];
      XM_LDSP(const WCounter)* f = &L->m[firstComp].counters[fi];
      const bool fRev = (f->seqAId & 1) != 0;
      const bool fQueryMatchReversed = (fRev == (firstComp % 2 == 0));
      if (fQueryMatchReversed != queryMatchReversed || f->contig != kContig) continue;
      const int fo = f->offset;
      if (fo < searchStart || fo > searchEnd) continue;
      int p = nn++;
      while (p > 0 && L->m[firstComp].counters[nearby[p - 1]].offset > fo) { nearby[p] = nearby[p - 1]; p--; }
      nearby[p] = (int8_t)fi;
    }
    const bool descending = queryMatchReversed && nn > 1;
    for (int t = 0; t < nn; t++) {
      const int fi = nearby[descending ? nn - 1 - t : t];
      const int c0 = lastComponentIsLargest ? fi : ci;
      const int c1 = lastComponentIsLargest ? ci : fi;
      if (L->nAssembled >= LDS::kQM) { { L->status = wOverflowStatus(L); L->why = 10; } return; }
      WQMatch q;
      q.n = 2;
      q.c[0] = wCounterMatch(L, 0, c0);
      q.c[1] = wCounterMatch(L, 1, c1);
      q.hint = wCounterNumDistinctMismatches(L, e, 0, c0) < wCounterNumDistinctMismatches(L, e, 1, c1) ? 1 : 0;
      // countPriority :314-334
      const int pa = L->m[0].counters[c0].priority, pb = L->m[1].counters[c1].priority;
      if (wSmStartB(q.c[0]) < wSmEndB(L, e, q.c[1]) && wSmEndB(L, e, q.c[0]) > wSmStartB(q.c[1])) q.priority = imax(imax(0, pa), pb);
      else q.priority = pa + pb;
      wStoreQMatch<LDS>(&L->assembled[L->nAssembled], q);
      L->nAssembled = L->nAssembled + 1;
    }
  }
  }
}
template <class LDS>
XM_INL void wPcMatch(WL_T L, const WEnv& e, const int* which) {  // :116-133
  bool same = L->havePrevious != 0;
  if (same) for (int i = 0; i < L->nMates; i++) if (L->prevListId[i] != wList(L, i, which[i])->id) { same = false; break; }
  if (!same) {
    wPcMatchWithoutCache(L, e, which);
    for (int i = 0; i < L->nMates; i++) L->prevListId[i] = wList(L, i, which[i])->id;
    L->havePrevious = 1;
  }
}
template <class LDS>
XM_INL void wPcFilterPriority(WL_T L, int priority) {  // :267-294
  int n = 0;
  const int nA = L->nAssembled;
  for (int i = 0; i < nA; i++) if (L->assembled[i].priority == priority) { wStoreQMatch<LDS>(&L->filtered[n], wLoadQMatch<LDS>(&L->assembled[i])); n++; }
  L->nFiltered = n;
}
template <class LDS>
WV_FN void wPcFindGoodPositionsHavingPriority(WL_T L, const WEnv& e, int numMismatches) {  // :21-24, :51-81
  int which[2] = {WLIST_HP, WLIST_HP};
  for (int i = 0; i < L->nMates; i++) {
    which[i] = wCompFindGoodPositionsHavingPriorityUpTo(L, e, i, numMismatches);
    if (L->status) { L->nFiltered = 0; return; }
    if (wList(L, i, which[i])->n >= 1) L->foundNonemptyResult = 1;
  }
  wPcMatch(L, e, which);
  if (L->status) { L->nFiltered = 0; return; }
  wPcFilterPriority(L, numMismatches);
}
template <class LDS>
WV_FN void wPcOptimisticGetBestMatches(WL_T L, const WEnv& e) {  // :84-98 (+ filterMatchesHavingMinPriority :296-304, sic: max)
  int which[2] = {WLIST_BEST, WLIST_BEST};
  for (int i = 0; i < L->nMates; i++) {
  WV_TIMER(e, WT_OPTIMISTIC);
    while (true) {
      which[i] = wCompGetBestMatches(L, e, i);
      if (wList(L, i, which[i])->n == 1 || !wCompStep(L, e, i)) break;
    }
    if (L->status) { L->nFiltered = 0; return; }
  }
  wPcMatch(L, e, which);
  if (L->status) { L->nFiltered = 0; return; }
  int mn = -1;
  const int nA = L->nAssembled;
  for (int i = 0; i < nA; i++) { const int p = L->assembled[i].priority; if (mn < 0 || mn < p) mn = p; }
  wPcFilterPriority(L, mn);
}
template <class LDS>
WV_FN void wPcFindPartiallyGoodPositions(WL_T L, const WEnv& e) {  // :26-49
  L->nFiltered = 0;
  if (L->nMates != 2) return;
  if constexpr (LDS::kMates < 2) return;
  if (!L->foundNonemptyResult) return;
  int which[2];
  bool foundGoodPosition = false, foundBadPosition = false;
  for (int i = 0; i < 2; i++) {
    int here = wCompFindGoodPositionsHavingPriorityUpTo(L, e, i, INT32_MAX);
    if (L->status) return;
    if (wList(L, i, here)->n == 0) { foundBadPosition = true; here = wCompGetAllPositions(L, e, i); }
    else foundGoodPosition = true;
    which[i] = here;
  }
  if (foundGoodPosition && foundBadPosition) {
    wPcMatch(L, e, which);
    if (L->status) return;
    const int nA = L->nAssembled;
    for (int i = 0; i < nA; i++) wStoreQMatch<LDS>(&L->filtered[i], wLoadQMatch<LDS>(&L->assembled[i]));
    L->nFiltered = nA;
  }
}
template <class LDS>
XM_INL int wPcGetNumBlocks(WL_T L) { int t = 0; for (int i = 0; i < L->nMates; i++) t += L->m[i].numBlocksMatchingAnywhere; return t; }  // :108-114

// QueryMatch helpers (M/QueryMatch.java)
XM_INL bool wQmReversed(const WQMatch& q) { return (q.c[0].seqAId & 1) != 0; }
template <class LDS>
XM_INL int wQmQueryTotalLength(WL_T L, const WQMatch& q) { int t = 0; for (int i = 0; i < q.n; i++) t += L->mateLen[q.c[i].seqAId >> 1]; return t; }
XM_INL int wQmStartIndexB(const WQMatch& q) { return imin(imax(0, q.c[0].offset), imax(0, q.c[q.n - 1].offset)); }
XM_INL int wQmEndIndexB(const WQMatch& q) { return imax(imax(0, q.c[0].offset), imax(0, q.c[q.n - 1].offset)); }  // (sic) :54-58
template <class LDS>
XM_INL int wQmTotalDistanceBetweenComponents(WL_T L, const WEnv& e, const WQMatch& q) {  // :70-79,123-132
  int total = 0;
  for (int i = 1; i < q.n; i++) {
    const WSeqMatch& a = q.c[i - 1];
    const WSeqMatch& b = q.c[i];
    int d;
    if (a.contig != b.contig) d = INT32_MAX;
    else if (wQmReversed(q)) d = imax(0, a.offset) - wSmEndB(L, e, b);
    else d = imax(0, b.offset) - wSmEndB(L, e, a);
    total = jadd(total, d);
  }
  return total;
}
XM_INL bool wQmSamePosition(const WQMatch& a, const WQMatch& b) {  // :81-93
  if (a.n != b.n) return false;
  for (int i = 0; i < a.n; i++) if (!wSmEquals(a.c[i], b.c[i])) return false;
  return true;
}

// ---------------------------------------------------------------- StraightAligner (M/StraightAligner.java) + alignMatch (M/QueryMatch_Aligner.java:412-462)
struct WSa {  // SequenceAlignment header in registers; its blocks are WaveLds::candBlocks[slot]
  int32_t nb, contig;
  int32_t referenceReversed, seqAId;
  double totalPenalty, alignedPenalty;
};

// sum of AlignmentParameters.getPenalty(query[qStart+i], reference[rStart+i]) over i in [0, n), added in index order (:106-126):
// 64 bases per round, one per lane; the terms that are not zero are then added one by one in position order
template <class LDS>
WV_FN double wUngappedPenalty(WL_T L, const WEnv& e, int seqAId, int contig, int qStart, int rStart, int n) {
  WV_TIMER(e, WT_UNGAPPED);
  const IndexView& ix = e.ix;
  double total = 0;
  for (int r0 = 0; r0 < n; r0 += 64) {
    WV_VAR(int, isMis);
    WV_VAR(int, isAmb);
    WV_PAR
      WV(isMis) = 0; WV(isAmb) = 0;
      const int i = r0 + wl;
      if (i >= n) continue;
      const uint8_t a = wSeqAt(L, seqAId, qStart + i);
      const uint8_t b = wRefAt(ix, contig, false, rStart + i);
      if (!bpCanMatch(b, a)) WV(isMis) = 1;
      else if (bpPop((uint8_t)(a | b)) > 1) WV(isAmb) = 1;
    WV_ENDPAR
    const unsigned long long mMask = WV_BALLOT(isMis), aMask = WV_BALLOT(isAmb);
    unsigned long long both = mMask | aMask;
    while (both) {
      const int bit = __builtin_ctzll(both);
      both &= both - 1;
      if ((mMask >> bit) & 1ull) total += e.params.MutationPenalty;
      else {
        const uint8_t a = wSeqAt(L, seqAId, qStart + r0 + bit);
        const uint8_t b = wRefAt(ix, contig, false, rStart + r0 + bit);
        total += e.params.AmbiguityPenalty * bpFalseNegativeRate((uint8_t)(a | b));
      }
    }
  }
  return total;
}

}  // namespace xm
#include "xm_wave_search.h"
#include "xm_wave_chain.h"
namespace xm {

// alignMatch :412-462 with the outermost StraightAligner (:13-71) done here; fromHashblockMatch is always true
template <class LDS>
WV_FN bool wAlignMatch(WL_T L, const WEnv& e, int seqAId, int contig, int offset, const Params& params, WSa& out, int slot) {
  WV_TIMER(e, WT_ALIGNMATCH);
  const IndexView& ix = e.ix;
  const int refLen = ix.contigLen[contig], qLen = L->mateLen[seqAId >> 1];
  const int startB = imax(0, offset), endB = imin(offset + qLen, refLen);
  const Section qs{startB - offset, endB - offset};
  const double maxInterestingPenalty = secLen(qs) * params.MaxErrorRate;
  const int maxShift = j2i(dmax(0.0, (maxInterestingPenalty - params.DeletionStart_Penalty) / params.DeletionExtension_Penalty));
  const Section rs{imax(0, startB - maxShift), imin(endB + maxShift, refLen)};
  Analysis an;
  an.matcher = nullptr;
  an.maxInsertionExtensionPenalty = maxInterestingPenalty - params.InsertionStart_Penalty;
  an.maxDeletionExtensionPenalty = maxInterestingPenalty - params.DeletionStart_Penalty;
  an.predictedBestOffset = offset;
  an.lastCheckedOffset = offset;  // StraightAligner.align :18
  an.confidentAboutBestOffset = true;
  if (e.dc) e.dc->refWindowBytes += (unsigned long long)((secLen(rs) + 1) / 2);
  // straightAlignment :73-94
  int queryStartIndex = qs.start, queryEndIndex = qs.end, referenceStartIndex = rs.start, referenceEndIndex = rs.end;
  if (queryStartIndex + offset > referenceStartIndex) referenceStartIndex = queryStartIndex + offset; else queryStartIndex = referenceStartIndex - offset;
  if (queryEndIndex + offset < referenceEndIndex) referenceEndIndex = queryEndIndex + offset; else queryEndIndex = referenceEndIndex - offset;
  const int n = queryEndIndex - queryStartIndex;
  const double simpleTotal = wUngappedPenalty(L, e, seqAId, contig, queryStartIndex, referenceStartIndex, n);  // alignedPenalty of the straight alignment
  WSa simple;
  simple.nb = 1; simple.contig = contig; simple.referenceReversed = seqAId & 1; simple.seqAId = seqAId;
  simple.alignedPenalty = simpleTotal;
  simple.totalPenalty = simpleTotal + (double)(qLen - n) * params.UnalignedPenalty;
  const ABlock simpleBlock{queryStartIndex, referenceStartIndex, n, referenceEndIndex - referenceStartIndex};
  const double indelPenalty = dmin(params.getStartingInsertionStartPenalty() + params.InsertionExtension_Penalty, params.DeletionStart_Penalty + params.DeletionExtension_Penalty);
  bool useSimple = false, result = false, decided = false;
  if (simpleTotal <= 0) { useSimple = true; result = true; decided = true; }
  else {
    if (simpleTotal <= indelPenalty || (an.maxInsertionExtensionPenalty <= 0 && an.maxDeletionExtensionPenalty <= 0)) {
      decided = true;
      if (simpleTotal <= maxInterestingPenalty) { useSimple = true; result = true; }
    } else if (indelPenalty > maxInterestingPenalty) decided = true;
  }
  if (!decided) {
    if constexpr (LDS::kTier == 0) {  // (the light kernels do not contain the gapped chain)
      L->status = XM_ST_WAVE_GAPPED;
      return false;
    } else {
      const double rate = simpleTotal / secLen(qs);
      Params sub = params;
      sub.MaxErrorRate = dmin(rate, params.MaxErrorRate);
      WChainCtx cx{seqAId, contig, qLen, refLen};
      WAn wan;
      wan.mslot = -1; wan.predictedBestOffset = an.predictedBestOffset; wan.lastCheckedOffset = an.lastCheckedOffset; wan.confident = 1;
      wan.maxIns = an.maxInsertionExtensionPenalty; wan.maxDel = an.maxDeletionExtensionPenalty;
      const bool have = wChainBehindStraight(L, e, cx, qs, rs, sub, wan, out, &L->candBlocks[slot][0]);
      if (L->status) return false;
      result = have;
      if (!have || out.alignedPenalty >= simpleTotal) {
        if (simpleTotal <= maxInterestingPenalty) { useSimple = true; result = true; }
      }
    }
  }
  if (useSimple) {
    out = simple;
    L->candBlocks[slot][0].startA = simpleBlock.startA; L->candBlocks[slot][0].startB = simpleBlock.startB;
    L->candBlocks[slot][0].lenA = simpleBlock.lenA; L->candBlocks[slot][0].lenB = simpleBlock.lenB;
    wvFence();
  }
  return result;
}

// ---------------------------------------------------------------- QueryMatch_Aligner (M/QueryMatch_Aligner.java)
template <class LDS>
XM_INL void wQmaInit(WL_T L, const WEnv& e, int ai, int nMates, int queryLength) {
  auto a = &L->al[ai];
  a->maxErrorRate = e.params.MaxErrorRate;
  a->bestPenalty = (double)INT32_MAX;
  a->nMates = nMates; a->queryLength = queryLength; a->nGood = 0; a->poolUsed = 0; a->nBest = 0;
}
struct WCand { int32_t nSeq, innerDistance; WSa seq[2]; double spacingPenalty, overlapMultiplier, duplicationBonus, totalPenalty; };

// doAlign :94-272 (mates that overlap on the reference, innerDistance < 0, are left to the lane-per-read kernel)
template <class LDS>
WV_FN bool wQmaDoAlign(WL_T L, const WEnv& e, int ai, const WQMatch& match, double extraSpacing, WCand& res) {
  if (e.dc) e.dc->candidatesExtended++;
  if (L->al[ai].nGood >= LDS::kGood) { { L->status = wOverflowStatus(L); L->why = 14; } return false; }
  Params parameters = e.params;
  parameters.MaxErrorRate = L->al[ai].maxErrorRate;
  const double innerDistance = (match.n < 2 ? 0 : wQmTotalDistanceBetweenComponents(L, e, match)) + extraSpacing;
  double spacingPenalty;  // computeSpacingPenalty :530-546
  if (innerDistance < 0 && innerDistance > -1 * L->al[ai].queryLength) spacingPenalty = 0;
  else spacingPenalty = (double)j2i(fabs(innerDistance - L->expectedInner) / L->deviation);
  const int queryTotalLengthI = wQmQueryTotalLength(L, match);
  const double maxAllowedPenalty = jnextUp(queryTotalLengthI * parameters.MaxErrorRate);
  if (innerDistance > 0) {
    const double minPossiblePenalty = spacingPenalty + match.priority * parameters.MutationPenalty;
    if (minPossiblePenalty > maxAllowedPenalty) return false;
  }
  if (innerDistance < 0) { { L->status = XM_ST_WAVE_FALLBACK; L->why = 15; } return false; }  // tryJoinQuerySequences / overlap bonus: not in the wave form
  double componentsPenalty = 0;
  bool remaining[2] = {true, match.n > 1};
  int numRemaining = match.n;
  int first, step, last;
  if (match.hint) { first = 0; step = 1; last = match.n; } else { first = match.n - 1; step = -1; last = -1; }
  const double maxTotalComponentPenalty = maxAllowedPenalty - spacingPenalty;
  while (true) {
    int numBases = 0;
    for (int i = 0; i < match.n; i++) if (remaining[i]) numBases += L->mateLen[match.c[i].seqAId >> 1];
    if (numBases < 1) break;
    Params prs = parameters;
    prs.MaxErrorRate = divideRoundUp(maxTotalComponentPenalty - componentsPenalty, numBases);
    bool foundAMatch = false;
    for (int i = first; i != last; i += step) {
      if (remaining[i]) {
        const bool ok = wAlignMatch(L, e, match.c[i].seqAId, match.c[i].contig, match.c[i].offset, prs, res.seq[i], i);
        if (L->status) return false;
        if (ok) {
          foundAMatch = true;
          remaining[i] = false;
          componentsPenalty += res.seq[i].totalPenalty;
          numRemaining--;
          break;
        }
      }
    }
    if (numRemaining < 1) break;
    if (!foundAMatch) return false;
  }
  res.nSeq = match.n;
  double totalUsedPenalty = componentsPenalty;
  totalUsedPenalty += spacingPenalty;
  if (totalUsedPenalty > maxAllowedPenalty) return false;
  if (res.nSeq > 1) {
    const int endB0 = L->candBlocks[0][res.seq[0].nb - 1].startB + L->candBlocks[0][res.seq[0].nb - 1].lenB;
    res.innerDistance = L->candBlocks[1][0].startB - endB0;
  } else res.innerDistance = 0;
  res.spacingPenalty = spacingPenalty;
  res.overlapMultiplier = 1;
  res.duplicationBonus = 0;
  res.totalPenalty = totalUsedPenalty;
  return true;
}

// align :35-54: returns the index of the alignment in the aligner's list, or -1 for null
template <class LDS>
WV_FN int wQmaAlign(WL_T L, const WEnv& e, int ai, const WQMatch& match, double extraSpacing) {
  WCand cand;
  if (!wQmaDoAlign(L, e, ai, match, extraSpacing, cand) || L->status) return -1;
  auto a = &L->al[ai];
  int needBlocks = 0;
  for (int k = 0; k < cand.nSeq; k++) needBlocks += cand.seq[k].nb;
  if (a->poolUsed + needBlocks > LDS::kPool) { { L->status = wOverflowStatus(L); L->why = 16; } return -1; }
  const int idx = a->nGood;
  a->nGood = idx + 1;
  XM_LDSP(WAl)* g = &a->good[idx];
  g->nSeq = cand.nSeq; g->innerDistance = cand.innerDistance;
  g->spacingPenalty = cand.spacingPenalty; g->overlapMultiplier = cand.overlapMultiplier; g->duplicationBonus = cand.duplicationBonus; g->totalPenalty = cand.totalPenalty;
  int used = a->poolUsed;
  for (int k = 0; k < cand.nSeq; k++) {
    g->seq[k].nb = cand.seq[k].nb; g->seq[k].contig = cand.seq[k].contig; g->seq[k].firstBlock = used;
    g->seq[k].referenceReversed = (uint8_t)cand.seq[k].referenceReversed; g->seq[k].seqAId = (uint8_t)cand.seq[k].seqAId;
    g->seq[k].totalPenalty = cand.seq[k].totalPenalty; g->seq[k].alignedPenalty = cand.seq[k].alignedPenalty;
    for (int i = 0; i < cand.seq[k].nb; i++) {
      a->pool[used].startA = L->candBlocks[k][i].startA; a->pool[used].startB = L->candBlocks[k][i].startB;
      a->pool[used].lenA = L->candBlocks[k][i].lenA; a->pool[used].lenB = L->candBlocks[k][i].lenB;
      used++;
    }
  }
  a->poolUsed = used;
  const double pen = cand.totalPenalty;
  if (pen < a->bestPenalty) {
    a->bestPenalty = pen;
    const double newTargetPenalty = pen + e.params.Max_PenaltySpan;
    const double newTargetErrorRate = divideRoundUp(newTargetPenalty, a->queryLength);
    if (newTargetErrorRate < a->maxErrorRate) a->maxErrorRate = newTargetErrorRate;
  }
  wvFence();
  return idx;
}

template <class LDS>
XM_INL bool wSaSame(WL_T L, int ai, int i, int j, int k) {  // [inferred] SequenceAlignment equality: same blocks
  auto a = &L->al[ai];
  XM_LDSP(const WSeqAl)* x = &a->good[i].seq[k];
  XM_LDSP(const WSeqAl)* y = &a->good[j].seq[k];
  if (x->referenceReversed != y->referenceReversed || x->nb != y->nb || x->contig != y->contig) return false;
  for (int b = 0; b < x->nb; b++) {
    XM_LDSP(const ABlock)* p = &a->pool[x->firstBlock + b];
    XM_LDSP(const ABlock)* q = &a->pool[y->firstBlock + b];
    if (p->startA != q->startA || p->startB != q->startB || p->lenA != q->lenA || p->lenB != q->lenB) return false;
  }
  return true;
}
// getBestAlignments :71-92 (withoutDuplicates keeps first occurrences)
template <class LDS>
WV_FN void wQmaGetBestAlignments(WL_T L, const WEnv& e, int ai) {
  auto a = &L->al[ai];
  const double maxInterestingPenaltyAnywhere = a->queryLength * a->maxErrorRate;
  double cutoffPenalty = a->bestPenalty + e.params.Max_PenaltySpan;
  if (cutoffPenalty > maxInterestingPenaltyAnywhere) cutoffPenalty = maxInterestingPenaltyAnywhere;
  int nBest = 0;
  const int nGood = a->nGood;
  for (int i = 0; i < nGood; i++) {
    if (!(a->good[i].totalPenalty <= cutoffPenalty)) continue;
    bool dup = false;
    for (int j = 0; j < nBest && !dup; j++) {
      const int u = a->bestIdx[j];
      if (a->good[u].nSeq != a->good[i].nSeq) continue;
      bool same = true;
      for (int k = 0; k < a->good[u].nSeq; k++) if (!wSaSame(L, ai, u, i, k)) { same = false; break; }
      if (same) dup = true;
    }
    if (!dup) a->bestIdx[nBest++] = (int8_t)i;
  }
  a->nBest = nBest;
  wvFence();
}

// ---------------------------------------------------------------- per-read driver (M/AlignerWorker.java:306-644)
XM_INL double wPenaltyLowerBound(const WEnv& e, int numMismatchedHashblocks) {  // :487-491
  const double mutationPenalty = numMismatchedHashblocks * e.params.MutationPenalty;
  const double indelPenalty = e.ix.minInterestingSize * numMismatchedHashblocks * e.params.DeletionExtension_Penalty;
  return dmin(mutationPenalty, indelPenalty);
}
// does any aligned base pair of the alignment involve an ambiguity code (SequenceAlignment.hasAmbiguousBasepairs, [inferred])
template <class LDS>
WV_FN bool wAlHasAmbiguous(WL_T L, const WEnv& e, int ai, int alIdx) {
  const IndexView& ix = e.ix;
  auto a = &L->al[ai];
  unsigned long long any = 0;
  const int nSeq = a->good[alIdx].nSeq;
  for (int k = 0; k < nSeq; k++) {
    const int nb = a->good[alIdx].seq[k].nb, fb = a->good[alIdx].seq[k].firstBlock, contig = a->good[alIdx].seq[k].contig, seqAId = a->good[alIdx].seq[k].seqAId;
    for (int b = 0; b < nb; b++) {
      const int lenA = a->pool[fb + b].lenA, lenB = a->pool[fb + b].lenB, sA = a->pool[fb + b].startA, sB = a->pool[fb + b].startB;
      if (lenA != lenB) continue;
      for (int r0 = 0; r0 < lenA; r0 += 64) {
        WV_VAR(int, amb);
        WV_PAR
          WV(amb) = 0;
          const int i = r0 + wl;
          if (i >= lenA) continue;
          if (bpIsAmbiguous(wSeqAt(L, seqAId, sA + i)) || bpIsAmbiguous(wRefAt(ix, contig, false, sB + i))) WV(amb) = 1;
        WV_ENDPAR
        any |= WV_BALLOT(amb);
      }
    }
  }
  return any != 0;
}
template <class LDS>
WV_FN bool wQuicklyConfidentInBestAlignment(WL_T L, const WEnv& e, int ai, int alIdx, const WQMatch& m) {  // :494-587
  if (alIdx < 0) return false;
  auto a = &L->al[ai];
  const int nSeq = a->good[alIdx].nSeq;
  for (int k = 0; k < nSeq; k++) {
  WV_TIMER(e, WT_CONFIDENT);
    const int nb = a->good[alIdx].seq[k].nb, fb = a->good[alIdx].seq[k].firstBlock;
    for (int b = 0; b < nb; b++) if (a->pool[fb + b].lenA != a->pool[fb + b].lenB) return false;  // hasIndel
  }
  const int contig = m.c[0].contig;
  const int matchStart = wQmStartIndexB(m), matchEnd = wQmEndIndexB(m);
  const double penalty = a->good[alIdx].totalPenalty;
  if (penalty <= 0 && e.params.Max_PenaltySpan < e.params.getMinPossibleNonzeroPenalty()) return true;
  // the pow / log term is the host's (IndexView::conf, xm_confidence.h); a key the table does not hold yet: the read leaves the wave form and the
  // lane-per-read pass that takes it over puts the key on the miss list
  double totalLengthForHighConfidence = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  if (!confLookup(e.ix.conf, e.ix.confMask, penalty, wQmQueryTotalLength(L, m), totalLengthForHighConfidence)) { { L->status = XM_ST_WAVE_FALLBACK; L->why = 19; } return false; }
#else
  totalLengthForHighConfidence = confidenceLengthOnHost(penalty, wQmQueryTotalLength(L, m), e.params.Max_PenaltySpan, e.params.MutationPenalty, e.ix.dupGranularity, e.ix.totalForwardAndReverseSize);
#endif
  const double matchMiddle = (double)((matchStart + matchEnd) / 2);
  const double interestingWindow = jmaxd(totalLengthForHighConfidence, (double)((matchEnd - matchStart + 1) / 2));
  const int windowStart = j2i(matchMiddle - interestingWindow);
  const int windowEnd = j2i(matchMiddle + interestingWindow);
  bool hasNearbyDuplication = false;
  if (mayContainDuplicationInRange(e.ix, contig, windowStart, windowEnd)) hasNearbyDuplication = true;
  else if (matchStart <= interestingWindow) hasNearbyDuplication = true;
  else if (matchEnd >= e.ix.contigLen[contig] - interestingWindow) hasNearbyDuplication = true;
  if (hasNearbyDuplication) return false;
  if (wAlHasAmbiguous(L, e, ai, alIdx)) return false;
  return true;
}

struct WResult {  // what alignToAncestralReference returns: up to 2 components, each a list of alignments of one aligner
  int32_t nComponents;
  int32_t aligner[2];  // index into WaveLds::al, -1 = none
  int32_t single[2];   // >= 0: the component is exactly this alignment (QueryAlignments.singleChoice)
  int32_t empty[2];
};

// getUnpairedAlignments :602-644 (the two sub-aligners take over both aligner slots: the paired aligner has no alignment by then)
template <class LDS>
WV_FN void wGetUnpairedAlignments(WL_T L, const WEnv& e, WResult& rr) {
  if constexpr (LDS::kMates < 2) { L->status = XM_ST_INTERNAL; return; }
  rr.nComponents = 2;
  const double expectedInnerDistance = L->expectedInner;
  for (int sequenceIndex = 0; sequenceIndex < 2; sequenceIndex++) {
    rr.single[sequenceIndex] = -1;
    rr.empty[sequenceIndex] = 0;
    const int len = L->mateLen[sequenceIndex];
    const double maxInterestingSubqueryPenalty = len * e.params.MaxErrorRate;
    const int maxNumMismatches = j2i(maxInterestingSubqueryPenalty / e.params.MutationPenalty);
    const int which = wCompFindGoodPositionsHavingPriorityUpTo(L, e, sequenceIndex, maxNumMismatches);  // findGoodComponentMatches
    if (L->status) return;
    wQmaInit(L, e, sequenceIndex, 1, len);
    wvFence();
    rr.aligner[sequenceIndex] = sequenceIndex;
    const int nLocs = wList(L, sequenceIndex, which)->n;
    for (int i = 0; i < nLocs; i++) {
      const int ci = wList(L, sequenceIndex, which)->items[i];
      const WSeqMatch sm = wCounterMatch(L, sequenceIndex, ci);
      int minInnerDistance;
      if (sequenceIndex % 2 == 1) minInnerDistance = imax(0, sm.offset);
      else minInnerDistance = e.ix.contigLen[sm.contig] - wSmEndB(L, e, sm);
      double innerDistance = minInnerDistance;
      if (innerDistance < expectedInnerDistance) innerDistance = expectedInnerDistance;
      const double spacingPenalty = innerDistance / L->deviation;
      if (spacingPenalty > maxInterestingSubqueryPenalty) continue;
      WQMatch qm;
      qm.n = 1; qm.priority = -1; qm.c[0] = sm; qm.c[1] = sm; qm.hint = 0;
      wQmaAlign(L, e, sequenceIndex, qm, innerDistance);
      if (L->status) return;
    }
    wQmaGetBestAlignments(L, e, sequenceIndex);
  }
}

// one mate into LDS and its Counting_HashBlockPath reset (M/Counting_HashBlockPath.java:20-37); false: the mate has an ambiguity code
template <class LDS>
WV_FN bool wMateInit(WL_T L, const WEnv& e, int mi, const uint8_t* codes, int len, bool reverseComplement) {
  WV_TIMER(e, WT_MATEINIT);
  auto M = &L->m[mi];
  XM_GLOBAL(const uint8_t)* const g = (XM_GLOBAL(const uint8_t)*)codes;
  unsigned long long bad = 0;
  for (int r0 = 0; r0 < len; r0 += 64) {
    WV_VAR(int, amb);
    WV_PAR
      WV(amb) = 0;
      const int i = r0 + wl;
      if (i >= len) continue;
      const uint8_t c = reverseComplement ? bpComplement(g[len - 1 - i]) : g[i];
      M->codes[i] = c;
      if (bpIsAmbiguous(c)) WV(amb) = 1;
    WV_ENDPAR
    bad |= WV_BALLOT(amb);
  }
  WV_PAR
    for (int i = wl; i < WV_MAXLEVELS * WV_MAXWIN; i += 64) { M->chunkOf[i / WV_MAXWIN][i % WV_MAXWIN] = 0xFF; M->exists[i / WV_MAXWIN][i % WV_MAXWIN] = 0; }
    if (wl < WV_MAXLEVELS) M->frontier[wl] = 0;
  WV_ENDPAR
  M->len = len;
  M->queryId = mi * 2 + (reverseComplement ? 1 : 0);
  M->rcId = mi * 2 + (reverseComplement ? 0 : 1);
  XM_LDSP(WPathState)* p = &M->path;  // M/HashBlockPath.java:15-24
  p->batchIndex = -1; p->curExists = 1;
  p->curStart = 0; p->curLen = 0; p->curFwd = 0; p->curRev = 0; p->curFlags = 0; p->curGapDir = 0; p->curExtraGap = 0;
  p->gapComputed = 0; p->gapStatus = 0; p->havePrev1 = 0; p->havePrev2 = 0; p->prevFwd1 = 0; p->prevFwd2 = 0;
  M->nCounters = 0; M->nGood = 0; M->foundGood = 0; M->done = 0; M->nHistory = 0; M->pendHead = 0; M->pendTail = 0;
  M->numBlocksMatchingAnywhere = 0; M->maxNonoverlappingBlockVisited = 0; M->numNonoverlappingBlocksVisited = 0; M->minNumDistinctMismatches = -1;
  M->nextBlockId = 0;
  M->hp.id = 0; M->best.id = 0; M->all.id = 0; M->hp.n = 0; M->best.n = 0; M->all.n = 0;
  const int maxPossibleIndel = j2i((len * e.params.MaxErrorRate - e.params.DeletionStart_Penalty) / e.params.DeletionExtension_Penalty);
  M->maxIndelLengthToConsider = maxPossibleIndel / 2;
  wvFence();
  return bad == 0;
}

// alignToAncestralReference :306-484
template <class LDS>
WV_FN void wAlignRead(WL_T L, const WEnv& e, const ReadIn& in, WResult& rr) {
  WV_TIMER(e, WT_TOTAL);
  rr.nComponents = 1; rr.single[0] = -1; rr.empty[0] = 0; rr.aligner[0] = -1; rr.single[1] = -1; rr.empty[1] = 0; rr.aligner[1] = -1;
  L->status = 0;
  L->why = 0;
  L->tier = e.tier;
  L->searchCursor = 0;
  L->nMates = in.nMates;
  L->listIdCounter = 0;
  L->nChunksUsed = 0;
  L->expectedInner = in.expectedInner;
  L->deviation = in.deviation;
  for (int m = 0; m < 2; m++) L->mateLen[m] = in.mateLen[m];
  wvFence();
  if (in.nMates > LDS::kMates) { L->status = XM_ST_WAVE_FALLBACK; return; }
  for (int m = 0; m < in.nMates; m++) {
    if (in.mateLen[m] > WV_MAXLEN || in.mateLen[m] < 1) { { L->status = XM_ST_WAVE_FALLBACK; L->why = 17; } return; }
    if (in.mateLen[m] > e.ix.maxHashedLength) { L->status = XM_ST_NEED_GROW; return; }
  }
  if (e.dc) { e.dc->reads++; for (int m = 0; m < in.nMates; m++) e.dc->readBytes += (unsigned long long)((in.mateLen[m] + 1) / 2); }
  int queryLength = 0;
  for (int m = 0; m < in.nMates; m++) queryLength += in.mateLen[m];
  const double maxInterestingPenalty = queryLength * e.params.MaxErrorRate;
  const int maxInnerDistance = j2i(maxInterestingPenalty * in.deviation + in.expectedInner);
  for (int i = 0; i < in.nMates; i++) {
    if (!wMateInit(L, e, i, in.mate[i], in.mateLen[i], i > 0)) { { L->status = XM_ST_WAVE_FALLBACK; L->why = 18; } return; }  // :317-318; ambiguity codes: lane-per-read kernel
  }
  L->maxOffsetBetweenComponents = jadd(maxInnerDistance, L->m[0].len);
  L->foundNonemptyResult = 0; L->havePrevious = 0; L->nAssembled = 0; L->nFiltered = 0;
  wvFence();
  int optimisticBestAlignment = -1;
  bool haveOptimisticMatch = false;
  WQMatch optimisticBestMatch;
  optimisticBestMatch.n = 0; optimisticBestMatch.priority = 0; optimisticBestMatch.hint = 0;
  int numMismatches = 0;
  wPcOptimisticGetBestMatches(L, e);
  if (L->status) return;
  wQmaInit(L, e, 0, in.nMates, queryLength);
  wvFence();
  rr.aligner[0] = 0;
  if (L->nFiltered == 1) {
    optimisticBestMatch = wLoadQMatch<LDS>(&L->filtered[0]);
    haveOptimisticMatch = true;
    optimisticBestAlignment = wQmaAlign(L, e, 0, optimisticBestMatch, 0);
    if (L->status) return;
    const bool quick = wQuicklyConfidentInBestAlignment(L, e, 0, optimisticBestAlignment, optimisticBestMatch);
    if (L->status) return;
    if (quick) {
      if (e.dc) e.dc->quickAccepts++;
      rr.single[0] = optimisticBestAlignment;
      return;
    }
  }
  if (optimisticBestAlignment >= 0) {
    while (true) {
      const double possiblePenalty = wPenaltyLowerBound(e, numMismatches);
      if (possiblePenalty > L->al[0].good[optimisticBestAlignment].totalPenalty + e.params.Max_PenaltySpan) {
        rr.single[0] = optimisticBestAlignment;
        return;
      }
      wPcFindGoodPositionsHavingPriority(L, e, numMismatches);
      if (L->status) return;
      numMismatches++;
      bool done = false;
      const int nF = L->nFiltered;
      for (int i = 0; i < nF; i++) if (!wQmSamePosition(optimisticBestMatch, wLoadQMatch<LDS>(&L->filtered[i]))) { done = true; break; }
      if (done) break;
    }
  }
  double bestPenalty = (double)INT32_MAX;
  int candidateNumMismatches = 0;
  while (true) {
    const double estimatedPenalty = wPenaltyLowerBound(e, candidateNumMismatches);
    if (estimatedPenalty > bestPenalty + e.params.Max_PenaltySpan) break;
    if (candidateNumMismatches > wPcGetNumBlocks(L)) break;
    wPcFindGoodPositionsHavingPriority(L, e, candidateNumMismatches);
    if (L->status) return;
    // (the candidate list is copied out: aligning a candidate does not touch it, but keep the reference's iteration over a fixed list explicit)
    const int nF = L->nFiltered;
    for (int i = 0; i < nF; i++) {
      const WQMatch cm = wLoadQMatch<LDS>(&L->filtered[i]);
      int al;
      if (haveOptimisticMatch && wQmSamePosition(cm, optimisticBestMatch)) al = optimisticBestAlignment;
      else al = wQmaAlign(L, e, 0, cm, 0);
      if (L->status) return;
      if (al >= 0) {
        const double penalty = L->al[0].good[al].totalPenalty;
        if (bestPenalty > penalty) bestPenalty = penalty;
      }
    }
    if (estimatedPenalty >= maxInterestingPenalty) break;
    candidateNumMismatches++;
  }
  wQmaGetBestAlignments(L, e, 0);
  if (L->al[0].nBest < 1 && in.nMates > 1) {
    wPcFindPartiallyGoodPositions(L, e);
    if (L->status) return;
    const int nF = L->nFiltered;
    for (int i = 0; i < nF; i++) {
      const WQMatch cm = wLoadQMatch<LDS>(&L->filtered[i]);
      const int al = wQmaAlign(L, e, 0, cm, 0);
      if (L->status) return;
      if (al >= 0) {
        const double penalty = L->al[0].good[al].totalPenalty;
        if (bestPenalty > penalty) bestPenalty = penalty;
      }
    }
  }
  wQmaGetBestAlignments(L, e, 0);
  {
    const int numBest = L->al[0].nBest;
    if (numBest < 1 && in.nMates > 1) {
      wGetUnpairedAlignments(L, e, rr);
      if (L->status) return;
    }
    if ((int64_t)numBest > (int64_t)e.params.MaxNumMatches) {  // :476-481
      rr.nComponents = 1; rr.single[0] = -1; rr.empty[0] = 1;
    }
  }
}

// ---------------------------------------------------------------- result streams (layout: include/xmapper_hip.h)
template <class LDS>
XM_INL void wCountAl(WL_T L, int ai, int idx, int64_t& ni, int64_t& nd) {
  ni += 2; nd += 4;
  const int nSeq = L->al[ai].good[idx].nSeq;
  for (int k = 0; k < nSeq; k++) { ni += 3 + 4 * L->al[ai].good[idx].seq[k].nb; nd += 2; }
}
template <class LDS>
XM_INL void wResultSize(WL_T L, const WResult& rr, int64_t& ni, int64_t& nd) {
  ni = 1; nd = 0;
  for (int c = 0; c < rr.nComponents; c++) {
    ni += 1;
    if (rr.empty[c] || rr.aligner[c] < 0) continue;
    if (rr.single[c] >= 0) { wCountAl(L, rr.aligner[c], rr.single[c], ni, nd); continue; }
    const int nBest = L->al[rr.aligner[c]].nBest;
    for (int i = 0; i < nBest; i++) wCountAl(L, rr.aligner[c], L->al[rr.aligner[c]].bestIdx[i], ni, nd);
  }
}
template <class LDS>
XM_INL void wWriteAl(WL_T L, int ai, int idx, int32_t* ints, double* dbls, int64_t& ni, int64_t& nd) {
  auto a = &L->al[ai];
  XM_LDSP(const WAl)* q = &a->good[idx];
  ints[ni++] = q->innerDistance;
  ints[ni++] = q->nSeq;
  dbls[nd++] = q->spacingPenalty; dbls[nd++] = q->overlapMultiplier; dbls[nd++] = q->duplicationBonus; dbls[nd++] = q->totalPenalty;
  for (int k = 0; k < q->nSeq; k++) {
    ints[ni++] = q->seq[k].contig; ints[ni++] = q->seq[k].referenceReversed; ints[ni++] = q->seq[k].nb;
    for (int b = 0; b < q->seq[k].nb; b++) {
      XM_LDSP(const ABlock)* bl = &a->pool[q->seq[k].firstBlock + b];
      ints[ni++] = bl->startA; ints[ni++] = bl->startB; ints[ni++] = bl->lenA; ints[ni++] = bl->lenB;
    }
    dbls[nd++] = q->seq[k].totalPenalty; dbls[nd++] = q->seq[k].alignedPenalty;
  }
}
template <class LDS>
XM_INL void wResultWrite(WL_T L, const WResult& rr, int32_t* ints, double* dbls, DevCounters* dc) {
  int64_t ni = 0, nd = 0;
  ints[ni++] = rr.nComponents;
  for (int c = 0; c < rr.nComponents; c++) {
    if (rr.empty[c] || rr.aligner[c] < 0) { ints[ni++] = 0; continue; }
    if (rr.single[c] >= 0) { ints[ni++] = 1; wWriteAl(L, rr.aligner[c], rr.single[c], ints, dbls, ni, nd); if (dc) dc->alignmentsOut++; continue; }
    const int nBest = L->al[rr.aligner[c]].nBest;
    ints[ni++] = nBest;
    for (int i = 0; i < nBest; i++) { wWriteAl(L, rr.aligner[c], L->al[rr.aligner[c]].bestIdx[i], ints, dbls, ni, nd); if (dc) dc->alignmentsOut++; }
  }
}

}  // namespace xm
