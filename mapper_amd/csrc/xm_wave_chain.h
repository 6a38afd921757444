// xmapper-hip device core, wave-per-read form: the gapped extension chain (included by xm_wave.h).
// Restates, over LDS-resident structures and with the data-parallel steps spread over the lanes of the read's wave:
//   M/SkipHighAmbiguity_Aligner.java, M/HashBlock_Aligner.java (+ M/HashBlock_Matcher.java, M/CountMap.java), M/BlockAligner.java,
//   the StraightAligners between them (M/StraightAligner.java); M/PathAligner.java stays the exact best-first emulation of
//   xm_extend.h (pathAlign), entered with every lane of the wave on the same search.
// HashBlock_Matcher without tables: a section's table entry for a k-mer is "its only position in the section / several / none"
// (M/HashBlock_Matcher.java:40-77); here the k-mer codes of the reference window and of the query are computed once per matcher,
// one lane per position, and an entry is evaluated when it is asked for by comparing the query's code with the section's codes, one lane
// per section position (ballot, count).  Which sections count as indexed follows the reference's lazy rule (:203-215) bit by bit.
// A reference window with an ambiguity code (the rolling code of :55-62 goes stale there) is left to the lane-per-read kernel.
#pragma once

namespace xm {

struct WChainCtx { int32_t seqAId, contig, qLen, refLen; };
struct WAn {  // AlignmentAnalysis; mslot: which matcher it carries (-1 none)
  int32_t mslot, predictedBestOffset, lastCheckedOffset, confident;
  double maxIns, maxDel;
};
enum { WM_SLOT_A = 0, WM_SLOT_B = 1, WM_SLOT_T = 2 };

template <class LDS>
XM_INL uint8_t wRefFwdAt(const WEnv& e, const WChainCtx& cx, int i) { return wRefAt(e.ix, cx.contig, false, i); }

// ---------------------------------------------------------------- HashBlock_Matcher (M/HashBlock_Matcher.java)
template <class LDS>
WV_FN void wMatcherInit(WL_T L, const WEnv& e, const WChainCtx& cx, int slot, const Section& rs, int sectionLength) {  // :14-29 + the codes of every position
  auto M = &L->mt[slot];
  if (sectionLength < 1) sectionLength = 1;
  int v = sectionLength * 5, k = 0;  // (int)(log(5*sectionLength)/log(4) + 1): 5*sectionLength is never a power of 4
  long long pw = 1;
  while (pw * 4 <= v) { pw *= 4; k++; }
  int blockLength = k + 1;
  if (blockLength < 3) blockLength = 3;
  const int referenceLength = secLen(rs);
  if (referenceLength > LDS::kMRef || blockLength > 7) { L->status = wOverflowStatus(L); L->why = 40; return; }
  M->referenceStart = rs.start; M->referenceLength = referenceLength; M->blockLength = blockLength; M->sectionLength = sectionLength;
  M->maxSectionIndex = (cx.refLen - 1 - rs.start) / sectionLength;
  M->nSections = 0; M->presentLo = 0; M->presentHi = 0;
  const int nRef = referenceLength - blockLength;  // positions that can be indexed: referenceStart + [0, nRef)
  unsigned long long amb = 0;
  for (int r0 = 0; r0 < nRef; r0 += 64) {
    WV_VAR(int, bad);
    WV_PAR
      WV(bad) = 0;
      const int j = r0 + wl;
      if (j >= nRef) continue;
      int sum = 0;
      bool unknown = false;
      for (int t = 0; t < blockLength; t++) {
        const uint8_t c = wRefAt(e.ix, cx.contig, false, rs.start + j + t);
        if (bpIsAmbiguous(c)) unknown = true;
        sum = sum * 4 + encodedCharToInt(c);
      }
      M->rCode[j] = (int16_t)(unknown ? -3 : sum);
      if (unknown) WV(bad) = 1;
    WV_ENDPAR
    amb |= WV_BALLOT(bad);
  }
  if (amb) { L->status = XM_ST_WAVE_FALLBACK; L->why = 41; return; }  // stale rolling codes after an ambiguity code (:55-62): lane-per-read kernel
  for (int r0 = 0; r0 < cx.qLen; r0 += 64) {
    WV_PAR
      const int i = r0 + wl;
      if (i >= cx.qLen) continue;
      int sum = -3;  // encodeBlock :79-91: UNKNOWN when the block runs off the sequence
      if (i + blockLength <= cx.qLen) {
        sum = 0;
        for (int t = 0; t < blockLength; t++) sum = sum * 4 + encodedCharToInt(wSeqAt(L, cx.seqAId, i + t));
      }
      M->qCode[i] = (int16_t)sum;
    WV_ENDPAR
  }
  wvFence();
}

// lookup :98-141
template <class LDS>
WV_FN int wMatcherLookup(WL_T L, const WEnv& e, const WChainCtx& cx, int slot, int queryIndex, int minReferenceIndex, int maxReferenceIndex) {
  auto M = &L->mt[slot];
  if (minReferenceIndex < 0) return M_UNKNOWN;
  if (maxReferenceIndex > cx.refLen) return M_UNKNOWN;
  const int encoded = M->qCode[queryIndex];
  if (encoded < 0) return M_UNKNOWN;
  const int S = M->sectionLength, referenceStart = M->referenceStart, k = M->blockLength;
  const int nRef = M->referenceLength - k;
  int matched = M_NO_MATCHES;
  const int minSectionIndex = imax(0, (minReferenceIndex - referenceStart) / S);
  const int maxSection = imin(M->maxSectionIndex, (maxReferenceIndex - referenceStart) / S);
  for (int sectionIndex = minSectionIndex; sectionIndex <= maxSection; sectionIndex++) {
    // getSection :203-215: a section is indexed when it is asked for as the new last one; sections a jump skipped stay "null"
    bool present;
    if (M->nSections > sectionIndex) {
      present = sectionIndex < 64 ? ((M->presentLo >> sectionIndex) & 1ull) != 0 : ((M->presentHi >> (sectionIndex - 64)) & 1ull) != 0;
    } else {
      if (sectionIndex >= 128) { L->status = wOverflowStatus(L); L->why = 42; return M_UNKNOWN; }
      M->nSections = sectionIndex + 1;
      if (sectionIndex < 64) M->presentLo = M->presentLo | (1ull << sectionIndex); else M->presentHi = M->presentHi | (1ull << (sectionIndex - 64));
      present = true;
      wvFence();
    }
    int lookedUp;
    if (S < 3) {  // scanSection :143-157
      lookedUp = M_NO_MATCHES;
      const int startIndex = referenceStart + sectionIndex * S;
      const int endIndex = startIndex + S;
      for (int i = startIndex; i < endIndex; i++) {
        bool can = !(i + k > referenceStart + M->referenceLength);  // canPositionsMatch :159-171
        for (int t = 0; can && t < k; t++) if (!bpCanMatch(wSeqAt(L, cx.seqAId, queryIndex + t), wRefAt(e.ix, cx.contig, false, i + t))) can = false;
        if (can) { if (lookedUp == M_NO_MATCHES) lookedUp = i; else { lookedUp = M_MULTIPLE; break; } }
      }
    } else if (present) {
      // the section's table entry for this code (:40-77): positions [start, min(start + S, nRef)) of the window that carry it
      const int start = sectionIndex * S;
      const int end = imin(start + S, nRef);
      int count = 0, firstPos = -1;
      for (int r0 = start; r0 < end && count < 2; r0 += 64) {
        WV_VAR(int, hit);
        WV_PAR
          const int j = r0 + wl;
          WV(hit) = (j < end && M->rCode[j] == encoded) ? 1 : 0;
        WV_ENDPAR
        const unsigned long long m = WV_BALLOT(hit);
        if (m) {
          if (firstPos < 0) firstPos = r0 + __builtin_ctzll(m);
          count += __builtin_popcountll(m);
        }
      }
      lookedUp = count == 0 ? (int)M_NO_MATCHES : (count > 1 ? (int)M_MULTIPLE : referenceStart + firstPos);
    } else {
      return M_UNKNOWN;
    }
    if (lookedUp == M_UNKNOWN) return M_UNKNOWN;
    if (lookedUp == M_MULTIPLE) return M_MULTIPLE;
    if (lookedUp == M_NO_MATCHES) continue;
    if (lookedUp < minReferenceIndex || lookedUp > maxReferenceIndex) continue;
    if (matched != M_NO_MATCHES) return M_MULTIPLE;
    matched = lookedUp;
  }
  return matched;
}

// how many positions t in [0, limit) can match before the first that cannot: query[ai + dir*t] against reference[bi + dir*t] (dir = +1 / -1);
// 64 positions per round, one per lane (the caller guarantees the indices are inside both sequences)
template <class LDS>
WV_FN int wMatchRun(WL_T L, const WEnv& e, const WChainCtx& cx, int ai, int bi, int limit, int dir) {
  for (int r0 = 0; r0 < limit; r0 += 64) {
    WV_VAR(int, miss);
    WV_PAR
      const int t = r0 + wl;
      WV(miss) = (t < limit && !bpCanMatch(wSeqAt(L, cx.seqAId, ai + dir * t), wRefAt(e.ix, cx.contig, false, bi + dir * t))) ? 1 : 0;
    WV_ENDPAR
    const unsigned long long m = WV_BALLOT(miss);
    if (m) return r0 + __builtin_ctzll(m);
  }
  return limit;
}

// ---------------------------------------------------------------- CountMap (M/CountMap.java) over WaveLds::cmKeys / cmVals
template <class LDS>
XM_INL void wCmPut(WL_T L, int key, int value) {
  const int n = L->cmN;
  for (int i = 0; i < n; i++) if (L->cmKeys[i] == key) { L->cmVals[i] = value; return; }
  if (n >= LDS::kCountMap) { L->status = wOverflowStatus(L); L->why = 43; return; }
  L->cmKeys[n] = key; L->cmVals[n] = value; L->cmN = n + 1;
}
template <class LDS>
XM_INL void wCmAdd(WL_T L, int key, int value, int& mostPopularKey, int& mostPopularCount, bool& haveCounts) {
  if (key == mostPopularKey || mostPopularCount == 0) {
    mostPopularCount += value;
    mostPopularKey = key;
    if (haveCounts) wCmPut(L, mostPopularKey, mostPopularCount);
  } else {
    if (!haveCounts) { haveCounts = true; wCmPut(L, mostPopularKey, mostPopularCount); }
    int count = value;
    const int n = L->cmN;
    for (int i = 0; i < n; i++) if (L->cmKeys[i] == key) { count = L->cmVals[i] + value; break; }
    wCmPut(L, key, count);
    if (count > mostPopularCount) { mostPopularKey = key; mostPopularCount = count; }
  }
  wvFence();
}

// ---------------------------------------------------------------- HashBlock_Aligner (M/HashBlock_Aligner.java)
// analyzePenalty :94-283.  storeSlot: where a matcher made for a matcher-less analysis lives; temporaries go to slot T.
template <class LDS>
WV_FN PenaltyAnalysis wHbaAnalyzePenalty(WL_T L, const WEnv& e, const WChainCtx& cx, const Section& qs, const Section& rs, const Params& p, WAn& an, int storeSlot) {
  PenaltyAnalysis result;
  result.minPossiblePenalty = 0; result.maxInsertionExtensionPenalty = 0; result.maxDeletionExtensionPenalty = 0;
  result.offsetWithMostHashblockMatches = 0; result.numHashBlockMatchesWithBestOffset = 0;
  int mslot = an.mslot;
  const double maxInterestingPenalty = p.MaxErrorRate * secLen(qs);
  int numMismatches = 0;
  int maxNonmatchingBlockEnd = qs.start;
  int mostPopularKey = 0, mostPopularCount = 0;
  bool haveCounts = false;
  L->cmN = 0;
  int numLateBlocksSupportingInsertion = 0, numLateBlocksSupportingDeletion = 0;
  const int minPossibleOffset = rs.start - qs.start;
  const int maxPossibleOffset = rs.end - qs.end;
  const int lookupUncertainty = maxPossibleOffset - minPossibleOffset;
  if (mslot < 0 || iabs(L->mt[mslot].sectionLength - lookupUncertainty) > lookupUncertainty / 2) {
    const int slot = an.mslot >= 0 ? (int)WM_SLOT_T : storeSlot;
    wMatcherInit(L, e, cx, slot, rs, lookupUncertainty);
    if (L->status) return result;
    mslot = slot;
    if (an.mslot < 0) an.mslot = slot;
  }
  const int blockLength = L->mt[mslot].blockLength;
  const int maxBlockStart = qs.end - blockLength;
  for (int blockStartIndex = qs.start; blockStartIndex <= maxBlockStart; blockStartIndex++) {
    if (blockStartIndex >= maxNonmatchingBlockEnd) {
      const int position = wMatcherLookup(L, e, cx, mslot, blockStartIndex, blockStartIndex + minPossibleOffset, blockStartIndex + maxPossibleOffset + 1);
      if (L->status) return result;
      const int offset = position - blockStartIndex;
      if (position == M_UNKNOWN || position == M_MULTIPLE) continue;
      if (position == M_NO_MATCHES) {
        numMismatches++;
        maxNonmatchingBlockEnd = blockStartIndex + blockLength;
        if (hbaMinIndelPenaltyForBlockMismatches(numMismatches, p) > maxInterestingPenalty) break;
        continue;
      }
      const int otherStartIndex = position;
      const int reverseCount = imin(blockStartIndex - maxNonmatchingBlockEnd, otherStartIndex);
      bool foundMismatch = false;
      if (reverseCount > 0 && wMatchRun(L, e, cx, blockStartIndex - 1, otherStartIndex - 1, reverseCount, -1) < reverseCount) {
        numMismatches++;
        foundMismatch = true;
        maxNonmatchingBlockEnd = blockStartIndex + blockLength;
      }
      if (!foundMismatch) {
        const int forwardShift = qs.end - blockStartIndex;
        const int limit = forwardShift - blockLength;  // positions i = blockLength .. forwardShift - 1
        if (limit > 0) {
          // (a reference position at or past rs.end counts as a base that matches nothing)
          const int inside = imax(0, imin(limit, rs.end - (otherStartIndex + blockLength)));
          const int run = wMatchRun(L, e, cx, blockStartIndex + blockLength, otherStartIndex + blockLength, inside, 1);
          if (run < limit) {
            numMismatches++;
            foundMismatch = true;
            maxNonmatchingBlockEnd = blockStartIndex + blockLength + run + 1;
          }
        }
        if (!foundMismatch) maxNonmatchingBlockEnd = qs.end;
        int numOther = 0;
        const int forwardShift2 = maxNonmatchingBlockEnd - blockStartIndex - blockLength;
        for (int i = blockLength; i < forwardShift2; i++) {
          const int indexA = blockStartIndex + i;
          const int lookupResult = wMatcherLookup(L, e, cx, mslot, indexA, indexA + minPossibleOffset, indexA + maxPossibleOffset + 1);
          if (L->status) return result;
          const int offset2 = lookupResult - indexA;
          if (lookupResult >= 0 && offset2 == offset) {
            numOther++;
            i = i - 1 + blockLength;
          }
        }
        if (offset != mostPopularKey && mostPopularCount > 0) {
          if (offset > mostPopularKey) numLateBlocksSupportingDeletion += numOther;
          else numLateBlocksSupportingInsertion += numOther;
        }
        wCmAdd(L, offset, numOther, mostPopularKey, mostPopularCount, haveCounts);
        if (L->status) return result;
      }
      if (foundMismatch) {
        if (hbaMinIndelPenaltyForBlockMismatches(numMismatches, p) > maxInterestingPenalty) break;
      } else {
        wCmAdd(L, offset, 1, mostPopularKey, mostPopularCount, haveCounts);
        if (L->status) return result;
      }
    }
  }
  int mostPopularOffset = mostPopularKey;
  const int mostPopularOffset_count = mostPopularCount;
  const double indelPenalty = hbaMinIndelPenaltyForBlockMismatches(numMismatches, p);
  result.minPossiblePenalty = indelPenalty;
  const bool couldDiffer = mostPopularOffset_count < 1 || an.lastCheckedOffset != mostPopularOffset;
  if (couldDiffer) {
    const double mismatchPenalty = numMismatches * p.MutationPenalty;
    if (result.minPossiblePenalty > mismatchPenalty) result.minPossiblePenalty = mismatchPenalty;
  }
  // setMaxExtensionPenalty :313-319
  const double longInsertion = hbaLongInsertion(numMismatches + numLateBlocksSupportingDeletion, maxInterestingPenalty, p, blockLength);
  const double manyInsertions = hbaManyInsertions(numMismatches + numLateBlocksSupportingInsertion, maxInterestingPenalty, p);
  result.maxInsertionExtensionPenalty = dmax(longInsertion, manyInsertions);
  result.maxDeletionExtensionPenalty = hbaManyDeletions(numMismatches + numLateBlocksSupportingInsertion, maxInterestingPenalty, p);
  if (result.maxInsertionExtensionPenalty > an.maxIns) result.maxInsertionExtensionPenalty = an.maxIns;
  if (result.maxDeletionExtensionPenalty > an.maxDel) result.maxDeletionExtensionPenalty = an.maxDel;
  if (mostPopularOffset_count < 1) mostPopularOffset = an.predictedBestOffset;
  result.offsetWithMostHashblockMatches = mostPopularOffset;
  result.numHashBlockMatchesWithBestOffset = mostPopularOffset_count;
  return result;
}

// ---------------------------------------------------------------- SequenceAlignments of the chain: header in registers, blocks in LDS
typedef XM_LDSP(ABlock)* WBlocksPtr;
XM_INL void wSetBlock(WBlocksPtr b, int i, int startA, int startB, int lenA, int lenB) { b[i].startA = startA; b[i].startB = startB; b[i].lenA = lenA; b[i].lenB = lenB; }
XM_INL void wCopyBlocks(WBlocksPtr dst, WBlocksPtr src, int n) { for (int i = 0; i < n; i++) { dst[i].startA = src[i].startA; dst[i].startB = src[i].startB; dst[i].lenA = src[i].lenA; dst[i].lenB = src[i].lenB; } }

// newSequenceAlignment :73-95 over blocks[0..nb): the per-block penalties (:106-126) are added in block order
template <class LDS>
WV_FN void wFinishSeqAl(WL_T L, const WEnv& e, const WChainCtx& cx, const Params& p, WBlocksPtr blocks, int nb, bool referenceReversed, WSa& out) {
  int alignedQueryLength = 0;
  double totalPenalty = 0;
  for (int i = 0; i < nb; i++) {
    const int sA = blocks[i].startA, sB = blocks[i].startB, lA = blocks[i].lenA, lB = blocks[i].lenB;
    double penalty = 0;
    if (lA == lB) penalty = wUngappedPenalty(L, e, cx.seqAId, cx.contig, sA, sB, lA);
    else if (lA > 0) { penalty += p.InsertionStart_Penalty; penalty += p.InsertionExtension_Penalty * lA; }
    else { penalty += p.DeletionStart_Penalty; penalty += p.DeletionExtension_Penalty * lB; }
    totalPenalty += penalty;
    alignedQueryLength += lA;
  }
  if (nb > 0 && p.StartingInsertionStartFree && blocks[0].lenB == 0) totalPenalty -= p.InsertionStart_Penalty;
  const double alignedPenalty = totalPenalty;
  if (nb > 0) totalPenalty += (double)(cx.qLen - alignedQueryLength) * p.UnalignedPenalty;
  out.nb = nb; out.contig = cx.contig; out.referenceReversed = referenceReversed ? 1 : 0; out.seqAId = cx.seqAId;
  out.totalPenalty = totalPenalty; out.alignedPenalty = alignedPenalty;
}

// PathAligner: the wave-cooperative best-first search of xm_wave_search.h.
template <class LDS>
WV_FN void wPyramidsForget(WL_T L) {  // every pyramid window is rebuilt when it is asked for again (wPyrEnsure)
  WV_PAR
    for (int mi = 0; mi < LDS::kMates; mi++) {
      for (int i = wl; i < WV_MAXLEVELS * WV_MAXWIN; i += 64) { L->m[mi].chunkOf[i / WV_MAXWIN][i % WV_MAXWIN] = 0xFF; L->m[mi].exists[i / WV_MAXWIN][i % WV_MAXWIN] = 0; }
      if (wl < WV_MAXLEVELS) L->m[mi].frontier[wl] = 0;
    }
  WV_ENDPAR
  L->nChunksUsed = 0;
  wvFence();
}
template <class LDS>
WV_FN bool wPathAlign(WL_T L, const WEnv& e, const WChainCtx& cx, const Section& qs, const Section& rs, const Params& p, const WAn& an, WSa& out, WBlocksPtr outBlocks) {
  WSearchReq q;
  q.seqAId = cx.seqAId; q.contig = cx.contig; q.qsStart = qs.start; q.qsEnd = qs.end; q.rsStart = rs.start; q.rsEnd = rs.end;
  q.predictedBestOffset = an.predictedBestOffset; q.confident = an.confident; q.startingInsertionStartFree = p.StartingInsertionStartFree; q.pad = 0;
  q.maxIns = an.maxIns; q.maxDel = an.maxDel; q.maxErrorRate = p.MaxErrorRate;
  WSearchResult res;
  res.ok = -1; res.status = XM_ST_OVERFLOW; res.nb = 0; res.nodesPut = 0; res.totalPenalty = 0; res.alignedPenalty = 0;
  // the read's own wave runs the search, its structures laid over the pyramid's chunk pool (the pyramid windows are rebuilt on demand afterwards)
  static_assert(sizeof(WSearchLdsInline) <= sizeof(int32_t) * 3 * 64 * (LDS::kTier >= 1 ? LDS::kChunks : 1000), "the chunk pool of a chain tier holds the inline search");
  if (e.searchNodes) {
    wPyramidsForget(L);
    XM_LDSP(WSearchLdsInline)* S = (XM_LDSP(WSearchLdsInline)*)&L->chunkFwd[0][0];
    wPathSearch(S, e.searchNodes, e.ix, e.params, e.mateBase[cx.seqAId >> 1], cx.qLen, q, res, e.dc);
    wvFence();
  }
  if (res.ok < 0 && res.status == XM_ST_OVERFLOW) {
    // it outgrew the inline capacities: a request to the search kernel through the read's memo (the k-th such search of a run of the read
    // takes result k if it is there; else the read stops with XM_ST_WAVE_SEARCH and runs again after the search kernel)
    WMemo* const M = e.memo;
    if (!M) { L->status = XM_ST_WAVE_FALLBACK; L->why = 44; return false; }
    const int k = L->searchCursor;
    L->searchCursor = k + 1;
    if (k < M->count) {
      const WSearchResult* r = &M->res[k];
      res.ok = r->ok; res.nb = r->nb; res.status = r->status; res.nodesPut = r->nodesPut; res.totalPenalty = r->totalPenalty; res.alignedPenalty = r->alignedPenalty;
      for (int i = 0; i < r->nb && i < WV_MAXBLOCKS; i++) res.blocks[i] = r->blocks[i];
    } else {
      if (k >= WV_MEMO_MAX) { L->status = XM_ST_WAVE_FALLBACK; L->why = 45; return false; }
      WV_LANE0 { M->req = q; M->pending = 1; }
      L->status = XM_ST_WAVE_SEARCH;
      return false;
    }
  }
  if (e.dc) { e.dc->pathAlignerCalls++; e.dc->pathAlignerNodes += (unsigned long long)res.nodesPut; }
  if (res.ok < 0) { L->status = res.status == XM_ST_OVERFLOW ? XM_ST_WAVE_FALLBACK : res.status; L->why = 46; return false; }
  if (!res.ok) return false;
  if (res.nb > WV_MAXBLOCKS) { L->status = XM_ST_WAVE_FALLBACK; L->why = 47; return false; }
  for (int i = 0; i < res.nb; i++) wSetBlock(outBlocks, i, res.blocks[i].startA, res.blocks[i].startB, res.blocks[i].lenA, res.blocks[i].lenB);
  out.nb = res.nb; out.contig = cx.contig; out.referenceReversed = cx.seqAId & 1; out.seqAId = cx.seqAId;
  out.totalPenalty = res.totalPenalty; out.alignedPenalty = res.alignedPenalty;
  wvFence();
  return true;
}

// StraightAligner.align :13-71 in front of `next` (0: HashBlock_Aligner #2 -> StraightAligner -> PathAligner, 1: PathAligner)
template <class LDS>
WV_FN bool wHashBlockAlign2(WL_T L, const WEnv& e, const WChainCtx& cx, const Section& qs, Section rs, const Params& p, WAn an, WSa& out, WBlocksPtr outBlocks);
template <class LDS>
WV_FN bool wStraightThen(WL_T L, const WEnv& e, const WChainCtx& cx, const Section& qs, const Section& rs, const Params& p, WAn& an, WSa& out, WBlocksPtr outBlocks, int next) {
  an.lastCheckedOffset = an.predictedBestOffset;
  // straightAlignment :73-94
  int queryStartIndex = qs.start, queryEndIndex = qs.end, referenceStartIndex = rs.start, referenceEndIndex = rs.end;
  const int off = an.predictedBestOffset;
  if (queryStartIndex + off > referenceStartIndex) referenceStartIndex = queryStartIndex + off; else queryStartIndex = referenceStartIndex - off;
  if (queryEndIndex + off < referenceEndIndex) referenceEndIndex = queryEndIndex + off; else queryEndIndex = referenceEndIndex - off;
  const int n = queryEndIndex - queryStartIndex, nB = referenceEndIndex - referenceStartIndex;
  // (the block has lengthA == lengthB by construction; its penalty is the ungapped sum, :106-126)
  double simpleAligned = wUngappedPenalty(L, e, cx.seqAId, cx.contig, queryStartIndex, referenceStartIndex, n);
  if (p.StartingInsertionStartFree && nB == 0) simpleAligned -= p.InsertionStart_Penalty;  // newSequenceAlignment :84-86 on an empty first block
  double simpleTotalWithUnaligned = simpleAligned + (double)(cx.qLen - n) * p.UnalignedPenalty;
  const double simpleTotal = simpleAligned;
  const double maxInterestingPenalty = secLen(qs) * p.MaxErrorRate;
  const double indelPenalty = dmin(p.getStartingInsertionStartPenalty() + p.InsertionExtension_Penalty, p.DeletionStart_Penalty + p.DeletionExtension_Penalty);
  bool useSimple = false, result = false, decided = false;
  if (simpleTotal <= 0) { useSimple = true; result = true; decided = true; }
  else if (an.confident) {
    if (simpleTotal <= indelPenalty || (an.maxIns <= 0 && an.maxDel <= 0)) {
      decided = true;
      if (simpleTotal <= maxInterestingPenalty) { useSimple = true; result = true; }
    } else if (indelPenalty > maxInterestingPenalty) decided = true;
  }
  if (!decided) {
    const double rate = simpleAligned / secLen(qs);
    Params sub = p;
    sub.MaxErrorRate = dmin(rate, p.MaxErrorRate);
    bool have;
    if (next == 0) have = wHashBlockAlign2(L, e, cx, qs, rs, sub, an, out, outBlocks);
    else have = wPathAlign(L, e, cx, qs, rs, sub, an, out, outBlocks);
    if (L->status) return false;
    result = have;
    if (!have || out.alignedPenalty >= simpleTotal) {
      if (simpleTotal <= maxInterestingPenalty) { useSimple = true; result = true; }
    }
  }
  if (useSimple) {
    wSetBlock(outBlocks, 0, queryStartIndex, referenceStartIndex, n, nB);
    out.nb = 1; out.contig = cx.contig; out.referenceReversed = cx.seqAId & 1; out.seqAId = cx.seqAId;
    out.alignedPenalty = simpleAligned; out.totalPenalty = simpleTotalWithUnaligned;
    wvFence();
  }
  return result;
}

// HashBlock_Aligner.align :21-81 (tail self-call as a loop); which: 1 = the outer one (next: BlockAligner), 2 = the inner one (next: StraightAligner -> PathAligner)
template <class LDS>
WV_FN bool wBlockAlign(WL_T L, const WEnv& e, const WChainCtx& cx, const Section& qs, const Section& rs, const Params& p, WAn& an, WSa& out, WBlocksPtr outBlocks);
template <class LDS, int WHICH>
WV_FN bool wHashBlockAlign(WL_T L, const WEnv& e, const WChainCtx& cx, const Section& qs, Section rs, const Params& p, WAn an, WSa& out, WBlocksPtr outBlocks) {
  while (true) {
    const double maxInterestingPenalty = p.MaxErrorRate * secLen(qs);
    if (secLen(qs) <= secLen(rs)) {
      const PenaltyAnalysis pa = wHbaAnalyzePenalty(L, e, cx, qs, rs, p, an, WHICH == 1 ? (int)WM_SLOT_A : (int)WM_SLOT_B);
      if (L->status) return false;
      if (pa.minPossiblePenalty > maxInterestingPenalty) return false;
      WAn sub = an;  // child()
      sub.maxIns = pa.maxInsertionExtensionPenalty;
      sub.maxDel = pa.maxDeletionExtensionPenalty;
      const double extra = pa.numHashBlockMatchesWithBestOffset * p.MutationPenalty + pa.minPossiblePenalty;
      if (extra > maxInterestingPenalty) {
        sub.predictedBestOffset = pa.offsetWithMostHashblockMatches;
        sub.confident = 1;
      } else if (!an.confident) {
        sub.predictedBestOffset = pa.offsetWithMostHashblockMatches;
      }
      if (an.confident && sub.predictedBestOffset == an.predictedBestOffset) sub.confident = 1;
      Section sec = rs;
      if (sub.confident) {
        const int maxDeletionLength = j2i((double)pa.maxDeletionExtensionPenalty / (double)p.DeletionExtension_Penalty);
        const int maxInsertionLength = j2i((double)pa.maxInsertionExtensionPenalty / (double)p.InsertionExtension_Penalty);
        const int maxIndelLength = imax(maxDeletionLength, maxInsertionLength);
        sec.start = imax(rs.start, qs.start + sub.predictedBestOffset - maxIndelLength);
        sec.end = imin(rs.end, qs.end + sub.predictedBestOffset + maxIndelLength);
      }
      if (secLen(sec) < secLen(rs)) { rs = sec; an = sub; continue; }
      rs = sec; an = sub;
    }
    if constexpr (WHICH == 1) return wBlockAlign(L, e, cx, qs, rs, p, an, out, outBlocks);
    else return wStraightThen(L, e, cx, qs, rs, p, an, out, outBlocks, 1);
  }
}
template <class LDS>
WV_FN bool wHashBlockAlign2(WL_T L, const WEnv& e, const WChainCtx& cx, const Section& qs, Section rs, const Params& p, WAn an, WSa& out, WBlocksPtr outBlocks) {
  return wHashBlockAlign<LDS, 2>(L, e, cx, qs, rs, p, an, out, outBlocks);
}

// ---------------------------------------------------------------- BlockAligner (M/BlockAligner.java)
// alignPiece :215-249 -> StraightAligner #2 -> HashBlock_Aligner #2 -> StraightAligner #3 -> PathAligner
template <class LDS>
WV_FN bool wAlignPiece(WL_T L, const WEnv& e, const WChainCtx& cx, const Section& qs, const Section& rs, double maxPenalty, const Params& p, bool firstPiece, const WAn& parent, WSa& out, WBlocksPtr outBlocks) {
  if (maxPenalty < 0) return false;
  Section sub = rs;
  if (parent.confident) {
    const int maxInsertionLength = j2i((double)parent.maxIns / (double)p.InsertionExtension_Penalty);
    const int maxDeletionLength = j2i((double)parent.maxDel / (double)p.DeletionExtension_Penalty);
    const int maxIndelLength = imax(maxInsertionLength, maxDeletionLength);
    const int referenceStart = imax(rs.start, qs.start + parent.predictedBestOffset - maxIndelLength);
    const int referenceEnd = imin(rs.end, qs.end + parent.predictedBestOffset + maxIndelLength);
    if (referenceEnd > referenceStart) { sub.start = referenceStart; sub.end = referenceEnd; }
  }
  Params sp = p;
  if (!firstPiece) sp.StartingInsertionStartFree = 1;
  sp.MaxErrorRate = maxPenalty / secLen(qs);
  WAn child = parent;
  child.confident = 0;
  return wStraightThen(L, e, cx, qs, sub, sp, child, out, outBlocks, 0);
}

template <class LDS>
WV_FN bool wBlockAlign(WL_T L, const WEnv& e, const WChainCtx& cx, const Section& qs, const Section& rs, const Params& p, WAn& an, WSa& out, WBlocksPtr outBlocks) {  // :17-36
  const double maxInterestingPenalty = p.MaxErrorRate * secLen(qs);
  // initialAlignments :39-96
  const double maxInterestingPenaltyWholeQuery = p.MaxErrorRate * cx.qLen;  // (sic) uses query.getLength()
  const int numBasesToEncodeReferencePosition = baNumBasesToEncodeReferencePosition(e.ix.baLogStep, secLen(rs));  // (sic) :48, the host's steps
  const int numHashblocks = secLen(qs) / numBasesToEncodeReferencePosition + 1;
  const int targetNumHashblocksPerBlock = j2i(sqrt((double)numHashblocks)) + 1;
  const int targetBlockSize = targetNumHashblocksPerBlock * numBasesToEncodeReferencePosition;
  const int numBlocks = secLen(qs) / targetBlockSize;
  if (numBlocks > LDS::kPieces) { L->status = wOverflowStatus(L); L->why = 48; return false; }
  if (numBlocks < 1) return false;  // "no initial alignments"
  WBlocksPtr scratchBlocks = &L->scratchBlocks[0];
  // piece lists 0 / 1 (ping-pong across joinAlignments rounds), blocks packed into one pool per list
  int used[2] = {0, 0};
  auto commit = [&](int list, int slot, const WSa& src, WBlocksPtr srcBlocks) -> bool {
    if (used[list] + src.nb > LDS::kPiecePool) { L->status = wOverflowStatus(L); L->why = 49; return false; }
    auto d = &L->pieces[list][slot];
    d->nb = src.nb; d->firstBlock = used[list]; d->referenceReversed = src.referenceReversed; d->totalPenalty = src.totalPenalty; d->alignedPenalty = src.alignedPenalty;
    wCopyBlocks(&L->piecePool[list][used[list]], srcBlocks, src.nb);
    used[list] += src.nb;
    wvFence();
    return true;
  };
  auto pieceSa = [&](int list, int slot) -> WSa {
    WSa s;
    auto d = &L->pieces[list][slot];
    s.nb = d->nb; s.contig = cx.contig; s.referenceReversed = d->referenceReversed; s.seqAId = cx.seqAId; s.totalPenalty = d->totalPenalty; s.alignedPenalty = d->alignedPenalty;
    return s;
  };
  unsigned have = 0;
  double usedPenalty = 0;
  int numRemainingAlignments = numBlocks;
  WSa scratch;
  while (true) {
    bool failedSubalignment = false, failedThenFound = false;
    int startPosition = qs.start;
    for (int i = 0; i < numBlocks; i++) {
      const int endPosition = qs.start + (secLen(qs) * (i + 1) / numBlocks);
      if (!((have >> i) & 1u)) {
        const Section sub{startPosition, endPosition};
        const double averagePenalty = (maxInterestingPenaltyWholeQuery - usedPenalty) / numRemainingAlignments;
        const bool ok = wAlignPiece(L, e, cx, sub, rs, averagePenalty, p, i == 0, an, scratch, scratchBlocks);
        if (L->status) return false;
        if (ok) {
          if (!commit(0, i, scratch, scratchBlocks)) return false;
          if (failedSubalignment) failedThenFound = true;
          numRemainingAlignments--;
          have |= 1u << i;
          usedPenalty += scratch.alignedPenalty;
        } else {
          failedSubalignment = true;
        }
      }
      startPosition = endPosition;
    }
    if (numRemainingAlignments < 1) break;
    if (!failedThenFound) return false;
  }
  // joinAlignments rounds :99-144
  int cur = 0, n = numBlocks;
  bool even = false;
  while (n > 1) {
    const int nxt = 1 - cur;
    int rn = 0;
    used[nxt] = 0;
    double usedP = 0;
    for (int i = 0; i < n; i++) usedP += L->pieces[cur][i].alignedPenalty;
    for (int i = 0; i < n; i += 2) {
      if (i + 1 < n) {
        // tryMerge :158-212
        auto l = &L->pieces[cur][i];
        auto r = &L->pieces[cur][i + 1];
        WBlocksPtr lb = &L->piecePool[cur][l->firstBlock];
        WBlocksPtr rb = &L->piecePool[cur][r->firstBlock];
        const int lnb = l->nb, rnb = r->nb;
        bool merged = false;
        {
          const int lEndB = lb[lnb - 1].startB + lb[lnb - 1].lenB;
          if (lEndB == rb[0].startB) {
            const int lsA = lb[lnb - 1].startA, lsB = lb[lnb - 1].startB, llA = lb[lnb - 1].lenA, llB = lb[lnb - 1].lenB;
            const int rsA = rb[0].startA, rsB = rb[0].startB, rlA = rb[0].lenA, rlB = rb[0].lenB;
            const int lt = llA == llB ? 0 : (llA > llB ? 1 : 2), rt = rlA == rlB ? 0 : (rlA > rlB ? 1 : 2);
            if (lt == rt && lsA + llA == rsA && lsB + llB == rsB) {
              if (lnb - 1 + 1 + rnb - 1 > WV_MAXBLOCKS) { L->status = wOverflowStatus(L); L->why = 50; return false; }
              int nb2 = 0;
              for (int k = 0; k < lnb - 1; k++) { wSetBlock(scratchBlocks, nb2, lb[k].startA, lb[k].startB, lb[k].lenA, lb[k].lenB); nb2++; }
              wSetBlock(scratchBlocks, nb2, lsA, lsB, llA + rlA, llB + rlB); nb2++;
              for (int k = 1; k < rnb; k++) { wSetBlock(scratchBlocks, nb2, rb[k].startA, rb[k].startB, rb[k].lenA, rb[k].lenB); nb2++; }
              wvFence();
              wFinishSeqAl(L, e, cx, p, scratchBlocks, nb2, l->referenceReversed != 0, scratch);
              merged = true;
            }
          }
        }
        if (!merged) {
          usedP -= L->pieces[cur][i].alignedPenalty;
          usedP -= L->pieces[cur][i + 1].alignedPenalty;
          const Section sub{lb[0].startA, rb[rnb - 1].startA + rb[rnb - 1].lenA};
          const bool ok = wAlignPiece(L, e, cx, sub, rs, maxInterestingPenalty - usedP, p, i == 0, an, scratch, scratchBlocks);
          if (L->status) return false;
          if (!ok) return false;
          if (!commit(nxt, rn, scratch, scratchBlocks)) return false;
          usedP += scratch.alignedPenalty;
          rn++;
        } else {
          if (!even) {  // !allowSimpleMerges: keep `left`, retry from its right neighbour
            if (!commit(nxt, rn, pieceSa(cur, i), lb)) return false;
            rn++;
            i--;
            continue;
          }
          if (!commit(nxt, rn, scratch, scratchBlocks)) return false;
          rn++;
        }
      } else {
        if (!commit(nxt, rn, pieceSa(cur, i), &L->piecePool[cur][L->pieces[cur][i].firstBlock])) return false;
        rn++;
      }
    }
    cur = nxt;
    n = rn;
    even = !even;
  }
  out = pieceSa(cur, 0);
  if (out.nb > WV_MAXBLOCKS) { L->status = wOverflowStatus(L); L->why = 51; return false; }
  wCopyBlocks(outBlocks, &L->piecePool[cur][L->pieces[cur][0].firstBlock], out.nb);
  wvFence();
  return true;
}

// SkipHighAmbiguity_Aligner :13-28 -> HashBlock_Aligner #1 (the rest of the chain behind the outermost StraightAligner)
template <class LDS>
WV_FN bool wChainBehindStraight(WL_T L, const WEnv& e, const WChainCtx& cx, const Section& qs, const Section& rs, const Params& p, WAn& an, WSa& out, WBlocksPtr outBlocks) {
  WV_TIMER(e, WT_CHAIN);
  unsigned numAmbiguities = 0;
  for (int r0 = rs.start; r0 < rs.end; r0 += 64) {
    WV_VAR(int, amb);
    WV_PAR
      const int i = r0 + wl;
      WV(amb) = (i < rs.end && bpIsAmbiguous(wRefAt(e.ix, cx.contig, false, i))) ? 1 : 0;
    WV_ENDPAR
    numAmbiguities += (unsigned)__builtin_popcountll(WV_BALLOT(amb));
  }
  if ((int)numAmbiguities >= secLen(rs) / 4) return false;
  return wHashBlockAlign<LDS, 1>(L, e, cx, qs, rs, p, an, out, outBlocks);
}

}  // namespace xm
