// xmapper-hip device core: candidate -> QueryAlignment and the per-read driver.
// Replaces (per read, on the GPU): M/QueryMatch_Aligner.java and M/AlignerWorker.java:306-644
// (alignToAncestralReference, getPenaltyLowerBound, quicklyConfidentInBestAlignment, getUnpairedAlignments) plus
// M/Readable_DuplicationDetector.java:28-47.  One read is processed start to finish by one lane; all state lives in that
// lane's scratch arena (persistent part bump-allocated from the bottom, stack-discipline temporaries above it).
#pragma once
#include "xm_extend.h"
#include "xm_confidence.h"

namespace xm {

struct QAl {  // QueryAlignment
  int32_t nSeq, innerDistance;
  SeqAl seq[2];
  double spacingPenalty, overlapMultiplier, duplicationBonus, totalPenalty;
};

struct ReadIn {  // one query
  int32_t nMates;
  const uint8_t* mate[2];
  int32_t mateLen[2];
  double expectedInner, deviation;
};

struct QMAligner {  // QueryMatch_Aligner
  Params parameters;
  int32_t nMates;        // 1 or 2 (query.getNumSequences())
  int32_t queryLength;   // query.getLength()
  double expectedInner, deviation;
  QAl* good; int32_t nGood, goodCap;  // blocks of accepted alignments live compactly in blockPool
  QAl cand;                      // the candidate doAlign is building (blocks: candBlocks[k], capacity caps.maxBlocks)
  ABlock* blockPool; int32_t poolUsed, poolCap;
  double bestPenalty;
  int16_t* bestIdx; int32_t nBest;  // result of getBestAlignments()
};

struct ReadResult {
  int32_t nComponents;
  QMAligner* aligner[2];
  int32_t single[2];  // >= 0: the component is exactly this one alignment of aligner[c] (QueryAlignments.singleChoice)
  int32_t empty[2];   // 1: the component is empty regardless of the aligner
};

// Where alignRead stands (AlignerWorker.alignToAncestralReference :306-484): everything its loops carry from one candidate to the next.
// A read that stops in the light pass because a candidate needs the gapped chain (XM_ST_NEED_HEAVY) is resumed in the gapped pass at
// that candidate, with its seeding state (pyramids, votes, candidate lists, accepted alignments) as the light pass left it.
struct AlignReadState {
  int32_t phase;  // 0 before the first candidate, 1 aligning the optimistic best match, 3 in the main loop at filtered[i], 4 in the partially-good loop at filtered[i], 5 after them (not resumable)
  int32_t optimisticBestAlignment, haveOptimisticMatch, numMismatches, candidateNumMismatches, i, queryLength;
  QMatch optimisticBestMatch;
  double bestPenalty, estimatedPenalty, maxInterestingPenalty;
  QMAligner* aligner;
  ReadResult rr;
  unsigned long long candidatesAtCall, refWindowBytesAtCall;  // counters when the resumable qmaAlign call started (the resumed run counts that candidate again)
};

struct ReadCtx {
  const IndexView* ix;
  Caps caps;
  DevCounters* dc;
  int32_t status;
  int32_t listIdCounter;
  ReadIn in;
  Params params;
  Arena persist;  // bottom part of the lane's arena
  Arena tmp;      // top part
  SeedEnv seed;
  Comp comps[2];
  PathsCounter pc;
  float heavyHint;      // light pass: largest straight-alignment penalty among the candidates that needed the gapped chain
  AlignReadState ar;
};

XM_INL SeqView queryView(const ReadCtx& cx, uint8_t seqAId) {
  SeqView v;
  int m = seqAId >> 1;
  v.base = cx.in.mate[m]; v.len = cx.in.mateLen[m]; v.rc = seqAId & 1; v.id = seqAId;
  return v;
}

XM_INL double divideRoundUp(double a, double b) {  // M/QueryMatch_Aligner.java:56-61
  double r = a / b;
  if (r * b < a) r = jnextUp(r);
  return r;
}

// goodCap: room for that many accepted alignments (the read's aligner: caps.maxGoodAlignments; the two aligners of getUnpairedAlignments half of it - the
// seeding state of a pair must still fit the region beside them)
XM_INL void qmaInit(ReadCtx& cx, QMAligner& a, int nMates, int queryLength, int goodCap) {
  a.parameters = cx.params;
  a.nMates = nMates;
  a.queryLength = queryLength;
  a.expectedInner = cx.in.expectedInner;
  a.deviation = cx.in.deviation;
  a.nGood = 0;
  a.nBest = 0;
  a.bestPenalty = (double)INT32_MAX;
  const int n = goodCap;
  a.goodCap = n;
  a.good = arenaArray<QAl>(cx.persist, n);
  a.poolCap = 6 * n + 4 * cx.caps.maxBlocks;  // (blocks of the accepted alignments: most have one to three per mate; the read's aligner had room for 16 alignments and 12 blocks
                                              // each - the same pool now serves 32 alignments, and what the aligners of a pair with ambiguity codes take together has not grown)
  a.poolUsed = 0;
  a.blockPool = arenaArray<ABlock>(cx.persist, (size_t)a.poolCap);
  ABlock* candBlocks = arenaArray<ABlock>(cx.persist, (size_t)2 * cx.caps.maxBlocks);
  a.bestIdx = arenaArray<int16_t>(cx.persist, n);
  if (cx.persist.overflow) { cx.status = XM_ST_OVERFLOW; return; }
  for (int k = 0; k < 2; k++) a.cand.seq[k].blocks = candBlocks + (size_t)k * cx.caps.maxBlocks;
}

XM_INL void makeExtEnv(ReadCtx& cx, ExtEnv& e, const SeqView& query, int contig) {
  e.caps = &cx.caps; e.dc = cx.dc; e.status = &cx.status; e.tmp = &cx.tmp;
  e.query = query;
  e.reference = refView(*cx.ix, contig, false);
  e.contig = contig;
  e.slotA = e.slotB = e.slotT = nullptr;
  e.heavyHint = &cx.heavyHint;
  e.baLogStep = cx.ix->baLogStep;
}

// alignMatch :412-462 (fromHashblockMatch is always true).  The matcher slots live in tmp for the duration of the call.
XM_NOINL bool qmaAlignMatch(ReadCtx& cx, const SeqView& seqA, int contig, int offset, const Params& params, SeqAl& out) {
  size_t mark = cx.tmp.used;
  ExtEnv e;
  makeExtEnv(cx, e, seqA, contig);
  Matcher* slots = arenaArray<Matcher>(cx.tmp, 3);
  if (cx.tmp.overflow) { cx.status = XM_ST_OVERFLOW; cx.tmp.used = mark; return false; }
  for (int i = 0; i < 3; i++) {
    uint8_t* const present = arenaArray<uint8_t>(cx.tmp, cx.caps.maxSections);
    int16_t* const tables = arenaArray<int16_t>(cx.tmp, cx.caps.matcherEntries);
    slots[i].present = present;
    slots[i].tables = tables;
    slots[i].tableCap = cx.caps.matcherEntries;
    slots[i].maxSections = cx.caps.maxSections;
    slots[i].nSections = 0;
    slots[i].presentMask = 0;
    slots[i].sectionLength = 0;
  }
  if (cx.tmp.overflow) { cx.status = XM_ST_OVERFLOW; cx.tmp.used = mark; return false; }
  e.slotA = &slots[0]; e.slotB = &slots[1]; e.slotT = &slots[2];
  int refLen = e.reference.len;
  int startB = imax(0, offset), endB = imin(offset + seqA.len, refLen);
  Section qs{startB - offset, endB - offset};
  double maxInterestingPenalty = secLen(qs) * params.MaxErrorRate;
  int maxIndelLength = j2i(dmax(0.0, (maxInterestingPenalty - params.DeletionStart_Penalty) / params.DeletionExtension_Penalty));
  int maxShift = maxIndelLength;
  Section rs{imax(0, startB - maxShift), imin(endB + maxShift, refLen)};
  Analysis an;
  an.matcher = nullptr;
  an.maxInsertionExtensionPenalty = maxInterestingPenalty - params.InsertionStart_Penalty;
  an.maxDeletionExtensionPenalty = maxInterestingPenalty - params.DeletionStart_Penalty;
  an.predictedBestOffset = offset;
  an.lastCheckedOffset = 0;
  an.confidentAboutBestOffset = true;
  if (cx.dc) cx.dc->refWindowBytes += (unsigned long long)((secLen(rs) + 1) / 2);
  bool ok = outerChain(e, qs, rs, params, an, out);
  cx.tmp.used = mark;
  return ok && cx.status == 0;
}

// ---- helpers over finished SequenceAlignments
XM_INL double saPenaltyInRange(ReadCtx& cx, const Params& p, const SeqAl& a, int startIndexB, int endIndexB) {  // M/AlignmentParameters.java:97-154
  SeqView q = queryView(cx, a.seqAId);
  SeqView r = refView(*cx.ix, a.contig, false);
  double total = 0;
  for (int k = 0; k < a.nb; k++) {
    const ABlock& b = a.blocks[k];
    double penalty = 0;
    if (b.lenA == b.lenB) {
      for (int i = 0; i < b.lenA; i++) {
        int bIndex = b.startB + i;
        if (bIndex >= startIndexB && bIndex < endIndexB) penalty += p.getPenalty(q.at(b.startA + i), r.at(bIndex));
      }
    } else if (b.startB < endIndexB && abEndB(b) > startIndexB) {
      if (b.lenA > 0) { penalty += p.InsertionStart_Penalty; penalty += p.InsertionExtension_Penalty * b.lenA; }
      else { penalty += p.DeletionStart_Penalty; penalty += p.DeletionExtension_Penalty * b.lenB; }
    }
    total += penalty;
  }
  return total;
}
XM_INL int saLengthA(const SeqAl& a) { int t = 0; for (int i = 0; i < a.nb; i++) t += a.blocks[i].lenA; return t; }
XM_INL int saInsertAOrBLength(const SeqAl& a) { int t = 0; for (int i = 0; i < a.nb; i++) if (a.blocks[i].lenA != a.blocks[i].lenB) t += a.blocks[i].lenA + a.blocks[i].lenB; return t; }
XM_INL int saLengthABefore(const SeqAl& a, int refIndex) {  // [inferred, see oracle/xmo_types.h]
  int t = 0;
  for (int i = 0; i < a.nb; i++) {
    const ABlock& b = a.blocks[i];
    if (abEndB(b) <= refIndex) t += b.lenA;
    else if (b.startB >= refIndex) t += 0;
    else if (b.lenA == b.lenB) t += refIndex - b.startB;
  }
  return t;
}
XM_INL int saLengthAAfter(const SeqAl& a, int refIndex) {
  int t = 0;
  for (int i = 0; i < a.nb; i++) {
    const ABlock& b = a.blocks[i];
    if (b.startB >= refIndex) t += b.lenA;
    else if (abEndB(b) <= refIndex) t += 0;
    else if (b.lenA == b.lenB) t += abEndB(b) - refIndex;
  }
  return t;
}
XM_INL bool saHasIndel(const SeqAl& a) { for (int i = 0; i < a.nb; i++) if (a.blocks[i].lenA != a.blocks[i].lenB) return true; return false; }
XM_INL bool saHasAmbiguous(ReadCtx& cx, const SeqAl& a) {
  SeqView q = queryView(cx, a.seqAId);
  SeqView r = refView(*cx.ix, a.contig, false);
  for (int k = 0; k < a.nb; k++) {
    const ABlock& b = a.blocks[k];
    if (b.lenA != b.lenB) continue;
    for (int i = 0; i < b.lenA; i++) if (bpIsAmbiguous(q.at(b.startA + i)) || bpIsAmbiguous(r.at(b.startB + i))) return true;
  }
  return false;
}
XM_INL bool saSame(const SeqAl& x, const SeqAl& y) {
  if (x.referenceReversed != y.referenceReversed || x.nb != y.nb || x.contig != y.contig) return false;
  for (int i = 0; i < x.nb; i++) {
    const ABlock &a = x.blocks[i], &b = y.blocks[i];
    if (a.startA != b.startA || a.startB != b.startB || a.lenA != b.lenA || a.lenB != b.lenB) return false;
  }
  return true;
}

// extract :362-405: cut [queryStart, queryEnd) of the joined alignment out for one mate
XM_NOINL bool qmaExtract(ReadCtx& cx, const Params& p, const SeqAl& joined, int queryStart, int queryEnd, uint8_t mateSeqAId, SeqAl& out) {
  bool reverse = (mateSeqAId & 1) != 0;
  bool referenceReversed = (joined.referenceReversed != 0) != reverse;
  int n = 0;
  for (int k = 0; k < joined.nb; k++) {
    const ABlock& b = joined.blocks[k];
    if (b.startA >= queryEnd) break;
    if (abEndA(b) <= queryStart) continue;
    int selectionStart = imax(b.startA, queryStart);
    int selectionEnd = imin(abEndA(b), queryEnd);
    int querySelectionLength = selectionEnd - selectionStart;
    int referenceSelectionLength, referenceStart;
    int blockOffset = b.startB - b.startA;
    if (b.lenA == b.lenB) { referenceSelectionLength = querySelectionLength; referenceStart = selectionStart + blockOffset; }
    else if (b.lenA > b.lenB) { referenceSelectionLength = 0; referenceStart = b.startB; }
    else { referenceSelectionLength = b.lenB; referenceStart = selectionStart + blockOffset; }
    if (n >= cx.caps.maxBlocks) { cx.status = XM_ST_OVERFLOW; return false; }
    out.blocks[n++] = ABlock{selectionStart - queryStart, referenceStart, querySelectionLength, referenceSelectionLength};
  }
  if (n < 1) return false;
  out.nb = n;
  ExtEnv e;
  makeExtEnv(cx, e, queryView(cx, mateSeqAId), joined.contig);
  finishSeqAl(e, p, out, referenceReversed);
  return true;
}

// doAlign :94-272.  Builds the candidate in a.cand and returns true when it is a valid alignment.
XM_NOINL bool qmaDoAlign(ReadCtx& cx, QMAligner& a, const QMatch& match, double extraSpacing) {
  const SeedEnv& se = cx.seed;
  if (cx.dc) cx.dc->candidatesExtended++;
  if (a.nGood >= a.goodCap) { cx.status = XM_ST_OVERFLOW; return false; }
  QAl& res = a.cand;
  size_t mark = cx.tmp.used;
  double innerDistance = (match.n < 2 ? 0 : qmTotalDistanceBetweenComponents(se, match)) + extraSpacing;
  // computeSpacingPenalty :530-546
  double spacingPenalty;
  if (innerDistance < 0 && innerDistance > -1 * a.queryLength) spacingPenalty = 0;
  else spacingPenalty = (double)j2i(fabs(innerDistance - a.expectedInner) / a.deviation);
  double overlapMultiplier = 1, duplicationBonus = 0;
  int queryTotalLengthI = qmQueryTotalLength(se, match);
  double maxAllowedPenalty = jnextUp(queryTotalLengthI * a.parameters.MaxErrorRate);
  if (innerDistance > 0) {
    double minPossiblePenalty = spacingPenalty + match.priority * a.parameters.MutationPenalty;
    if (minPossiblePenalty > maxAllowedPenalty) return false;
  }
  bool haveComponents = false;
  double componentsPenalty = 0;
  if (match.n > 1 && innerDistance < 0) {
    // tryJoinQuerySequences :274-319
    const SeqMatch& m1 = match.c[0];
    const SeqMatch& m2 = match.c[1];
    int offset = m2.offset - m1.offset;
    SeqView s1 = queryView(cx, offset >= 0 ? m1.seqAId : m2.seqAId);
    SeqView s2 = queryView(cx, offset >= 0 ? m2.seqAId : m1.seqAId);
    int off = offset >= 0 ? offset : -offset;
    int suffixStartIndex = s1.len - off;
    bool joinable = suffixStartIndex >= 0;
    if (joinable) {
      int match2IndexEnd = imin(s2.len, s1.len - off);
      for (int i = 0; i < match2IndexEnd; i++) if (s1.at(i + off) != s2.at(i)) { joinable = false; break; }
    }
    if (joinable) {
      int endIndex = s2.len;
      if (endIndex - suffixStartIndex < 0) { cx.status = XM_ST_INTERNAL; return false; }  // getRange with a negative length throws
      int jl = s1.len + (endIndex - suffixStartIndex);
      if (jl > cx.caps.maxJoined) { cx.status = XM_ST_OVERFLOW; return false; }
      uint8_t* jb = arenaArray<uint8_t>(cx.tmp, jl);
      ABlock* jblocks = arenaArray<ABlock>(cx.tmp, cx.caps.maxBlocks);
      if (cx.tmp.overflow) { cx.status = XM_ST_OVERFLOW; cx.tmp.used = mark; return false; }
      for (int i = 0; i < s1.len; i++) jb[i] = s1.at(i);
      for (int i = suffixStartIndex; i < endIndex; i++) jb[s1.len + i - suffixStartIndex] = s2.at(i);
      SeqView joined;
      joined.base = jb; joined.len = jl; joined.rc = 0; joined.id = 4;
      // computeJoinedAlignment :321-330
      int joinedOffset = imin(m1.offset, m2.offset);
      Params subp = a.parameters;
      subp.MaxErrorRate = jnextUp(subp.MaxErrorRate);
      SeqAl jal;
      jal.blocks = jblocks;
      bool ok = qmaAlignMatch(cx, joined, m1.contig, joinedOffset, subp, jal);
      if (cx.status) { cx.tmp.used = mark; return false; }
      // splitAlignment :332-360
      if (!ok) { cx.tmp.used = mark; return false; }
      int l1 = seqALen(se, m1.seqAId), l2 = seqALen(se, m2.seqAId);
      bool ok1, ok2;
      if (offset >= 0) {
        ok1 = qmaExtract(cx, a.parameters, jal, 0, l1, m1.seqAId, res.seq[0]);
        ok2 = qmaExtract(cx, a.parameters, jal, offset, l2 + offset, m2.seqAId, res.seq[1]);
      } else {
        ok2 = qmaExtract(cx, a.parameters, jal, 0, l2, m2.seqAId, res.seq[1]);
        ok1 = qmaExtract(cx, a.parameters, jal, -offset, l1 - offset, m1.seqAId, res.seq[0]);
      }
      cx.tmp.used = mark;
      if (cx.status) return false;
      if (!ok1 || !ok2) return false;
      haveComponents = true;
      componentsPenalty += res.seq[0].totalPenalty;
      componentsPenalty += res.seq[1].totalPenalty;
    }
  }
  if (!haveComponents) {
    bool remaining[2] = {true, match.n > 1};
    int numRemaining = match.n;
    int first, step, last;
    if (match.hint) { first = 0; step = 1; last = match.n; } else { first = match.n - 1; step = -1; last = -1; }
    double maxTotalComponentPenalty;
    if (innerDistance < 0 && match.n > 1) {
      double queryTotalLength = queryTotalLengthI;
      double estimatedOverlap = dmin(-1 * innerDistance, (double)imin(seqALen(se, match.c[0].seqAId), seqALen(se, match.c[1].seqAId)));
      double estimatedUniqueLength = queryTotalLength - estimatedOverlap;
      maxTotalComponentPenalty = divideRoundUp(maxAllowedPenalty - spacingPenalty, queryTotalLength) * estimatedUniqueLength * 2;
    } else {
      maxTotalComponentPenalty = maxAllowedPenalty - spacingPenalty;
    }
    while (true) {
      int numBases = 0;
      for (int i = 0; i < match.n; i++) if (remaining[i]) numBases += seqALen(se, match.c[i].seqAId);
      if (numBases < 1) break;
      Params prs = a.parameters;
      prs.MaxErrorRate = divideRoundUp(maxTotalComponentPenalty - componentsPenalty, numBases);
      bool foundAMatch = false;
      for (int i = first; i != last; i += step) {
        if (remaining[i]) {
          bool ok = qmaAlignMatch(cx, queryView(cx, match.c[i].seqAId), match.c[i].contig, match.c[i].offset, prs, res.seq[i]);
          if (cx.status) return false;
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
  }
  res.nSeq = match.n;
  double totalUsedPenalty = componentsPenalty;
  if (innerDistance < 0) {
    // computeDuplicationBonus :506-520
    if (res.nSeq >= 2) {
      const SeqAl& x = res.seq[0];
      const SeqAl& y = res.seq[1];
      double overlappingLength = imin(saEndB(x), saEndB(y)) - imax(saStartB(x), saStartB(y));
      if (!(overlappingLength < 0))
        duplicationBonus = (saPenaltyInRange(cx, a.parameters, x, saStartB(y), saEndB(y)) + saPenaltyInRange(cx, a.parameters, y, saStartB(x), saEndB(x))) / 2;
    }
    totalUsedPenalty -= duplicationBonus;
    // multiplyPenaltyForOverlap :464-504
    double multipliedPenalty = totalUsedPenalty;
    if (res.nSeq >= 2) {
      const SeqAl& f = res.seq[0];
      const SeqAl& s = res.seq[1];
      double overlappingLengthB = imin(saEndB(f), saEndB(s)) - imax(saStartB(f), saStartB(s));
      if (overlappingLengthB > 0) {
        int uniqueLengthA;
        if (saStartB(f) <= saStartB(s)) uniqueLengthA = saLengthABefore(f, saStartB(s)) + saLengthA(s) + saLengthAAfter(f, saEndB(s));
        else uniqueLengthA = saLengthABefore(s, saStartB(f)) + saLengthA(f) + saLengthAAfter(s, saEndB(f));
        double deletion = imin(saInsertAOrBLength(f), saInsertAOrBLength(s));
        uniqueLengthA = j2i((double)uniqueLengthA - deletion);
        if (uniqueLengthA > 0) {
          int totalLengthA = saLengthA(f) + saLengthA(s);
          multipliedPenalty = divideRoundUp(totalUsedPenalty, uniqueLengthA) * totalLengthA;
        }
      }
    }
    if (totalUsedPenalty != 0) overlapMultiplier = multipliedPenalty / totalUsedPenalty; else overlapMultiplier = 1;
    totalUsedPenalty = multipliedPenalty;
  }
  totalUsedPenalty += spacingPenalty;
  if (totalUsedPenalty > maxAllowedPenalty) return false;
  res.innerDistance = res.nSeq > 1 ? saStartB(res.seq[1]) - saEndB(res.seq[0]) : 0;
  res.spacingPenalty = spacingPenalty;
  res.overlapMultiplier = overlapMultiplier;
  res.duplicationBonus = duplicationBonus;
  res.totalPenalty = totalUsedPenalty;
  return true;
}

// align :35-54.  returns the index of the alignment in a.good, or -1 for null
XM_INL int qmaAlign(ReadCtx& cx, QMAligner& a, const QMatch& match, double extraSpacing) {
  if (!qmaDoAlign(cx, a, match, extraSpacing) || cx.status) return -1;
  int needBlocks = 0;
  for (int k = 0; k < a.cand.nSeq; k++) needBlocks += a.cand.seq[k].nb;
  if (a.poolUsed + needBlocks > a.poolCap) { cx.status = XM_ST_OVERFLOW; return -1; }
  int idx = a.nGood++;
  a.good[idx] = a.cand;
  for (int k = 0; k < a.cand.nSeq; k++) {
    a.good[idx].seq[k].blocks = a.blockPool + a.poolUsed;
    for (int i = 0; i < a.cand.seq[k].nb; i++) a.blockPool[a.poolUsed++] = a.cand.seq[k].blocks[i];
  }
  double pen = a.good[idx].totalPenalty;
  if (pen < a.bestPenalty) {
    a.bestPenalty = pen;
    double newTargetPenalty = pen + a.parameters.Max_PenaltySpan;
    double newTargetErrorRate = divideRoundUp(newTargetPenalty, a.queryLength);
    if (newTargetErrorRate < a.parameters.MaxErrorRate) a.parameters.MaxErrorRate = newTargetErrorRate;
  }
  return idx;
}

// getBestAlignments :71-92 (withoutDuplicates keeps first occurrences; HashSet order is unpinned in the reference)
XM_NOINL void qmaGetBestAlignments(QMAligner& a) {
  double maxInterestingPenaltyAnywhere = a.queryLength * a.parameters.MaxErrorRate;
  double cutoffPenalty = a.bestPenalty + a.parameters.Max_PenaltySpan;
  if (cutoffPenalty > maxInterestingPenaltyAnywhere) cutoffPenalty = maxInterestingPenaltyAnywhere;
  a.nBest = 0;
  for (int i = 0; i < a.nGood; i++) {
    if (!(a.good[i].totalPenalty <= cutoffPenalty)) continue;
    bool dup = false;
    for (int j = 0; j < a.nBest && !dup; j++) {
      const QAl& u = a.good[a.bestIdx[j]];
      const QAl& v = a.good[i];
      if (u.nSeq != v.nSeq) continue;
      bool same = true;
      for (int k = 0; k < u.nSeq; k++) if (!saSame(u.seq[k], v.seq[k])) { same = false; break; }
      if (same) dup = true;
    }
    if (!dup) a.bestIdx[a.nBest++] = (int16_t)i;
  }
}

// ---------------------------------------------------------------- duplication map (M/Readable_DuplicationDetector.java:28-47)
XM_INL bool mayContainDuplicationInRange(const IndexView& ix, int contig, int startIndex, int endIndex) {
  int w = ix.dupWindow;
  int windowStart = startIndex / w, windowEnd = endIndex / w;
  const int32_t* keys = ix.dupKeys + ix.dupKeyStart[contig];
  int n = (int)(ix.dupKeyStart[contig + 1] - ix.dupKeyStart[contig]);
  if (n == 0) return false;
  int lo = 0, hi = n;  // floorEntry(endIndex): last key <= endIndex
  while (lo < hi) { int mid = (lo + hi) >> 1; if (keys[mid] <= endIndex) lo = mid + 1; else hi = mid; }
  if (lo > 0) {
    int pw = keys[lo - 1] / w;
    if (pw >= windowStart && pw <= windowEnd) return true;
  }
  lo = 0; hi = n;  // ceilingEntry(startIndex): first key >= startIndex
  while (lo < hi) { int mid = (lo + hi) >> 1; if (keys[mid] >= startIndex) hi = mid; else lo = mid + 1; }
  if (lo < n) {
    int nw = keys[lo] / w;
    if (nw >= windowStart && nw <= windowEnd) return true;
  }
  return false;
}

// ---------------------------------------------------------------- result streams (layout: include/xmapper_hip.h)
struct OutWriter { int32_t* ints; double* dbls; int64_t ni, nd; };
XM_INL void countQAl(const QAl& q, int64_t& ni, int64_t& nd) {
  ni += 2; nd += 4;
  for (int k = 0; k < q.nSeq; k++) { ni += 3 + 4 * q.seq[k].nb; nd += 2; }
}
XM_INL void writeQAl(OutWriter& w, const QAl& q) {
  w.ints[w.ni++] = q.innerDistance;
  w.ints[w.ni++] = q.nSeq;
  w.dbls[w.nd++] = q.spacingPenalty; w.dbls[w.nd++] = q.overlapMultiplier; w.dbls[w.nd++] = q.duplicationBonus; w.dbls[w.nd++] = q.totalPenalty;
  for (int k = 0; k < q.nSeq; k++) {
    const SeqAl& s = q.seq[k];
    w.ints[w.ni++] = s.contig; w.ints[w.ni++] = s.referenceReversed; w.ints[w.ni++] = s.nb;
    for (int b = 0; b < s.nb; b++) { w.ints[w.ni++] = s.blocks[b].startA; w.ints[w.ni++] = s.blocks[b].startB; w.ints[w.ni++] = s.blocks[b].lenA; w.ints[w.ni++] = s.blocks[b].lenB; }
    w.dbls[w.nd++] = s.totalPenalty; w.dbls[w.nd++] = s.alignedPenalty;
  }
}

// what alignToAncestralReference returns: up to 2 components, each a list of alignments of one aligner
XM_INL double penaltyLowerBound(const ReadCtx& cx, int numMismatchedHashblocks) {  // M/AlignerWorker.java:487-491
  double mutationPenalty = numMismatchedHashblocks * cx.params.MutationPenalty;
  double indelPenalty = cx.ix->minInterestingSize * numMismatchedHashblocks * cx.params.DeletionExtension_Penalty;
  return dmin(mutationPenalty, indelPenalty);
}

// totalLengthForHighConfidence of quicklyConfidentInBestAlignment.  On the device: from the host's table (IndexView::conf); a key the table does not
// hold goes to the miss list and the read stops with XM_ST_NEED_CONF.  (Host simulation of the tests: evaluated in place, by the same host function.)
XM_INL bool confidenceLength(ReadCtx& cx, double penalty, int queryTotalLength, double& value) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (confLookup(cx.ix->conf, cx.ix->confMask, penalty, queryTotalLength, value)) return true;
  ConfMiss* miss = cx.ix->confMiss;
  if (miss) {
    const unsigned long long at = atomicAdd(&miss->n, 1ull);
    if (at < miss->cap) { uint64_t bits; __builtin_memcpy(&bits, &penalty, 8); miss->keys[at].penaltyBits = bits; miss->keys[at].queryLength = queryTotalLength; miss->keys[at].pad = 0; }
  }
  cx.status = XM_ST_NEED_CONF;
  return false;
#else
  value = confidenceLengthOnHost(penalty, queryTotalLength, cx.params.Max_PenaltySpan, cx.params.MutationPenalty, cx.ix->dupGranularity, cx.ix->totalForwardAndReverseSize);
  return true;
#endif
}

XM_NOINL bool quicklyConfidentInBestAlignment(ReadCtx& cx, QMAligner& a, int alIdx, const QMatch& m) {  // :494-587
  if (alIdx < 0) return false;
  XM_TIC(t0);
  const QAl& al = a.good[alIdx];
  for (int k = 0; k < al.nSeq; k++) if (saHasIndel(al.seq[k])) return false;
  int contig = m.c[0].contig;
  int matchStart = qmStartIndexB(m), matchEnd = qmEndIndexB(m);
  double penalty = al.totalPenalty;
  if (penalty <= 0 && cx.params.Max_PenaltySpan < cx.params.getMinPossibleNonzeroPenalty()) return true;
  // :532-540: 1 - (1 - rate)^granularity, the two logarithms and their quotient are the host's (xm_confidence.h): table lookup by (penalty, length)
  double totalLengthForHighConfidence = 0;
  if (!confidenceLength(cx, penalty, qmQueryTotalLength(cx.seed, m), totalLengthForHighConfidence)) return false;  // (status set: the read runs again once the host has the value)
  double matchMiddle = (double)((matchStart + matchEnd) / 2);
  double interestingWindow = jmaxd(totalLengthForHighConfidence, (double)((matchEnd - matchStart + 1) / 2));
  int windowStart = j2i(matchMiddle - interestingWindow);
  int windowEnd = j2i(matchMiddle + interestingWindow);
  bool hasNearbyDuplication = false;
  if (mayContainDuplicationInRange(*cx.ix, contig, windowStart, windowEnd)) hasNearbyDuplication = true;
  else if (matchStart <= interestingWindow) hasNearbyDuplication = true;
  else if (matchEnd >= cx.ix->contigLen[contig] - interestingWindow) hasNearbyDuplication = true;
  XM_TOC(cx.dc, T_CONFIDENT, t0);
  if (hasNearbyDuplication) return false;
  for (int k = 0; k < al.nSeq; k++) if (saHasAmbiguous(cx, al.seq[k])) return false;
  return true;
}

// any base of the mate that is not exactly one of A C G T?  Eight codes per load: a code is unambiguous iff exactly one of its four bits is set
XM_INL bool mateHasAmbiguousBase(const SeqView& q) {
  XM_GLOBAL(const uint8_t)* const g = (XM_GLOBAL(const uint8_t)*)q.base;
  uint64_t bad = 0;
  int i = 0;
  for (; i + 8 <= q.len; i += 8) {
    uint64_t x;
    __builtin_memcpy(&x, (const void*)(q.base + i), 8);
    x &= 0x0F0F0F0F0F0F0F0Full;
    uint64_t pairs = (x & 0x0505050505050505ull) + ((x >> 1) & 0x0505050505050505ull);   // bit counts of the two 2-bit halves
    uint64_t pop = (pairs & 0x0303030303030303ull) + ((pairs >> 2) & 0x0303030303030303ull);  // per byte: 0..4
    bad |= pop ^ 0x0101010101010101ull;
  }
  for (; i < q.len; i++) if (bpIsAmbiguous(g[i])) bad = 1;
  return bad != 0;
}

XM_INL void compInit(ReadCtx& cx, Comp& c, const SeqView& query, const SeqView& rcQuery) {  // M/Counting_HashBlockPath.java:20-37
  const Caps& k = cx.caps;
  Arena& A = cx.persist;
  PBlock* blocks = arenaArray<PBlock>(A, k.maxPyramidBlocks);
  int32_t* ls = arenaArray<int32_t>(A, k.maxLevels + 2);
  c.counters = arenaArray<Counter>(A, k.maxCounters);
  c.good = arenaArray<int16_t>(A, k.maxCounters);
  c.history = arenaArray<QBlock>(A, k.maxHistory);
  c.pending = arenaArray<QBlock>(A, k.maxPending);
  c.hp.items = arenaArray<int16_t>(A, k.maxCounters);
  c.best.items = arenaArray<int16_t>(A, k.maxCounters);
  c.all.items = arenaArray<int16_t>(A, k.maxCounters);
  if (A.overflow) { cx.status = XM_ST_OVERFLOW; return; }
  c.pyr.init(query, blocks, k.maxPyramidBlocks, ls, k.maxLevels, &cx.status);
  c.pyr.dc = cx.dc;
  if (mateHasAmbiguousBase(query)) {  // blocks over an ambiguous base are lists of possibilities (MultiHashBlock): xm_seed.h, MultiStore
    // (any number of ambiguous bases: a mate of nothing but N is walked like any other - the reference bounds the combinations per block, not the bases)
    MultiStore* ms = arenaArray<MultiStore>(A, 1);
    const MultiCaps mc = multiCaps(k.scale);
    const int poolCap = mc.pool, condCap = mc.conds, stackCap = mc.stack, framesCap = mc.frames;
    Poss* pool = arenaArray<Poss>(A, (size_t)poolCap);
    CondEnt* conds = arenaArray<CondEnt>(A, (size_t)condCap);
    CondEnt* stack = arenaArray<CondEnt>(A, (size_t)stackCap);
    MFrame* frames = arenaArray<MFrame>(A, (size_t)framesCap);
    if (A.overflow) { cx.status = XM_ST_OVERFLOW; return; }
    ms->pool = pool; ms->poolUsed = 0; ms->poolCap = poolCap;
    ms->conds = conds; ms->condUsed = 0; ms->condCap = condCap;
    ms->stack = stack; ms->stackCap = stackCap;
    ms->frames = frames; ms->framesCap = framesCap;
    c.pyr.ms = ms;
  }
  pathInit(c.path);
  c.query = query; c.rcQuery = rcQuery;
  c.nCounters = 0; c.nGood = 0; c.foundGood = false; c.done = false; c.nHistory = 0; c.pendHead = c.pendTail = 0;
  c.numBlocksMatchingAnywhere = 0; c.maxNonoverlappingBlockVisited = 0; c.numNonoverlappingBlocksVisited = 0; c.minNumDistinctMismatches = -1;
  c.nextBlockId = 0;
  c.hp.id = c.best.id = c.all.id = 0; c.hp.n = c.best.n = c.all.n = 0;
  int maxPossibleIndel = j2i((query.len * cx.params.MaxErrorRate - cx.params.DeletionStart_Penalty) / cx.params.DeletionExtension_Penalty);
  c.maxIndelLengthToConsider = maxPossibleIndel / 2;
}

// getUnpairedAlignments :602-644
XM_NOINL void getUnpairedAlignments(ReadCtx& cx, ReadResult& rr) {
  rr.nComponents = 2;
  double expectedInnerDistance = cx.in.expectedInner;
  for (int sequenceIndex = 0; sequenceIndex < 2; sequenceIndex++) {
    rr.single[sequenceIndex] = -1;
    rr.empty[sequenceIndex] = 0;
    int len = cx.in.mateLen[sequenceIndex];
    double maxInterestingSubqueryPenalty = len * cx.params.MaxErrorRate;
    int maxNumMismatches = j2i(maxInterestingSubqueryPenalty / cx.params.MutationPenalty);
    ListRef locs = compFindGoodPositionsHavingPriorityUpTo(cx.comps[sequenceIndex], cx.seed, maxNumMismatches);  // findGoodComponentMatches
    if (cx.status) return;
    QMAligner* sub = arenaArray<QMAligner>(cx.persist, 1);
    if (cx.persist.overflow) { cx.status = XM_ST_OVERFLOW; return; }
    qmaInit(cx, *sub, 1, len, cx.caps.maxGoodAlignments / 2);
    if (cx.status) return;
    rr.aligner[sequenceIndex] = sub;
    for (int i = 0; i < locs.n; i++) {
      const Counter& k = cx.comps[sequenceIndex].counters[locs.items[i]];
      SeqMatch sm = counterMatch(k);
      int minInnerDistance;
      if (sequenceIndex % 2 == 1) minInnerDistance = smStartB(sm);
      else minInnerDistance = cx.ix->contigLen[sm.contig] - smEndB(cx.seed, sm);
      double innerDistance = minInnerDistance;
      if (innerDistance < expectedInnerDistance) innerDistance = expectedInnerDistance;
      double spacingPenalty = innerDistance / cx.in.deviation;
      if (spacingPenalty > maxInterestingSubqueryPenalty) continue;
      QMatch qm;
      qm.n = 1; qm.priority = -1; qm.c[0] = sm; qm.hint = 0;
      qmaAlign(cx, *sub, qm, innerDistance);
      if (cx.status) return;
    }
    qmaGetBestAlignments(*sub);
  }
}

// alignToAncestralReference :306-484
// resume: continue a read that stopped with XM_ST_NEED_HEAVY at the candidate recorded in cx.ar (cx, its persistent arena and cx.ar.rr are
// as that run left them; the caller has reset cx.status, re-pointed cx.tmp and raised cx.caps.heavyAllowed)
XM_NOINL void alignRead(ReadCtx& cx, ReadResult& rr, bool resume = false) {
  const ReadIn& in = cx.in;
  AlignReadState& st = cx.ar;
  PathsCounter& pc = cx.pc;
  const SeedEnv& se = cx.seed;
  QMAligner* aligner = nullptr;
  int al = -1;
  if (resume) {
    rr = st.rr;
    aligner = st.aligner;
    if (st.phase == 1) goto resume_optimistic;
    if (st.phase == 3) goto resume_main;
    if (st.phase == 4) goto resume_partial;
    cx.status = XM_ST_INTERNAL;  // (the caller only resumes phases 1, 3 and 4)
    return;
  }
  st.phase = 0;
  if (cx.dc) { cx.dc->reads++; for (int m = 0; m < in.nMates; m++) cx.dc->readBytes += (unsigned long long)((in.mateLen[m] + 1) / 2); }
  rr.nComponents = 1; rr.single[0] = -1; rr.empty[0] = 0; rr.aligner[0] = nullptr;
  for (int m = 0; m < in.nMates; m++) if (in.mateLen[m] > cx.ix->maxHashedLength) { cx.status = XM_ST_NEED_GROW; return; }
  st.queryLength = 0;
  for (int m = 0; m < in.nMates; m++) st.queryLength += in.mateLen[m];
  st.maxInterestingPenalty = st.queryLength * cx.params.MaxErrorRate;
  {
    int maxInnerDistance = j2i(st.maxInterestingPenalty * in.deviation + in.expectedInner);
    cx.seed.ix = cx.ix; cx.seed.caps = &cx.caps; cx.seed.dc = cx.dc; cx.seed.status = &cx.status; cx.seed.listIdCounter = &cx.listIdCounter;
    cx.seed.mateLen = cx.in.mateLen;
    cx.listIdCounter = 0;
    for (int i = 0; i < in.nMates; i++) {
      SeqView fwd = queryView(cx, (uint8_t)(i * 2)), rc = queryView(cx, (uint8_t)(i * 2 + 1));
      if (i > 0) compInit(cx, cx.comps[i], rc, fwd); else compInit(cx, cx.comps[i], fwd, rc);  // :317-318
      if (cx.status) return;
    }
    pc.comps = cx.comps; pc.nComps = in.nMates;
    pc.maxOffsetBetweenComponents = jadd(maxInnerDistance, cx.comps[0].query.len);
    pc.foundNonemptyResult = false; pc.havePrevious = false; pc.nAssembled = 0; pc.nFiltered = 0;
    pc.assembled = arenaArray<QMatch>(cx.persist, cx.caps.maxQM);
    pc.filtered = arenaArray<QMatch>(cx.persist, cx.caps.maxQM);
    pc.nearby = arenaArray<int16_t>(cx.persist, cx.caps.maxCounters);
    aligner = arenaArray<QMAligner>(cx.persist, 1);
    if (cx.persist.overflow) { cx.status = XM_ST_OVERFLOW; return; }
  }
  st.aligner = aligner;
  st.optimisticBestAlignment = -1;
  st.haveOptimisticMatch = 0;
  st.numMismatches = 0;
  pcOptimisticGetBestMatches(pc, se);
  if (cx.status) return;
  qmaInit(cx, *aligner, in.nMates, st.queryLength, cx.caps.maxGoodAlignments);
  if (cx.status) return;
  rr.aligner[0] = aligner;
  if (pc.nFiltered == 1) {
    st.optimisticBestMatch = pc.filtered[0];
    st.haveOptimisticMatch = 1;
    st.phase = 1;
    st.rr = rr;
resume_optimistic:
    if (cx.dc) { st.candidatesAtCall = cx.dc->candidatesExtended; st.refWindowBytesAtCall = cx.dc->refWindowBytes; }  // (behind the label: a resumed run counts into another lane's counters)
    st.optimisticBestAlignment = qmaAlign(cx, *aligner, st.optimisticBestMatch, 0);
    if (cx.status) return;
    st.phase = 2;
    const bool quick = quicklyConfidentInBestAlignment(cx, *aligner, st.optimisticBestAlignment, st.optimisticBestMatch);
    if (cx.status) return;  // (XM_ST_NEED_CONF: the host has a value to add to the confidence table)
    if (quick) {
      if (cx.dc) cx.dc->quickAccepts++;
      rr.single[0] = st.optimisticBestAlignment;
      return;
    }
  }
  st.phase = 2;
  if (st.optimisticBestAlignment >= 0) {
    while (true) {
      double possiblePenalty = penaltyLowerBound(cx, st.numMismatches);
      if (possiblePenalty > aligner->good[st.optimisticBestAlignment].totalPenalty + cx.params.Max_PenaltySpan) {
        rr.single[0] = st.optimisticBestAlignment;
        return;
      }
      pcFindGoodPositionsHavingPriority(pc, se, st.numMismatches);
      if (cx.status) return;
      st.numMismatches++;
      bool done = false;
      for (int i = 0; i < pc.nFiltered; i++) if (!qmSamePosition(st.optimisticBestMatch, pc.filtered[i])) { done = true; break; }
      if (done) break;
    }
  }
  st.bestPenalty = (double)INT32_MAX;
  st.candidateNumMismatches = 0;
  while (true) {
    st.estimatedPenalty = penaltyLowerBound(cx, st.candidateNumMismatches);
    if (st.estimatedPenalty > st.bestPenalty + cx.params.Max_PenaltySpan) break;
    if (st.candidateNumMismatches > pcGetNumBlocks(pc)) break;
    pcFindGoodPositionsHavingPriority(pc, se, st.candidateNumMismatches);
    if (cx.status) return;
    for (st.i = 0; st.i < pc.nFiltered; st.i++) {
      if (st.haveOptimisticMatch && qmSamePosition(pc.filtered[st.i], st.optimisticBestMatch)) {
        al = st.optimisticBestAlignment;
      } else {
        st.phase = 3;
        st.rr = rr;
resume_main:
        if (cx.dc) { st.candidatesAtCall = cx.dc->candidatesExtended; st.refWindowBytesAtCall = cx.dc->refWindowBytes; }
        al = qmaAlign(cx, *aligner, pc.filtered[st.i], 0);
        if (cx.status) return;
        st.phase = 2;
      }
      if (cx.status) return;
      if (al >= 0) {
        double penalty = aligner->good[al].totalPenalty;
        if (st.bestPenalty > penalty) st.bestPenalty = penalty;
      }
    }
    if (st.estimatedPenalty >= st.maxInterestingPenalty) break;
    st.candidateNumMismatches++;
  }
  qmaGetBestAlignments(*aligner);
  if (aligner->nBest < 1 && in.nMates > 1) {
    pcFindPartiallyGoodPositions(pc, se);
    if (cx.status) return;
    for (st.i = 0; st.i < pc.nFiltered; st.i++) {
      st.phase = 4;
      st.rr = rr;
resume_partial:
      if (cx.dc) { st.candidatesAtCall = cx.dc->candidatesExtended; st.refWindowBytesAtCall = cx.dc->refWindowBytes; }
      al = qmaAlign(cx, *aligner, pc.filtered[st.i], 0);
      if (cx.status) return;
      st.phase = 2;
      if (al >= 0) {
        double penalty = aligner->good[al].totalPenalty;
        if (st.bestPenalty > penalty) st.bestPenalty = penalty;
      }
    }
  }
  st.phase = 5;
  qmaGetBestAlignments(*aligner);
  {
    int numBest = aligner->nBest;
    if (numBest < 1 && in.nMates > 1) {
      getUnpairedAlignments(cx, rr);
      if (cx.status) return;
    }
    if ((int64_t)numBest > (int64_t)cx.params.MaxNumMatches) {  // :476-481
      rr.nComponents = 1; rr.single[0] = -1; rr.empty[0] = 1;
    }
  }
}

XM_INL void resultSize(const ReadResult& rr, int64_t& ni, int64_t& nd) {
  ni = 1; nd = 0;
  for (int c = 0; c < rr.nComponents; c++) {
    ni += 1;
    if (rr.empty[c] || !rr.aligner[c]) continue;
    if (rr.single[c] >= 0) { countQAl(rr.aligner[c]->good[rr.single[c]], ni, nd); continue; }
    for (int i = 0; i < rr.aligner[c]->nBest; i++) countQAl(rr.aligner[c]->good[rr.aligner[c]->bestIdx[i]], ni, nd);
  }
}
XM_INL void resultWrite(const ReadResult& rr, OutWriter& w, DevCounters* dc) {
  w.ints[w.ni++] = rr.nComponents;
  for (int c = 0; c < rr.nComponents; c++) {
    if (rr.empty[c] || !rr.aligner[c]) { w.ints[w.ni++] = 0; continue; }
    if (rr.single[c] >= 0) { w.ints[w.ni++] = 1; writeQAl(w, rr.aligner[c]->good[rr.single[c]]); if (dc) dc->alignmentsOut++; continue; }
    w.ints[w.ni++] = rr.aligner[c]->nBest;
    for (int i = 0; i < rr.aligner[c]->nBest; i++) { writeQAl(w, rr.aligner[c]->good[rr.aligner[c]->bestIdx[i]]); if (dc) dc->alignmentsOut++; }
  }
}

// ---------------------------------------------------------------- light pass -> gapped pass hand-over
// In the light pass a read's persistent arena is a region of its own (not the lane's), so when the read stops with XM_ST_NEED_HEAVY its
// seeding state survives: the context is copied next to it (SavedRead) and the gapped pass continues from there on another lane instead
// of seeding the read again (a quarter of the gapped pass's wave time went into that).
struct SavedRead {
  int32_t valid;       // 1: cx can be resumed (alignRead phases 1, 3, 4)
  int32_t pad;
  DevCounters partial; // what the stopped run had counted for this read
  ReadCtx cx;
};
XM_INL void dcAccumulate(DevCounters& a, const DevCounters& b, bool subtract) {
  unsigned long long* x = (unsigned long long*)&a;
  const unsigned long long* y = (const unsigned long long*)&b;
  for (size_t i = 0; i < sizeof(DevCounters) / sizeof(unsigned long long); i++) x[i] = subtract ? x[i] - y[i] : x[i] + y[i];
}
// the usable part of a read's region when its tail holds the SavedRead
XM_INL size_t retainedPersistBytes(size_t regionBytes) { return (regionBytes - sizeof(SavedRead)) & ~(size_t)15; }
XM_INL SavedRead* savedReadOf(void* region, size_t regionBytes) { return (SavedRead*)((uint8_t*)region + retainedPersistBytes(regionBytes)); }
// an arena of runRead splits 5 : 7 into the persistent part and the temporaries; a read's region = the persistent part + its SavedRead
XM_INL size_t arenaPersistBytes(size_t arenaBytes) { return (arenaBytes * 5 / 12) & ~(size_t)15; }
XM_INL size_t retainedRegionBytes(size_t arenaBytes) { return arenaPersistBytes(arenaBytes) + ((sizeof(SavedRead) + 15) & ~(size_t)15); }

// what the extension chain allocates in the temporaries follows the scale of the pass that runs the chain; what lives in the read's
// persistent arena (maxBlocks included: the accepted alignments are stored there) keeps the sizes of the scale the read was seeded with
// Chains of long reads (scale 16 and up: mates over 320 bases): XM_LONG_CHAIN_NODES times the search nodes of the scale, twice its matcher sections and
// its matcher tables whole.  A read that outgrows a capacity of the gapped pass is run again from its start at four times the scratch in a pass of its own,
// one read per wave, that lasts as long as its slowest read (0.7 s and 3.2 s behind gapped passes of 4.2 s and 7.8 s per 100 k 1 kb pieces of 10 kb reads);
// what those reads outgrew was the 640 sections of a matcher (windows of over 5 000 bases) and, fewer, the 24 576 nodes of a search
// (profiles/r03/NOTES.md 19).  chainExtraTmpBytes: what this adds to a lane's temporaries.
#ifndef XM_LONG_CHAIN_NODES
#define XM_LONG_CHAIN_NODES 4
#endif
XM_INL size_t chainExtraTmpBytes(int chainScale) {
  if (chainScale < 16) return 0;
  const Caps g = makeCaps(chainScale);
  return ((size_t)(XM_LONG_CHAIN_NODES - 1) * ((size_t)g.maxNodes * (sizeof(PNode) + 8) + (size_t)g.nodeHash * 4)  // nodes + list entries, cell hash
          + 3 * ((size_t)g.maxSections + (size_t)g.matcherEntries) + 4095) & ~(size_t)4095;                       // three matchers: sections, the other half of the tables
}
// the matcher part of chainExtraTmpBytes (what a lane of the scheduler kernel adds: its searches have their own arrays)
XM_INL size_t chainExtraMatcherBytes(int chainScale) {
  if (chainScale < 16) return 0;
  const Caps g = makeCaps(chainScale);
  return (3 * ((size_t)g.maxSections + (size_t)g.matcherEntries) + 4095) & ~(size_t)4095;
}
XM_INL void applyChainCaps(Caps& c, int chainScale) {
  const Caps g = makeCaps(chainScale);
  c.maxNodes = g.maxNodes; c.nodeHash = g.nodeHash; c.gridCap = g.gridCap; c.maxBuckets = g.maxBuckets; c.bucketHash = g.bucketHash;
  if (chainScale >= 16) { c.maxNodes *= XM_LONG_CHAIN_NODES; c.nodeHash *= XM_LONG_CHAIN_NODES; }
  // (the matcher tables of a 150 bp read's windows take ~10 KB each: half of the scale's 48 KB is room enough, and it is a third of a lane's temporaries)
  c.matcherEntries = (chainScale >= 4 && chainScale < 16) ? g.matcherEntries / 2 : g.matcherEntries; c.maxSections = chainScale >= 16 ? 2 * g.maxSections : g.maxSections;
  c.maxPieces = g.maxPieces; c.maxCountMap = g.maxCountMap;
  c.maxJoined = g.maxJoined;
}

// A read whose persistent arena is a region of its own and whose temporaries are the lane's arena.  Light pass (heavyAllowed < 2): a read
// that stops in front of the gapped chain leaves a SavedRead at the tail of the region.  Gapped pass, read without saved state
// (heavyAllowed 2, chainScale = the gapped scale): seeded at `scale`, chain scratch of the gapped pass.
XM_INL void runReadRetaining(ReadCtx& cx, const IndexView* ix, const Params& params, const ReadIn& in, int scale, void* region, size_t regionBytes, void* laneArena, size_t laneArenaBytes,
                             DevCounters* dc, ReadResult& rr, int heavyAllowed, int chainScale = 0) {
  cx.ix = ix; cx.caps = makeCaps(scale); cx.caps.heavyAllowed = heavyAllowed; cx.dc = dc; cx.status = XM_OK; cx.in = in; cx.params = params;
  if (chainScale > 0 && chainScale != scale) applyChainCaps(cx.caps, chainScale);
  cx.heavyHint = 0;
  cx.params.StartingInsertionStartFree = 0;
  cx.persist.init(region, retainedPersistBytes(regionBytes));
  cx.tmp.init(laneArena, laneArenaBytes);
  SavedRead* sv = savedReadOf(region, regionBytes);
  sv->valid = 0;
  DevCounters before = dc ? *dc : DevCounters();
  XM_TIC(t0);
  alignRead(cx, rr);
  XM_TOC(dc, T_TOTAL, t0);
  if (cx.status == XM_ST_NEED_HEAVY && (cx.ar.phase == 1 || cx.ar.phase == 3 || cx.ar.phase == 4)) {
    sv->cx = cx;
    if (dc) {
      sv->partial = *dc;
      dcAccumulate(sv->partial, before, true);
      sv->partial.candidatesExtended -= dc->candidatesExtended - cx.ar.candidatesAtCall;  // the stopped candidate is counted by the run that finishes it
      sv->partial.refWindowBytes -= dc->refWindowBytes - cx.ar.refWindowBytesAtCall;
    }
    sv->valid = 1;
  }
}
// gapped pass, a read the light pass handed over: the context back on this lane, pointers into the context re-seated, the chain's scratch
// capacities of the gapped pass, temporaries in this lane's arena; then on from the candidate that needed the chain, to the read's end.  The saved
// state is consumed (pyramid levels, hit lists and the aligner advance in place): a read is resumed once.
XM_INL void runReadResumed(ReadCtx& cx, SavedRead* sv, const IndexView* ix, int chainScale, void* laneArena, size_t laneArenaBytes, DevCounters* dc, ReadResult& rr) {
  cx = sv->cx;
  cx.ix = ix; cx.dc = dc; cx.status = XM_OK;
  cx.seed.ix = ix; cx.seed.caps = &cx.caps; cx.seed.dc = dc; cx.seed.status = &cx.status; cx.seed.listIdCounter = &cx.listIdCounter; cx.seed.mateLen = cx.in.mateLen;
  for (int m = 0; m < 2; m++) { cx.comps[m].pyr.status = &cx.status; cx.comps[m].pyr.dc = dc; }
  cx.pc.comps = cx.comps;
  cx.tmp.init(laneArena, laneArenaBytes);
  if (chainScale > 0) applyChainCaps(cx.caps, chainScale);
  cx.caps.heavyAllowed = 2;
  if (dc) dcAccumulate(*dc, sv->partial, false);
  XM_TIC(t0);
  alignRead(cx, rr, true);
  XM_TOC(dc, T_TOTAL, t0);
  sv->valid = 0;
}

// Carve a lane's arena and align one read.  `arena` must be 16-byte aligned.
XM_INL void runRead(ReadCtx& cx, const IndexView* ix, const Params& params, const ReadIn& in, int scale, void* arena, size_t arenaBytes, DevCounters* dc, ReadResult& rr, int heavyAllowed = 2) {
  cx.ix = ix; cx.caps = makeCaps(scale); cx.caps.heavyAllowed = heavyAllowed; cx.dc = dc; cx.status = XM_OK; cx.in = in; cx.params = params;
  cx.heavyHint = 0;
  cx.params.StartingInsertionStartFree = 0;
  const size_t persistBytes = arenaPersistBytes(arenaBytes);
  cx.persist.init(arena, persistBytes);
  cx.tmp.init((uint8_t*)arena + persistBytes, arenaBytes - persistBytes);
  XM_TIC(t0);
  alignRead(cx, rr);
  XM_TOC(dc, T_TOTAL, t0);
}

}  // namespace xm
