// xmapper-hip device core: the extension chain.
// Replaces (per candidate, on the GPU): M/StraightAligner.java, M/SkipHighAmbiguity_Aligner.java, M/HashBlock_Aligner.java,
// M/HashBlock_Matcher.java, M/CountMap.java, M/BlockAligner.java, M/PathAligner.java (+ _Runner, AlignmentNode,
// AlignmentAnalysis, PenaltyAnalysis), M/AlignmentParameters.java:73-180.
// The decorator chain of M/QueryMatch_Aligner.java:18-29 is unrolled into straight1 -> skipHighAmbiguity -> hashBlock<1> ->
// blockAligner -> straight2 -> hashBlock<2> -> straight3 -> pathAligner; HashBlock_Aligner's tail self-call is a loop.
// All penalties are IEEE doubles evaluated in the reference's order; PathAligner is an exact emulation of the reference's
// best-first search (bucket-by-exact-double, insertion order inside a bucket, stale re-exploration, == tie-breaks).
#pragma once
#include "xm_seed.h"

namespace xm {

struct alignas(16) ABlock { int32_t startA, startB, lenA, lenB; };  // AlignedBlock; moved as one 16-byte word
XM_INL int abEndA(const ABlock& b) { return b.startA + b.lenA; }
XM_INL int abEndB(const ABlock& b) { return b.startB + b.lenB; }
XM_INL int abIndelType(const ABlock& b) { return b.lenA == b.lenB ? 0 : (b.lenA > b.lenB ? 1 : 2); }

struct SeqAl {  // SequenceAlignment; `blocks` has room for caps.maxBlocks entries
  ABlock* blocks;
  int32_t nb;
  int32_t contig;
  uint8_t referenceReversed;
  uint8_t seqAId;
  double totalPenalty, alignedPenalty;
};
XM_INL int saStartA(const SeqAl& a) { return a.blocks[0].startA; }
XM_INL int saEndA(const SeqAl& a) { return abEndA(a.blocks[a.nb - 1]); }
XM_INL int saStartB(const SeqAl& a) { return a.blocks[0].startB; }
XM_INL int saEndB(const SeqAl& a) { return abEndB(a.blocks[a.nb - 1]); }
XM_INL void saCopy(SeqAl& dst, const SeqAl& src) {
  dst.nb = src.nb; dst.contig = src.contig; dst.referenceReversed = src.referenceReversed; dst.seqAId = src.seqAId;
  dst.totalPenalty = src.totalPenalty; dst.alignedPenalty = src.alignedPenalty;
  for (int i = 0; i < src.nb; i++) dst.blocks[i] = src.blocks[i];
}

struct Section { int32_t start, end; };  // SequenceSection over a known sequence
XM_INL int secLen(const Section& s) { return s.end - s.start; }

struct Matcher {  // HashBlock_Matcher
  int32_t referenceStart, referenceLength, blockLength, sectionLength, maxSectionIndex, numPossibilities, maxPossibility;
  int32_t nSections;   // locations.size()
  uint64_t presentMask; // bit i = present[i] for i < 64: which sections exist is asked for every lookup, and from a register it costs no trip to memory
  uint8_t* present;    // [maxSections] 0 = "null" entry
  int16_t* tables;     // [nSections][numPossibilities], value = position - referenceStart, or -1 / -2
  int32_t tableCap, maxSections;
};
enum : int { M_NO_MATCHES = -1, M_MULTIPLE = -2, M_UNKNOWN = -3 };

struct Analysis {  // AlignmentAnalysis
  Matcher* matcher;
  int32_t predictedBestOffset, lastCheckedOffset;
  bool confidentAboutBestOffset;
  double maxInsertionExtensionPenalty, maxDeletionExtensionPenalty;
};

struct ExtEnv {  // everything the chain needs
  const Caps* caps;
  DevCounters* dc;
  int32_t* status;
  Arena* tmp;        // stack-discipline scratch (mark = tmp->used, release = restore)
  SeqView query;     // sequenceA
  SeqView reference; // sequenceB (forward contig)
  int32_t contig;
  Matcher* slotA; Matcher* slotB; Matcher* slotT;
  const int32_t* baLogStep; // IndexView::baLogStep (BlockAligner's log term, evaluated on the host)
  float* heavyHint;         // light pass: where a read that stops with XM_ST_NEED_HEAVY leaves its cost hint
};

// ---------------------------------------------------------------- penalties (M/AlignmentParameters.java)
XM_INL double blockPenalty(const SeqView& q, const SeqView& r, const Params& p, const ABlock& b) {  // :106-126
  double penalty = 0;
  if (b.lenA == b.lenB) {
    // Eight positions per pair of loads (a byte-by-byte loop waits for memory once per base).  A position whose two codes are the same single base
    // adds getPenalty = AmbiguityPenalty * 0 = +0.0, which leaves the sum as it is; only the others are added, in order.
    int i = 0;
    while (i + 8 <= b.lenA && b.startA + i >= 0 && b.startB + i >= 0 && b.startA + i + 8 <= q.len && b.startB + i + 8 <= r.len) {
      const uint64_t wa = seqWord8(q, b.startA + i), wb = seqWord8(r, b.startB + i);
      uint64_t todo = seqZeroBytes(wa & wb) | seqAmbiguousBytes(wa | wb);
      while (todo) {
        const int k = __builtin_ctzll(todo) >> 3;
        penalty += p.getPenalty((uint8_t)(wa >> (8 * k)), (uint8_t)(wb >> (8 * k)));
        todo &= todo - 1;
      }
      i += 8;
    }
    for (; i < b.lenA; i++) penalty += p.getPenalty(q.at(b.startA + i), r.at(b.startB + i));
  } else if (b.lenA > 0) {
    penalty += p.InsertionStart_Penalty;
    penalty += p.InsertionExtension_Penalty * b.lenA;
  } else {
    penalty += p.DeletionStart_Penalty;
    penalty += p.DeletionExtension_Penalty * b.lenB;
  }
  return penalty;
}
// newSequenceAlignment :73-95 over out.blocks[0..nb)
XM_INL void finishSeqAl(const ExtEnv& e, const Params& pIn, SeqAl& out, bool referenceReversed) {
  const Params p = pIn;
  const SeqView q = e.query, r = e.reference;
  const ABlock* const blocks = out.blocks;
  const int nbL = out.nb;
  int alignedQueryLength = 0;
  double totalPenalty = 0;
  for (int i = 0; i < nbL; i++) {
    const ABlock b = blocks[i];
    totalPenalty += blockPenalty(q, r, p, b);
    alignedQueryLength += b.lenA;
  }
  if (out.nb > 0 && p.StartingInsertionStartFree && out.blocks[0].lenB == 0) totalPenalty -= p.InsertionStart_Penalty;
  double alignedPenalty = totalPenalty;
  if (out.nb > 0) {
    int unalignedQueryLength = e.query.len - alignedQueryLength;
    totalPenalty += (double)unalignedQueryLength * p.UnalignedPenalty;
  }
  out.referenceReversed = referenceReversed ? 1 : 0;
  out.seqAId = e.query.id;
  out.contig = e.contig;
  out.totalPenalty = totalPenalty;
  out.alignedPenalty = alignedPenalty;
}

// ---------------------------------------------------------------- HashBlock_Matcher (M/HashBlock_Matcher.java)
XM_INL int encodedCharToInt(uint8_t b) { return b == 1 ? 0 : b == 2 ? 1 : b == 4 ? 2 : 3; }  // callers guarantee unambiguous
XM_INL int matcherSectionIndex(const Matcher& m, int referenceIndex) { return (referenceIndex - m.referenceStart) / m.sectionLength; }
XM_INL void matcherInit(Matcher& m, const ExtEnv& e, const Section& referenceSection, int sectionLength) {  // :14-29
  if (sectionLength < 1) sectionLength = 1;
  // (int)(log(5*sectionLength)/log(4) + 1): 5*sectionLength is never a power of 4, so this is floor(log4(v)) + 1
  int v = sectionLength * 5, k = 0;
  long long pw = 1;
  while (pw * 4 <= v) { pw *= 4; k++; }
  m.blockLength = k + 1;
  if (m.blockLength < 3) m.blockLength = 3;
  m.referenceStart = referenceSection.start;
  m.referenceLength = secLen(referenceSection);
  m.sectionLength = sectionLength;
  m.maxSectionIndex = matcherSectionIndex(m, e.reference.len - 1);
  m.numPossibilities = 1 << (2 * m.blockLength);
  m.maxPossibility = m.numPossibilities - 1;
  m.nSections = 0;
  m.presentMask = 0;
}
XM_INL int matcherEncodeBlock(const Matcher& m, const SeqView& s, int index) {  // :79-91
  if (index + m.blockLength > s.len) return M_UNKNOWN;
  int sum = 0;
  for (int i = 0; i < m.blockLength; i++) {
    uint8_t here = s.at(index + i);
    if (bpIsAmbiguous(here)) return M_UNKNOWN;
    sum = sum * 4 + encodedCharToInt(here);
  }
  return sum;
}
// GPU note: everything below takes the sequences and the matcher BY VALUE / as locals.  Structures reached through a pointer
// (ExtEnv, Matcher in the arena) cost a dependent memory round trip per field on this hardware, and the compiler must reload
// them after every store it cannot disambiguate; locals stay in registers.
XM_INL int xmGroupShift();
#if defined(__HIP_DEVICE_COMPILE__) && !defined(XM_WAVE_UNIFORM)
// The same table built by the EIGHT lanes that run a long read (xmSetPairMode 3).  What the reference's loop leaves in an entry does not depend on the order of the
// positions - the position of a code that occurs once, M_MULTIPLE for one that occurs more often - so every lane takes an eighth of the section (a run of
// consecutive positions: the code rolls from one to the next), and a round handles one position of each lane: eight table entries read, codes that meet in a round
// found by comparing across the lanes (data-parallel primitives, no memory), eight entries written.  Lanes of one wave: their memory operations complete in program
// order, so a round sees what the round before wrote.  false = not done (an ambiguous base in the section: the (sic) of the loop below - a block with an ambiguous
// last base leaves the rolling code at its stale value - is order-dependent; the caller runs that loop).
XM_INL bool matcherIndexSectionEight(const Matcher& m, const SeqView& ref, int sectionIndex, int16_t* section) {
  const int g = (int)__lane_id() & 7;
  const int startIndex = m.referenceStart + sectionIndex * m.sectionLength;
  const int endIndex = imin(startIndex + m.sectionLength, m.referenceStart + m.referenceLength - m.blockLength);
  if (endIndex > startIndex && endIndex - 1 + m.blockLength > ref.len) return false;
  {
    struct alignas(16) W { uint32_t w[4]; };
    W* const t = (W*)section;
    W v;
    v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0xFFFFFFFFu;
    const int nW = m.numPossibilities / 8;
    for (int i = g; i < nW; i += 8) t[i] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  XM_GLOBAL(int16_t)* const table = (XM_GLOBAL(int16_t)*)section;
  const int count = imax(endIndex - startIndex, 0), chunk = (count + 7) >> 3;
  int i = startIndex + g * chunk;
  const int iEnd = imin(i + chunk, endIndex);
  bool amb = false;
  int code = 0;
  if (i < iEnd) {
    for (int b = 0; b < m.blockLength - 1; b++) { const uint8_t c = ref.at(i + b); amb |= bpIsAmbiguous(c); code = code * 4 + encodedCharToInt(c); }
  }
  for (int r = 0; r < chunk; r++, i++) {
    const bool valid = i < iEnd;
    int enc = -1 - g;                                        // (no position: a value no other lane holds)
    if (valid) {
      const uint8_t c = ref.at(i + m.blockLength - 1);
      amb |= bpIsAmbiguous(c);
      code = ((code * 4) & m.maxPossibility) + encodedCharToInt(c);
      enc = code;
    }
    const int16_t cur = valid ? table[enc] : (int16_t)0;
    // the codes of the other seven lanes: the three other lanes of the quad, then the other quad (row_half_mirror) and its three rotations
    auto other = [](int v, auto ctrl) { int r = __builtin_amdgcn_update_dpp(v, v, decltype(ctrl)::value, 0xF, 0xF, false); asm volatile("" : "+v"(r)); return r; };  // (pinned: xm_bound.h, BoundGroup::dpp)
    using C39 = std::integral_constant<int, 0x39>; using C4E = std::integral_constant<int, 0x4E>; using C93 = std::integral_constant<int, 0x93>; using C141 = std::integral_constant<int, 0x141>;
    const int q1 = other(enc, C39()), q2 = other(enc, C4E()), q3 = other(enc, C93()), h0 = other(enc, C141());
    const int h1 = other(h0, C39()), h2 = other(h0, C4E()), h3 = other(h0, C93());
    const bool dup = (enc == q1) | (enc == q2) | (enc == q3) | (enc == h0) | (enc == h1) | (enc == h2) | (enc == h3);
    if (valid) table[enc] = (dup || cur != (int16_t)M_NO_MATCHES) ? (int16_t)M_MULTIPLE : (int16_t)(i - m.referenceStart);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const uint32_t anyAmb = (uint32_t)(__ballot(amb ? 1 : 0) >> ((int)__lane_id() & ~7)) & 0xFFu;
  return anyAmb == 0;
}
#endif
XM_NOINL void matcherIndexSection(const Matcher m, const SeqView ref, int sectionIndex, int16_t* section, DevCounters* dc) {  // :40-77
  XM_TIC(t0);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(XM_WAVE_UNIFORM)
  if (xmGroupShift() == 3 && matcherIndexSectionEight(m, ref, sectionIndex, section)) { XM_TOC(dc, T_MATCHER_INDEX, t0); return; }
#endif
  {  // numPossibilities is a power of four >= 64 and the tables are 16-byte aligned: fill with 16-byte stores
    struct alignas(16) W { uint32_t w[4]; };
    W* const t = (W*)section;
    W v;
    v.w[0] = v.w[1] = v.w[2] = v.w[3] = 0xFFFFFFFFu;  // M_NO_MATCHES (-1) in every int16
    const int nW = m.numPossibilities / 8;
    for (int i = 0; i < nW; i++) t[i] = v;
  }
  int previousEncoded = M_UNKNOWN;
  int startIndex = m.referenceStart + sectionIndex * m.sectionLength;
  int endIndex = imin(startIndex + m.sectionLength, m.referenceStart + m.referenceLength - m.blockLength);
  // Eight positions per round: their codes first (registers), then the eight table entries with independent loads (one memory round
  // trip instead of eight dependent ones: the table is in the lane's arena in HBM), then the updates in position order; a position
  // whose code already occurred earlier in the same round takes that position's result instead of the stale table entry.
  XM_GLOBAL(int16_t)* const table = (XM_GLOBAL(int16_t)*)section;
  for (int i0 = startIndex; i0 < endIndex; i0 += 8) {
    int enc[8];
    int16_t cur[8], neu[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = i0 + k;
      int encoded = M_UNKNOWN;
      if (i < endIndex) {
        if (previousEncoded == M_UNKNOWN) {
          encoded = matcherEncodeBlock(m, ref, i);
        } else {
          uint8_t nextChar = ref.at(i + m.blockLength - 1);
          if (bpIsAmbiguous(nextChar)) encoded = M_UNKNOWN;
          else encoded = ((previousEncoded * 4) & m.maxPossibility) + encodedCharToInt(nextChar);
        }
        if (encoded != M_UNKNOWN) previousEncoded = encoded;  // (sic) an unknown block leaves previousEncoded at its stale value
      }
      enc[k] = encoded;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) cur[k] = enc[k] >= 0 ? table[enc[k]] : (int16_t)M_NO_MATCHES;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int16_t existing = cur[k];
#pragma unroll
      for (int j = 0; j < k; j++) if (enc[j] >= 0 && enc[j] == enc[k]) existing = neu[j];
      neu[k] = (existing == M_NO_MATCHES) ? (int16_t)(i0 + k - m.referenceStart) : (int16_t)M_MULTIPLE;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) if (enc[k] >= 0) table[enc[k]] = neu[k];
  }
  XM_TOC(dc, T_MATCHER_INDEX, t0);
}
// getSection :203-215; returns -1 for a "null" entry (a section skipped by an earlier jump), else the section slot
XM_INL int matcherGetSection(Matcher& m, const SeqView& ref, int index, bool& overflow, DevCounters* dc) {
  if (m.nSections > index) return (index < 64 ? ((m.presentMask >> index) & 1) != 0 : m.present[index] != 0) ? index : -1;
  if (index >= m.maxSections || (long long)(index + 1) * m.numPossibilities > m.tableCap) { overflow = true; return -1; }
  while (m.nSections <= index) m.present[m.nSections++] = 0;
  m.present[index] = 1;
  if (index < 64) m.presentMask |= 1ull << index;
  matcherIndexSection(m, ref, index, m.tables + (size_t)index * m.numPossibilities, dc);
  return index;
}
XM_INL bool matcherCanPositionsMatch(const Matcher& m, const SeqView& query, const SeqView& ref, int queryIndex, int referenceIndex) {  // :159-171
  if (referenceIndex + m.blockLength > m.referenceStart + m.referenceLength) return false;
  for (int i = 0; i < m.blockLength; i++) if (!bpCanMatch(query.at(queryIndex + i), ref.at(referenceIndex + i))) return false;
  return true;
}
// encodeBlock(query, index) with a one-entry rolling cache (consecutive block starts differ by one base)
struct EncodeCache { int32_t index, code; };
XM_INL int matcherEncodeQuery(const Matcher& m, const SeqView& query, int index, EncodeCache& c) {
  int code;
  if (c.code >= 0 && index == c.index + 1 && index + m.blockLength <= query.len) {
    uint8_t here = query.at(index + m.blockLength - 1);
    code = bpIsAmbiguous(here) ? (int)M_UNKNOWN : (((c.code * 4) & m.maxPossibility) + encodedCharToInt(here));
  } else {
    code = matcherEncodeBlock(m, query, index);
  }
  c.index = index;
  c.code = code;
  return code;
}
XM_INL int matcherLookup(Matcher& m, const SeqView& query, const SeqView& ref, int queryIndex, int minReferenceIndex, int maxReferenceIndex, EncodeCache& cache,
                         bool& overflow, DevCounters* dc) {  // :98-141
  if (minReferenceIndex < 0) return M_UNKNOWN;
  if (maxReferenceIndex > ref.len) return M_UNKNOWN;
  int encoded = matcherEncodeQuery(m, query, queryIndex, cache);
  if (encoded < 0) return M_UNKNOWN;
  int matched = M_NO_MATCHES;
  int minSectionIndex = imax(0, matcherSectionIndex(m, minReferenceIndex));
  int maxSection = imin(m.maxSectionIndex, matcherSectionIndex(m, maxReferenceIndex));
  // the table entries of the (up to four) sections that exist already: read together, before the loop that looks at them one after the other
  int16_t ahead[4] = {0, 0, 0, 0};
  bool haveAhead[4] = {false, false, false, false};
  if (m.sectionLength >= 3) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int si = minSectionIndex + k;
      if (si <= maxSection && si < m.nSections && si < 64 && ((m.presentMask >> si) & 1)) { ahead[k] = m.tables[(size_t)si * m.numPossibilities + encoded]; haveAhead[k] = true; }
    }
  }
  for (int sectionIndex = minSectionIndex; sectionIndex <= maxSection; sectionIndex++) {
    int slot = matcherGetSection(m, ref, sectionIndex, overflow, dc);
    if (overflow) return M_UNKNOWN;
    int lookedUp;
    if (m.sectionLength < 3) {  // scanSection :143-157
      lookedUp = M_NO_MATCHES;
      int startIndex = m.referenceStart + sectionIndex * m.sectionLength;
      int endIndex = startIndex + m.sectionLength;
      for (int i = startIndex; i < endIndex; i++) {
        if (matcherCanPositionsMatch(m, query, ref, queryIndex, i)) {
          if (lookedUp == M_NO_MATCHES) lookedUp = i; else { lookedUp = M_MULTIPLE; break; }
        }
      }
    } else if (slot >= 0) {
      const int k = sectionIndex - minSectionIndex;
      int16_t v = (k < 4 && haveAhead[k]) ? ahead[k] : m.tables[(size_t)slot * m.numPossibilities + encoded];
      lookedUp = v >= 0 ? (int)v + m.referenceStart : (int)v;
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

// ---------------------------------------------------------------- CountMap (M/CountMap.java)
struct CountMap {
  int32_t mostPopularKey, mostPopularCount, n, cap;
  bool haveCounts;
  int32_t* keys; int32_t* vals;
  int32_t* status;
  XM_INL void put(int key, int value) {
    for (int i = 0; i < n; i++) if (keys[i] == key) { vals[i] = value; return; }
    if (n >= cap) { *status = XM_ST_OVERFLOW; return; }
    keys[n] = key; vals[n] = value; n++;
  }
  XM_INL void add(int key, int value) {
    if (key == mostPopularKey || mostPopularCount == 0) {
      mostPopularCount += value;
      mostPopularKey = key;
      if (haveCounts) put(mostPopularKey, mostPopularCount);
    } else {
      if (!haveCounts) { haveCounts = true; put(mostPopularKey, mostPopularCount); }
      int count = value;
      for (int i = 0; i < n; i++) if (keys[i] == key) { count = vals[i] + value; break; }
      put(key, count);
      if (count > mostPopularCount) { mostPopularKey = key; mostPopularCount = count; }
    }
  }
};

// ---------------------------------------------------------------- PathAligner (M/PathAligner.java)
struct PNode {  // AlignmentNode, 32 bytes = two 16-byte loads
  double pen, insX, insY;
  int16_t x, y;
  uint8_t fl;  // 1 reachedMainDiagonal, 2 reachedOtherDiagonal
  uint8_t pad[3];
};

// Two storage modes share one implementation (template parameter LDS):
//  - LDS == false: every search structure lives in the lane's scratch arena in HBM (any size up to the scale's capacities);
//  - LDS == true: the lookup structures (cell hash, bucket table, bucket heap, node lists, both texts) live in the wave's slot of the
//    CU's local data share and only the 32-byte node payloads stay in HBM.  The gapped search is a chain of dependent lookups that
//    only one or two lanes of a wave execute at a time, so its cost is the latency of each lookup: ~100 cycles in LDS, >1000 in HBM.
//    The slot is sized for the piece-wise searches BlockAligner issues (<= 1008 nodes on <= 768 cells, <= 112 distinct priorities,
//    texts <= 128 x 253: the windows of paired reads are up to ~190 bases);
//    a search that outgrows it goes on in HBM mode (PaResume).  One slot per wave: pathAlignAny runs the lanes of a wave through it in turn.
#ifndef XM_PAL_SMALL  // (experiment knob: a smaller slot lets more workgroups share a CU's LDS)
constexpr int XM_PAL_HASH_BITS = 10, XM_PAL_HASH = 1024, XM_PAL_CELLS = 768, XM_PAL_NODES = 1008, XM_PAL_BUCKETS = 112, XM_PAL_BHASH = 256, XM_PAL_TEXTA = 128, XM_PAL_TEXTB = 256;
#else
constexpr int XM_PAL_HASH_BITS = 9, XM_PAL_HASH = 512, XM_PAL_CELLS = 384, XM_PAL_NODES = 448, XM_PAL_BUCKETS = 64, XM_PAL_BHASH = 128, XM_PAL_TEXTA = 64, XM_PAL_TEXTB = 128;
#endif
constexpr int XM_PAL_OFF_HASH = 0;                                      // uint32[1024]: (x << 8 | y) << 16 | node index + 1; at most 768 cells
constexpr int XM_PAL_OFF_XY = XM_PAL_OFF_HASH + XM_PAL_HASH * 4;        // uint16[nodes]: x << 8 | y of node i (= list entry i)
constexpr int XM_PAL_OFF_NEXT = XM_PAL_OFF_XY + XM_PAL_NODES * 2;       // uint16[nodes]: next list entry, 0xFFFF = none
constexpr int XM_PAL_OFF_BKEY = XM_PAL_OFF_NEXT + XM_PAL_NODES * 2;     // double[112]
constexpr int XM_PAL_OFF_BHEAD = XM_PAL_OFF_BKEY + XM_PAL_BUCKETS * 8;  // uint16[112]
constexpr int XM_PAL_OFF_BTAIL = XM_PAL_OFF_BHEAD + XM_PAL_BUCKETS * 2; // uint16[112]
constexpr int XM_PAL_OFF_BHASH = XM_PAL_OFF_BTAIL + XM_PAL_BUCKETS * 2; // uint8[256]: bucket + 1
constexpr int XM_PAL_OFF_TEXTA = XM_PAL_OFF_BHASH + XM_PAL_BHASH;       // uint8[128]  (no heap in LDS mode: the smallest live key is found by a scan)
constexpr int XM_PAL_OFF_TEXTB = XM_PAL_OFF_TEXTA + XM_PAL_TEXTA;       // uint8[256]
constexpr int XM_PAL_SLOT_BYTES = (XM_PAL_OFF_TEXTB + XM_PAL_TEXTB + 63) / 64 * 64;
static_assert(XM_PAL_OFF_BKEY % 8 == 0, "bucket keys must be 8-byte aligned");
static_assert(XM_PAL_SLOT_BYTES * 4 + 432 + 16 <= 40 * 1024, "four waves per workgroup (+ the 432-byte merge-rule table), four workgroups per CU, 160 KB of LDS");

struct PNode;
#if defined(__HIP_DEVICE_COMPILE__)
#ifndef XM_PAL_WAVES
#define XM_PAL_WAVES 4  // waves per workgroup of the kernel this is compiled into (one slot each)
#endif
__shared__ __attribute__((aligned(16))) uint8_t xm_pal_lds[XM_PAL_WAVES * XM_PAL_SLOT_BYTES];
XM_INL uint8_t* palSlot() { return xm_pal_lds + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * XM_PAL_SLOT_BYTES; }
// The node payloads of an LDS-mode search live in a buffer of the WAVE, not of the lane (one search at a time per wave): 4096 waves x 34 KB
// stay in the Infinity Cache, where the lanes' arenas (131 072 x 1.2 MB) never do.  The kernel leaves the buffer's base here.
__shared__ PNode* xm_pal_wave_nodes;
XM_INL void xmSetWaveNodes(PNode* base) { if (threadIdx.x == 0) xm_pal_wave_nodes = base; }  // (before the block's first barrier)
XM_INL PNode* palWaveNodes();
// Pair mode (gapped pass): every read is run by TWO adjacent lanes that execute the same instructions on the same data (a wave of the
// gapped pass holds at most 32 reads, so the other 32 lanes would idle anyway).  The second lane costs nothing while both do the same,
// and wherever a step of the read has two independent halves, each lane takes one and they swap results (PathAligner: the first two
// updates of an explored node).  Atomics and result writes are the first lane's.
// Round 6: the value is a shift - a read is run by 2^shift adjacent lanes (1: the pair above; 3: passes of long reads, whose waves hold 8 reads at most - the
// lanes beyond the first two repeat what the pair does, and the rejection filter's recurrence (xm_bound.h) spreads the cells of a column over all of them).
__shared__ int xm_pair_mode;
XM_INL void xmSetPairMode(int shift) { if (threadIdx.x == 0) xm_pair_mode = shift; }  // (before the block's first barrier)
XM_INL bool xmPairMode() { return __builtin_amdgcn_readfirstlane(xm_pair_mode) != 0; }
XM_INL int xmGroupShift() { return __builtin_amdgcn_readfirstlane(xm_pair_mode); }
// The arrays of an HBM-mode search (nodes, grid or hash, buckets, lists: 480 KB at the gapped pass's scale) are needed by under one search in a
// hundred, so a lane does not own them: every WAVE owns one such buffer (SearchPool: buffer w belongs to wave w of the launch), used by the read
// whose turn it is at the wave's search slot (pathAlign: HBM-mode searches take turns like the LDS-mode ones, so no claim is needed).  A lane's
// temporaries then hold the chain's structures only (matchers, piece lists: ~200 KB), which is what lets a context run all its lanes out of a
// few tens of GiB of scratch.  The buffer is the wave's own because memory written by one CU and reused by a CU of another XCD inside one
// launch is not coherent (each XCD has its own write-back L2).  No pool (batches of long reads, whose searches all run in HBM mode): the
// searches stay in the lanes' temporaries, side by side, as before.
struct SearchPool { uint8_t* base; unsigned long long bufBytes; int32_t n; int32_t pad; };
__shared__ SearchPool xm_search_pool;
XM_INL void xmSetSearchPool(const SearchPool& p) { if (threadIdx.x == 0) xm_search_pool = p; }  // (before the block's first barrier)
XM_INL bool xmHaveSearchPool() { return __builtin_amdgcn_readfirstlane(xm_search_pool.n) > 0; }
// the wave's buffer as an arena (only inside a turn at the wave's slot)
XM_INL bool xmWaveSearchBuffer(Arena& a) {
  const SearchPool p = xm_search_pool;
  const int wave = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  if (wave >= p.n) return false;
  a.init(p.base + (unsigned long long)wave * p.bufBytes, (size_t)p.bufBytes);
  return true;
}
#else
XM_INL uint8_t* palSlot() { static thread_local double slot[XM_PAL_SLOT_BYTES / 8]; return (uint8_t*)slot; }  // host simulation (tests only)
XM_INL void xmSetWaveNodes(PNode*) {}
XM_INL PNode* palWaveNodes();
XM_INL void xmSetPairMode(int) {}
XM_INL bool xmPairMode() { return false; }
XM_INL int xmGroupShift() { return 0; }
struct SearchPool { uint8_t* base; unsigned long long bufBytes; int32_t n; int32_t pad; };
XM_INL void xmSetSearchPool(const SearchPool&) {}
XM_INL bool xmHaveSearchPool() { return false; }
XM_INL bool xmWaveSearchBuffer(Arena&) { return false; }
#endif

#if defined(__HIP_DEVICE_COMPILE__)
XM_INL PNode* palWaveNodes() {
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  return xm_pal_wave_nodes + wave * XM_PAL_NODES;
}
#else
XM_INL PNode* palWaveNodes() { static thread_local double buf[XM_PAL_NODES * 4]; return (PNode*)buf; }
#endif

}  // namespace xm
#include "xm_bound.h"  // the rejection filter in front of the search (uses the wave's slot)
namespace xm {

// XM_PROFILE builds: where a search step spends its time (hash lookups / node loads / arithmetic / putNode), summed into t[12..15]
#if defined(XM_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define XM_PA_TIC(var) unsigned long long var = clock64()
#define XM_PA_TOC(field, var) do { __builtin_amdgcn_s_waitcnt(0); unsigned long long n_ = clock64(); field += n_ - var; var = n_; } while (0)
#else
#define XM_PA_TIC(var) do { } while (0)
#define XM_PA_TOC(field, var) do { } while (0)
#endif

template <bool LDS>
struct PathAlignerT {
  static constexpr double disallowed = 1000000.0;
  unsigned long long tLook, tLoad, tCompute, tPut;
  int32_t lastBucket, lastTail; double lastKey;  // the bucket the previous node went to
  // (the payloads are in HBM in either mode - the wave's node buffer, the lane's temporaries - and the pointer says so: a load through a generic pointer
  // is a flat_load, whose result the compiler must take for different in every lane; everything computed from a node - the keys, the buckets, the
  // counters, the loops over them - then needs execution-mask bookkeeping although one lane runs the search)
  XM_GLOBAL(PNode)* nodes; int32_t nNodes, maxNodes;  // node i is also list entry i (putNode appends exactly one of each)
  XM_INL PNode nodeAt(int i) const { PNode n; __builtin_memcpy(&n, (const void*)(nodes + i), sizeof(PNode)); return n; }
  XM_INL void setNode(int i, const PNode& n) { __builtin_memcpy((void*)(nodes + i), &n, sizeof(PNode)); }
  // locatedNodes: (x,y) -> latest node.  HBM mode: dense grid of node indices when (textA+2)*(textB+2) fits (the four neighbour
  // lookups of an update are then four INDEPENDENT loads, issued together), open-addressing hash otherwise.  LDS mode: hash.
  XM_GLOBAL(int32_t)* grid; int32_t gridH, gridW; bool useGrid;
  XM_GLOBAL(int32_t)* hash; int32_t hashMask;  // (x,y) -> latest node index + 1
  // prioritizedNodes: bucket per exact double key; list entries in insertion order.
  // buckets are never recycled (a removed key cannot reappear: new estimates are clamped to the active key); lookup by exact
  // key bits through an open-addressing hash, priorities.poll() through a binary min-heap of bucket ids
  // (HBM-mode arrays: pointers known to address HBM, like `nodes` - what a turn at the wave's buffer loads is then the same in every active lane for the compiler too)
  XM_GLOBAL(double)* bkey; XM_GLOBAL(int32_t)* bhead; XM_GLOBAL(int32_t)* btail; int32_t nBuckets, maxBuckets;
  XM_GLOBAL(int32_t)* bhash; int32_t bhashMask, bhashCap; XM_GLOBAL(int32_t)* heap; int32_t heapSize;
  XM_GLOBAL(int16_t)* lx; XM_GLOBAL(int16_t)* ly; XM_GLOBAL(int32_t)* lnext;
  // LDS mode
  uint32_t* Lhash; int32_t nCells; uint16_t* Lxy; uint16_t* Lnext; double* Lbkey; uint16_t* Lbhead; uint16_t* Lbtail; uint8_t* Lbhash; uint8_t* Lheap;
  uint8_t* LtextA; uint8_t* LtextB;
  bool ldsOverflow;
  // problem
  Params parameters;
  // register-resident copies (the struct is only used inside pathAlign with every method force-inlined, so it never
  // has to live in scratch memory)
  const uint8_t* qBase; int32_t qLen; bool qRc; const uint8_t* rBase;
  bool confident; double maxInsExt, maxDelExt;
  bool overflow; int32_t predictedBestOffset; unsigned long long nodesPut;
  int32_t startIndexA, endIndexA, startIndexB, endIndexB, textALength, textBLength;
  int32_t startX, startY, goalX, goalY, diagonal, stepDelta;
  double maxInterestingPenalty, activePenalty;
  bool mayQueryExtendPastEndOfReference, searchReverse;

  XM_INL uint8_t charAGlobal(int i) const {
    XM_GLOBAL(const uint8_t)* const g = (XM_GLOBAL(const uint8_t)*)qBase;
    int k = startIndexA + i;
    return qRc ? bpComplement(g[qLen - 1 - k]) : g[k];
  }
  XM_INL uint8_t charBGlobal(int i) const { return ((XM_GLOBAL(const uint8_t)*)rBase)[startIndexB + i]; }
  XM_INL uint8_t charA(int i) const { if constexpr (LDS) return LtextA[i]; else return charAGlobal(i); }
  XM_INL uint8_t charB(int i) const { if constexpr (LDS) return LtextB[i]; else return charBGlobal(i); }
  XM_INL int signedDist(int x, int y) const { return x - y - diagonal; }

  // ---- storage accessors
  XM_INL double bucketKey(int b) const { if constexpr (LDS) return Lbkey[b]; else return bkey[b]; }
  XM_INL int bucketHead(int b) const { if constexpr (LDS) { uint16_t v = Lbhead[b]; return v == 0xFFFF ? -1 : (int)v; } else return bhead[b]; }
  XM_INL int bucketTail(int b) const { if constexpr (LDS) { uint16_t v = Lbtail[b]; return v == 0xFFFF ? -1 : (int)v; } else return btail[b]; }
  XM_INL void setBucketHead(int b, int v) { if constexpr (LDS) Lbhead[b] = (uint16_t)v; else bhead[b] = v; }
  XM_INL void setBucketTail(int b, int v) { if constexpr (LDS) Lbtail[b] = (uint16_t)v; else btail[b] = v; }
  XM_INL int heapAt(int i) const { if constexpr (LDS) return Lheap[i]; else return heap[i]; }
  XM_INL void setHeap(int i, int b) { if constexpr (LDS) Lheap[i] = (uint8_t)b; else heap[i] = b; }
  XM_INL int listNext(int li) const { if constexpr (LDS) { uint16_t v = Lnext[li]; return v == 0xFFFF ? -1 : (int)v; } else return lnext[li]; }
  XM_INL void setListNext(int li, int v) { if constexpr (LDS) Lnext[li] = (uint16_t)v; else lnext[li] = v; }
  XM_INL int listX(int li) const { if constexpr (LDS) return Lxy[li] >> 8; else return lx[li]; }
  XM_INL int listY(int li) const { if constexpr (LDS) return Lxy[li] & 0xFF; else return ly[li]; }
  XM_INL void setListXY(int li, int x, int y) { if constexpr (LDS) Lxy[li] = (uint16_t)((x << 8) | y); else { lx[li] = (int16_t)x; ly[li] = (int16_t)y; } }

  XM_INL int findNodeHash(int x, int y) const {
    uint32_t key = ((uint32_t)x << 16) | (uint32_t)(y & 0xFFFF);
    uint32_t h = (key * 2654435761u) & (uint32_t)hashMask;
    while (true) {
      int32_t v = hash[h];
      if (v == 0) return -1;
      int idx = v - 1;
      if (nodes[idx].x == x && nodes[idx].y == y) return idx;
      h = (h + 1) & (uint32_t)hashMask;
    }
  }
  // LDS cell hash: one 32-bit word per cell, key and node index together, so a lookup is one LDS read unless it collides
  XM_INL static uint32_t ldsCellHash(uint32_t key) { return (key * 2654435761u) >> (32 - XM_PAL_HASH_BITS); }  // top bits
  XM_INL void ldsResolve(uint32_t key, uint32_t& h, uint32_t& v) const {  // from the first probe (h, v) to the cell's slot or the empty slot that ends its run
    while (v != 0 && (v >> 16) != key) { h = (h + 1) & (XM_PAL_HASH - 1); v = Lhash[h]; }
  }
  XM_INL int findNode(int x, int y) const {  // getNode :541-553
    if (x < 0 || y < 0) return -1;
    if constexpr (LDS) {
      if (x >= gridW || y >= gridH) return -1;
      const uint32_t key = ((uint32_t)x << 8) | (uint32_t)y;
      uint32_t h = ldsCellHash(key);
      uint32_t v = Lhash[h];
      ldsResolve(key, h, v);
      return (int)(v & 0xFFFFu) - 1;
    } else {
      if (useGrid) {
        if (x >= gridW || y >= gridH) return -1;
        return grid[x * gridH + y] - 1;
      }
      return findNodeHash(x, y);
    }
  }
  // :523-539 (overwrites the node at (x,y)).  LDS mode: slot >= 0 is the cell's slot as ldsResolve left it (the caller just looked the cell up)
  XM_INL void saveNode(int idx, int x, int y, int slot = -1, bool slotTaken = false) {
    if (x < 0 || y < 0) return;
    if constexpr (LDS) {
      if (x >= gridW || y >= gridH) return;
      const uint32_t key = ((uint32_t)x << 8) | (uint32_t)y;
      uint32_t h, v;
      if (slot >= 0) { h = (uint32_t)slot; v = slotTaken ? 1u : 0u; }  // (the caller's lookup already knows whether the cell exists)
      else { h = ldsCellHash(key); v = Lhash[h]; ldsResolve(key, h, v); }
      if (v == 0) {
        if (nCells >= XM_PAL_CELLS) { ldsOverflow = true; overflow = true; return; }
        nCells++;
      }
      Lhash[h] = (key << 16) | (uint32_t)(idx + 1);
    } else {
      if (useGrid) {
        if (x < gridW && y < gridH) grid[x * gridH + y] = idx + 1;
        return;
      }
      uint32_t key = ((uint32_t)x << 16) | (uint32_t)(y & 0xFFFF);
      uint32_t h = (key * 2654435761u) & (uint32_t)hashMask;
      while (true) {
        int32_t v = hash[h];
        if (v == 0) { hash[h] = idx + 1; return; }
        int o = v - 1;
        if (nodes[o].x == x && nodes[o].y == y) { hash[h] = idx + 1; return; }
        h = (h + 1) & (uint32_t)hashMask;
      }
    }
  }
  XM_INL static uint32_t mixKey(double key) {
    uint64_t kb;
    __builtin_memcpy(&kb, &key, 8);
    return (uint32_t)((kb ^ (kb >> 29)) * 0x9E3779B97F4A7C15ull >> 40);
  }
  XM_INL double estimateOverallPenalty(int x, int y, double pen, double insX, double insY, uint8_t fl) const {  // :475-521
    if (!confident) return pen;
    int sd = signedDist(x, y);
    if (fl & 1) {
      if (sd * stepDelta > 0) {
        double ext = fabs(sd * parameters.InsertionExtension_Penalty);
        if (ext > maxInsExt) return disallowed;
      } else {
        double ext = fabs(sd * parameters.DeletionExtension_Penalty);
        if (ext > maxDelExt) return disallowed;
      }
      if (fl & 2) return pen;
      double indelPenalty = dmin(parameters.InsertionStart_Penalty + parameters.InsertionExtension_Penalty, parameters.DeletionStart_Penalty + parameters.DeletionExtension_Penalty);
      return pen + indelPenalty;
    }
    if (sd * stepDelta < 0) {
      double ext = fabs(sd * parameters.InsertionExtension_Penalty);
      if (ext > maxInsExt) return disallowed;
      double startP = dmin(parameters.InsertionStart_Penalty, insX - pen);
      return pen + startP + ext;
    } else {
      double ext = fabs(sd * parameters.DeletionExtension_Penalty);
      if (ext > maxDelExt) return disallowed;
      double startP = dmin(parameters.DeletionStart_Penalty, insY - pen);
      return pen + startP + ext;
    }
  }
  XM_INL void putNode(int x, int y, double pen, double insX, double insY, uint8_t fl, int cellSlot = -1, bool cellTaken = false) {  // :446-473
    double est = estimateOverallPenalty(x, y, pen, insX, insY, fl);
    if (est < activePenalty) est = activePenalty;
    if (nNodes >= maxNodes) { if constexpr (LDS) ldsOverflow = true; overflow = true; return; }
    if constexpr (LDS) {
      if ((x | y) < 0 || x > 255 || y > 255) { ldsOverflow = true; overflow = true; return; }  // does not pack into a list entry
    }
    int b = -1;
    int tail = -1;
    // most nodes go where the previous one went (estimates are clamped to the active key): the last bucket and its tail are kept in
    // registers, which saves the three dependent lookups below for them
    if (lastBucket >= 0 && est == lastKey) { b = lastBucket; tail = lastTail; }
    else {
    const uint32_t mixed = mixKey(est);
    uint32_t h;
    if constexpr (LDS) {
      h = mixed & (XM_PAL_BHASH - 1);
      while (true) {
        uint32_t v = Lbhash[h];
        if (v == 0) break;
        if (Lbkey[v - 1] == est) { b = (int)v - 1; break; }
        h = (h + 1) & (XM_PAL_BHASH - 1);
      }
    } else {
      h = mixed & (uint32_t)bhashMask;
      while (true) {
        int32_t v = bhash[h];
        if (v == 0) break;
        if (bkey[v - 1] == est) { b = v - 1; break; }
        h = (h + 1) & (uint32_t)bhashMask;
      }
    }
    if (b < 0) {
      if (nBuckets >= maxBuckets) { if constexpr (LDS) ldsOverflow = true; overflow = true; return; }
      b = nBuckets++;
      if constexpr (LDS) { Lbkey[b] = est; Lbhead[b] = 0xFFFF; Lbtail[b] = 0xFFFF; Lbhash[h] = (uint8_t)(b + 1); }
      else {
        bkey[b] = est; bhead[b] = -1; btail[b] = -1; bhash[h] = b + 1;
        // the key table starts small (a search clears it before it starts, and most searches use a few dozen keys) and grows with the keys
        if (nBuckets * 4 > bhashMask + 1 && bhashMask + 1 < bhashCap) {
          const int newSize = imin(bhashCap, (bhashMask + 1) * 4);
          for (int i = 0; i < newSize; i++) bhash[i] = 0;
          bhashMask = newSize - 1;
          for (int k = 0; k < nBuckets; k++) {
            uint32_t hh = mixKey(bkey[k]) & (uint32_t)bhashMask;
            while (bhash[hh] != 0) hh = (hh + 1) & (uint32_t)bhashMask;
            bhash[hh] = k + 1;
          }
        }
      }
      if constexpr (LDS) {
        heapSize++;  // LDS mode keeps no heap (the search loop scans the <= 112 keys): heapSize counts the live buckets
      } else {
        int i = heapSize++;  // sift up
        while (i > 0) {
          int parent = (i - 1) >> 1;
          int pb = heapAt(parent);
          if (bucketKey(pb) <= est) break;
          setHeap(i, pb);
          i = parent;
        }
        setHeap(i, b);
      }
    } else {
      tail = bucketTail(b);
    }
    lastBucket = b; lastKey = est;
    }
    const int idx = nNodes++;  // = list entry
    setListXY(idx, x, y); setListNext(idx, -1);
    if (tail >= 0) setListNext(tail, idx); else setBucketHead(b, idx);
    setBucketTail(b, idx);
    lastTail = idx;
    PNode n;
    n.pen = pen; n.insX = insX; n.insY = insY; n.x = (int16_t)x; n.y = (int16_t)y; n.fl = fl; n.pad[0] = n.pad[1] = n.pad[2] = 0;
    setNode(idx, n);
    saveNode(idx, x, y, cellSlot, cellTaken);
    nodesPut++;
  }
  // What an update decided: the node to put (putNode's arguments), or nothing.  Computing it reads the search structures only, so the
  // first two updates of an explored node, which look at disjoint cells, can be computed side by side (pair mode, pathSearchT).
  struct UpdateOut { int32_t put, x, y, fl, cellSlot, existing; double pen, insX, insY; };
  XM_INL void update(int x, int y) {  // :555-571 + computeUpdated :573-719
    UpdateOut o;
    computeUpdate(x, y, o);
    if (o.put) {
      XM_PA_TIC(t1);
      putNode(o.x, o.y, o.pen, o.insX, o.insY, (uint8_t)o.fl, o.cellSlot, o.existing != 0);
      XM_PA_TOC(tPut, t1);
    }
  }
  XM_INL void computeUpdate(int x, int y, UpdateOut& out) {
    out.put = 0;
    if (x <= 0 || x > textALength) return;
    if (y <= 0 || y > textBLength) return;
    // the four lookups first (independent loads), then the four nodes (index 0 stands in for "null": node 0 always exists)
    XM_PA_TIC(t0);
    int existing, left, up, diag, cellSlot = -1;
    if constexpr (LDS) {
      // all four cells are inside the grid here; the four first probes are independent LDS reads, issued together
      const uint32_t xs = (uint32_t)(x - stepDelta), ys = (uint32_t)(y - stepDelta);
      const uint32_t kE = ((uint32_t)x << 8) | (uint32_t)y, kL = (xs << 8) | (uint32_t)y, kU = ((uint32_t)x << 8) | ys, kD = (xs << 8) | ys;
      uint32_t hE = ldsCellHash(kE), hL = ldsCellHash(kL), hU = ldsCellHash(kU), hD = ldsCellHash(kD);
      uint32_t vE = Lhash[hE], vL = Lhash[hL], vU = Lhash[hU], vD = Lhash[hD];
      ldsResolve(kE, hE, vE); ldsResolve(kL, hL, vL); ldsResolve(kU, hU, vU); ldsResolve(kD, hD, vD);
      existing = (int)(vE & 0xFFFFu) - 1; left = (int)(vL & 0xFFFFu) - 1; up = (int)(vU & 0xFFFFu) - 1; diag = (int)(vD & 0xFFFFu) - 1;
      cellSlot = (int)hE;
    } else {
      existing = findNode(x, y);
      left = findNode(x - stepDelta, y);
      up = findNode(x, y - stepDelta);
      diag = findNode(x - stepDelta, y - stepDelta);
    }
    XM_PA_TOC(tLook, t0);
    const PNode nE = nodeAt(existing >= 0 ? existing : 0);
    const PNode nL = nodeAt(left >= 0 ? left : 0);
    const PNode nU = nodeAt(up >= 0 ? up : 0);
    const PNode nD = nodeAt(diag >= 0 ? diag : 0);
    XM_PA_TOC(tLoad, t0);
    computeFromNodes(x, y, existing, left, up, diag, nE, nL, nU, nD, cellSlot, out);
  }
  // computeUpdated :573-719 proper: what to put at (x, y), given the latest nodes of the cell and of its left, upper and diagonal neighbours
  // (an index < 0: no node there; its payload is then not looked at).  The caller has checked that (x, y) is inside the texts.
  XM_INL void computeFromNodes(int x, int y, int existing, int left, int up, int diag, const PNode& nE, const PNode& nL, const PNode& nU, const PNode& nD, int cellSlot, UpdateOut& out) {
    out.put = 0;
    XM_PA_TIC(t0);
    // the six bases the three transitions can look at (three consecutive ones of either text): read up front, together, instead of
    // one dependent read per branch; indices outside a text are clamped for the read and never used (the range tests below stay)
    const int ia = x - 1, ib = y - 1;
    const uint8_t a0 = charA(ia), b0 = charB(ib);
    const uint8_t aPrev = charA(iclamp(ia - stepDelta, 0, textALength - 1)), aNext = charA(iclamp(ia + stepDelta, 0, textALength - 1));
    const uint8_t bPrev = charB(iclamp(ib - stepDelta, 0, textBLength - 1)), bNext = charB(iclamp(ib + stepDelta, 0, textBLength - 1));
    double insertXPenalty = disallowed, insertYPenalty = disallowed, overlayPenalty = disallowed;
    if (diag >= 0) overlayPenalty = nD.pen + parameters.getPenalty(a0, b0);
    if (left >= 0) {
      if (y == goalY && mayQueryExtendPastEndOfReference) {
        insertXPenalty = nL.pen + parameters.UnalignedPenalty;
      } else {
        bool allowed = true;
        int prevA = x - 1 - stepDelta, prevB = y - 1;
        if (prevA >= 0 && prevA < textALength && prevB >= 0 && prevB < textBLength) {
          if (!bpCanMatch(aPrev, b0)) allowed = false;
        }
        if (allowed) {
          int nextA = x - 1, nextB = y - 1 + stepDelta;
          if (nextA >= 0 && nextA < textALength && nextB >= 0 && nextB < textBLength) {
            uint8_t a = a0, b = bNext;
            if (parameters.getPenalty(a, b) == 0) allowed = false;
            else if (bpIsFullyAmbiguous(a) || bpIsFullyAmbiguous(b)) allowed = false;
          }
        }
        double newInsertX = allowed ? nL.pen + parameters.InsertionStart_Penalty + parameters.InsertionExtension_Penalty : disallowed;
        double extendInsertX = nL.insX + parameters.InsertionExtension_Penalty;
        insertXPenalty = dmin(extendInsertX, newInsertX);
      }
    }
    if (up >= 0) {
      bool allowed = true;
      int prevA = x - 1, prevB = y - 1 - stepDelta;
      if (prevA >= 0 && prevA < textALength && prevB >= 0 && prevB < textBLength) {
        if (!bpCanMatch(a0, bPrev)) allowed = false;
      }
      if (allowed) {
        int nextA = x - 1 + stepDelta, nextB = y - 1;
        if (nextA >= 0 && nextA < textALength && nextB >= 0 && nextB < textBLength) {
          uint8_t a = aNext, b = b0;
          if (parameters.getPenalty(a, b) == 0) allowed = false;
          else if (bpIsFullyAmbiguous(a) || bpIsFullyAmbiguous(b)) allowed = false;
        }
      }
      double newInsertY = allowed ? nU.pen + parameters.DeletionStart_Penalty + parameters.DeletionExtension_Penalty : disallowed;
      double extendInsertY = nU.insY + parameters.DeletionExtension_Penalty;
      insertYPenalty = dmin(extendInsertY, newInsertY);
    }
    double bestPenalty = dmin(dmin(overlayPenalty, insertXPenalty), insertYPenalty);
    if (existing < 0 || bestPenalty < nE.pen || insertXPenalty < nE.insX || insertYPenalty < nE.insY) {
      uint8_t fl = 0;
      if (bestPenalty != disallowed) {
        if (bestPenalty == overlayPenalty) fl = nD.fl;
        else if (bestPenalty == insertXPenalty) fl = nL.fl;
        else fl = nU.fl;
        if (iabs(signedDist(x, y)) == 0) fl |= 1; else fl |= 2;
      }
      XM_PA_TOC(tCompute, t0);
      out.put = 1; out.x = x; out.y = y; out.pen = bestPenalty; out.insX = insertXPenalty; out.insY = insertYPenalty; out.fl = fl; out.cellSlot = cellSlot;
      out.existing = existing >= 0 ? 1 : 0;
    } else {
      XM_PA_TOC(tCompute, t0);
    }
  }
  XM_INL bool chooseSearchReverse() const {  // :17-53
    int sumMis = 0, numMis = 0, sumMatch = 0, numMatch = 0;
    int offset = predictedBestOffset;
    int s = imax(startIndexA, startIndexB - offset);
    int t = imin(endIndexA, endIndexB - offset);
    int length = t - s;
    for (int i = 0; i < length; i++) {
      int j = i - diagonal;
      if (j >= 0 && j < textBLength) {
        if (!bpCanMatch(charA(i), charB(j))) { sumMis += i; numMis++; } else { sumMatch += i; numMatch++; }
      }
    }
    if (numMis > 1 && numMatch > 1) return (sumMis / numMis) > (sumMatch / numMatch);
    return true;
  }
};

XM_INL bool paCanRemoveSection(const ABlock& b) {  // :358-366
  if (b.lenA <= 0 && b.lenB <= 0) return true;
  if ((b.startA <= 0 && b.lenA <= 0) || (b.startB <= 0 && b.lenB <= 0)) return true;
  return false;
}

// LDS-mode searches run one lane at a time (pathAlign's loop over the lanes of the wave), so everything the search computes is the
// same in all its active lanes.  Passing the inputs through readfirstlane tells the compiler so: branches become scalar branches
// (no exec-mask bookkeeping) and the integer bookkeeping moves to the scalar unit.
#if defined(__HIP_DEVICE_COMPILE__)
XM_INL int32_t uniI(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
XM_INL uint64_t uniU64(uint64_t v) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
#else
XM_INL int32_t uniI(int32_t v) { return v; }
XM_INL uint64_t uniU64(uint64_t v) { return v; }
#endif
XM_INL double uniD(double v) { uint64_t b; __builtin_memcpy(&b, &v, 8); b = uniU64(b); __builtin_memcpy(&v, &b, 8); return v; }
template <typename T> XM_INL T* uniP(T* p) { return (T*)uniU64((uint64_t)p); }

// What PathAligner.align is given: the two texts (as views), the sections, the parameters and the analysis scalars
struct PaProblem {
  const uint8_t* qBase; int32_t qLen; bool qRc; const uint8_t* rBase; int32_t referenceLen;
  Section qs, rs;
  Params params;
  bool confident; double maxInsExt, maxDelExt; int32_t predictedBestOffset;
};

// An LDS-mode search that is about to outgrow the wave's slot does not start over in HBM mode: it stops in front of the list entry it
// was going to explore (nothing of that entry done yet) and the HBM-mode search takes the whole state over from the slot and the wave's
// node buffer - nodes, cells, lists, buckets - and continues at that entry.  (The searches that do not fit are only a little larger than
// the slot: ~990 nodes when they stop, ~1060 when they finish.)
struct PaResume { int32_t valid, li, bucket, nNodes, nBuckets; unsigned long long nodesPut; };

// PathAligner.align :55-293 up to and including justify: the search, the traceback and the final block list (outBlocks[0..nbOut)).
// false = null (or *status set).  LDS mode: *ldsOverflow = true means "does not fit the slot, nothing decided".
// UNI: one search at a time per wave (a turn at the wave's slot or buffer), so everything it computes is the same in all its active lanes
template <bool LDS, bool UNI = LDS>
XM_INL bool pathSearchT(const PaProblem& prIn, Arena& tmpIn, const Caps& capsIn, int32_t* status, DevCounters* dc, ABlock* const outBlocksIn, int32_t& nbOut, bool* ldsOverflow, bool pair = false, PaResume* resume = nullptr) {
  XM_TIC(tPath);
  // LDS mode: the arguments too are the same in every active lane, and the compiler must know it - an argument of an out-of-line function counts as
  // different per lane, a branch on one (`pair`, `resume`, `dc`) as divergent, and everything the two arms of such a branch assign - the whole search
  // state - as divergent behind it: the search then runs on vector registers with execution-mask bookkeeping around every loop and branch
  Arena* tmpP = &tmpIn;
  if constexpr (UNI) {
    pair = uniI(pair ? 1 : 0) != 0; resume = uniP(resume); dc = uniP(dc); status = uniP(status); ldsOverflow = uniP(ldsOverflow); tmpP = uniP(tmpP);
  }
  Arena& tmp = *tmpP;
  // by-value copies: anything read through a reference inside a loop that also stores would be re-loaded (and waited for) on
  // every iteration, because the compiler cannot prove the store does not alias it
  PaProblem pr = prIn;
  Caps caps = capsIn;
  ABlock* outBlocks = outBlocksIn;
  if constexpr (UNI) {
    pr.qBase = uniP(pr.qBase); pr.rBase = uniP(pr.rBase); pr.qLen = uniI(pr.qLen); pr.qRc = uniI(pr.qRc ? 1 : 0) != 0; pr.referenceLen = uniI(pr.referenceLen);
    pr.qs.start = uniI(pr.qs.start); pr.qs.end = uniI(pr.qs.end); pr.rs.start = uniI(pr.rs.start); pr.rs.end = uniI(pr.rs.end);
    Params& q = pr.params;
    q.MutationPenalty = uniD(q.MutationPenalty); q.InsertionStart_Penalty = uniD(q.InsertionStart_Penalty); q.InsertionExtension_Penalty = uniD(q.InsertionExtension_Penalty);
    q.DeletionStart_Penalty = uniD(q.DeletionStart_Penalty); q.DeletionExtension_Penalty = uniD(q.DeletionExtension_Penalty); q.MaxErrorRate = uniD(q.MaxErrorRate);
    q.UnalignedPenalty = uniD(q.UnalignedPenalty); q.AmbiguityPenalty = uniD(q.AmbiguityPenalty); q.Max_PenaltySpan = uniD(q.Max_PenaltySpan);
    q.MaxNumMatches = uniI(q.MaxNumMatches); q.StartingInsertionStartFree = uniI(q.StartingInsertionStartFree);
    pr.confident = uniI(pr.confident ? 1 : 0) != 0; pr.maxInsExt = uniD(pr.maxInsExt); pr.maxDelExt = uniD(pr.maxDelExt); pr.predictedBestOffset = uniI(pr.predictedBestOffset);
    caps.maxNodes = uniI(caps.maxNodes); caps.maxBuckets = uniI(caps.maxBuckets); caps.maxBlocks = uniI(caps.maxBlocks);
    caps.gridCap = uniI(caps.gridCap); caps.nodeHash = uniI(caps.nodeHash); caps.bucketHash = uniI(caps.bucketHash);
    outBlocks = uniP(outBlocks);
  }
  const Section qs = pr.qs, rs = pr.rs;
  const Params params = pr.params;
  size_t mark = tmp.used;
  nbOut = 0;
  PathAlignerT<LDS> pa;
  pa.parameters = params;
  pa.qBase = pr.qBase; pa.qLen = pr.qLen; pa.qRc = pr.qRc; pa.rBase = pr.rBase;
  pa.confident = pr.confident; pa.maxInsExt = pr.maxInsExt; pa.maxDelExt = pr.maxDelExt;
  pa.predictedBestOffset = pr.predictedBestOffset; pa.overflow = false; pa.nodesPut = 0; pa.ldsOverflow = false;
  pa.tLook = pa.tLoad = pa.tCompute = pa.tPut = 0;
  pa.lastBucket = -1; pa.lastTail = -1; pa.lastKey = 0;
  const int referenceLen = pr.referenceLen;
  pa.gridW = secLen(qs) + 2; pa.gridH = secLen(rs) + 2;
  pa.grid = nullptr; pa.hash = nullptr; pa.hashMask = 0; pa.useGrid = false;
  pa.bkey = nullptr; pa.bhead = nullptr; pa.btail = nullptr; pa.bhash = nullptr; pa.bhashMask = 0; pa.bhashCap = 0; pa.heap = nullptr; pa.lx = nullptr; pa.ly = nullptr; pa.lnext = nullptr;
  pa.Lhash = nullptr; pa.Lxy = nullptr; pa.Lnext = nullptr; pa.Lbkey = nullptr; pa.Lbhead = nullptr; pa.Lbtail = nullptr; pa.Lbhash = nullptr; pa.Lheap = nullptr;
  pa.LtextA = nullptr; pa.LtextB = nullptr;
  pa.startIndexA = qs.start; pa.endIndexA = qs.end; pa.startIndexB = rs.start; pa.endIndexB = rs.end;
  pa.textALength = secLen(qs); pa.textBLength = secLen(rs);
  if constexpr (LDS) {
    // (a list entry packs x and y into a byte each: y runs to textB + 1)
    if (pa.textALength > XM_PAL_TEXTA || pa.textBLength > XM_PAL_TEXTB - 3 || pa.textALength < 0 || pa.textBLength < 0) { *ldsOverflow = true; return false; }
    pa.maxNodes = imin(caps.maxNodes, XM_PAL_NODES);
    pa.maxBuckets = imin(caps.maxBuckets, XM_PAL_BUCKETS);
    pa.nodes = (XM_GLOBAL(PNode)*)palWaveNodes();
    uint8_t* const slot = palSlot();
    pa.Lhash = (uint32_t*)(slot + XM_PAL_OFF_HASH); pa.nCells = 0; pa.Lxy = (uint16_t*)(slot + XM_PAL_OFF_XY); pa.Lnext = (uint16_t*)(slot + XM_PAL_OFF_NEXT);
    pa.Lbkey = (double*)(slot + XM_PAL_OFF_BKEY); pa.Lbhead = (uint16_t*)(slot + XM_PAL_OFF_BHEAD); pa.Lbtail = (uint16_t*)(slot + XM_PAL_OFF_BTAIL);
    pa.Lbhash = slot + XM_PAL_OFF_BHASH; pa.Lheap = nullptr; pa.LtextA = slot + XM_PAL_OFF_TEXTA; pa.LtextB = slot + XM_PAL_OFF_TEXTB;
    {
      uint64_t* const z1 = (uint64_t*)pa.Lhash;
      for (int i = 0; i < XM_PAL_HASH * 4 / 8; i++) z1[i] = 0;
      uint64_t* const z2 = (uint64_t*)pa.Lbhash;
      for (int i = 0; i < XM_PAL_BHASH / 8; i++) z2[i] = 0;
      for (int i0 = 0; i0 < pa.textALength; i0 += 8) {  // eight loads in flight per round
        uint8_t c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = pa.charAGlobal(imin(i0 + k, pa.textALength - 1));
#pragma unroll
        for (int k = 0; k < 8; k++) if (i0 + k < pa.textALength) pa.LtextA[i0 + k] = c[k];
      }
      for (int i0 = 0; i0 < pa.textBLength; i0 += 8) {
        uint8_t c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = pa.charBGlobal(imin(i0 + k, pa.textBLength - 1));
#pragma unroll
        for (int k = 0; k < 8; k++) if (i0 + k < pa.textBLength) pa.LtextB[i0 + k] = c[k];
      }
    }
  } else {
    pa.maxNodes = caps.maxNodes;
    // (UNI: the arena's base comes out of the caller's frame - private memory, whose loads count as different per lane - so the pointers are made uniform too)
    auto U = [](auto* q) { if constexpr (UNI) return uniP(q); else return q; };
    pa.nodes = (XM_GLOBAL(PNode)*)U(arenaArray<PNode>(tmp, caps.maxNodes));
    pa.useGrid = (long long)pa.gridW * pa.gridH <= (long long)caps.gridCap;
    if (pa.useGrid) pa.grid = (XM_GLOBAL(int32_t)*)U(arenaArray<int32_t>(tmp, (size_t)pa.gridW * pa.gridH));
    else { pa.hash = (XM_GLOBAL(int32_t)*)U(arenaArray<int32_t>(tmp, caps.nodeHash)); pa.hashMask = caps.nodeHash - 1; }
    pa.maxBuckets = caps.maxBuckets;
    pa.bkey = (XM_GLOBAL(double)*)U(arenaArray<double>(tmp, caps.maxBuckets)); pa.bhead = (XM_GLOBAL(int32_t)*)U(arenaArray<int32_t>(tmp, caps.maxBuckets)); pa.btail = (XM_GLOBAL(int32_t)*)U(arenaArray<int32_t>(tmp, caps.maxBuckets));
    pa.bhash = (XM_GLOBAL(int32_t)*)U(arenaArray<int32_t>(tmp, caps.bucketHash)); pa.bhashCap = caps.bucketHash; pa.bhashMask = imin(caps.bucketHash, 2048) - 1;
    pa.heap = (XM_GLOBAL(int32_t)*)U(arenaArray<int32_t>(tmp, caps.maxBuckets));
    pa.lx = (XM_GLOBAL(int16_t)*)U(arenaArray<int16_t>(tmp, caps.maxNodes)); pa.ly = (XM_GLOBAL(int16_t)*)U(arenaArray<int16_t>(tmp, caps.maxNodes)); pa.lnext = (XM_GLOBAL(int32_t)*)U(arenaArray<int32_t>(tmp, caps.maxNodes));
    if (tmp.overflow) { *status = XM_ST_OVERFLOW; tmp.used = mark; return false; }
    XM_GLOBAL(int32_t)* const h1 = pa.useGrid ? pa.grid : pa.hash; const int n1 = pa.useGrid ? pa.gridW * pa.gridH : caps.nodeHash;
    for (int i = 0; i < n1; i++) h1[i] = 0;
    XM_GLOBAL(int32_t)* const h2 = pa.bhash; const int n2 = pa.bhashMask + 1;
    for (int i = 0; i < n2; i++) h2[i] = 0;
  }
  XM_TOC(dc, T_PATH_INIT, tPath);
  pa.heapSize = 0;
  pa.nNodes = 0; pa.nBuckets = 0;
  pa.activePenalty = 0;

  pa.maxInterestingPenalty = secLen(qs) * params.MaxErrorRate;
  if (pa.textALength + 2 > 32000 || pa.textBLength + 2 > 32000) { *status = XM_ST_OVERFLOW; tmp.used = mark; return false; }
  pa.diagonal = pa.startIndexB - (pa.startIndexA + pr.predictedBestOffset);
  pa.stepDelta = 1;
  pa.searchReverse = pa.chooseSearchReverse();
  if (pa.searchReverse) { pa.stepDelta = -1; pa.mayQueryExtendPastEndOfReference = pa.startIndexB == 0; }
  else { pa.stepDelta = 1; pa.mayQueryExtendPastEndOfReference = pa.endIndexB == referenceLen; }
  int width = pa.textALength + 2, height = pa.endIndexB - pa.startIndexB + 2;
  if (pa.searchReverse) { pa.startX = width - 1; pa.startY = height - 1; pa.goalX = 1; pa.goalY = 1; }
  else { pa.startX = 0; pa.startY = 0; pa.goalX = width - 2; pa.goalY = height - 2; }
  const double disallowed = PathAlignerT<LDS>::disallowed;
  int resumeLi = -2;  // (HBM mode taking over an LDS-mode search: the list entry to go on with)
  bool importing = false;
  // (the caller's PaResume lives in its frame: private memory, whose loads count as different per lane - UNI makes what is read of it uniform)
  auto UI = [](int v) { if constexpr (UNI) return uniI(v); else return v; };
  if constexpr (!LDS) importing = resume && UI(resume->valid) != 0;
  if (importing) {
    if constexpr (!LDS) {
      const uint8_t* const slot = palSlot();
      const uint32_t* const Lh = (const uint32_t*)(slot + XM_PAL_OFF_HASH);
      const uint16_t* const Lxy = (const uint16_t*)(slot + XM_PAL_OFF_XY);
      const uint16_t* const Lnx = (const uint16_t*)(slot + XM_PAL_OFF_NEXT);
      const double* const Lbk = (const double*)(slot + XM_PAL_OFF_BKEY);
      const uint16_t* const Lbh = (const uint16_t*)(slot + XM_PAL_OFF_BHEAD);
      const uint16_t* const Lbt = (const uint16_t*)(slot + XM_PAL_OFF_BTAIL);
      const PNode* const waveNodes = palWaveNodes();
      const int nN = UI(resume->nNodes), nB = UI(resume->nBuckets);
      if (nN > pa.maxNodes || nB > pa.maxBuckets) { *status = XM_ST_OVERFLOW; tmp.used = mark; return false; }
      for (int i = 0; i < nN; i++) {
        pa.setNode(i, waveNodes[i]);
        const uint16_t xy = Lxy[i], nx = Lnx[i];
        pa.lx[i] = (int16_t)(xy >> 8); pa.ly[i] = (int16_t)(xy & 0xFF); pa.lnext[i] = nx == 0xFFFF ? -1 : (int32_t)nx;
      }
      pa.nNodes = nN;
      for (int h = 0; h < XM_PAL_HASH; h++) {  // cells: (x, y) -> latest node
        const uint32_t v = Lh[h];
        if (v) pa.saveNode((int)(v & 0xFFFFu) - 1, (int)(v >> 24), (int)((v >> 16) & 0xFFu));
      }
      pa.nBuckets = nB;
      for (int k = 0; k < nB; k++) {
        const double key = Lbk[k];
        const uint16_t hd = Lbh[k], tl = Lbt[k];
        pa.bkey[k] = key; pa.bhead[k] = hd == 0xFFFF ? -1 : (int32_t)hd; pa.btail[k] = tl == 0xFFFF ? -1 : (int32_t)tl;
        if (key != HUGE_VAL) {  // a live bucket: into the key table and the heap (a removed key cannot come back)
          uint32_t hh = pa.mixKey(key) & (uint32_t)pa.bhashMask;
          while (pa.bhash[hh] != 0) hh = (hh + 1) & (uint32_t)pa.bhashMask;
          pa.bhash[hh] = k + 1;
          int i = pa.heapSize++;
          while (i > 0) {
            const int parent = (i - 1) >> 1;
            const int pb = pa.heapAt(parent);
            if (pa.bucketKey(pb) <= key) break;
            pa.setHeap(i, pb);
            i = parent;
          }
          pa.setHeap(i, k);
        }
      }
      pa.nodesPut = UNI ? uniU64(resume->nodesPut) : resume->nodesPut;
      resumeLi = UI(resume->li);
      if (pa.heapSize < 1 || pa.heapAt(0) != UI(resume->bucket)) { *status = XM_ST_INTERNAL; tmp.used = mark; return false; }
    }
  } else
  if (pa.textBLength >= pa.textALength) {
    double startingInsertionStartPenalty = params.getStartingInsertionStartPenalty();
    if (!pa.mayQueryExtendPastEndOfReference) startingInsertionStartPenalty = disallowed;
    int initialDeletionCount = imax(0, pa.textBLength - pa.textALength) + 1;
    for (int i = 0; i < initialDeletionCount && !pa.overflow; i++) pa.putNode(pa.startX, pa.startY + i * pa.stepDelta, 0, startingInsertionStartPenalty, disallowed, 0);
  } else {
    int initialInsertionCount = imax(0, pa.textALength - pa.textBLength) + 1;
    for (int i = 0; i < initialInsertionCount && !pa.overflow; i++) pa.putNode(pa.startX + i * pa.stepDelta, pa.startY, 0, disallowed, disallowed, 0);
  }
  if (!importing && pa.mayQueryExtendPastEndOfReference) {
    int initialInsertionCount = j2i(pr.maxInsExt / params.DeletionExtension_Penalty);
    for (int i = 1; i < initialInsertionCount && !pa.overflow; i++) pa.putNode(pa.startX + i * pa.stepDelta, pa.startY, i * params.UnalignedPenalty, disallowed, disallowed, 0);
  }
  bool haveLast = false;
  int lastX = 0, lastY = 0;
  // every exit below goes through `leave` so that the locally accumulated counters and the overflow flag are published once
  auto leave = [&](bool r) -> bool {
    tmp.used = mark;
    if constexpr (LDS) {
      if (pa.ldsOverflow) {
#ifdef XM_PA_STATS
        fprintf(stderr, "PAFALLBACK nodes %d cells %d buckets %d textA %d textB %d\n", pa.nNodes, pa.nCells, pa.nBuckets, pa.textALength, pa.textBLength);
#endif
        *ldsOverflow = true; return false;  // nothing decided, nothing counted: the HBM-mode search redoes it
      }
    }
    if (pa.overflow) *status = XM_ST_OVERFLOW;
#ifdef XM_PA_STATS
    fprintf(stderr, "PASTAT nodes %d buckets %d gridW %d gridH %d scale %d lds %d\n", pa.nNodes, pa.nBuckets, pa.gridW, pa.gridH, caps.scale, LDS ? 1 : 0);
#endif
    if (dc) { dc->pathAlignerCalls++; dc->pathAlignerNodes += pa.nodesPut; }
#if defined(XM_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if (dc) { dc->t[13] += pa.tLook + pa.tLoad; dc->t[14] += pa.tCompute; dc->t[15] += pa.tPut; }  // (t[12] is T_BOUND since round 6: lookups and loads share a slot)
#endif
    pa.nodesPut = 0;
    return r && !pa.overflow;
  };
  while (!haveLast) {
    if (pa.overflow) return leave(false);
    // priorities.poll(): smallest live key
    if (pa.heapSize < 1) { *status = XM_ST_INTERNAL; return leave(false); }  // Java: NullPointerException
    int b;
    if constexpr (LDS) {
      // the live bucket with the smallest key: a scan of the keys (independent LDS reads, a removed bucket's key is +inf) is shorter
      // than the dependent reads of a heap for this few buckets
      b = 0;
      double best = pa.Lbkey[0];
      for (int k = 1; k < pa.nBuckets; k++) { const double key = pa.Lbkey[k]; if (key < best) { best = key; b = k; } }
    } else {
      b = pa.heapAt(0);
    }
    pa.activePenalty = pa.bucketKey(b);
    int li = resumeLi != -2 ? resumeLi : pa.bucketHead(b);
    resumeLi = -2;
    while (li >= 0) {
      int x = pa.listX(li), y = pa.listY(li);
      if (pa.activePenalty > pa.maxInterestingPenalty + 0.000001) return leave(false);
      if (x == pa.goalX) { haveLast = true; lastX = x; lastY = y; break; }
      if constexpr (LDS) {
        // exploring an entry puts at most three nodes (cells, priorities): if that may not fit the slot any more, the HBM-mode search goes on from here
        if (resume && (pa.nNodes + 3 > pa.maxNodes || pa.nCells + 3 > XM_PAL_CELLS || pa.nBuckets + 3 > pa.maxBuckets)) {
          resume->valid = 1; resume->li = li; resume->bucket = b; resume->nNodes = pa.nNodes; resume->nBuckets = pa.nBuckets; resume->nodesPut = pa.nodesPut;
          tmp.used = mark;
          *ldsOverflow = true;
          return false;
        }
      }
#if defined(__HIP_DEVICE_COMPILE__)
      if (pair) {
        // explore :722-729 with two lanes: (x+d, y) and (x, y+d) read disjoint cells and neither reads what the other puts, so the two
        // lanes of the pair compute one each, swap what they decided, and then both put the two nodes in the reference's order (every
        // lane keeps the whole search state).  (x+d, y+d) looks at both new nodes: computed by both lanes after the puts.
        const bool second = ((int)__lane_id() & 1) != 0;
        typename PathAlignerT<LDS>::UpdateOut mine, other, a, b;
        pa.computeUpdate(second ? x : x + pa.stepDelta, second ? y + pa.stepDelta : y, mine);
        if constexpr (UNI) {
          // one search at a time per wave: its two lanes are 2k and 2k + 1, and what either decided is read straight into scalar registers
          // (v_readlane with the lane in an SGPR: no exchange through the LDS crossbar, no selects, nothing to make uniform afterwards)
          const int laneA = uniI((int)__lane_id() & ~1), laneB = laneA + 1;
          auto rdI = [](int v, int lane) { return __builtin_amdgcn_readlane(v, lane); };
          auto rdD = [](double v, int lane) { uint64_t b; __builtin_memcpy(&b, &v, 8); const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
                                             b = ((uint64_t)hi << 32) | lo; __builtin_memcpy(&v, &b, 8); return v; };
          a.put = rdI(mine.put, laneA); a.x = rdI(mine.x, laneA); a.y = rdI(mine.y, laneA); a.fl = rdI(mine.fl, laneA); a.cellSlot = rdI(mine.cellSlot, laneA); a.existing = rdI(mine.existing, laneA);
          a.pen = rdD(mine.pen, laneA); a.insX = rdD(mine.insX, laneA); a.insY = rdD(mine.insY, laneA);
          b.put = rdI(mine.put, laneB); b.x = rdI(mine.x, laneB); b.y = rdI(mine.y, laneB); b.fl = rdI(mine.fl, laneB); b.cellSlot = rdI(mine.cellSlot, laneB); b.existing = rdI(mine.existing, laneB);
          b.pen = rdD(mine.pen, laneB); b.insX = rdD(mine.insX, laneB); b.insY = rdD(mine.insY, laneB);
        } else {
          // (searches side by side: several pairs of the wave may be searching at once, every pair with its own values)
          other.put = __shfl_xor(mine.put, 1); other.x = __shfl_xor(mine.x, 1); other.y = __shfl_xor(mine.y, 1); other.fl = __shfl_xor(mine.fl, 1);
          other.cellSlot = __shfl_xor(mine.cellSlot, 1); other.existing = __shfl_xor(mine.existing, 1);
          other.pen = __shfl_xor(mine.pen, 1); other.insX = __shfl_xor(mine.insX, 1); other.insY = __shfl_xor(mine.insY, 1);
          a = second ? other : mine; b = second ? mine : other;
        }
        if (a.put) pa.putNode(a.x, a.y, a.pen, a.insX, a.insY, (uint8_t)a.fl, a.cellSlot, a.existing != 0);
        // the first put may have taken the empty slot the second one's lookup ended at: look the cell up again then
        if (b.put) pa.putNode(b.x, b.y, b.pen, b.insX, b.insY, (uint8_t)b.fl, (a.put && !a.existing && !b.existing) ? -1 : b.cellSlot, b.existing != 0);
        pa.update(x + pa.stepDelta, y + pa.stepDelta);
      } else
#endif
#pragma unroll 1  // one copy of update() in the loop: the code of this search has to stay in the instruction cache
      for (int mv = 0; mv < 3; mv++) {  // explore :722-729: (x+d, y), (x, y+d), (x+d, y+d)
        int ux = (mv == 1) ? x : x + pa.stepDelta;
        int uy = (mv == 0) ? y : y + pa.stepDelta;
        pa.update(ux, uy);
      }
      if (pa.overflow) return leave(false);
      li = pa.listNext(li);
    }
    // prioritizedNodes.remove(activePenalty) + the poll(): pop the heap root (the active bucket is still the minimum: every key
    // inserted meanwhile is >= activePenalty and distinct keys are distinct buckets)
    if constexpr (LDS) {
      pa.Lbkey[b] = HUGE_VAL;
      pa.heapSize--;
      if (pa.lastBucket == b) pa.lastBucket = -1;
    } else {
      const int last = pa.heapAt(--pa.heapSize);
      const double lastKey = pa.bucketKey(last);
      int i = 0;
      while (true) {
        int l = 2 * i + 1, r = l + 1;
        if (l >= pa.heapSize) break;
        int bl = pa.heapAt(l);
        double kl = pa.bucketKey(bl);
        int c = l, bc = bl;
        double kc = kl;
        if (r < pa.heapSize) {
          int br = pa.heapAt(r);
          double kr = pa.bucketKey(br);
          if (kr < kl) { c = r; bc = br; kc = kr; }
        }
        if (kc >= lastKey) break;
        pa.setHeap(i, bc);
        i = c;
      }
      if (pa.heapSize > 0) pa.setHeap(i, last);
    }
  }
  // traceback :195-264
  int i = lastX, j = lastY;
  int nb = 0;
  const int sd = pa.stepDelta;
  const int sA = pa.startIndexA, sB = pa.startIndexB;
  while (i != pa.startX && j != pa.startY) {
    if (nb >= caps.maxBlocks) { pa.overflow = true; return leave(false); }
    int node = pa.findNode(i, j);
    double bestPenalty = pa.nodes[node].pen, insertXPenalty = pa.nodes[node].insX, insertYPenalty = pa.nodes[node].insY;
    ABlock blk;
    if (bestPenalty == insertXPenalty) {
      int oldI = i;
      i -= sd;
      while (i != pa.startX) {
        int other = pa.findNode(i, j);
        double otherNew = pa.nodes[other].pen + params.InsertionStart_Penalty + params.InsertionExtension_Penalty;
        double otherExtend = pa.nodes[other].insX + params.InsertionExtension_Penalty;
        if (otherNew < otherExtend) break;
        i -= sd;
      }
      if (pa.searchReverse) blk = ABlock{sA + oldI - 1, sB + j - 1, i - oldI, 0};
      else blk = ABlock{sA + i, sB + j, oldI - i, 0};
    } else if (bestPenalty == insertYPenalty) {
      int oldJ = j;
      j -= sd;
      while (j != pa.startY) {
        int other = pa.findNode(i, j);
        double otherNew = pa.nodes[other].pen + params.DeletionStart_Penalty + params.DeletionExtension_Penalty;
        double otherExtend = pa.nodes[other].insY + params.DeletionExtension_Penalty;
        if (otherNew < otherExtend) break;
        j -= sd;
      }
      if (pa.searchReverse) blk = ABlock{sA + i - 1, sB + oldJ - 1, 0, j - oldJ};
      else blk = ABlock{sA + i, sB + j, 0, oldJ - j};
    } else {
      int oldI = i, oldJ = j;
      i -= sd;
      j -= sd;
      while (i != pa.startX && j != pa.startY) {
        int other = pa.findNode(i, j);
        if (pa.nodes[other].pen == pa.nodes[other].insX || pa.nodes[other].pen == pa.nodes[other].insY) break;
        i -= sd;
        j -= sd;
      }
      if (pa.searchReverse) blk = ABlock{sA + oldI - 1, sB + oldJ - 1, i - oldI, j - oldJ};
      else blk = ABlock{sA + i, sB + j, oldI - i, oldJ - j};
    }
    outBlocks[nb++] = blk;
  }
  leave(true);  // the search structures are dead from here on
  if (!pa.searchReverse) for (int a = 0, b2 = nb - 1; a < b2; a++, b2--) { ABlock t = outBlocks[a]; outBlocks[a] = outBlocks[b2]; outBlocks[b2] = t; }
  if (nb < 1) return false;
  // justify :307-352
  ABlock* s = outBlocks;
  SeqView jq, jr;
  jq.base = pr.qBase; jq.len = pr.qLen; jq.rc = pr.qRc ? 1 : 0; jq.id = 0;
  jr.base = pr.rBase; jr.len = pr.referenceLen; jr.rc = 0; jr.id = 0;
  for (int k = 1; k < nb - 1; k++) {
    while (true) {
      ABlock left = s[k - 1], middle = s[k], right = s[k + 1];
      if ((middle.lenA > 0) == (middle.lenB > 0)) break;
      if (left.lenA == 0 || left.lenB == 0) break;
      if (right.lenA == 0 || right.lenB == 0) break;
      if (middle.lenA > 0) { if (jq.at(abEndA(left) - 1) != jq.at(abEndA(middle) - 1)) break; }
      else { if (jr.at(abEndB(left) - 1) != jr.at(abEndB(middle) - 1)) break; }
      s[k - 1] = ABlock{left.startA, left.startB, left.lenA - 1, left.lenB - 1};
      s[k] = ABlock{middle.startA - 1, middle.startB - 1, middle.lenA, middle.lenB};
      s[k + 1] = ABlock{right.startA - 1, right.startB - 1, right.lenA + 1, right.lenB + 1};
    }
  }
  int drop = 0;
  while (drop < nb && paCanRemoveSection(s[drop])) drop++;
  if (drop >= nb) { *status = XM_ST_INTERNAL; return false; }  // Java: IndexOutOfBoundsException
  if (drop > 0) { for (int k = drop; k < nb; k++) s[k - drop] = s[k]; nb -= drop; }
  nbOut = nb;
  return true;
}

XM_NOINL bool pathSearchHbm(const PaProblem& pr, Arena& tmp, const Caps& caps, int32_t* status, DevCounters* dc, ABlock* outBlocks, int32_t& nb, bool pair = false, PaResume* resume = nullptr) {
  return pathSearchT<false, false>(pr, tmp, caps, status, dc, outBlocks, nb, nullptr, pair, resume);
}
// the HBM-mode search of the read whose turn it is at the wave's slot: its arrays in the wave's buffer when the launch has a pool, else in the
// lane's temporaries
// The wave's buffer is sized for XM_POOL_NODE_FACTOR times the node capacities of the pass's scale (searchPoolCaps; the host sizes it with the same
// function): the searches that need it are few, but on a reference with repeats some of them put eight thousand nodes and more, and a read whose search
// outgrows the pass's capacities is run again from its start, alone on a wave, in a pass that lasts as long as the slowest such read (profiles/r04/NOTES.md).
constexpr int XM_POOL_NODE_FACTOR = 4;
XM_INL Caps searchPoolCaps(const Caps& caps) {
  Caps c = caps;
  c.maxNodes = caps.maxNodes * XM_POOL_NODE_FACTOR; c.nodeHash = caps.nodeHash * XM_POOL_NODE_FACTOR;
  c.maxBuckets = caps.maxBuckets * 2; c.bucketHash = caps.bucketHash * 2;
  return c;
}
XM_INL size_t searchPoolBytes(const Caps& passCaps) {
  const Caps c = searchPoolCaps(passCaps);
  const size_t cells = (size_t)(c.gridCap > c.nodeHash ? c.gridCap : c.nodeHash);
  return ((size_t)c.maxNodes * 32 + cells * 4 + (size_t)c.maxBuckets * 20 + (size_t)c.bucketHash * 4 + (size_t)c.maxNodes * 8 + (size_t)c.maxBlocks * 16 * 4 + 4096 + 4095) & ~(size_t)4095;
}
// (a turn at the wave's buffer: one search at a time per wave, like a turn at the slot - the uniform instantiation)
XM_NOINL bool pathSearchHbmTurn(const PaProblem& pr, Arena& tmp, const Caps& caps, int32_t* status, DevCounters* dc, ABlock* outBlocks, int32_t& nb, bool pair, PaResume* resume) {
  return pathSearchT<false, true>(pr, tmp, caps, status, dc, outBlocks, nb, nullptr, pair, resume);
}
XM_INL bool pathSearchHbmInTurn(const PaProblem& pr, Arena& tmp, const Caps& caps, int32_t* status, DevCounters* dc, ABlock* outBlocks, int32_t& nb, bool pair, PaResume* resume = nullptr) {
  Arena wb;
  if (xmWaveSearchBuffer(wb)) return pathSearchHbmTurn(pr, wb, searchPoolCaps(caps), status, dc, outBlocks, nb, pair, resume);
  return pathSearchHbm(pr, tmp, caps, status, dc, outBlocks, nb, pair, resume);
}
XM_NOINL bool pathSearchLds(const PaProblem& pr, Arena& tmp, const Caps& caps, int32_t* status, DevCounters* dc, ABlock* outBlocks, int32_t& nb, bool* ldsOverflow, bool pair, PaResume* resume = nullptr) {
  return pathSearchT<true>(pr, tmp, caps, status, dc, outBlocks, nb, ldsOverflow, pair, resume);
}

// One turn at the wave's slot: the LDS-mode search, and when it stops in front of an entry it has no room for, the HBM-mode search that
// takes its state over (while the slot and the wave's node buffer still hold it).  ldsOverflow stays set for the searches that could not
// even start in the slot (texts too long, too many start nodes): those are done in HBM mode from the beginning, after the turns.
XM_INL bool pathSearchSlot(const PaProblem& pr, Arena& tmp, const Caps& caps, int32_t* status, DevCounters* dc, ABlock* outBlocks, int32_t& nb, bool* ldsOverflow, bool pair) {
  PaResume rs;
  rs.valid = 0; rs.li = -1; rs.bucket = -1; rs.nNodes = 0; rs.nBuckets = 0; rs.nodesPut = 0;
  bool found = pathSearchLds(pr, tmp, caps, status, dc, outBlocks, nb, ldsOverflow, pair, &rs);
  if (*ldsOverflow && rs.valid) {
    *ldsOverflow = false;
    found = pathSearchHbmInTurn(pr, tmp, caps, status, dc, outBlocks, nb, pair, &rs);
  }
  return found;
}

// The search in the form of xm_wsearch.h (lane-private tables built for few dependent trips to memory: four per explored entry against about
// thirteen of PathAlignerT<false>), run from start to end in the lane's temporaries: what a chain whose searches start in HBM mode (scale 16 and up: long
// reads, and the reruns of reads that outgrew scale 4 - a handful of reads whose searches put tens of thousands of nodes and whose pass lasts as long
// as the slowest of them) uses instead of pathSearchHbm.  XM_WSEARCH_FROM (a compile-time define): the chain scale from which it does.
XM_NOINL_DECL bool pathSearchW(const PaProblem& pr, Arena& tmp, const Caps& caps, int32_t* status, DevCounters* dc, ABlock* outBlocks, int32_t& nb);
#ifndef XM_WSEARCH_FROM
#define XM_WSEARCH_FROM 16
#endif
#if defined(__HIPCC__) && defined(XM_PROFILE)
// profile builds: how many reads of a wave stand at a PathAligner call together ([0] arrivals, [1] reads in them, [2] arrivals of four reads or more, [3] reads in those)
__device__ unsigned long long xm_arrive_prof[16];
// (diagnostic) pair mode: the two lanes of a read must hold the same value wherever they stand together; [4 + k]: times they did not at check point k
#define XM_PAIR_CHECK(k, v) do { if (xmPairMode()) { const long long v_ = (long long)(v); const int lo_ = __shfl_xor((int)v_, 1), hi_ = __shfl_xor((int)(v_ >> 32), 1); \
  if (lo_ != (int)v_ || hi_ != (int)(v_ >> 32)) atomicAdd(&xm_arrive_prof[4 + (k)], 1ull); } } while (0)
#endif
#if !(defined(__HIP_DEVICE_COMPILE__) && defined(XM_PROFILE))
#undef XM_PAIR_CHECK
#define XM_PAIR_CHECK(k, v) do { } while (0)
#endif
XM_INL int wideSearchFrom() { return XM_WSEARCH_FROM; }

// PathAligner.align: LDS-mode search first (the lanes of the wave that arrive here together take the wave's slot one after the other),
// the searches that do not fit are then redone in HBM mode.
XM_INL bool pathAlign(const ExtEnv& e, const Section& qs, const Section& rs, const Params& p, Analysis& an, SeqAl& out) {
  int32_t nb = 0;
  bool found = false;
  {
    PaProblem pr;
    pr.qBase = e.query.base; pr.qLen = e.query.len; pr.qRc = e.query.rc != 0; pr.rBase = e.reference.base; pr.referenceLen = e.reference.len;
    pr.qs = qs; pr.rs = rs; pr.params = p;
    pr.confident = an.confidentAboutBestOffset; pr.maxInsExt = an.maxInsertionExtensionPenalty; pr.maxDelExt = an.maxDeletionExtensionPenalty;
    pr.predictedBestOffset = an.predictedBestOffset;
    // chains of long reads (scale 16 and up: mates over 320 bases, and the reruns of reads that outgrew scale 4): their searches outgrow the slot
    // (hundreds to thousands of entries), so the turn at it only adds the take-over; they start in HBM mode, side by side where lanes arrive together
    // (1 kb queries with 3 % substitutions, half of them with an indel: 3 738 -> 3 294 ms per 200 k; profiles/r03/NOTES.md 13).  XM_HBM_ONLY_FROM: experiment define
#ifndef XM_HBM_ONLY_FROM
#define XM_HBM_ONLY_FROM 16
#endif
    XM_PAIR_CHECK(2, ((long long)qs.start << 32) ^ (long long)rs.start ^ ((long long)qs.end << 16) ^ ((long long)rs.end << 48));
    if (xmBoundFilter()) {
      // the rejection filter (xm_bound.h): a search it proves null is not run (PathAligner_Runner.align was still called: the call counts)
      BoundProblem bp;
      bp.qBase = pr.qBase; bp.qLen = pr.qLen; bp.qRc = pr.qRc; bp.rBase = pr.rBase; bp.referenceLen = pr.referenceLen;
      bp.startA = qs.start; bp.endA = qs.end; bp.startB = rs.start; bp.endB = rs.end; bp.predictedBestOffset = pr.predictedBestOffset;
      bp.mutation = p.MutationPenalty; bp.insStart = p.InsertionStart_Penalty; bp.insExt = p.InsertionExtension_Penalty; bp.delStart = p.DeletionStart_Penalty;
      bp.delExt = p.DeletionExtension_Penalty; bp.maxErrorRate = p.MaxErrorRate; bp.ambiguity = p.AmbiguityPenalty;
      bp.budget = secLen(qs) * p.MaxErrorRate; bp.piece = 0;  // :60
      bool taken = false;
      unsigned long long cells = 0;
      XM_TIC(tBound);
      const bool rejected = boundRejects(bp, xmGroupShift(), *e.tmp, taken, cells);
      XM_TOC(e.dc, T_BOUND, tBound);
      if (e.dc && taken) { e.dc->boundChecks++; e.dc->boundCells += cells; }
      if (rejected) {
        if (e.dc) { e.dc->boundRejects++; e.dc->pathAlignerCalls++; }
        return false;
      }
    }
    bool ldsOverflow = e.caps->searchInHbmOnly != 0 || e.caps->scale >= XM_HBM_ONLY_FROM;  // (searchInHbmOnly: the test entry; 2 = in the form of xm_wsearch.h)
    if (!ldsOverflow) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(XM_WAVE_UNIFORM)
    // wave-per-read kernels (xm_wave_kernel.hip): every lane of the wave is on the same search with the same values, so the wave's slot
    // is simply used (all lanes write the same words)
    found = pathSearchSlot(pr, *e.tmp, *e.caps, e.status, e.dc, out.blocks, nb, &ldsOverflow, false);
#elif defined(__HIP_DEVICE_COMPILE__)
    unsigned long long pending = __ballot(1);
    const int lane = (int)__lane_id();
#ifdef XM_PROFILE
    if (lane == __ffsll((long long)pending) - 1) {
      const unsigned long long reads = (unsigned long long)((__popcll(pending) + (1 << xmGroupShift()) - 1) >> xmGroupShift());
      atomicAdd(&xm_arrive_prof[0], 1ull); atomicAdd(&xm_arrive_prof[1], reads);
      if (reads >= 4) { atomicAdd(&xm_arrive_prof[2], 1ull); atomicAdd(&xm_arrive_prof[3], reads); }
    }
#endif
    if (xmPairMode()) {  // the lanes of a read take the slot together
      const int gm = (1 << xmGroupShift()) - 1;
      while (pending) {
        const int leader = (__ffsll((long long)pending) - 1) & ~gm;
        if ((lane & ~gm) == leader) found = pathSearchSlot(pr, *e.tmp, *e.caps, e.status, e.dc, out.blocks, nb, &ldsOverflow, true);
        pending &= ~(((2ull << gm) - 1ull) << leader);
      }
    } else {
      while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        if (lane == leader) found = pathSearchSlot(pr, *e.tmp, *e.caps, e.status, e.dc, out.blocks, nb, &ldsOverflow, false);
        pending &= pending - 1;
      }
    }
#else
    found = pathSearchSlot(pr, *e.tmp, *e.caps, e.status, e.dc, out.blocks, nb, &ldsOverflow, false);
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(XM_WAVE_UNIFORM)
    if (xmHaveSearchPool()) {
      // the searches that could not start in the slot (texts too long for it): one after the other through the wave's buffer, like the turns above
      unsigned long long waiting = __ballot(ldsOverflow ? 1 : 0);
      const int lane2 = (int)__lane_id();
      const bool pm = xmPairMode();
      const int gm2 = (1 << xmGroupShift()) - 1;
      while (waiting) {
        const int leader = (__ffsll((long long)waiting) - 1) & ~gm2;
        if (ldsOverflow && (lane2 & ~gm2) == leader) found = pathSearchHbmInTurn(pr, *e.tmp, *e.caps, e.status, e.dc, out.blocks, nb, pm);
        waiting &= ~(((2ull << gm2) - 1ull) << leader);
      }
    } else
#endif
    if (ldsOverflow) {
      if ((e.caps->scale >= wideSearchFrom() && !e.caps->searchInHbmOnly) || e.caps->searchInHbmOnly == 2) found = pathSearchW(pr, *e.tmp, *e.caps, e.status, e.dc, out.blocks, nb);
      else found = pathSearchHbm(pr, *e.tmp, *e.caps, e.status, e.dc, out.blocks, nb, xmPairMode());
    }
  }
  XM_PAIR_CHECK(3, ((long long)(found ? 1 : 0) << 40) ^ ((long long)*e.status << 20) ^ (long long)nb);
  if (!found || *e.status) return false;
  out.nb = nb;
  finishSeqAl(e, p, out, e.query.rc != 0);
  if (out.alignedPenalty > secLen(qs) * p.MaxErrorRate) return false;
  return true;
}

// ---------------------------------------------------------------- StraightAligner (M/StraightAligner.java)
XM_INL void straightAlignment(const ExtEnv& e, const Section& qs, const Section& rs, const Params& p, const Analysis& an, SeqAl& out) {  // :73-94
  int queryStartIndex = qs.start, queryEndIndex = qs.end, referenceStartIndex = rs.start, referenceEndIndex = rs.end;
  int off = an.predictedBestOffset;
  if (queryStartIndex + off > referenceStartIndex) referenceStartIndex = queryStartIndex + off; else queryStartIndex = referenceStartIndex - off;
  if (queryEndIndex + off < referenceEndIndex) referenceEndIndex = queryEndIndex + off; else queryEndIndex = referenceEndIndex - off;
  out.nb = 1;
  out.blocks[0] = ABlock{queryStartIndex, referenceStartIndex, queryEndIndex - queryStartIndex, referenceEndIndex - referenceStartIndex};
  finishSeqAl(e, p, out, e.query.rc != 0);
}

typedef bool (*NextAligner)(const ExtEnv&, const Section&, const Section&, const Params&, Analysis&, SeqAl&);

// :13-71; `next` is the rest of the chain
template <typename Next>
XM_INL bool straightAlign(const ExtEnv& e, const Section& qs, const Section& rs, const Params& p, Analysis& an, SeqAl& out, Next next) {
  an.lastCheckedOffset = an.predictedBestOffset;
  ABlock simpleBlock[1];
  SeqAl simple;
  simple.blocks = simpleBlock;
  XM_TIC(tS);
  straightAlignment(e, qs, rs, p, an, simple);
  XM_TOC(e.dc, T_STRAIGHT, tS);
  double simpleTotal = simple.alignedPenalty;
  double maxInterestingPenalty = secLen(qs) * p.MaxErrorRate;
  double indelPenalty = dmin(p.getStartingInsertionStartPenalty() + p.InsertionExtension_Penalty, p.DeletionStart_Penalty + p.DeletionExtension_Penalty);
  if (simpleTotal <= 0) { saCopy(out, simple); return true; }
  if (an.confidentAboutBestOffset) {
    if (simpleTotal <= indelPenalty || (an.maxInsertionExtensionPenalty <= 0 && an.maxDeletionExtensionPenalty <= 0)) {
      if (simpleTotal <= maxInterestingPenalty) { saCopy(out, simple); return true; }
      return false;
    }
    if (indelPenalty > maxInterestingPenalty) return false;
  }
  double rate = simple.alignedPenalty / secLen(qs);
  Params sub = p;
  sub.MaxErrorRate = dmin(rate, p.MaxErrorRate);
  bool have = next(e, qs, rs, sub, an, out);
  if (*e.status) return false;
  if (!have || out.alignedPenalty >= simpleTotal) {
    if (simpleTotal <= maxInterestingPenalty) { saCopy(out, simple); return true; }
  }
  return have;
}

// ---------------------------------------------------------------- HashBlock_Aligner (M/HashBlock_Aligner.java)
struct PenaltyAnalysis { double minPossiblePenalty, maxInsertionExtensionPenalty, maxDeletionExtensionPenalty; int32_t offsetWithMostHashblockMatches, numHashBlockMatchesWithBestOffset; };

XM_INL double hbaMinIndelPenaltyForBlockMismatches(int numMismatches, const Params& p) {  // :286-310
  numMismatches = imax(1, numMismatches);
  double minPenaltyPerInitialIndel = dmin(p.getStartingInsertionStartPenalty() + p.InsertionExtension_Penalty, p.DeletionStart_Penalty + p.DeletionExtension_Penalty);
  double minPenaltyPerExtension = dmin(p.InsertionExtension_Penalty, p.DeletionExtension_Penalty);
  double minPenaltyPerSubsequentIndel = dmin(p.InsertionStart_Penalty + p.InsertionExtension_Penalty, p.DeletionStart_Penalty + p.DeletionExtension_Penalty);
  double minPenaltyPerSubsequentChange = dmin(p.MutationPenalty, minPenaltyPerSubsequentIndel);
  if (numMismatches <= 1) return minPenaltyPerInitialIndel;
  if (numMismatches <= 2) return minPenaltyPerInitialIndel + minPenaltyPerExtension;
  return minPenaltyPerInitialIndel + minPenaltyPerExtension + (numMismatches - 2) * minPenaltyPerSubsequentChange;
}
XM_INL double hbaLongInsertion(int numMismatches, double totalPenalty, const Params& p, int blockLength) {  // :322-354
  double availablePenalty = totalPenalty - p.getStartingInsertionStartPenalty();
  double penaltyOfOnlySNPs = numMismatches * p.MutationPenalty;
  double penaltyPerBlockExtension = blockLength * p.InsertionExtension_Penalty;
  double extraPenaltyPerBlockExtension = penaltyPerBlockExtension - p.MutationPenalty;
  if (extraPenaltyPerBlockExtension <= 0) return availablePenalty;
  if (numMismatches < 2) return availablePenalty;
  double penaltyOfShortExtension = 2 * p.InsertionExtension_Penalty;
  if (penaltyOfShortExtension > availablePenalty) return availablePenalty;
  double penaltyOfShortSNPs = 2 * p.MutationPenalty;
  double maxAllowedPenaltyIncreasePastAllSNPs = availablePenalty - penaltyOfOnlySNPs;
  double maxAllowedPenaltyForBlockExtensions = maxAllowedPenaltyIncreasePastAllSNPs + penaltyOfShortSNPs - penaltyOfShortExtension;
  double maxNumBlockExtensions = maxAllowedPenaltyForBlockExtensions / extraPenaltyPerBlockExtension;
  double r = (maxNumBlockExtensions * blockLength + 2) * p.InsertionExtension_Penalty;
  r = dmin(r, availablePenalty);
  if (r < penaltyOfShortExtension) r = 0;
  return r;
}
XM_INL double hbaManyInsertions(int numMismatches, double totalPenalty, const Params& p) {  // :356-376
  double availablePenalty = totalPenalty + (p.InsertionStart_Penalty - p.getStartingInsertionStartPenalty());
  double penaltyOfOnlySNPs = numMismatches * p.MutationPenalty;
  double penaltyPerShortIndel = p.InsertionStart_Penalty + 2 * p.InsertionExtension_Penalty;
  double extra = penaltyPerShortIndel - 2 * p.MutationPenalty;
  if (extra <= 0) return availablePenalty;
  double maxNumShortIndels = (availablePenalty - penaltyOfOnlySNPs) / extra;
  if (maxNumShortIndels < 1) maxNumShortIndels = 0;
  double r = maxNumShortIndels * 2 * p.InsertionExtension_Penalty;
  return dmin(r, availablePenalty);
}
XM_INL double hbaManyDeletions(int numMismatches, double totalPenalty, const Params& p) {  // :378-400
  double availablePenalty = totalPenalty;
  double penaltyOfOnlySNPs = numMismatches * p.MutationPenalty;
  double penaltyPerShortIndel = p.DeletionStart_Penalty + 2 * p.DeletionExtension_Penalty;
  double extra = penaltyPerShortIndel - 2 * p.MutationPenalty;
  if (extra <= 0) return availablePenalty;
  double maxNumShortIndels = (availablePenalty - penaltyOfOnlySNPs) / extra;
  if (maxNumShortIndels < 1) maxNumShortIndels = 0;
  double r = maxNumShortIndels * 2 * p.DeletionExtension_Penalty;
  r = dmin(r, availablePenalty);
  if (r < 0) r = 0;
  return r;
}

// analyzePenalty :94-283.  storeSlot: where a matcher created for a matcher-less analysis lives (A for the outer
// HashBlock_Aligner, B for the inner one); temporaries go to slot T.
XM_NOINL PenaltyAnalysis hbaAnalyzePenalty(const ExtEnv& e, const Section& qsIn, const Section& rsIn, const Params& pIn, Analysis& an, Matcher* storeSlot) {
  XM_TIC(tA);
  const Section qs = qsIn, rs = rsIn;  // by value (see pathAlign)
  const Params p = pIn;
  Arena& tmp = *e.tmp;
  size_t mark = tmp.used;
  PenaltyAnalysis result;
  result.minPossiblePenalty = 0; result.maxInsertionExtensionPenalty = 0; result.maxDeletionExtensionPenalty = 0;
  result.offsetWithMostHashblockMatches = 0; result.numHashBlockMatchesWithBestOffset = 0;
  Matcher* matcher = an.matcher;
  const SeqView query = e.query, reference = e.reference;  // register-resident copies
  DevCounters* const dc = e.dc;
  bool overflow = false;
  EncodeCache cache; cache.index = -2; cache.code = -1;
  double maxInterestingPenalty = p.MaxErrorRate * secLen(qs);
  int numMismatches = 0;
  int maxNonmatchingBlockEnd = qs.start;
  CountMap counts;
  counts.mostPopularKey = 0; counts.mostPopularCount = 0; counts.n = 0; counts.cap = e.caps->maxCountMap; counts.haveCounts = false;
  counts.keys = arenaArray<int32_t>(tmp, counts.cap); counts.vals = arenaArray<int32_t>(tmp, counts.cap); counts.status = e.status;
  if (tmp.overflow) { *e.status = XM_ST_OVERFLOW; tmp.used = mark; return result; }
  int numLateBlocksSupportingInsertion = 0, numLateBlocksSupportingDeletion = 0;
  int minPossibleOffset = rs.start - qs.start;
  int maxPossibleOffset = rs.end - qs.end;
  int lookupUncertainty = maxPossibleOffset - minPossibleOffset;
  if (!matcher || iabs(matcher->sectionLength - lookupUncertainty) > lookupUncertainty / 2) {
    Matcher* slot = an.matcher ? e.slotT : storeSlot;
    matcherInit(*slot, e, rs, lookupUncertainty);
    if (slot->referenceLength > 32000) { *e.status = XM_ST_OVERFLOW; tmp.used = mark; return result; }
    matcher = slot;
    if (!an.matcher) an.matcher = matcher;
  }
  Matcher mm = *matcher;  // by value: scalars stay in registers, tables/present are pointers into the arena
  int blockLength = mm.blockLength;
  int maxBlockStart = qs.end - blockLength;
  for (int blockStartIndex = qs.start; blockStartIndex <= maxBlockStart; blockStartIndex++) {
    if (blockStartIndex >= maxNonmatchingBlockEnd) {
      int position = matcherLookup(mm, query, reference, blockStartIndex, blockStartIndex + minPossibleOffset, blockStartIndex + maxPossibleOffset + 1, cache, overflow, dc);
      if (overflow) break;
      int offset = position - blockStartIndex;
      if (position == M_UNKNOWN || position == M_MULTIPLE) continue;
      if (position == M_NO_MATCHES) {
        numMismatches++;
        maxNonmatchingBlockEnd = blockStartIndex + blockLength;
        if (hbaMinIndelPenaltyForBlockMismatches(numMismatches, p) > maxInterestingPenalty) break;
        continue;
      }
      int otherStartIndex = position;
      int reverseCount = imin(blockStartIndex - maxNonmatchingBlockEnd, otherStartIndex);
      bool foundMismatch = false;
      // the two base-by-base extensions of the reference (:167-200), eight bases per pair of loads (seqMatchRun, xm_defs.h)
      if (reverseCount > 0 && seqMatchRunBack(query, blockStartIndex - 1, reference, otherStartIndex - 1, reverseCount) < reverseCount) {
        numMismatches++;
        foundMismatch = true;
        maxNonmatchingBlockEnd = blockStartIndex + blockLength;
      }
      if (!foundMismatch) {
        int forwardShift = qs.end - blockStartIndex;
        const int limit = forwardShift - blockLength;  // positions i = blockLength .. forwardShift - 1
        if (limit > 0) {
          // (a reference position at or past rs.end counts as a base that matches nothing)
          const int inside = imax(0, imin(limit, rs.end - (otherStartIndex + blockLength)));
          const int run = seqMatchRun(query, blockStartIndex + blockLength, reference, otherStartIndex + blockLength, inside);
          if (run < limit) {
            numMismatches++;
            foundMismatch = true;
            maxNonmatchingBlockEnd = blockStartIndex + blockLength + run + 1;
          }
        }
        if (!foundMismatch) maxNonmatchingBlockEnd = qs.end;
        int numOther = 0;
        int forwardShift2 = maxNonmatchingBlockEnd - blockStartIndex - blockLength;
        for (int i = blockLength; i < forwardShift2; i++) {
          int indexA = blockStartIndex + i;
          int lookupResult = matcherLookup(mm, query, reference, indexA, indexA + minPossibleOffset, indexA + maxPossibleOffset + 1, cache, overflow, dc);
          if (overflow) break;
          int offset2 = lookupResult - indexA;
          if (lookupResult >= 0 && offset2 == offset) {
            numOther++;
            i = i - 1 + blockLength;
          }
        }
        if (offset != counts.mostPopularKey && counts.mostPopularCount > 0) {
          if (offset > counts.mostPopularKey) numLateBlocksSupportingDeletion += numOther;
          else numLateBlocksSupportingInsertion += numOther;
        }
        counts.add(offset, numOther);
      }
      if (foundMismatch) {
        if (hbaMinIndelPenaltyForBlockMismatches(numMismatches, p) > maxInterestingPenalty) break;
      } else {
        counts.add(offset, 1);
      }
      if (overflow) break;
    }
  }
  matcher->nSections = mm.nSections; matcher->presentMask = mm.presentMask;  // write back the scalars that change
  if (overflow) *e.status = XM_ST_OVERFLOW;
  if (*e.status) { tmp.used = mark; return result; }
  int mostPopularOffset = counts.mostPopularKey;
  int mostPopularOffset_count = counts.mostPopularCount;
  tmp.used = mark;
  XM_TOC(e.dc, T_ANALYZE, tA);
  double indelPenalty = hbaMinIndelPenaltyForBlockMismatches(numMismatches, p);
  result.minPossiblePenalty = indelPenalty;
  bool couldDiffer = mostPopularOffset_count < 1 || an.lastCheckedOffset != mostPopularOffset;
  if (couldDiffer) {
    double mismatchPenalty = numMismatches * p.MutationPenalty;
    if (result.minPossiblePenalty > mismatchPenalty) result.minPossiblePenalty = mismatchPenalty;
  }
  // setMaxExtensionPenalty :313-319
  double longInsertion = hbaLongInsertion(numMismatches + numLateBlocksSupportingDeletion, maxInterestingPenalty, p, blockLength);
  double manyInsertions = hbaManyInsertions(numMismatches + numLateBlocksSupportingInsertion, maxInterestingPenalty, p);
  result.maxInsertionExtensionPenalty = dmax(longInsertion, manyInsertions);
  result.maxDeletionExtensionPenalty = hbaManyDeletions(numMismatches + numLateBlocksSupportingInsertion, maxInterestingPenalty, p);
  if (result.maxInsertionExtensionPenalty > an.maxInsertionExtensionPenalty) result.maxInsertionExtensionPenalty = an.maxInsertionExtensionPenalty;
  if (result.maxDeletionExtensionPenalty > an.maxDeletionExtensionPenalty) result.maxDeletionExtensionPenalty = an.maxDeletionExtensionPenalty;
  if (mostPopularOffset_count < 1) mostPopularOffset = an.predictedBestOffset;
  result.offsetWithMostHashblockMatches = mostPopularOffset;
  result.numHashBlockMatchesWithBestOffset = mostPopularOffset_count;
  return result;
}

// HashBlock_Aligner.align :21-81 with the tail self-call as a loop
template <typename Next>
XM_INL bool hashBlockAlign(const ExtEnv& e, const Section& qs, Section rs, const Params& p, Analysis an, SeqAl& out, Matcher* storeSlot, Next next) {
  while (true) {
    double maxInterestingPenalty = p.MaxErrorRate * secLen(qs);
    if (secLen(qs) > secLen(rs)) return next(e, qs, rs, p, an, out);
    PenaltyAnalysis pa = hbaAnalyzePenalty(e, qs, rs, p, an, storeSlot);
    if (*e.status) return false;
    if (pa.minPossiblePenalty > maxInterestingPenalty) return false;
    Analysis sub = an;  // child()
    sub.maxInsertionExtensionPenalty = pa.maxInsertionExtensionPenalty;
    sub.maxDeletionExtensionPenalty = pa.maxDeletionExtensionPenalty;
    double extra = pa.numHashBlockMatchesWithBestOffset * p.MutationPenalty + pa.minPossiblePenalty;
    if (extra > maxInterestingPenalty) {
      sub.predictedBestOffset = pa.offsetWithMostHashblockMatches;
      sub.confidentAboutBestOffset = true;
    } else if (!an.confidentAboutBestOffset) {
      sub.predictedBestOffset = pa.offsetWithMostHashblockMatches;
    }
    if (an.confidentAboutBestOffset && sub.predictedBestOffset == an.predictedBestOffset) sub.confidentAboutBestOffset = true;
    Section sec = rs;
    if (sub.confidentAboutBestOffset) {
      int maxDeletionLength = j2i((double)pa.maxDeletionExtensionPenalty / (double)p.DeletionExtension_Penalty);
      int maxInsertionLength = j2i((double)pa.maxInsertionExtensionPenalty / (double)p.InsertionExtension_Penalty);
      int maxIndelLength = imax(maxDeletionLength, maxInsertionLength);
      sec.start = imax(rs.start, qs.start + sub.predictedBestOffset - maxIndelLength);
      sec.end = imin(rs.end, qs.end + sub.predictedBestOffset + maxIndelLength);
    }
    if (secLen(sec) < secLen(rs)) { rs = sec; an = sub; continue; }
    return next(e, qs, sec, p, sub, out);
  }
}

// ---------------------------------------------------------------- the inner chain: straight3 -> pathAligner, hashBlock<2>, straight2
struct NextPath {
  XM_INL bool operator()(const ExtEnv& e, const Section& qs, const Section& rs, const Params& p, Analysis& an, SeqAl& out) const {
    XM_TIC(t0);
    bool r = pathAlign(e, qs, rs, p, an, out);
    XM_TOC(e.dc, T_PATH, t0);
    return r;
  }
};
struct NextStraight3 {
  XM_INL bool operator()(const ExtEnv& e, const Section& qs, const Section& rs, const Params& p, Analysis& an, SeqAl& out) const { return straightAlign(e, qs, rs, p, an, out, NextPath()); }
};
struct NextHashBlock2 {
  XM_INL bool operator()(const ExtEnv& e, const Section& qs, const Section& rs, const Params& p, Analysis& an, SeqAl& out) const { return hashBlockAlign(e, qs, rs, p, an, out, e.slotB, NextStraight3()); }
};
XM_NOINL bool innerChain(const ExtEnv& e, const Section& qs, const Section& rs, const Params& p, Analysis& an, SeqAl& out) {  // StraightAligner #2 of :18-29
  return straightAlign(e, qs, rs, p, an, out, NextHashBlock2());
}

// ---------------------------------------------------------------- BlockAligner (M/BlockAligner.java)
XM_NOINL bool baAlignPiece(const ExtEnv& e, const Section& qs, const Section& rs, double maxPenalty, const Params& p, bool firstPiece, const Analysis& parent, SeqAl& out) {  // :215-249
  if (maxPenalty < 0) return false;
  Section sub = rs;
  if (parent.confidentAboutBestOffset) {
    int maxInsertionLength = j2i((double)parent.maxInsertionExtensionPenalty / (double)p.InsertionExtension_Penalty);
    int maxDeletionLength = j2i((double)parent.maxDeletionExtensionPenalty / (double)p.DeletionExtension_Penalty);
    int maxIndelLength = imax(maxInsertionLength, maxDeletionLength);
    int referenceStart = imax(rs.start, qs.start + parent.predictedBestOffset - maxIndelLength);
    int referenceEnd = imin(rs.end, qs.end + parent.predictedBestOffset + maxIndelLength);
    if (referenceEnd > referenceStart) { sub.start = referenceStart; sub.end = referenceEnd; }
  }
  if (xmBoundFilter() && boundPieceApplies(qs.start, qs.end, sub.start, sub.end, e.reference.len, parent.predictedBestOffset, parent.matcher != nullptr, parent.matcher ? parent.matcher->sectionLength : 0)) {
    // the rejection filter over the piece's whole chain (xm_bound.h, boundPieceApplies): a piece it proves unalignable within maxPenalty is not sent down the chain -
    // no hash-block analysis, no straight alignments, no search; alignPiece returns null as it would have
    BoundProblem bp;
    bp.qBase = e.query.base; bp.qLen = e.query.len; bp.qRc = e.query.rc != 0; bp.rBase = e.reference.base; bp.referenceLen = e.reference.len;
    bp.startA = qs.start; bp.endA = qs.end; bp.startB = sub.start; bp.endB = sub.end; bp.predictedBestOffset = parent.predictedBestOffset;
    bp.mutation = p.MutationPenalty; bp.insStart = p.InsertionStart_Penalty; bp.insExt = p.InsertionExtension_Penalty; bp.delStart = p.DeletionStart_Penalty;
    bp.delExt = p.DeletionExtension_Penalty; bp.maxErrorRate = p.MaxErrorRate; bp.ambiguity = p.AmbiguityPenalty;
    bp.budget = maxPenalty; bp.piece = 1;
    bool taken = false;
    unsigned long long cells = 0;
    XM_TIC(tBound);
    const bool rejected = boundRejects(bp, xmGroupShift(), *e.tmp, taken, cells);
    XM_TOC(e.dc, T_BOUND, tBound);
    if (e.dc && taken) { e.dc->boundPieceChecks++; e.dc->boundCells += cells; }
    if (rejected) {
      if (e.dc) e.dc->boundPieceRejects++;
      return false;
    }
  }
  Params sp = p;
  if (!firstPiece) sp.StartingInsertionStartFree = 1;
  sp.MaxErrorRate = maxPenalty / secLen(qs);
  Analysis child = parent;
  child.confidentAboutBestOffset = false;
  return innerChain(e, qs, sub, sp, child, out);
}

XM_INL bool baTryMerge(const ExtEnv& e, const SeqAl& left, const SeqAl& right, const Params& p, SeqAl& out) {  // :158-212
  if (saEndB(left) != saStartB(right)) return false;
  const ABlock& l = left.blocks[left.nb - 1];
  const ABlock& r = right.blocks[0];
  if (abIndelType(l) != abIndelType(r)) return false;
  if (abEndA(l) != r.startA) return false;
  if (abEndB(l) != r.startB) return false;
  if (left.nb - 1 + 1 + right.nb - 1 > e.caps->maxBlocks) { *e.status = XM_ST_OVERFLOW; return false; }
  int n = 0;
  for (int i = 0; i < left.nb - 1; i++) out.blocks[n++] = left.blocks[i];
  out.blocks[n++] = ABlock{l.startA, l.startB, l.lenA + r.lenA, l.lenB + r.lenB};
  for (int i = 1; i < right.nb; i++) out.blocks[n++] = right.blocks[i];
  out.nb = n;
  finishSeqAl(e, p, out, left.referenceReversed != 0);
  return true;
}

XM_NOINL bool blockAlign(const ExtEnv& e, const Section& qsIn, const Section& rsIn, const Params& pIn, Analysis& an, SeqAl& out) {  // :17-36
  const Section qs = qsIn, rs = rsIn;
  const Params p = pIn;
  Arena& tmp = *e.tmp;
  size_t mark = tmp.used;
  const Caps caps = *e.caps;
  double maxInterestingPenalty = p.MaxErrorRate * secLen(qs);
  // initialAlignments :39-96
  double maxInterestingPenaltyWholeQuery = p.MaxErrorRate * e.query.len;  // (sic) uses query.getLength()
  int numBasesToEncodeReferencePosition = e.baLogStep ? baNumBasesToEncodeReferencePosition(e.baLogStep, secLen(rs))  // (sic) :48, the host's steps
                                                      : 3;  // (only the component test entry runs the chain without an index; it never reaches BlockAligner)
  int numHashblocks = secLen(qs) / numBasesToEncodeReferencePosition + 1;
  int targetNumHashblocksPerBlock = j2i(sqrt((double)numHashblocks)) + 1;
  int targetBlockSize = targetNumHashblocksPerBlock * numBasesToEncodeReferencePosition;
  int numBlocks = secLen(qs) / targetBlockSize;
  if (numBlocks > caps.maxPieces) { *e.status = XM_ST_OVERFLOW; return false; }
  if (numBlocks < 1) return false;  // "no initial alignments"
  // two piece lists (ping-pong across joinAlignments rounds); blocks are packed into one pool per list
  const int poolCap = 4 * caps.maxBlocks + 4 * caps.maxPieces;
  SeqAl* listA = arenaArray<SeqAl>(tmp, caps.maxPieces);
  SeqAl* listB = arenaArray<SeqAl>(tmp, caps.maxPieces);
  ABlock* poolA = arenaArray<ABlock>(tmp, (size_t)poolCap);
  ABlock* poolB = arenaArray<ABlock>(tmp, (size_t)poolCap);
  ABlock* scratchBlocks = arenaArray<ABlock>(tmp, (size_t)caps.maxBlocks);
  uint8_t* have = arenaArray<uint8_t>(tmp, caps.maxPieces);
  if (tmp.overflow) { *e.status = XM_ST_OVERFLOW; tmp.used = mark; return false; }
  for (int i = 0; i < caps.maxPieces; i++) have[i] = 0;
  SeqAl scratch;
  scratch.blocks = scratchBlocks;
  int usedA = 0, usedB = 0;
  // commit `src` into list entry `dst`, taking its blocks from `pool`
  auto commit = [&](SeqAl& dst, const SeqAl& src, ABlock* pool, int& used) -> bool {
    if (used + src.nb > poolCap) { *e.status = XM_ST_OVERFLOW; return false; }
    dst.blocks = pool + used;
    used += src.nb;
    saCopy(dst, src);
    return true;
  };
  double usedPenalty = 0;
  int numRemainingAlignments = numBlocks;
  while (true) {
    bool failedSubalignment = false, failedThenFound = false;
    int startPosition = qs.start;
    for (int i = 0; i < numBlocks; i++) {
      int endPosition = qs.start + (secLen(qs) * (i + 1) / numBlocks);
      if (!have[i]) {
        Section sub{startPosition, endPosition};
        double averagePenalty = (maxInterestingPenaltyWholeQuery - usedPenalty) / numRemainingAlignments;
        bool ok = baAlignPiece(e, sub, rs, averagePenalty, p, i == 0, an, scratch);
        if (*e.status) { tmp.used = mark; return false; }
        if (ok) {
          if (!commit(listA[i], scratch, poolA, usedA)) { tmp.used = mark; return false; }
          if (failedSubalignment) failedThenFound = true;
          numRemainingAlignments--;
          have[i] = 1;
          usedPenalty += listA[i].alignedPenalty;
        } else {
          failedSubalignment = true;
        }
      }
      startPosition = endPosition;
    }
    if (numRemainingAlignments < 1) break;
    if (!failedThenFound) { tmp.used = mark; return false; }
  }
  // joinAlignments rounds :99-144
  SeqAl* cur = listA;
  SeqAl* nxt = listB;
  ABlock* nxtPool = poolB;
  ABlock* curPool = poolA;
  int n = numBlocks;
  bool even = false;
  while (n > 1) {
    int rn = 0;
    int nxtUsed = 0;
    double used = 0;
    for (int i = 0; i < n; i++) used += cur[i].alignedPenalty;
    for (int i = 0; i < n; i += 2) {
      if (i + 1 < n) {
        bool merged = baTryMerge(e, cur[i], cur[i + 1], p, scratch);
        if (*e.status) { tmp.used = mark; return false; }
        if (!merged) {
          used -= cur[i].alignedPenalty;
          used -= cur[i + 1].alignedPenalty;
          Section sub{saStartA(cur[i]), saEndA(cur[i + 1])};
          bool ok = baAlignPiece(e, sub, rs, maxInterestingPenalty - used, p, i == 0, an, scratch);
          if (*e.status) { tmp.used = mark; return false; }
          if (!ok) { tmp.used = mark; return false; }
          if (!commit(nxt[rn], scratch, nxtPool, nxtUsed)) { tmp.used = mark; return false; }
          used += nxt[rn].alignedPenalty;
          rn++;
        } else {
          if (!even) {  // !allowSimpleMerges: keep `left`, retry from its right neighbour
            if (!commit(nxt[rn], cur[i], nxtPool, nxtUsed)) { tmp.used = mark; return false; }
            rn++;
            i--;
            continue;
          }
          if (!commit(nxt[rn], scratch, nxtPool, nxtUsed)) { tmp.used = mark; return false; }
          rn++;
        }
      } else {
        if (!commit(nxt[rn], cur[i], nxtPool, nxtUsed)) { tmp.used = mark; return false; }
        rn++;
      }
    }
    SeqAl* t = cur; cur = nxt; nxt = t;
    ABlock* tp = curPool; curPool = nxtPool; nxtPool = tp;
    n = rn;
    even = !even;
  }
  (void)usedB; (void)curPool;
  saCopy(out, cur[0]);
  tmp.used = mark;
  return true;
}

// ---------------------------------------------------------------- the outer chain (M/QueryMatch_Aligner.java:18-29)
// XM_LIGHT_ONLY: a translation unit that only ever runs the light pass (its kernel never enters the gapped chain, so the chain's code and
// registers are not compiled into it)
#ifdef XM_LIGHT_ONLY
#define XM_LIGHT_ONLY_FLAG true
#else
#define XM_LIGHT_ONLY_FLAG false
#endif
struct NextBlock {
  XM_INL bool operator()(const ExtEnv& e, const Section& qs, const Section& rs, const Params& p, Analysis& an, SeqAl& out) const {
    if (XM_LIGHT_ONLY_FLAG || e.caps->heavyAllowed < 2) { *e.status = XM_ST_NEED_HEAVY; return false; }
    XM_TIC(t0);
    bool r = blockAlign(e, qs, rs, p, an, out);
    XM_TOC(e.dc, T_BLOCK, t0);
    return r;
  }
};
struct NextHashBlock1 {
  XM_INL bool operator()(const ExtEnv& e, const Section& qs, const Section& rs, const Params& p, Analysis& an, SeqAl& out) const {
    if (XM_LIGHT_ONLY_FLAG || e.caps->heavyAllowed < 1) {
      // cost hint for the gapped pass: the penalty of the straight alignment that was not good enough (p.MaxErrorRate is that
      // alignment's error rate here, StraightAligner :59-61).  A read with an indel mismatches on one whole side of it.
      if (e.heavyHint) *e.heavyHint = (float)(p.MaxErrorRate * secLen(qs));
      *e.status = XM_ST_NEED_HEAVY;
      return false;
    }
    // SkipHighAmbiguity_Aligner :13-28
    const int numAmbiguities = seqCountAmbiguous(e.reference, rs.start, rs.end);
    if (numAmbiguities >= secLen(rs) / 4) return false;
    return hashBlockAlign(e, qs, rs, p, an, out, e.slotA, NextBlock());
  }
};
XM_NOINL bool outerChain(const ExtEnv& e, const Section& qs, const Section& rs, const Params& p, Analysis& an, SeqAl& out) {
  XM_TIC(t0);
  bool r = straightAlign(e, qs, rs, p, an, out, NextHashBlock1());
  XM_TOC(e.dc, T_OUTER, t0);
  return r;
}

}  // namespace xm

#include "xm_wsearch.h"
