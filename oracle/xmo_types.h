// ORACLE — test infrastructure only.  CPU restatement of the X-Mapper (mathjeff/Mapper) per-read
// seed-and-extend path.  Nothing under oracle/ is linked into, imported by, or executed from the
// product path (mapper_amd/, libxmapper_hip.so); only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py use it.
//
// This header restates the value types the hot path needs.  About half of them live in the
// un-vendored submodule deps/QuickVariants (empty in the reference snapshot), so
// their behaviour is *inferred* from call sites and from the reference's JUnit tests; every inferred
// contract is marked [inferred] and listed in DESIGN.md ("parity unpinned" items).
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <algorithm>
#include <stdexcept>
#include <limits>

namespace xmo {

// ---------------------------------------------------------------- Java arithmetic helpers
static inline int32_t jwrap(int64_t v) { return (int32_t)(uint32_t)(uint64_t)v; }
static inline int32_t jadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t jmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
// (int)double in Java: truncation toward zero, saturating, NaN -> 0
static inline int32_t j2i(double d) {
  if (d != d) return 0;
  if (d >= 2147483647.0) return INT32_MAX;
  if (d <= -2147483648.0) return INT32_MIN;
  return (int32_t)d;
}
static inline int64_t j2l(double d) {
  if (d != d) return 0;
  if (d >= 9223372036854775807.0) return INT64_MAX;
  if (d <= -9223372036854775808.0) return INT64_MIN;
  return (int64_t)d;
}
static inline int32_t jabs(int32_t v) { return v == INT32_MIN ? v : (v < 0 ? -v : v); }
static inline double jnextUp(double d) { return std::nextafter(d, std::numeric_limits<double>::infinity()); }

// ---------------------------------------------------------------- Basepairs  [inferred: QuickVariants]
// bitmask A=1 C=2 G=4 T=8 (M/HashBlock_Matcher.java:184-196), union = OR, canMatch = (a&b)!=0
// (consistent with encodedCharB=0 never matching, M/HashBlock_Aligner.java:185-186).
struct Basepairs {
  static uint8_t encode(char c) {
    switch (c) {
      case 'A': case 'a': return 1;  case 'C': case 'c': return 2;
      case 'G': case 'g': return 4;  case 'T': case 't': case 'U': case 'u': return 8;
      case 'R': case 'r': return 1|4;   case 'Y': case 'y': return 2|8;
      case 'S': case 's': return 2|4;   case 'W': case 'w': return 1|8;
      case 'K': case 'k': return 4|8;   case 'M': case 'm': return 1|2;
      case 'B': case 'b': return 2|4|8; case 'D': case 'd': return 1|4|8;
      case 'H': case 'h': return 1|2|8; case 'V': case 'v': return 1|2|4;
      default: return 15;  // N and anything else
    }
  }
  static char decode(uint8_t b) {
    static const char* table = "?ACMGRSVTWYHKDBN";
    return table[b & 15];
  }
  static uint8_t complement(uint8_t b) {
    return (uint8_t)(((b & 1) << 3) | ((b & 2) << 1) | ((b & 4) >> 1) | ((b & 8) >> 3));
  }
  static bool canMatch(uint8_t a, uint8_t b) { return (a & b) != 0; }
  static uint8_t unionOf(uint8_t a, uint8_t b) { return a | b; }
  static int popcount(uint8_t b) { return __builtin_popcount(b & 15); }
  static bool isAmbiguous(uint8_t b) { return popcount(b) != 1; }
  static bool isFullyAmbiguous(uint8_t b) { return (b & 15) == 15; }
  // T/BasepairsTest.java:26-38: N -> 1, 2-way -> 1/3 (AmbiguityPenalty * rate must equal AmbiguityPenalty / 3)
  static double getMutationFalseNegativeRate(uint8_t b) {
    int n = popcount(b);
    if (n <= 1) return 0.0;
    return (double)(n - 1) / 3.0;
  }
};

// ---------------------------------------------------------------- Sequence  [inferred: QuickVariants]
struct Sequence {
  std::string name;
  std::vector<uint8_t> codes;             // one Basepairs code per base
  const Sequence* complementedFrom = nullptr;  // non-null <=> this is a reverse complement
  int contigIndex = -1;                   // index of the forward contig in the SequenceDatabase (-1: a query)
  // T/RepeatingSequence.java:9-12: a Sequence of one repeated base, so that the reference's tests can "allocate" contigs of 2^30 .. 2^31 bases
  // (T/SequenceDatabase_Test.java:117, T/PackedMap_Test.java:59); repeatedLength >= 0 selects it
  int repeatedLength = -1;
  uint8_t repeatedCode = 1;
  int getLength() const { return repeatedLength >= 0 ? repeatedLength : (int)codes.size(); }
  uint8_t encodedCharAt(int i) const { return repeatedLength >= 0 ? repeatedCode : codes[(size_t)i]; }
  char charAt(int i) const { return Basepairs::decode(codes[(size_t)i]); }
  std::string getRange(int start, int len) const {
    std::string s((size_t)len, '?');
    for (int i = 0; i < len; i++) s[(size_t)i] = charAt(start + i);
    return s;
  }
  std::string getText() const { return getRange(0, getLength()); }
  const Sequence* getComplementedFrom() const { return complementedFrom; }
};

static inline std::unique_ptr<Sequence> makeSequence(const std::string& name, const std::string& text) {
  std::unique_ptr<Sequence> s(new Sequence());
  s->name = name;
  s->codes.resize(text.size());
  for (size_t i = 0; i < text.size(); i++) s->codes[i] = Basepairs::encode(text[i]);
  return s;
}
static inline std::unique_ptr<Sequence> makeReverseComplement(const Sequence& f) {
  std::unique_ptr<Sequence> s(new Sequence());
  s->name = f.name;
  int n = f.getLength();
  if (f.repeatedLength >= 0) { s->repeatedLength = f.repeatedLength; s->repeatedCode = Basepairs::complement(f.repeatedCode); }
  else {
    s->codes.resize((size_t)n);
    for (int i = 0; i < n; i++) s->codes[(size_t)i] = Basepairs::complement(f.codes[(size_t)(n - 1 - i)]);
  }
  s->complementedFrom = &f;
  s->contigIndex = f.contigIndex;
  return s;
}

struct SequencePosition {
  const Sequence* sequence;
  int startIndex;
  const Sequence* getSequence() const { return sequence; }
  int getStartIndex() const { return startIndex; }
};

// ---------------------------------------------------------------- SequenceDatabase  [inferred: QuickVariants]
// Holds fwd,rev interleaved (M/Mapper.java:1164-1165).  Position codec: cumulative start of the sequence
// (in database order fwd0,rev0,fwd1,rev1,...) + start index; numBitsPerPosition = log2RoundUp(total).
struct SequenceDatabase {
  std::vector<std::unique_ptr<Sequence>> owned;
  std::vector<const Sequence*> all;      // fwd0, rev0, fwd1, rev1, ...
  std::vector<int64_t> cumulativeStart;  // per entry of `all`
  int64_t totalForwardSize = 0;

  void addForward(std::unique_ptr<Sequence> f) {
    f->contigIndex = (int)all.size() / 2;
    std::unique_ptr<Sequence> r = makeReverseComplement(*f);
    cumulativeStart.push_back(totalForwardSize * 2);
    cumulativeStart.push_back(totalForwardSize * 2 + f->getLength());
    totalForwardSize += f->getLength();
    all.push_back(f.get());
    all.push_back(r.get());
    owned.push_back(std::move(f));
    owned.push_back(std::move(r));
  }
  int numContigs() const { return (int)all.size() / 2; }
  const Sequence* forward(int contig) const { return all[(size_t)contig * 2]; }
  const Sequence* reverse(int contig) const { return all[(size_t)contig * 2 + 1]; }
  int64_t getTotalForwardSize() const { return totalForwardSize; }
  int64_t getTotalForwardAndReverseSize() const { return totalForwardSize * 2; }
  const Sequence* getReverseComplement(const Sequence* s) const {
    return s->complementedFrom ? s->complementedFrom : reverse(s->contigIndex);
  }
  static int log2RoundUp(int64_t v) {
    int bits = 0;
    int64_t p = 1;
    while (p < v) { p <<= 1; bits++; }
    return bits;
  }
  int getNumBitsPerPosition() const { return log2RoundUp(getTotalForwardAndReverseSize()); }
  int64_t encodePosition(const Sequence* s, int startIndex) const {
    size_t idx = (size_t)s->contigIndex * 2 + (s->complementedFrom ? 1 : 0);
    return cumulativeStart[idx] + startIndex;
  }
  SequencePosition decodePosition(int64_t encoded) const {
    size_t idx = (size_t)(std::upper_bound(cumulativeStart.begin(), cumulativeStart.end(), encoded) - cumulativeStart.begin()) - 1;
    return SequencePosition{all[idx], (int)(encoded - cumulativeStart[idx])};
  }
};

// ---------------------------------------------------------------- AlignedBlock / SequenceAlignment / QueryAlignment [inferred]
struct AlignedBlock {
  const Sequence* sequenceA; const Sequence* sequenceB;
  int startIndexA, startIndexB, lengthA, lengthB;
  AlignedBlock(const Sequence* a, const Sequence* b, int sa, int sb, int la, int lb)
      : sequenceA(a), sequenceB(b), startIndexA(sa), startIndexB(sb), lengthA(la), lengthB(lb) {}
  int getStartIndexA() const { return startIndexA; }
  int getStartIndexB() const { return startIndexB; }
  int getLengthA() const { return lengthA; }
  int getLengthB() const { return lengthB; }
  int getEndIndexA() const { return startIndexA + lengthA; }
  int getEndIndexB() const { return startIndexB + lengthB; }
  int getOffset() const { return startIndexB - startIndexA; }
  const Sequence* getSequenceA() const { return sequenceA; }
  const Sequence* getSequenceB() const { return sequenceB; }
  uint8_t getLastEncodedCharA() const { return sequenceA->encodedCharAt(getEndIndexA() - 1); }
  uint8_t getLastEncodedCharB() const { return sequenceB->encodedCharAt(getEndIndexB() - 1); }
  // 0 = 1-1, 1 = insertion (lenB==0), 2 = deletion (lenA==0)
  int indelType() const { return lengthA == lengthB ? 0 : (lengthA > lengthB ? 1 : 2); }
  bool sameIndelType(const AlignedBlock& o) const { return indelType() == o.indelType(); }
};

struct SequenceAlignment {
  std::vector<AlignedBlock> sections;
  bool referenceReversed = false;
  double totalPenalty = 0, alignedPenalty = 0;
  const std::vector<AlignedBlock>& getSections() const { return sections; }
  bool isReferenceReversed() const { return referenceReversed; }
  double getPenalty() const { return totalPenalty; }
  double getAlignedPenalty() const { return alignedPenalty; }
  const Sequence* getSequenceA() const { return sections[0].sequenceA; }
  const Sequence* getSequenceB() const { return sections[0].sequenceB; }
  int getStartIndexA() const { return sections.front().startIndexA; }
  int getEndIndexA() const { return sections.back().getEndIndexA(); }
  int getStartIndexB() const { return sections.front().startIndexB; }
  int getEndIndexB() const { return sections.back().getEndIndexB(); }
  int getLengthA() const { int t = 0; for (auto& b : sections) t += b.lengthA; return t; }
  int getStartOffset() const { return sections.front().getOffset(); }
  bool hasIndel() const { for (auto& b : sections) if (b.lengthA != b.lengthB) return true; return false; }
  int countNumIndels() const { int n = 0; for (auto& b : sections) if (b.lengthA != b.lengthB) n++; return n; }
  // [inferred] total number of inserted bases on either side
  int getInsertAOrBLength() const { int t = 0; for (auto& b : sections) if (b.lengthA != b.lengthB) t += b.lengthA + b.lengthB; return t; }
  // [inferred] number of query bases aligned to reference positions < refIndex
  int getLengthABefore(int refIndex) const {
    int t = 0;
    for (auto& b : sections) {
      if (b.getEndIndexB() <= refIndex) t += b.lengthA;
      else if (b.startIndexB >= refIndex) t += 0;
      else if (b.lengthA == b.lengthB) t += refIndex - b.startIndexB;
    }
    return t;
  }
  // [inferred] number of query bases aligned to reference positions >= refIndex
  int getLengthAAfter(int refIndex) const {
    int t = 0;
    for (auto& b : sections) {
      if (b.startIndexB >= refIndex) t += b.lengthA;
      else if (b.getEndIndexB() <= refIndex) t += 0;
      else if (b.lengthA == b.lengthB) t += b.getEndIndexB() - refIndex;
    }
    return t;
  }
  bool hasAmbiguousBasepairs() const {
    for (auto& b : sections) {
      if (b.lengthA != b.lengthB) continue;
      for (int i = 0; i < b.lengthA; i++) {
        if (Basepairs::isAmbiguous(b.sequenceA->encodedCharAt(b.startIndexA + i))) return true;
        if (Basepairs::isAmbiguous(b.sequenceB->encodedCharAt(b.startIndexB + i))) return true;
      }
    }
    return false;
  }
  std::string getAlignedTextA() const {
    std::string s;
    for (auto& b : sections) {
      if (b.lengthA > 0) s += b.sequenceA->getRange(b.startIndexA, b.lengthA);
      else s += std::string((size_t)b.lengthB, '-');
    }
    return s;
  }
  std::string getAlignedTextB() const {
    std::string s;
    for (auto& b : sections) {
      if (b.lengthB > 0) s += b.sequenceB->getRange(b.startIndexB, b.lengthB);
      else s += std::string((size_t)b.lengthA, '-');
    }
    return s;
  }
  bool sameAs(const SequenceAlignment& o) const {
    if (referenceReversed != o.referenceReversed || sections.size() != o.sections.size()) return false;
    if (getSequenceB() != o.getSequenceB()) return false;
    for (size_t i = 0; i < sections.size(); i++) {
      const AlignedBlock &x = sections[i], &y = o.sections[i];
      if (x.startIndexA != y.startIndexA || x.startIndexB != y.startIndexB || x.lengthA != y.lengthA || x.lengthB != y.lengthB) return false;
    }
    return true;
  }
};
typedef std::shared_ptr<SequenceAlignment> SequenceAlignmentP;

struct QueryAlignment {
  std::vector<SequenceAlignmentP> components;
  double spacingPenalty = 0, overlapMultiplier = 1, duplicationBonus = 0, totalPenalty = 0;
  int innerDistance = 0;
  double getPenalty() const { return totalPenalty; }
  const SequenceAlignment& getComponent(int i) const { return *components[(size_t)i]; }
  bool hasIndel() const { for (auto& c : components) if (c->hasIndel()) return true; return false; }
  bool hasAmbiguousBasepairs() const { for (auto& c : components) if (c->hasAmbiguousBasepairs()) return true; return false; }
  // [inferred] QueryAlignment.equals: same blocks in every component
  bool sameAs(const QueryAlignment& o) const {
    if (components.size() != o.components.size()) return false;
    for (size_t i = 0; i < components.size(); i++) if (!components[i]->sameAs(*o.components[i])) return false;
    return true;
  }
};
typedef std::shared_ptr<QueryAlignment> QueryAlignmentP;

// QueryAlignments: numComponents == 1 normally, 2 when paired reads fall back to unpaired (M/AlignerWorker.java:643)
struct QueryAlignments {
  std::vector<std::vector<QueryAlignmentP>> components;
};

// ---------------------------------------------------------------- Query  [inferred: QuickVariants]
// Single-end: expectedInnerDistance = 0 and any positive deviation => spacing penalty 0
// (pinned by AS:f:0.0 in T/SamWriter_Test.java:26).
struct Query {
  std::vector<const Sequence*> sequences;
  double expectedInnerDistance = 0;
  double spacingDeviationPerUnitPenalty = 1;
  int getNumSequences() const { return (int)sequences.size(); }
  const Sequence* getSequence(int i) const { return sequences[(size_t)i]; }
  int getLength() const { int t = 0; for (auto s : sequences) t += s->getLength(); return t; }
  double getExpectedInnerDistance() const { return expectedInnerDistance; }
  double getSpacingDeviationPerUnitPenalty() const { return spacingDeviationPerUnitPenalty; }
  Query subquery(int i) const { Query q; q.sequences.push_back(sequences[(size_t)i]); q.expectedInnerDistance = expectedInnerDistance; q.spacingDeviationPerUnitPenalty = spacingDeviationPerUnitPenalty; return q; }
};

// ---------------------------------------------------------------- AlignmentParameters  (M/AlignmentParameters.java:6-182)
struct AlignmentParameters {
  double MutationPenalty = 0;
  double InsertionStart_Penalty = 0, InsertionExtension_Penalty = 0;
  double DeletionStart_Penalty = 0, DeletionExtension_Penalty = 0;
  double MaxErrorRate = 0;
  double UnalignedPenalty = 0;
  double AmbiguityPenalty = 0;
  int MaxNumMatches = INT32_MAX;
  double Max_PenaltySpan = 0;
  bool StartingInsertionStartFree = false;

  double getStartingInsertionStartPenalty() const { return StartingInsertionStartFree ? 0 : InsertionStart_Penalty; }  // :36-40
  double getMinPossibleNonzeroPenalty() const {  // :42-47
    double result = MutationPenalty;
    result = std::min(result, getStartingInsertionStartPenalty() + InsertionStart_Penalty);
    result = std::min(result, DeletionStart_Penalty + DeletionExtension_Penalty);
    return result;
  }
  double getPenalty(uint8_t encodedQuery, uint8_t encodedReference) const {  // :156-180
    if (!Basepairs::canMatch(encodedReference, encodedQuery)) return MutationPenalty;
    uint8_t u = Basepairs::unionOf(encodedQuery, encodedReference);
    return AmbiguityPenalty * Basepairs::getMutationFalseNegativeRate(u);
  }
  double getPenalty(const AlignedBlock& block) const {  // :106-126
    double penalty = 0;
    if (block.lengthA == block.lengthB) {
      for (int i = 0; i < block.lengthA; i++) {
        uint8_t a = block.sequenceA->encodedCharAt(block.startIndexA + i);
        uint8_t b = block.sequenceB->encodedCharAt(block.startIndexB + i);
        penalty += getPenalty(a, b);
      }
    } else if (block.lengthA > 0) {
      penalty += InsertionStart_Penalty;
      penalty += InsertionExtension_Penalty * block.lengthA;
    } else {
      penalty += DeletionStart_Penalty;
      penalty += DeletionExtension_Penalty * block.lengthB;
    }
    return penalty;
  }
  double getPenalty(const AlignedBlock& block, int startIndexB, int endIndexB) const {  // :128-154
    double penalty = 0;
    if (block.lengthA == block.lengthB) {
      for (int i = 0; i < block.lengthA; i++) {
        int bIndex = block.startIndexB + i;
        if (bIndex >= startIndexB && bIndex < endIndexB) {
          uint8_t a = block.sequenceA->encodedCharAt(block.startIndexA + i);
          uint8_t b = block.sequenceB->encodedCharAt(bIndex);
          penalty += getPenalty(a, b);
        }
      }
    } else if (block.startIndexB < endIndexB && block.getEndIndexB() > startIndexB) {
      if (block.lengthA > 0) {
        penalty += InsertionStart_Penalty;
        penalty += InsertionExtension_Penalty * block.lengthA;
      } else {
        penalty += DeletionStart_Penalty;
        penalty += DeletionExtension_Penalty * block.lengthB;
      }
    }
    return penalty;
  }
  double getPenalty(const SequenceAlignment& alignment, int startIndexB, int endIndexB) const {  // :97-103
    double total = 0;
    for (auto& b : alignment.sections) total += getPenalty(b, startIndexB, endIndexB);
    return total;
  }
  SequenceAlignmentP newSequenceAlignment(const std::vector<AlignedBlock>& sections, bool referenceReversed) const {  // :73-95
    int alignedQueryLength = 0;
    double totalPenalty = 0;
    for (auto& block : sections) {
      totalPenalty += getPenalty(block);
      alignedQueryLength += block.lengthA;
    }
    if (!sections.empty()) {
      if (StartingInsertionStartFree && sections[0].lengthB == 0) totalPenalty -= InsertionStart_Penalty;
    }
    double alignedPenalty = totalPenalty;
    if (!sections.empty()) {
      int unalignedQueryLength = sections[0].sequenceA->getLength() - alignedQueryLength;
      double unalignedPenalty = (double)unalignedQueryLength * UnalignedPenalty;
      totalPenalty += unalignedPenalty;
    }
    SequenceAlignmentP r(new SequenceAlignment());
    r->sections = sections;
    r->referenceReversed = referenceReversed;
    r->totalPenalty = totalPenalty;
    r->alignedPenalty = alignedPenalty;
    return r;
  }
};

struct SequenceSection {  // M/SequenceSection.java
  const Sequence* sequence; int startIndex, endIndex;
  SequenceSection(const Sequence* s, int a, int b) : sequence(s), startIndex(a), endIndex(b) {}
  const Sequence* getSequence() const { return sequence; }
  int getStartIndex() const { return startIndex; }
  int getEndIndex() const { return endIndex; }
  int getLength() const { return endIndex - startIndex; }
};

// counters used for the roofline accounting of SURVEY.md §8(d)
struct Counters {
  int64_t reads = 0, headerProbes = 0, bucketFetches = 0, hitsFetched = 0, flankChecks = 0,
          candidatesExtended = 0, ungappedOnly = 0, pathAlignerCalls = 0, pathAlignerNodes = 0,
          quickAccepts = 0, blocksOut = 0;
  // observer of the product's rejection filter in front of PathAligner (xmo_extend.h PathAligner::boundObserve; off unless xmo_observe_bound(1)): it never changes what the oracle returns
  int64_t pathNullSearches = 0, pathNullNodes = 0, pathBoundChecks = 0, pathBoundRejects = 0, pathBoundRejectNodes = 0;
  // (pieces: BlockAligner.alignPiece calls the piece-level filter takes / rejects; PathAligner calls and nodes the reference spent inside rejected pieces)
  int64_t pieceChecks = 0, pieceRejects = 0, skippedCalls = 0, skippedNodes = 0;
};

}  // namespace xmo
