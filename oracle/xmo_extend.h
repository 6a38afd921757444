// ORACLE — test infrastructure only (see xmo_types.h).
// Restates the extension chain: M/LocalAligner.java, M/StraightAligner.java, M/SkipHighAmbiguity_Aligner.java,
// M/HashBlock_Aligner.java, M/HashBlock_Matcher.java, M/CountMap.java, M/BlockAligner.java, M/PathAligner.java,
// M/PathAligner_Runner.java, M/AlignmentNode.java, M/AlignmentAnalysis.java, M/PenaltyAnalysis.java,
// M/QueryMatch_Aligner.java.
#pragma once
#include "xmo_seed.h"

namespace xmo {

// ---------------------------------------------------------------- HashBlock_Matcher (M/HashBlock_Matcher.java)
struct HashBlock_Matcher {
  static constexpr int NO_MATCHES = -1, MULTIPLE_MATCHES = -2, UNKNOWN = -3;
  const Sequence* query; const Sequence* reference;
  int referenceStart, referenceLength, blockLength, sectionLength, maxSectionIndex, numPossibilities, maxPossibility;
  struct Section { bool present = false; std::vector<int> table; };
  std::vector<Section> locations;

  HashBlock_Matcher(const Sequence* query, const SequenceSection& referenceSection, int sectionLength) {  // :14-29
    if (sectionLength < 1) sectionLength = 1;
    blockLength = j2i(std::log((double)(sectionLength * 5)) / std::log(4.0) + 1);
    if (blockLength < 3) blockLength = 3;
    reference = referenceSection.getSequence();
    referenceStart = referenceSection.getStartIndex();
    referenceLength = referenceSection.getLength();
    this->sectionLength = sectionLength;
    this->query = query;
    maxSectionIndex = getSectionIndex(reference->getLength() - 1);
    numPossibilities = j2i(std::pow(4.0, (double)blockLength));
    maxPossibility = numPossibilities - 1;
  }
  int getBlockLength() const { return blockLength; }
  int getSectionLength() const { return sectionLength; }
  static int encodedCharToInt(uint8_t b) {  // :184-197
    switch (b) { case 1: return 0; case 2: return 1; case 4: return 2; case 8: return 3; default: throw std::runtime_error("invalid encoded char"); }
  }
  int getSectionIndex(int referenceIndex) const { return (referenceIndex - referenceStart) / sectionLength; }  // :199-201
  int encodeBlock(const Sequence* sequence, int index) const {  // :79-91
    if (index + blockLength > sequence->getLength()) return UNKNOWN;
    int sum = 0;
    for (int i = 0; i < blockLength; i++) {
      uint8_t here = sequence->encodedCharAt(index + i);
      if (Basepairs::isAmbiguous(here)) return UNKNOWN;
      sum = sum * 4 + encodedCharToInt(here);
    }
    return sum;
  }
  void indexSection(int sectionIndex, std::vector<int>& section) {  // :40-77
    section.assign((size_t)numPossibilities, NO_MATCHES);
    int previousEncoded = UNKNOWN;
    int startIndex = referenceStart + sectionIndex * sectionLength;
    int endIndex = std::min(startIndex + sectionLength, referenceStart + referenceLength - blockLength);
    for (int i = startIndex; i < endIndex; i++) {
      int encoded;
      if (previousEncoded == UNKNOWN) {
        encoded = encodeBlock(reference, i);
      } else {
        uint8_t nextChar = reference->encodedCharAt(i + blockLength - 1);
        if (Basepairs::isAmbiguous(nextChar)) encoded = UNKNOWN;
        else encoded = ((previousEncoded * 4) & maxPossibility) + encodedCharToInt(nextChar);
      }
      if (encoded == UNKNOWN) continue;  // (sic) previousEncoded keeps its stale value
      int existing = section[(size_t)encoded];
      section[(size_t)encoded] = (existing == NO_MATCHES) ? i : MULTIPLE_MATCHES;
      previousEncoded = encoded;
    }
  }
  Section& getSection(int index) {  // :203-215 (sections skipped by a jump stay "null" forever)
    if ((int)locations.size() > index) return locations[(size_t)index];
    while ((int)locations.size() <= index) locations.push_back(Section());
    locations[(size_t)index].present = true;
    indexSection(index, locations[(size_t)index].table);
    return locations[(size_t)index];
  }
  bool canPositionsMatch(int queryIndex, int referenceIndex) const {  // :159-171
    if (referenceIndex + blockLength > referenceStart + referenceLength) return false;
    for (int i = 0; i < blockLength; i++) {
      uint8_t a = query->encodedCharAt(queryIndex++);
      uint8_t b = reference->encodedCharAt(referenceIndex++);
      if (!Basepairs::canMatch(a, b)) return false;
    }
    return true;
  }
  int scanSection(int queryIndex, int sectionIndex) const {  // :143-157
    int result = NO_MATCHES;
    int startIndex = referenceStart + sectionIndex * sectionLength;
    int endIndex = startIndex + sectionLength;
    for (int i = startIndex; i < endIndex; i++) {
      if (canPositionsMatch(queryIndex, i)) {
        if (result == NO_MATCHES) result = i; else return MULTIPLE_MATCHES;
      }
    }
    return result;
  }
  int lookup(int queryIndex, int minReferenceIndex, int maxReferenceIndex) {  // :98-141
    if (minReferenceIndex < 0) return UNKNOWN;
    if (maxReferenceIndex > reference->getLength()) return UNKNOWN;
    int encoded = encodeBlock(query, queryIndex);
    if (encoded < 0) return UNKNOWN;
    int matched = NO_MATCHES;
    int minSectionIndex = std::max(0, getSectionIndex(minReferenceIndex));
    int maxSection = std::min(maxSectionIndex, getSectionIndex(maxReferenceIndex));
    for (int sectionIndex = minSectionIndex; sectionIndex <= maxSection; sectionIndex++) {
      Section& section = getSection(sectionIndex);
      int lookedUp;
      if (sectionLength < 3) lookedUp = scanSection(queryIndex, sectionIndex);
      else if (section.present) lookedUp = section.table[(size_t)encoded];
      else return UNKNOWN;
      if (lookedUp == UNKNOWN) return UNKNOWN;
      if (lookedUp == MULTIPLE_MATCHES) return MULTIPLE_MATCHES;
      if (lookedUp == NO_MATCHES) continue;
      if (lookedUp < minReferenceIndex || lookedUp > maxReferenceIndex) continue;
      if (matched != NO_MATCHES) return MULTIPLE_MATCHES;
      matched = lookedUp;
    }
    return matched;
  }
};

struct AlignmentAnalysis {  // M/AlignmentAnalysis.java
  std::shared_ptr<HashBlock_Matcher> hashBlock_matcher;
  int predictedBestOffset = 0, lastCheckedOffset = 0;
  bool confidentAboutBestOffset = false;
  double maxInsertionExtensionPenalty = 1000000, maxDeletionExtensionPenalty = 1000000;
  AlignmentAnalysis child() const { return *this; }
};

struct PenaltyAnalysis {  // M/PenaltyAnalysis.java
  double minPossiblePenalty = 0, maxInsertionExtensionPenalty = 0, maxDeletionExtensionPenalty = 0;
  int offsetWithMostHashblockMatches = 0, numHashBlockMatchesWithBestOffset = 0;
};

struct CountMap {  // M/CountMap.java
  int mostPopularKey = 0, mostPopularKey_count = 0;
  bool haveCounts = false;
  std::map<int, int> counts;
  void add(int key, int value) {
    if (key == mostPopularKey || mostPopularKey_count == 0) {
      mostPopularKey_count += value;
      mostPopularKey = key;
      if (haveCounts) counts[mostPopularKey] = mostPopularKey_count;
    } else {
      if (!haveCounts) { haveCounts = true; counts[mostPopularKey] = mostPopularKey_count; }
      int count;
      auto it = counts.find(key);
      if (it == counts.end()) count = value; else count = it->second + value;
      counts[key] = count;
      if (count > mostPopularKey_count) { mostPopularKey = key; mostPopularKey_count = count; }
    }
  }
  int getMaxPopularity() const { return mostPopularKey_count; }
  int getMostPopularKey() const { return mostPopularKey; }
};

struct LocalAligner {  // M/LocalAligner.java
  virtual SequenceAlignmentP align(const SequenceSection& querySection, const SequenceSection& referenceSection, const AlignmentParameters& parameters, AlignmentAnalysis& alignmentAnalysis) = 0;
  virtual ~LocalAligner() {}
  Counters* counters = nullptr;
};

// ---------------------------------------------------------------- the recurrence of the product's rejection filter (observer side: xm_bound.h has the product's form)
// Test infrastructure, not part of the reference.  Prices and budget on the filter's grid of 1/60 penalty unit (rounded down), and the plain affine-gap recurrence over the whole
// rectangle in 64-bit integers: a(i) / b(j) give the query's and the window's bases in the direction of the sweep.  freeStart: a start node at every row of column 0 with insertion
// state 0 (the bound over a piece's whole chain); otherwise the start nodes of PathAligner.java:120-139 for a window that does not let the query extend past the reference.
struct BoundGrid {
  static constexpr int SCALE = 60, KMAX = 2048, MMAX = 4096;
  int64_t thr = 0, mut = 0, isie = 0, ie = 0, dsde = 0, de = 0, amb[4] = {0, 0, 0, 0};
  bool prices(const AlignmentParameters& p, double budget) {
    const double s = (double)SCALE;
    const double t = std::floor((budget + 0.000001 + 0.0000001) * s);
    if (!(t >= 0 && t < 60000.0)) return false;
    thr = (int64_t)t;
    mut = (int64_t)std::floor(p.MutationPenalty * s);
    isie = (int64_t)std::floor((p.InsertionStart_Penalty + p.InsertionExtension_Penalty) * s); ie = (int64_t)std::floor(p.InsertionExtension_Penalty * s);
    dsde = (int64_t)std::floor((p.DeletionStart_Penalty + p.DeletionExtension_Penalty) * s); de = (int64_t)std::floor(p.DeletionExtension_Penalty * s);
    for (int j = 1; j < 4; j++) amb[j] = (int64_t)std::floor(p.AmbiguityPenalty * ((double)j / 3.0) * s);
    return !(mut < 0 || isie < 1 || ie < 1 || dsde < 1 || de < 1 || amb[1] < 0 || mut > 30000 || isie > 30000 || dsde > 30000 || amb[3] > 30000);
  }
  // the band the product would need (it decides which problems the filter takes)
  bool bandFits(int n, int m, bool freeStart) const {
    const int64_t maxIns = freeStart ? thr / ie + 1 : (thr < isie ? 0 : (thr - isie) / ie + 1), maxDel = thr < dsde ? 0 : (thr - dsde) / de + 1;
    const int64_t d0 = freeStart ? 0 : (m >= n ? 0 : -(n - m)), d1 = freeStart ? m : (m >= n ? m - n : 0);
    const int64_t dlo = std::max<int64_t>(d0 - maxIns, -n), dhi = std::min<int64_t>(d1 + maxDel, m);
    return dhi - dlo + 1 <= KMAX;
  }
  template <typename A, typename B>
  bool exceedsBudget(int n, int m, A a, B b, bool freeStart) const {
    auto sub = [&](uint8_t qa, uint8_t rb) -> int64_t {
      if (!Basepairs::canMatch(rb, qa)) return mut;
      return amb[Basepairs::popcount(Basepairs::unionOf(qa, rb)) - 1];
    };
    const int64_t INF = (int64_t)1 << 40;
    std::vector<int64_t> H((size_t)m + 1, INF), E((size_t)m + 1, INF), Hn((size_t)m + 1), En((size_t)m + 1);
    if (freeStart) { for (int y = 0; y <= m; y++) { H[(size_t)y] = 0; E[(size_t)y] = 0; } }
    else for (int y = 0; y <= std::max(m - n, 0); y++) H[(size_t)y] = 0;
    for (int x = 1; x <= n; x++) {
      Hn[0] = (!freeStart && x <= n - m) ? 0 : INF; En[0] = INF;
      int64_t F = INF;
      for (int y = 1; y <= m; y++) {
        const int64_t diag = H[(size_t)y - 1] + sub(a(x - 1), b(y - 1));
        const int64_t e = std::min(E[(size_t)y] + ie, H[(size_t)y] + isie);
        F = std::min(F + de, Hn[(size_t)y - 1] + dsde);
        En[(size_t)y] = e;
        Hn[(size_t)y] = std::min(std::min(diag, e), F);
      }
      H.swap(Hn); E.swap(En);
    }
    int64_t best = INF;
    for (int y = 0; y <= m; y++) best = std::min(best, H[(size_t)y]);
    return best > thr;
  }
};

// ---------------------------------------------------------------- PathAligner (M/PathAligner.java)
struct PathAligner {
  static constexpr double disallowed = 1000000.0;  // :771
  struct Node { int x, y; double penalty, insertXPenalty, insertYPenalty; bool reachedMainDiagonal, reachedOtherDiagonal; };
  AlignmentParameters parameters;
  const Sequence* query; const Sequence* reference;
  std::vector<uint8_t> queryEncodedChars, referenceEncodedChars;
  int startIndexB, endIndexB, startIndexA, endIndexA, textALength, textBLength;
  int startX, startY, goalX, goalY, diagonal;
  double maxInterestingPenalty;
  bool mayQueryExtendPastEndOfReference;
  std::map<double, std::vector<std::pair<int, int>>> prioritizedNodes;  // HashMap<Double,List<Node>> + PriorityQueue<Double>
  std::vector<std::vector<int>> locatedNodes;                           // [x][encodedXY] -> index into pool, -1 = null
  std::vector<Node> pool;
  double activePenalty = 0;
  int stepDelta;
  bool searchReverse;
  AlignmentAnalysis* alignmentAnalysis;
  Counters* counters = nullptr;

  uint8_t getEncodedCharA(int i) const { return queryEncodedChars[(size_t)i]; }
  uint8_t getEncodedCharB(int i) const { return referenceEncodedChars[(size_t)i]; }
  int getSignedDistanceFromDiagonal(int x, int y) const { return x - y - diagonal; }
  int getDistanceFromDiagonal(int x, int y) const { return std::abs(getSignedDistanceFromDiagonal(x, y)); }

  bool chooseSearchReverse() {  // :17-53
    int sumOfMismatchingIndices = 0, numMismatches = 0, sumOfMatchingIndices = 0, numMatches = 0;
    int offset = alignmentAnalysis->predictedBestOffset;
    int startIndex = std::max(startIndexA, startIndexB - offset);
    int endIndex = std::min(endIndexA, endIndexB - offset);
    int length = endIndex - startIndex;
    for (int i = 0; i < length; i++) {
      int j = i - diagonal;
      if (j >= 0 && j < (int)referenceEncodedChars.size()) {
        uint8_t a = getEncodedCharA(i);
        uint8_t b = getEncodedCharB(j);
        if (!Basepairs::canMatch(a, b)) { sumOfMismatchingIndices += i; numMismatches++; }
        else { sumOfMatchingIndices += i; numMatches++; }
      }
    }
    if (numMismatches > 1 && numMatches > 1) {
      int averageMismatchIndex = sumOfMismatchingIndices / numMismatches;
      int averageMatchIndex = sumOfMatchingIndices / numMatches;
      return averageMismatchIndex > averageMatchIndex;
    }
    return true;
  }

  const Node* getNode(int x, int y) const {  // :541-553
    if (x < 0 || (int)locatedNodes.size() <= x) return nullptr;
    const std::vector<int>& diag = locatedNodes[(size_t)x];
    int encodedXY = (y - x) * 2;
    if (encodedXY < 0) encodedXY = -encodedXY - 1;
    if (encodedXY >= (int)diag.size()) return nullptr;
    int idx = diag[(size_t)encodedXY];
    return idx < 0 ? nullptr : &pool[(size_t)idx];
  }
  void saveNode(const Node& node) {  // :523-539
    int x = node.x, y = node.y;
    if (x < 0 || y < 0) return;
    while ((int)locatedNodes.size() <= x) locatedNodes.push_back(std::vector<int>());
    std::vector<int>& diag = locatedNodes[(size_t)x];
    int encodedXY = (y - x) * 2;
    if (encodedXY < 0) encodedXY = -encodedXY - 1;
    while ((int)diag.size() <= encodedXY) diag.push_back(-1);
    pool.push_back(node);
    diag[(size_t)encodedXY] = (int)pool.size() - 1;
  }
  double estimateOverallPenalty(const Node& node) const {  // :475-521
    if (!alignmentAnalysis->confidentAboutBestOffset) return node.penalty;
    int signedDistanceFromDiagonal = getSignedDistanceFromDiagonal(node.x, node.y);
    if (node.reachedMainDiagonal) {
      if (signedDistanceFromDiagonal * stepDelta > 0) {
        double insertionExtensionPenalty = std::fabs(signedDistanceFromDiagonal * parameters.InsertionExtension_Penalty);
        if (insertionExtensionPenalty > alignmentAnalysis->maxInsertionExtensionPenalty) return disallowed;
      } else {
        double deletionExtensionPenalty = std::fabs(signedDistanceFromDiagonal * parameters.DeletionExtension_Penalty);
        if (deletionExtensionPenalty > alignmentAnalysis->maxDeletionExtensionPenalty) return disallowed;
      }
      if (node.reachedOtherDiagonal) return node.penalty;
      double indelPenalty = std::min(parameters.InsertionStart_Penalty + parameters.InsertionExtension_Penalty, parameters.DeletionStart_Penalty + parameters.DeletionExtension_Penalty);
      return node.penalty + indelPenalty;
    }
    if (signedDistanceFromDiagonal * stepDelta < 0) {
      double insertionExtensionPenalty = std::fabs(signedDistanceFromDiagonal * parameters.InsertionExtension_Penalty);
      if (insertionExtensionPenalty > alignmentAnalysis->maxInsertionExtensionPenalty) return disallowed;
      double insertionStartPenalty = std::min(parameters.InsertionStart_Penalty, node.insertXPenalty - node.penalty);
      return node.penalty + insertionStartPenalty + insertionExtensionPenalty;
    } else {
      double deletionExtensionPenalty = std::fabs(signedDistanceFromDiagonal * parameters.DeletionExtension_Penalty);
      if (deletionExtensionPenalty > alignmentAnalysis->maxDeletionExtensionPenalty) return disallowed;
      double deletionStartPenalty = std::min(parameters.DeletionStart_Penalty, node.insertYPenalty - node.penalty);
      return node.penalty + deletionStartPenalty + deletionExtensionPenalty;
    }
  }
  void putNode(const Node& node) {  // :446-473
    double estimatedTotalPenalty = estimateOverallPenalty(node);
    if (estimatedTotalPenalty < activePenalty) estimatedTotalPenalty = activePenalty;
    prioritizedNodes[estimatedTotalPenalty].push_back(std::make_pair(node.x, node.y));
    saveNode(node);
    if (counters) counters->pathAlignerNodes++;
  }
  bool computeUpdated(int x, int y, Node& out) const {  // :573-719
    const Node* existing = getNode(x, y);
    const Node* left = getNode(x - stepDelta, y);
    const Node* up = getNode(x, y - stepDelta);
    const Node* diag = getNode(x - stepDelta, y - stepDelta);
    double insertXPenalty, insertYPenalty, overlayPenalty, newOverlayPenalty;
    insertXPenalty = insertYPenalty = overlayPenalty = newOverlayPenalty = disallowed;
    if (diag) {
      uint8_t a = getEncodedCharA(x - 1);
      uint8_t b = getEncodedCharB(y - 1);
      newOverlayPenalty = parameters.getPenalty(a, b);
      overlayPenalty = diag->penalty + newOverlayPenalty;
    }
    if (left) {
      if (y == goalY && mayQueryExtendPastEndOfReference) {
        insertXPenalty = left->penalty + parameters.UnalignedPenalty;
      } else {
        bool newInsertionAllowed = true;
        {
          int prevAIndex = x - 1 - stepDelta;
          int prevBIndex = y - 1;
          if (prevAIndex >= 0 && prevAIndex < textALength && prevBIndex >= 0 && prevBIndex < textBLength) {
            if (!Basepairs::canMatch(getEncodedCharA(prevAIndex), getEncodedCharB(prevBIndex))) newInsertionAllowed = false;
          }
        }
        if (newInsertionAllowed) {
          int nextAIndex = x - 1;
          int nextBIndex = y - 1 + stepDelta;
          if (nextAIndex >= 0 && nextAIndex < textALength && nextBIndex >= 0 && nextBIndex < textBLength) {
            uint8_t nextA = getEncodedCharA(nextAIndex), nextB = getEncodedCharB(nextBIndex);
            if (parameters.getPenalty(nextA, nextB) == 0) newInsertionAllowed = false;
            else if (Basepairs::isFullyAmbiguous(nextA) || Basepairs::isFullyAmbiguous(nextB)) newInsertionAllowed = false;
          }
        }
        double newInsertXPenalty = newInsertionAllowed ? left->penalty + parameters.InsertionStart_Penalty + parameters.InsertionExtension_Penalty : disallowed;
        double extendInsertXPenalty = left->insertXPenalty + parameters.InsertionExtension_Penalty;
        insertXPenalty = std::min(extendInsertXPenalty, newInsertXPenalty);
      }
    }
    if (up) {
      bool newInsertionAllowed = true;
      {
        int prevAIndex = x - 1;
        int prevBIndex = y - 1 - stepDelta;
        if (prevAIndex >= 0 && prevAIndex < textALength && prevBIndex >= 0 && prevBIndex < textBLength) {
          if (!Basepairs::canMatch(getEncodedCharA(prevAIndex), getEncodedCharB(prevBIndex))) newInsertionAllowed = false;
        }
      }
      if (newInsertionAllowed) {
        int nextAIndex = x - 1 + stepDelta;
        int nextBIndex = y - 1;
        if (nextAIndex >= 0 && nextAIndex < textALength && nextBIndex >= 0 && nextBIndex < textBLength) {
          uint8_t nextA = getEncodedCharA(nextAIndex), nextB = getEncodedCharB(nextBIndex);
          if (parameters.getPenalty(nextA, nextB) == 0) newInsertionAllowed = false;
          else if (Basepairs::isFullyAmbiguous(nextA) || Basepairs::isFullyAmbiguous(nextB)) newInsertionAllowed = false;
        }
      }
      double newInsertYPenalty = newInsertionAllowed ? up->penalty + parameters.DeletionStart_Penalty + parameters.DeletionExtension_Penalty : disallowed;
      double extendInsertYPenalty = up->insertYPenalty + parameters.DeletionExtension_Penalty;
      insertYPenalty = std::min(extendInsertYPenalty, newInsertYPenalty);
    }
    double bestPenalty = std::min(std::min(overlayPenalty, insertXPenalty), insertYPenalty);
    if (!existing || bestPenalty < existing->penalty || insertXPenalty < existing->insertXPenalty || insertYPenalty < existing->insertYPenalty) {
      bool reachedMainDiagonal = false, reachedOtherDiagonal = false;
      if (bestPenalty == disallowed) {
      } else {
        if (bestPenalty == overlayPenalty) { reachedMainDiagonal = diag->reachedMainDiagonal; reachedOtherDiagonal = diag->reachedOtherDiagonal; }
        else if (bestPenalty == insertXPenalty) { reachedMainDiagonal = left->reachedMainDiagonal; reachedOtherDiagonal = left->reachedOtherDiagonal; }
        else { reachedMainDiagonal = up->reachedMainDiagonal; reachedOtherDiagonal = up->reachedOtherDiagonal; }
        if (getDistanceFromDiagonal(x, y) == 0) reachedMainDiagonal = true; else reachedOtherDiagonal = true;
      }
      out = Node{x, y, bestPenalty, insertXPenalty, insertYPenalty, reachedMainDiagonal, reachedOtherDiagonal};
      return true;
    }
    return false;
  }
  void update(int x, int y) {  // :555-571
    if (x <= 0 || x > textALength) return;
    if (y <= 0 || y > textBLength) return;
    Node n;
    if (computeUpdated(x, y, n)) putNode(n);
  }
  void explore(int x, int y) {  // :722-729
    update(x + stepDelta, y);
    update(x, y + stepDelta);
    update(x + stepDelta, y + stepDelta);
  }
  static bool canRemoveSection(const AlignedBlock& block) {  // :358-366
    if (block.lengthA <= 0 && block.lengthB <= 0) return true;
    if ((block.startIndexA <= 0 && block.lengthA <= 0) || (block.startIndexB <= 0 && block.lengthB <= 0)) return true;
    return false;
  }
  SequenceAlignmentP justify(std::vector<AlignedBlock>& sections) {  // :307-352
    for (int i = 1; i < (int)sections.size() - 1; i++) {
      while (true) {
        AlignedBlock left = sections[(size_t)i - 1], middle = sections[(size_t)i], right = sections[(size_t)i + 1];
        if ((middle.lengthA > 0) == (middle.lengthB > 0)) break;
        if (left.lengthA == 0 || left.lengthB == 0) break;
        if (right.lengthA == 0 || right.lengthB == 0) break;
        if (middle.lengthA > 0) { if (left.getLastEncodedCharA() != middle.getLastEncodedCharA()) break; }
        else { if (left.getLastEncodedCharB() != middle.getLastEncodedCharB()) break; }
        sections[(size_t)i - 1] = AlignedBlock(left.sequenceA, left.sequenceB, left.startIndexA, left.startIndexB, left.lengthA - 1, left.lengthB - 1);
        sections[(size_t)i] = AlignedBlock(middle.sequenceA, middle.sequenceB, middle.startIndexA - 1, middle.startIndexB - 1, middle.lengthA, middle.lengthB);
        sections[(size_t)i + 1] = AlignedBlock(right.sequenceA, right.sequenceB, right.startIndexA - 1, right.startIndexB - 1, right.lengthA + 1, right.lengthB + 1);
      }
    }
    while (canRemoveSection(sections[0])) sections.erase(sections.begin());  // (Java would throw on an empty list)
    return parameters.newSequenceAlignment(sections, query->getComplementedFrom() != nullptr);
  }

  // OBSERVER of the product's rejection filter (mapper_amd/csrc/xm_bound.h) - test infrastructure, not part of the reference, off unless a test turns it on
  // (xmo_observe_bound).  It never changes what align() returns: it evaluates, beside the search, the claim the product's filter makes - "the plain affine-gap
  // recurrence over this problem (every move the search of :555-719 can make, at the price :573-719 charges for it or less, from the start nodes of :120-150)
  // stays above maxInterestingPenalty + 1e-6 in every cell of column goalX, so align() returns null" - THROWS when a search it would have rejected returns an
  // alignment, and counts the searches it rejects and the nodes the reference spent in them, so that the product's counters stay comparable with the oracle's.
  // Why the claim holds: a node's three penalties are sums of move prices along one path from a start node, so none is below the recurrence's value of its
  // cell (induction over putNode calls); the loop of :153-192 ends with an answer only through a node at goalX taken from a bucket whose key is <= max + 1e-6
  // (:169,180), and a node's key is never below its penalty (:475-521 only adds to it, :458-461 only raises it).
  // The recurrence is evaluated as the product does it - on an integer grid of 1/60 penalty unit with prices rounded down and the budget
  // floor((max + 1e-6 + 1e-7) * 60) - but over the whole rectangle, in 64-bit integers, without the product's band and interval bookkeeping: the two must agree
  // on every search, which the tests check through the counters (BoundGrid above has the recurrence and the filter's limits).
  // -> 0: the filter does not take the problem, 1: taken, not rejected, 2: rejected
  int boundObserve() const {
    const int n = textALength, m = textBLength;
    BoundGrid g;
    if (!g.prices(parameters, maxInterestingPenalty)) return 0;
    if (mayQueryExtendPastEndOfReference || n < 1 || m < 1 || m > BoundGrid::MMAX) return 0;
    if (!g.bandFits(n, m, false)) return 0;
    // search coordinates: x' query bases consumed, y' window bases consumed, in the direction the search runs
    auto a = [&](int i) { return searchReverse ? queryEncodedChars[(size_t)(n - 1 - i)] : queryEncodedChars[(size_t)i]; };
    auto b = [&](int j) { return searchReverse ? referenceEncodedChars[(size_t)(m - 1 - j)] : referenceEncodedChars[(size_t)j]; };
    return g.exceedsBudget(n, m, a, b, false) ? 2 : 1;
  }
  static int& boundObserver() { static int on = 0; return on; }  // xmo_observe_bound (xmo_capi.cpp)

  SequenceAlignmentP align(const SequenceSection& querySection, const SequenceSection& referenceSection, const AlignmentParameters& params, AlignmentAnalysis& analysis) {
    if (!boundObserver()) return doAlign(querySection, referenceSection, params, analysis);
    const int64_t nodes0 = counters ? counters->pathAlignerNodes : 0;
    SequenceAlignmentP result = doAlign(querySection, referenceSection, params, analysis);
    const int verdict = boundObserve();
    if (verdict == 2 && result) throw std::runtime_error("the rejection filter's bound is not a lower bound: a search it rejects returned an alignment");
    if (counters) {
      const int64_t spent = counters->pathAlignerNodes - nodes0;
      if (!result) { counters->pathNullSearches++; counters->pathNullNodes += spent; }
      if (verdict >= 1) counters->pathBoundChecks++;
      if (verdict == 2) { counters->pathBoundRejects++; counters->pathBoundRejectNodes += spent; }
    }
    if (const char* f = getenv("XMO_SEARCH_LOG")) {  // (scratch analysis: one line per search)
      static FILE* fp = fopen(f, "w");
      fprintf(fp, "%p %d %d %.4f %d %d %d %lld\n", (const void*)query, textALength, textBLength, maxInterestingPenalty, (int)analysis.confidentAboutBestOffset, result ? 1 : 0, verdict,
              (long long)(counters ? counters->pathAlignerNodes - nodes0 : 0));
    }
    return result;
  }

  SequenceAlignmentP doAlign(const SequenceSection& querySection, const SequenceSection& referenceSection, const AlignmentParameters& params, AlignmentAnalysis& analysis) {  // :55-293
    parameters = params;
    maxInterestingPenalty = querySection.getLength() * parameters.MaxErrorRate;
    query = querySection.getSequence();
    startIndexA = querySection.getStartIndex();
    endIndexA = querySection.getEndIndex();
    queryEncodedChars.assign(query->codes.begin() + startIndexA, query->codes.begin() + endIndexA);
    reference = referenceSection.getSequence();
    startIndexB = referenceSection.getStartIndex();
    endIndexB = referenceSection.getEndIndex();
    referenceEncodedChars.assign(reference->codes.begin() + startIndexB, reference->codes.begin() + endIndexB);
    textALength = querySection.getLength();
    textBLength = referenceSection.getLength();
    alignmentAnalysis = &analysis;
    diagonal = startIndexB - (startIndexA + analysis.predictedBestOffset);
    searchReverse = chooseSearchReverse();
    if (searchReverse) { stepDelta = -1; mayQueryExtendPastEndOfReference = startIndexB == 0; }
    else { stepDelta = 1; mayQueryExtendPastEndOfReference = endIndexB == reference->getLength(); }
    const Sequence* sequenceA = query; const Sequence* sequenceB = reference;
    int width = textALength + 2;
    int height = endIndexB - startIndexB + 2;
    if (searchReverse) { startX = width - 1; startY = height - 1; goalX = 1; goalY = 1; }
    else { startX = 0; startY = 0; goalX = width - 2; goalY = height - 2; }

    if (textBLength >= textALength) {
      double startingInsertionStartPenalty = parameters.getStartingInsertionStartPenalty();
      if (!mayQueryExtendPastEndOfReference) startingInsertionStartPenalty = disallowed;
      int initialDeletionCount = std::max(0, textBLength - textALength) + 1;
      for (int i = 0; i < initialDeletionCount; i++) {
        int ya = startY + i * stepDelta;
        putNode(Node{startX, ya, 0, startingInsertionStartPenalty, disallowed, false, false});
      }
    } else {
      int initialInsertionCount = std::max(0, textALength - textBLength) + 1;
      for (int i = 0; i < initialInsertionCount; i++) {
        int xa = startX + i * stepDelta;
        putNode(Node{xa, startY, 0, disallowed, disallowed, false, false});
      }
    }
    if (mayQueryExtendPastEndOfReference) {
      int initialInsertionCount = j2i(analysis.maxInsertionExtensionPenalty / parameters.DeletionExtension_Penalty);
      for (int i = 1; i < initialInsertionCount; i++) {
        int xa = startX + i * stepDelta;
        double penalty = i * parameters.UnalignedPenalty;
        putNode(Node{xa, startY, penalty, disallowed, disallowed, false, false});
      }
    }

    bool haveLast = false;
    int lastX = 0, lastY = 0;
    while (!haveLast) {
      if (prioritizedNodes.empty()) throw std::runtime_error("PathAligner: priority queue empty (Java: NullPointerException)");
      activePenalty = prioritizedNodes.begin()->first;
      for (size_t i = 0; i < prioritizedNodes.begin()->second.size(); i++) {
        std::pair<int, int> xy = prioritizedNodes.begin()->second[i];
        int x = xy.first, y = xy.second;
        if (activePenalty > maxInterestingPenalty + 0.000001) return nullptr;
        if (x == goalX) { haveLast = true; lastX = x; lastY = y; break; }
        explore(x, y);
      }
      prioritizedNodes.erase(prioritizedNodes.begin());
    }
    int i = lastX, j = lastY;
    std::vector<AlignedBlock> blocks;
    while (i != startX && j != startY) {
      const Node* node = getNode(i, j);
      double bestPenalty = node->penalty, insertXPenalty = node->insertXPenalty, insertYPenalty = node->insertYPenalty;
      if (bestPenalty == insertXPenalty) {
        int oldI = i;
        i -= stepDelta;
        while (i != startX) {
          const Node* other = getNode(i, j);
          double otherNewInsertionPenalty = other->penalty + parameters.InsertionStart_Penalty + parameters.InsertionExtension_Penalty;
          double otherExtendInsertionPenalty = other->insertXPenalty + parameters.InsertionExtension_Penalty;
          if (otherNewInsertionPenalty < otherExtendInsertionPenalty) break;
          i -= stepDelta;
        }
        if (searchReverse) blocks.push_back(AlignedBlock(sequenceA, sequenceB, startIndexA + oldI - 1, startIndexB + j - 1, i - oldI, 0));
        else blocks.push_back(AlignedBlock(sequenceA, sequenceB, startIndexA + i, startIndexB + j, oldI - i, 0));
      } else if (bestPenalty == insertYPenalty) {
        int oldJ = j;
        j -= stepDelta;
        while (j != startY) {
          const Node* other = getNode(i, j);
          double otherNewDeletionPenalty = other->penalty + parameters.DeletionStart_Penalty + parameters.DeletionExtension_Penalty;
          double otherExtendDeletionPenalty = other->insertYPenalty + parameters.DeletionExtension_Penalty;
          if (otherNewDeletionPenalty < otherExtendDeletionPenalty) break;
          j -= stepDelta;
        }
        if (searchReverse) blocks.push_back(AlignedBlock(sequenceA, sequenceB, startIndexA + i - 1, startIndexB + oldJ - 1, 0, j - oldJ));
        else blocks.push_back(AlignedBlock(sequenceA, sequenceB, startIndexA + i, startIndexB + j, 0, oldJ - j));
      } else {
        int oldI = i, oldJ = j;
        i -= stepDelta;
        j -= stepDelta;
        while (i != startX && j != startY) {
          const Node* other = getNode(i, j);
          if (other->penalty == other->insertXPenalty || other->penalty == other->insertYPenalty) break;
          i -= stepDelta;
          j -= stepDelta;
        }
        if (searchReverse) blocks.push_back(AlignedBlock(sequenceA, sequenceB, startIndexA + oldI - 1, startIndexB + oldJ - 1, i - oldI, j - oldJ));
        else blocks.push_back(AlignedBlock(sequenceA, sequenceB, startIndexA + i, startIndexB + j, oldI - i, oldJ - j));
      }
    }
    if (!searchReverse) std::reverse(blocks.begin(), blocks.end());
    if (blocks.empty()) return nullptr;
    SequenceAlignmentP result = justify(blocks);
    if (result->getAlignedPenalty() > maxInterestingPenalty) return nullptr;
    return result;
  }
};

struct PathAligner_Runner : LocalAligner {  // M/PathAligner_Runner.java
  SequenceAlignmentP align(const SequenceSection& q, const SequenceSection& r, const AlignmentParameters& p, AlignmentAnalysis& a) override {
    PathAligner pa;
    pa.counters = counters;
    if (counters) counters->pathAlignerCalls++;
    return pa.align(q, r, p, a);
  }
};

// ---------------------------------------------------------------- StraightAligner (M/StraightAligner.java)
struct StraightAligner : LocalAligner {
  LocalAligner* nextAligner;
  explicit StraightAligner(LocalAligner* next) : nextAligner(next) {}
  static SequenceAlignmentP straightAlignment(const SequenceSection& querySection, const SequenceSection& referenceSection, const AlignmentParameters& parameters, const AlignmentAnalysis& alignmentAnalysis) {  // :73-94
    int queryStartIndex = querySection.getStartIndex(), queryEndIndex = querySection.getEndIndex();
    int referenceStartIndex = referenceSection.getStartIndex(), referenceEndIndex = referenceSection.getEndIndex();
    int predictedBestOffset = alignmentAnalysis.predictedBestOffset;
    if (queryStartIndex + predictedBestOffset > referenceStartIndex) referenceStartIndex = queryStartIndex + predictedBestOffset;
    else queryStartIndex = referenceStartIndex - predictedBestOffset;
    if (queryEndIndex + predictedBestOffset < referenceEndIndex) referenceEndIndex = queryEndIndex + predictedBestOffset;
    else queryEndIndex = referenceEndIndex - predictedBestOffset;
    const Sequence* query = querySection.getSequence();
    const Sequence* reference = referenceSection.getSequence();
    std::vector<AlignedBlock> blocks(1, AlignedBlock(query, reference, queryStartIndex, referenceStartIndex, queryEndIndex - queryStartIndex, referenceEndIndex - referenceStartIndex));
    return parameters.newSequenceAlignment(blocks, query->getComplementedFrom() != nullptr);
  }
  SequenceAlignmentP align(const SequenceSection& querySection, const SequenceSection& referenceSection, const AlignmentParameters& parameters, AlignmentAnalysis& alignmentAnalysis) override {  // :13-71
    alignmentAnalysis.lastCheckedOffset = alignmentAnalysis.predictedBestOffset;
    SequenceAlignmentP simpleAlignment = straightAlignment(querySection, referenceSection, parameters, alignmentAnalysis);
    double simpleAlignment_totalPenalty = simpleAlignment->getAlignedPenalty();
    double maxInterestingPenalty = querySection.getLength() * parameters.MaxErrorRate;
    double indelPenalty = std::min(parameters.getStartingInsertionStartPenalty() + parameters.InsertionExtension_Penalty, parameters.DeletionStart_Penalty + parameters.DeletionExtension_Penalty);
    if (simpleAlignment_totalPenalty <= 0) return simpleAlignment;
    if (alignmentAnalysis.confidentAboutBestOffset) {
      if (simpleAlignment_totalPenalty <= indelPenalty || (alignmentAnalysis.maxInsertionExtensionPenalty <= 0 && alignmentAnalysis.maxDeletionExtensionPenalty <= 0)) {
        if (simpleAlignment_totalPenalty <= maxInterestingPenalty) return simpleAlignment;
        return nullptr;
      }
      if (indelPenalty > maxInterestingPenalty) return nullptr;
    }
    double simpleAlignment_penaltyRate = simpleAlignment->getAlignedPenalty() / querySection.getLength();
    AlignmentParameters subParameters = parameters;
    subParameters.MaxErrorRate = std::min(simpleAlignment_penaltyRate, parameters.MaxErrorRate);
    SequenceAlignmentP alignment = nextAligner->align(querySection, referenceSection, subParameters, alignmentAnalysis);
    if (!alignment || alignment->getAlignedPenalty() >= simpleAlignment_totalPenalty) {
      if (simpleAlignment_totalPenalty <= maxInterestingPenalty) return simpleAlignment;
    }
    return alignment;
  }
};

struct SkipHighAmbiguity_Aligner : LocalAligner {  // M/SkipHighAmbiguity_Aligner.java
  LocalAligner* nextAligner;
  explicit SkipHighAmbiguity_Aligner(LocalAligner* next) : nextAligner(next) {}
  SequenceAlignmentP align(const SequenceSection& querySection, const SequenceSection& referenceSection, const AlignmentParameters& parameters, AlignmentAnalysis& alignmentAnalysis) override {
    int numAmbiguities = 0;
    const Sequence* reference = referenceSection.getSequence();
    for (int i = referenceSection.getStartIndex(); i < referenceSection.getEndIndex(); i++)
      if (Basepairs::isAmbiguous(reference->encodedCharAt(i))) numAmbiguities++;
    if (numAmbiguities >= referenceSection.getLength() / 4) return nullptr;
    return nextAligner->align(querySection, referenceSection, parameters, alignmentAnalysis);
  }
};

// ---------------------------------------------------------------- HashBlock_Aligner (M/HashBlock_Aligner.java)
struct HashBlock_Aligner : LocalAligner {
  LocalAligner* nextAligner;
  explicit HashBlock_Aligner(LocalAligner* next) : nextAligner(next) {}

  SequenceAlignmentP align(const SequenceSection& querySection, const SequenceSection& referenceSection, const AlignmentParameters& parameters, AlignmentAnalysis& alignmentAnalysis) override {  // :21-81
    double maxInterestingPenalty = parameters.MaxErrorRate * querySection.getLength();
    if (querySection.getLength() > referenceSection.getLength())
      return nextAligner->align(querySection, referenceSection, parameters, alignmentAnalysis);
    PenaltyAnalysis penaltyAnalysis = analyzePenalty(querySection, referenceSection, parameters, alignmentAnalysis);
    if (penaltyAnalysis.minPossiblePenalty > maxInterestingPenalty) return nullptr;
    int offsetWithMostHashblockMatches = penaltyAnalysis.offsetWithMostHashblockMatches;
    int numHashblocksWithBestOffset = penaltyAnalysis.numHashBlockMatchesWithBestOffset;
    AlignmentAnalysis subAnalysis = alignmentAnalysis.child();
    subAnalysis.maxInsertionExtensionPenalty = penaltyAnalysis.maxInsertionExtensionPenalty;
    subAnalysis.maxDeletionExtensionPenalty = penaltyAnalysis.maxDeletionExtensionPenalty;
    double extraPenaltyForMissingAllHashblockMatches = numHashblocksWithBestOffset * parameters.MutationPenalty + penaltyAnalysis.minPossiblePenalty;
    if (extraPenaltyForMissingAllHashblockMatches > maxInterestingPenalty) {
      subAnalysis.predictedBestOffset = offsetWithMostHashblockMatches;
      subAnalysis.confidentAboutBestOffset = true;
    } else {
      if (!alignmentAnalysis.confidentAboutBestOffset) subAnalysis.predictedBestOffset = offsetWithMostHashblockMatches;
    }
    if (alignmentAnalysis.confidentAboutBestOffset && subAnalysis.predictedBestOffset == alignmentAnalysis.predictedBestOffset)
      subAnalysis.confidentAboutBestOffset = true;
    SequenceSection referenceSubsection = referenceSection;
    if (subAnalysis.confidentAboutBestOffset) {
      int maxDeletionLength = j2i((double)penaltyAnalysis.maxDeletionExtensionPenalty / (double)parameters.DeletionExtension_Penalty);
      int maxInsertionLength = j2i((double)penaltyAnalysis.maxInsertionExtensionPenalty / (double)parameters.InsertionExtension_Penalty);
      int maxIndelLength = std::max(maxDeletionLength, maxInsertionLength);
      int referenceStart = std::max(referenceSection.getStartIndex(), querySection.getStartIndex() + subAnalysis.predictedBestOffset - maxIndelLength);
      int referenceEnd = std::min(referenceSection.getEndIndex(), querySection.getEndIndex() + subAnalysis.predictedBestOffset + maxIndelLength);
      referenceSubsection = SequenceSection(referenceSection.getSequence(), referenceStart, referenceEnd);
    }
    if (referenceSubsection.getLength() < referenceSection.getLength())
      return this->align(querySection, referenceSubsection, parameters, subAnalysis);
    return nextAligner->align(querySection, referenceSubsection, parameters, subAnalysis);
  }

  bool isTooManyMismatches(int numMismatches, const AlignmentParameters& parameters, double maxInterestingPenalty) const {  // :83-92
    return getMinIndelPenaltyForBlockMismatches(numMismatches, parameters) > maxInterestingPenalty;
  }

  PenaltyAnalysis analyzePenalty(const SequenceSection& querySection, const SequenceSection& referenceSection, const AlignmentParameters& parameters, AlignmentAnalysis& alignmentAnalysis) {  // :94-283
    const Sequence* query = querySection.getSequence();
    const Sequence* reference = referenceSection.getSequence();
    std::shared_ptr<HashBlock_Matcher> matcher = alignmentAnalysis.hashBlock_matcher;
    double maxInterestingPenalty = parameters.MaxErrorRate * querySection.getLength();
    int numMismatches = 0;
    int maxNonmatchingBlockEnd = querySection.getStartIndex();
    CountMap counts;
    int numLateBlocksSupportingInsertion = 0, numLateBlocksSupportingDeletion = 0;
    int minPossibleOffset = referenceSection.getStartIndex() - querySection.getStartIndex();
    int maxPossibleOffset = referenceSection.getEndIndex() - querySection.getEndIndex();
    int lookupUncertainty = maxPossibleOffset - minPossibleOffset;
    if (!matcher || std::abs(matcher->getSectionLength() - lookupUncertainty) > lookupUncertainty / 2) {
      matcher.reset(new HashBlock_Matcher(query, referenceSection, lookupUncertainty));
      if (!alignmentAnalysis.hashBlock_matcher) alignmentAnalysis.hashBlock_matcher = matcher;
    }
    int blockLength = matcher->getBlockLength();
    int maxBlockStart = querySection.getEndIndex() - blockLength;
    for (int blockStartIndex = querySection.getStartIndex(); blockStartIndex <= maxBlockStart; blockStartIndex++) {
      if (blockStartIndex >= maxNonmatchingBlockEnd) {
        int position = matcher->lookup(blockStartIndex, blockStartIndex + minPossibleOffset, blockStartIndex + maxPossibleOffset + 1);
        int offset = position - blockStartIndex;
        if (position == HashBlock_Matcher::UNKNOWN || position == HashBlock_Matcher::MULTIPLE_MATCHES) continue;
        if (position == HashBlock_Matcher::NO_MATCHES) {
          numMismatches++;
          maxNonmatchingBlockEnd = blockStartIndex + blockLength;
          if (isTooManyMismatches(numMismatches, parameters, maxInterestingPenalty)) break;
          continue;
        }
        int otherStartIndex = position;
        int reverseCount = std::min(blockStartIndex - maxNonmatchingBlockEnd, otherStartIndex);
        bool foundMismatch = false;
        for (int i = 1; i <= reverseCount; i++) {
          int indexA = blockStartIndex - i;
          int indexB = otherStartIndex - i;
          if (!Basepairs::canMatch(query->encodedCharAt(indexA), reference->encodedCharAt(indexB))) {
            numMismatches++;
            foundMismatch = true;
            maxNonmatchingBlockEnd = blockStartIndex + blockLength;
            break;
          }
        }
        if (!foundMismatch) {
          int forwardShift = querySection.getEndIndex() - blockStartIndex;
          for (int i = blockLength; i < forwardShift; i++) {
            int indexA = blockStartIndex + i;
            int indexB = otherStartIndex + i;
            uint8_t encodedCharA = query->encodedCharAt(indexA);
            uint8_t encodedCharB = (indexB < referenceSection.getEndIndex()) ? reference->encodedCharAt(indexB) : 0;
            if (!Basepairs::canMatch(encodedCharA, encodedCharB)) {
              numMismatches++;
              foundMismatch = true;
              maxNonmatchingBlockEnd = indexA + 1;
              break;
            }
          }
          if (!foundMismatch) maxNonmatchingBlockEnd = querySection.getEndIndex();
          int numOtherContainedUniqueHashblockMatches = 0;
          int forwardShift2 = maxNonmatchingBlockEnd - blockStartIndex - blockLength;
          for (int i = blockLength; i < forwardShift2; i++) {
            int indexA = blockStartIndex + i;
            int lookupResult = matcher->lookup(indexA, indexA + minPossibleOffset, indexA + maxPossibleOffset + 1);
            int offset2 = lookupResult - indexA;
            if (lookupResult >= 0 && offset2 == offset) {
              numOtherContainedUniqueHashblockMatches++;
              i = i - 1 + blockLength;
            }
          }
          if (offset != counts.getMostPopularKey() && counts.getMaxPopularity() > 0) {
            if (offset > counts.getMostPopularKey()) numLateBlocksSupportingDeletion += numOtherContainedUniqueHashblockMatches;
            else numLateBlocksSupportingInsertion += numOtherContainedUniqueHashblockMatches;
          }
          counts.add(offset, numOtherContainedUniqueHashblockMatches);
        }
        if (foundMismatch) {
          if (isTooManyMismatches(numMismatches, parameters, maxInterestingPenalty)) break;
        } else {
          counts.add(offset, 1);
        }
      }
    }
    int mostPopularOffset = counts.getMostPopularKey();
    int mostPopularOffset_count = counts.getMaxPopularity();
    PenaltyAnalysis result;
    double indelPenalty = getMinIndelPenaltyForBlockMismatches(numMismatches, parameters);
    result.minPossiblePenalty = indelPenalty;
    bool couldBestOffsetBeDifferentThanPreviouslyExpected = mostPopularOffset_count < 1 || alignmentAnalysis.lastCheckedOffset != mostPopularOffset;
    if (couldBestOffsetBeDifferentThanPreviouslyExpected) {
      double mismatchPenalty = numMismatches * parameters.MutationPenalty;
      if (result.minPossiblePenalty > mismatchPenalty) result.minPossiblePenalty = mismatchPenalty;
    }
    setMaxExtensionPenalty(numMismatches, numLateBlocksSupportingInsertion, numLateBlocksSupportingDeletion, maxInterestingPenalty, parameters, blockLength, result);
    if (result.maxInsertionExtensionPenalty > alignmentAnalysis.maxInsertionExtensionPenalty) result.maxInsertionExtensionPenalty = alignmentAnalysis.maxInsertionExtensionPenalty;
    if (result.maxDeletionExtensionPenalty > alignmentAnalysis.maxDeletionExtensionPenalty) result.maxDeletionExtensionPenalty = alignmentAnalysis.maxDeletionExtensionPenalty;
    if (mostPopularOffset_count < 1) mostPopularOffset = alignmentAnalysis.predictedBestOffset;
    result.offsetWithMostHashblockMatches = mostPopularOffset;
    result.numHashBlockMatchesWithBestOffset = mostPopularOffset_count;
    return result;
  }

  static double getMinIndelPenaltyForBlockMismatches(int numMismatches, const AlignmentParameters& parameters) {  // :286-310
    numMismatches = std::max(1, numMismatches);
    double minPenaltyPerInitialIndel = std::min(parameters.getStartingInsertionStartPenalty() + parameters.InsertionExtension_Penalty, parameters.DeletionStart_Penalty + parameters.DeletionExtension_Penalty);
    double minPenaltyPerExtension = std::min(parameters.InsertionExtension_Penalty, parameters.DeletionExtension_Penalty);
    double minPenaltyPerSubsequentIndel = std::min(parameters.InsertionStart_Penalty + parameters.InsertionExtension_Penalty, parameters.DeletionStart_Penalty + parameters.DeletionExtension_Penalty);
    double minPenaltyPerSubsequentChange = std::min(parameters.MutationPenalty, minPenaltyPerSubsequentIndel);
    if (numMismatches <= 1) return minPenaltyPerInitialIndel;
    if (numMismatches <= 2) return minPenaltyPerInitialIndel + minPenaltyPerExtension;
    return minPenaltyPerInitialIndel + minPenaltyPerExtension + (numMismatches - 2) * minPenaltyPerSubsequentChange;
  }
  static void setMaxExtensionPenalty(int numMismatches, int numBlocksSupportingInsertion, int numBlocksSupportingDeletion, double totalPenalty, const AlignmentParameters& parameters, int blockLength, PenaltyAnalysis& pa) {  // :313-319
    double longInsertion = getMaxExtensionPenaltyOfLongInsertion(numMismatches + numBlocksSupportingDeletion, totalPenalty, parameters, blockLength);
    double manyInsertions = getMaxExtensionPenaltyOfManyInsertions(numMismatches + numBlocksSupportingInsertion, totalPenalty, parameters);
    pa.maxInsertionExtensionPenalty = std::max(longInsertion, manyInsertions);
    pa.maxDeletionExtensionPenalty = getMaxExtensionPenaltyOfManyDeletions(numMismatches + numBlocksSupportingInsertion, totalPenalty, parameters);
  }
  static double getMaxExtensionPenaltyOfLongInsertion(int numMismatches, double totalPenalty, const AlignmentParameters& parameters, int blockLength) {  // :322-354
    double availablePenalty = totalPenalty - parameters.getStartingInsertionStartPenalty();
    double penaltyOfOnlySNPs = numMismatches * parameters.MutationPenalty;
    double penaltyPerBlockExtension = blockLength * parameters.InsertionExtension_Penalty;
    double extraPenaltyPerBlockExtension = penaltyPerBlockExtension - parameters.MutationPenalty;
    if (extraPenaltyPerBlockExtension <= 0) return availablePenalty;
    if (numMismatches < 2) return availablePenalty;
    double penaltyOfShortExtension = 2 * parameters.InsertionExtension_Penalty;
    if (penaltyOfShortExtension > availablePenalty) return availablePenalty;
    double penaltyOfShortSNPs = 2 * parameters.MutationPenalty;
    double maxAllowedPenaltyIncreasePastAllSNPs = availablePenalty - penaltyOfOnlySNPs;
    double maxAllowedPenaltyForBlockExtensions = maxAllowedPenaltyIncreasePastAllSNPs + penaltyOfShortSNPs - penaltyOfShortExtension;
    double maxNumBlockExtensions = maxAllowedPenaltyForBlockExtensions / extraPenaltyPerBlockExtension;
    double maxExtensionPenaltyOfLongInsertion = (maxNumBlockExtensions * blockLength + 2) * parameters.InsertionExtension_Penalty;
    maxExtensionPenaltyOfLongInsertion = std::min(maxExtensionPenaltyOfLongInsertion, availablePenalty);
    if (maxExtensionPenaltyOfLongInsertion < penaltyOfShortExtension) maxExtensionPenaltyOfLongInsertion = 0;
    return maxExtensionPenaltyOfLongInsertion;
  }
  static double getMaxExtensionPenaltyOfManyInsertions(int numMismatches, double totalPenalty, const AlignmentParameters& parameters) {  // :356-376
    double availablePenalty = totalPenalty + (parameters.InsertionStart_Penalty - parameters.getStartingInsertionStartPenalty());
    double penaltyOfOnlySNPs = numMismatches * parameters.MutationPenalty;
    double penaltyPerShortIndel = parameters.InsertionStart_Penalty + 2 * parameters.InsertionExtension_Penalty;
    double extraPenaltyPerShortIndel = penaltyPerShortIndel - 2 * parameters.MutationPenalty;
    if (extraPenaltyPerShortIndel <= 0) return availablePenalty;
    double maxNumShortIndels = (availablePenalty - penaltyOfOnlySNPs) / extraPenaltyPerShortIndel;
    if (maxNumShortIndels < 1) maxNumShortIndels = 0;
    double result = maxNumShortIndels * 2 * parameters.InsertionExtension_Penalty;
    return std::min(result, availablePenalty);
  }
  static double getMaxExtensionPenaltyOfManyDeletions(int numMismatches, double totalPenalty, const AlignmentParameters& parameters) {  // :378-400
    double availablePenalty = totalPenalty;
    double penaltyOfOnlySNPs = numMismatches * parameters.MutationPenalty;
    double penaltyPerShortIndel = parameters.DeletionStart_Penalty + 2 * parameters.DeletionExtension_Penalty;
    double extraPenaltyPerShortIndel = penaltyPerShortIndel - 2 * parameters.MutationPenalty;
    if (extraPenaltyPerShortIndel <= 0) return availablePenalty;
    double maxNumShortIndels = (availablePenalty - penaltyOfOnlySNPs) / extraPenaltyPerShortIndel;
    if (maxNumShortIndels < 1) maxNumShortIndels = 0;
    double result = maxNumShortIndels * 2 * parameters.DeletionExtension_Penalty;
    result = std::min(result, availablePenalty);
    if (result < 0) result = 0;
    return result;
  }
};

// ---------------------------------------------------------------- BlockAligner (M/BlockAligner.java)
struct BlockAligner : LocalAligner {
  LocalAligner* nextAligner;
  explicit BlockAligner(LocalAligner* next) : nextAligner(next) {}

  SequenceAlignmentP align(const SequenceSection& querySection, const SequenceSection& reference, const AlignmentParameters& parameters, AlignmentAnalysis& alignmentAnalysis) override {  // :17-36
    double maxInterestingPenalty = parameters.MaxErrorRate * querySection.getLength();
    std::vector<SequenceAlignmentP> alignments;
    if (!initialAlignments(querySection, reference, parameters, alignmentAnalysis, alignments) || alignments.empty()) return nullptr;
    bool even = false;
    while (alignments.size() > 1) {
      if (!joinAlignments(alignments, reference, parameters, maxInterestingPenalty, alignmentAnalysis, even)) return nullptr;
      even = !even;
    }
    return alignments[0];
  }

  bool initialAlignments(const SequenceSection& querySection, const SequenceSection& referenceSection, const AlignmentParameters& alignmentParameters, AlignmentAnalysis& alignmentAnalysis, std::vector<SequenceAlignmentP>& result) {  // :39-96
    const Sequence* query = querySection.getSequence();
    double maxInterestingPenalty = alignmentParameters.MaxErrorRate * query->getLength();
    int numBasesToEncodeReferencePosition = j2i(std::log((double)referenceSection.getLength() / std::log(4.0))) + 1;  // (sic) :48
    int numHashblocks = querySection.getLength() / numBasesToEncodeReferencePosition + 1;
    int targetNumHashblocksPerBlock = j2i(std::sqrt((double)numHashblocks)) + 1;
    int targetBlockSize = targetNumHashblocksPerBlock * numBasesToEncodeReferencePosition;
    int numBlocks = querySection.getLength() / targetBlockSize;
    result.assign((size_t)numBlocks, nullptr);
    double usedPenalty = 0;
    int numRemainingAlignments = numBlocks;
    while (true) {
      bool failedSubalignment = false, failedSubalignmentThenFoundSubalignment = false;
      int startPosition = querySection.getStartIndex();
      for (int i = 0; i < numBlocks; i++) {
        int endPosition = querySection.getStartIndex() + (querySection.getLength() * (i + 1) / numBlocks);
        if (!result[(size_t)i]) {
          SequenceSection querySubsection(query, startPosition, endPosition);
          double averagePenalty = (maxInterestingPenalty - usedPenalty) / numRemainingAlignments;
          SequenceAlignmentP subAlignment = alignPiece(querySubsection, referenceSection, averagePenalty, alignmentParameters, i == 0, alignmentAnalysis);
          if (subAlignment) {
            if (failedSubalignment) failedSubalignmentThenFoundSubalignment = true;
            numRemainingAlignments--;
            result[(size_t)i] = subAlignment;
            usedPenalty += subAlignment->getAlignedPenalty();
          } else {
            failedSubalignment = true;
          }
        }
        startPosition = endPosition;
      }
      if (numRemainingAlignments < 1) return true;
      if (!failedSubalignmentThenFoundSubalignment) return false;
    }
  }

  bool joinAlignments(std::vector<SequenceAlignmentP>& alignments, const SequenceSection& referenceSection, const AlignmentParameters& alignmentParameters, double maxInterestingPenalty, AlignmentAnalysis& alignmentAnalysis, bool allowSimpleMerges) {  // :99-144
    std::vector<SequenceAlignmentP> result;
    double usedPenalty = 0;
    for (auto& a : alignments) usedPenalty += a->getAlignedPenalty();
    for (int i = 0; i < (int)alignments.size(); i += 2) {
      SequenceAlignmentP merge;
      SequenceAlignmentP left = alignments[(size_t)i];
      if (i + 1 < (int)alignments.size()) {
        SequenceAlignmentP right = alignments[(size_t)i + 1];
        merge = doTryMerge(*left, *right, alignmentParameters);
        if (!merge) {
          usedPenalty -= left->getAlignedPenalty();
          usedPenalty -= right->getAlignedPenalty();
          SequenceSection querySubsection(left->getSequenceA(), left->getStartIndexA(), right->getEndIndexA());
          merge = alignPiece(querySubsection, referenceSection, maxInterestingPenalty - usedPenalty, alignmentParameters, i == 0, alignmentAnalysis);
          if (!merge) return false;
          usedPenalty += merge->getAlignedPenalty();
        } else {
          if (!allowSimpleMerges) {
            result.push_back(left);
            i--;
            continue;
          }
        }
      } else {
        merge = left;
      }
      result.push_back(merge);
    }
    alignments.swap(result);
    return true;
  }

  SequenceAlignmentP doTryMerge(const SequenceAlignment& left, const SequenceAlignment& right, const AlignmentParameters& parameters) {  // :158-190
    if (left.getEndIndexB() != right.getStartIndexB()) return nullptr;
    const std::vector<AlignedBlock>& leftSections = left.getSections();
    const std::vector<AlignedBlock>& rightSections = right.getSections();
    const AlignedBlock& l = leftSections.back();
    const AlignedBlock& r = rightSections.front();
    // tryMergeBlocks :192-212
    if (!l.sameIndelType(r)) return nullptr;
    if (l.getEndIndexA() != r.getStartIndexA()) return nullptr;
    if (l.getEndIndexB() != r.getStartIndexB()) return nullptr;
    AlignedBlock middleBlock(l.sequenceA, l.sequenceB, l.startIndexA, l.startIndexB, l.lengthA + r.lengthA, l.lengthB + r.lengthB);
    std::vector<AlignedBlock> sections;
    for (size_t i = 0; i + 1 < leftSections.size(); i++) sections.push_back(leftSections[i]);
    sections.push_back(middleBlock);
    for (size_t i = 1; i < rightSections.size(); i++) sections.push_back(rightSections[i]);
    return parameters.newSequenceAlignment(sections, left.isReferenceReversed());
  }

  SequenceAlignmentP alignPiece(const SequenceSection& querySection, const SequenceSection& referenceSection, double maxPenalty, const AlignmentParameters& parameters, bool firstPiece, const AlignmentAnalysis& parentAlignmentAnalysis) {  // :215-249
    if (maxPenalty < 0) return nullptr;
    SequenceSection referenceSubsection = referenceSection;
    if (parentAlignmentAnalysis.confidentAboutBestOffset) {
      int maxInsertionLength = j2i((double)parentAlignmentAnalysis.maxInsertionExtensionPenalty / (double)parameters.InsertionExtension_Penalty);
      int maxDeletionLength = j2i((double)parentAlignmentAnalysis.maxDeletionExtensionPenalty / (double)parameters.DeletionExtension_Penalty);
      int maxIndelLength = std::max(maxInsertionLength, maxDeletionLength);
      int referenceStart = std::max(referenceSection.getStartIndex(), querySection.getStartIndex() + parentAlignmentAnalysis.predictedBestOffset - maxIndelLength);
      int referenceEnd = std::min(referenceSection.getEndIndex(), querySection.getEndIndex() + parentAlignmentAnalysis.predictedBestOffset + maxIndelLength);
      if (referenceEnd > referenceStart) referenceSubsection = SequenceSection(referenceSection.getSequence(), referenceStart, referenceEnd);
    }
    AlignmentParameters subParameters = parameters;
    if (!firstPiece) subParameters.StartingInsertionStartFree = true;
    subParameters.MaxErrorRate = maxPenalty / querySection.getLength();
    AlignmentAnalysis childAnalysis = parentAlignmentAnalysis.child();
    childAnalysis.confidentAboutBestOffset = false;
    if (!PathAligner::boundObserver() || !counters) return nextAligner->align(querySection, referenceSubsection, subParameters, childAnalysis);
    // OBSERVER of the product's piece-level rejection (xm_bound.h boundPieceApplies / boundRejects with piece = 1) - test infrastructure, never acts on the verdict:
    // does the recurrence over the piece (its first and last base left out) and this window, a start node at every row, stay above maxPenalty?  If so the chain below must
    // return null - THROWS if it does not - and everything the reference does inside it is what the product skips: its PathAligner calls and nodes are counted apart, and what
    // the search-level observer saw inside is taken back out (the product never gets there).
    const int verdict = pieceObserve(querySection, referenceSubsection, maxPenalty, parameters, parentAlignmentAnalysis);
    const Counters before = *counters;
    SequenceAlignmentP result = nextAligner->align(querySection, referenceSubsection, subParameters, childAnalysis);
    if (verdict == 2 && result) throw std::runtime_error("the rejection filter's bound over a piece is not a lower bound: a piece it rejects aligned");
    if (verdict >= 1) counters->pieceChecks++;
    if (verdict == 2) {
      counters->pieceRejects++;
      counters->skippedCalls += counters->pathAlignerCalls - before.pathAlignerCalls;
      counters->skippedNodes += counters->pathAlignerNodes - before.pathAlignerNodes;
      counters->pathNullSearches = before.pathNullSearches; counters->pathNullNodes = before.pathNullNodes; counters->pathBoundChecks = before.pathBoundChecks;
      counters->pathBoundRejects = before.pathBoundRejects; counters->pathBoundRejectNodes = before.pathBoundRejectNodes;
    }
    return result;
  }
  // -> 0: the piece-level filter does not take the piece, 1: taken, not rejected, 2: rejected (the conditions: xm_bound.h boundPieceApplies)
  int pieceObserve(const SequenceSection& q, const SequenceSection& w, double maxPenalty, const AlignmentParameters& parameters, const AlignmentAnalysis& parent) const {
    const int n = q.getLength(), m = w.getLength();
    if (n < 8 || m < n || w.getStartIndex() <= 0 || w.getEndIndex() >= w.getSequence()->getLength()) return 0;
    const int minOff = w.getStartIndex() - q.getStartIndex(), maxOff = w.getEndIndex() - q.getEndIndex(), u = maxOff - minOff;
    if (parent.predictedBestOffset < minOff || parent.predictedBestOffset > maxOff + 1) return 0;
    if (parent.hashBlock_matcher && !(parent.hashBlock_matcher->getSectionLength() > u + u / 2)) return 0;
    BoundGrid g;
    if (!g.prices(parameters, maxPenalty)) return 0;
    if (m > BoundGrid::MMAX || !g.bandFits(n - 2, m, true)) return 0;
    const Sequence* qs = q.getSequence(); const Sequence* ws = w.getSequence();
    const int qa = q.getStartIndex() + 1, wb = w.getStartIndex();
    auto a = [&](int i) { return qs->encodedCharAt(qa + i); };
    auto b = [&](int j) { return ws->encodedCharAt(wb + j); };
    return g.exceedsBudget(n - 2, m, a, b, true) ? 2 : 1;
  }
};

// ---------------------------------------------------------------- QueryMatch_Aligner (M/QueryMatch_Aligner.java)
struct QueryMatch_Aligner {
  AlignmentParameters parameters;
  Query query;
  std::vector<QueryAlignmentP> goodAlignments;
  double bestPenalty = INT32_MAX;
  // chain of :18-29 (outermost first): Straight -> SkipHighAmbiguity -> HashBlock -> Block -> Straight -> HashBlock -> Straight -> PathAligner_Runner
  PathAligner_Runner a0; StraightAligner a1; HashBlock_Aligner a2; StraightAligner a3; BlockAligner a4; HashBlock_Aligner a5; SkipHighAmbiguity_Aligner a6; StraightAligner a7;
  LocalAligner* aligner;
  Counters* counters;
  std::vector<std::unique_ptr<Sequence>> joinedSequences;  // owns "joined" query sequences

  QueryMatch_Aligner(const Query& query, const AlignmentParameters& initialParameters, Counters* counters)
      : parameters(initialParameters), query(query), a1(&a0), a2(&a1), a3(&a2), a4(&a3), a5(&a4), a6(&a5), a7(&a6), aligner(&a7), counters(counters) {
    a0.counters = counters;
    a4.counters = counters;  // (BlockAligner: the observer of the product's piece-level rejection)
  }

  static double divideRoundUp(double a, double b) {  // :56-61
    double result = a / b;
    if (result * b < a) result = jnextUp(result);
    return result;
  }

  QueryAlignmentP align(const QueryMatch& match, double extraSpacing = 0) {  // :35-54
    QueryAlignmentP alignment = doAlign(match, extraSpacing);
    if (alignment) {
      if (alignment->getPenalty() < bestPenalty) {
        bestPenalty = alignment->getPenalty();
        double newTargetPenalty = alignment->getPenalty() + parameters.Max_PenaltySpan;
        double newTargetErrorRate = divideRoundUp(newTargetPenalty, query.getLength());
        if (newTargetErrorRate < parameters.MaxErrorRate) parameters.MaxErrorRate = newTargetErrorRate;
      }
      goodAlignments.push_back(alignment);
    }
    return alignment;
  }

  std::vector<QueryAlignmentP> getBestAlignments() const {  // :71-92
    double maxInterestingPenaltyAnywhere = query.getLength() * parameters.MaxErrorRate;
    double cutoffPenalty = bestPenalty + parameters.Max_PenaltySpan;
    if (cutoffPenalty > maxInterestingPenaltyAnywhere) cutoffPenalty = maxInterestingPenaltyAnywhere;
    std::vector<QueryAlignmentP> bestAlignments;
    for (auto& a : goodAlignments) if (a->getPenalty() <= cutoffPenalty) bestAlignments.push_back(a);
    if (bestAlignments.size() <= 1) return bestAlignments;
    // withoutDuplicates: HashSet<QueryAlignment> [inferred equals; parity unpinned order => first-occurrence order]
    std::vector<QueryAlignmentP> unique;
    for (auto& a : bestAlignments) {
      bool dup = false;
      for (auto& u : unique) if (u->sameAs(*a)) { dup = true; break; }
      if (!dup) unique.push_back(a);
    }
    return unique;
  }

  int getSpacing(const QueryMatch& match) const { return match.getNumSequences() < 2 ? 0 : match.getTotalDistanceBetweenComponents(); }  // :522-526
  double computeSpacingPenalty(double innerDistance) const {  // :530-546
    double expected = query.getExpectedInnerDistance();
    int totalLength = query.getLength();
    if (innerDistance < 0 && innerDistance > -1 * totalLength) return 0;
    double deviationPerPenalty = query.getSpacingDeviationPerUnitPenalty();
    int penalty = j2i(std::fabs(innerDistance - expected) / deviationPerPenalty);
    return (double)penalty;
  }
  static int countQueryLength(const std::vector<SequenceMatchP>& components) {  // :548-555
    int total = 0;
    for (auto& m : components) if (m) total += m->getSequenceA()->getLength();
    return total;
  }

  QueryAlignmentP doAlign(const QueryMatch& match, double extraSpacing) {  // :94-272
    if (counters) counters->candidatesExtended++;
    double innerDistance = getSpacing(match) + extraSpacing;
    double spacingPenalty = computeSpacingPenalty(innerDistance);
    double overlapMultiplier = 1, duplicationBonus = 0;
    double maxAllowedPenalty = match.getQueryTotalLength() * parameters.MaxErrorRate;
    maxAllowedPenalty = jnextUp(maxAllowedPenalty);
    if (innerDistance > 0) {
      double minPossiblePenalty = spacingPenalty + match.getPriority() * parameters.MutationPenalty;
      if (minPossiblePenalty > maxAllowedPenalty) return nullptr;
    }
    std::vector<SequenceAlignmentP> resultComponents;
    bool haveResultComponents = false;
    double componentsPenalty = 0;
    if (match.getNumSequences() > 1 && innerDistance < 0) {
      const Sequence* joined = tryJoinQuerySequences(match);
      if (joined) {
        SequenceAlignmentP joinedAlignment = computeJoinedAlignment(joined, match);
        if (!splitAlignment(joinedAlignment, match, resultComponents)) return nullptr;
        haveResultComponents = true;
        for (auto& c : resultComponents) componentsPenalty += c->getPenalty();
      }
    }
    if (!haveResultComponents) {
      resultComponents.assign(match.components.size(), nullptr);
      std::vector<SequenceMatchP> remainingQueryComponents = match.components;
      int numRemainingComponents = (int)match.components.size();
      bool checkComponentsInForwardOrder = match.hintCheckComponentsInForwardOrder;
      int firstComponentIndex, componentIndexStep, lastComponentIndex;
      if (checkComponentsInForwardOrder) { firstComponentIndex = 0; componentIndexStep = 1; lastComponentIndex = match.getNumSequences(); }
      else { firstComponentIndex = match.getNumSequences() - 1; componentIndexStep = -1; lastComponentIndex = -1; }
      double maxTotalComponentPenalty;
      if (innerDistance < 0 && match.getNumSequences() > 1) {
        double queryTotalLength = match.getQueryTotalLength();
        double estimatedOverlap = std::min(-1 * innerDistance, (double)std::min(match.getComponent(0).getSequenceA()->getLength(), match.getComponent(1).getSequenceA()->getLength()));
        double estimatedUniqueLength = queryTotalLength - estimatedOverlap;
        maxTotalComponentPenalty = divideRoundUp(maxAllowedPenalty - spacingPenalty, queryTotalLength) * estimatedUniqueLength * 2;
      } else {
        maxTotalComponentPenalty = maxAllowedPenalty - spacingPenalty;
      }
      while (true) {
        int numBases = countQueryLength(remainingQueryComponents);
        if (numBases < 1) break;
        double averagePenaltyPerRemainingItem = divideRoundUp(maxTotalComponentPenalty - componentsPenalty, numBases);
        AlignmentParameters parametersForRemainingSequences = parameters;
        parametersForRemainingSequences.MaxErrorRate = averagePenaltyPerRemainingItem;
        bool foundAMatch = false;
        for (int i = firstComponentIndex; i != lastComponentIndex; i += componentIndexStep) {
          SequenceMatchP componentMatch = remainingQueryComponents[(size_t)i];
          if (componentMatch) {
            SequenceAlignmentP sequenceAlignment = alignMatch(*componentMatch, parametersForRemainingSequences);
            if (sequenceAlignment) {
              resultComponents[(size_t)i] = sequenceAlignment;
              foundAMatch = true;
              remainingQueryComponents[(size_t)i] = nullptr;
              componentsPenalty += sequenceAlignment->getPenalty();
              numRemainingComponents--;
              break;
            }
          }
        }
        if (numRemainingComponents < 1) break;
        if (!foundAMatch) return nullptr;
      }
    }
    double totalUsedPenalty = componentsPenalty;
    if (innerDistance < 0) {
      duplicationBonus = computeDuplicationBonus(resultComponents);
      totalUsedPenalty -= duplicationBonus;
      double multipliedPenalty = multiplyPenaltyForOverlap(resultComponents, totalUsedPenalty);
      if (totalUsedPenalty != 0) overlapMultiplier = multipliedPenalty / totalUsedPenalty;
      else overlapMultiplier = 1;
      totalUsedPenalty = multipliedPenalty;
    }
    totalUsedPenalty += spacingPenalty;
    if (totalUsedPenalty > maxAllowedPenalty) return nullptr;
    int actualInnerDistance = resultComponents.size() > 1 ? resultComponents[1]->getStartIndexB() - resultComponents[0]->getEndIndexB() : 0;
    QueryAlignmentP result(new QueryAlignment());
    result->components = resultComponents;
    result->spacingPenalty = spacingPenalty;
    result->overlapMultiplier = overlapMultiplier;
    result->duplicationBonus = duplicationBonus;
    result->totalPenalty = totalUsedPenalty;
    result->innerDistance = actualInnerDistance;
    return result;
  }

  const Sequence* tryJoinQuerySequences(const QueryMatch& match) {  // :274-284
    const SequenceMatch& match1 = match.getComponent(0);
    const SequenceMatch& match2 = match.getComponent(1);
    int offset = match2.getOffset() - match1.getOffset();
    if (offset >= 0) return tryJoinQuerySequences(match1.getSequenceA(), match2.getSequenceA(), offset);
    return tryJoinQuerySequences(match2.getSequenceA(), match1.getSequenceA(), -offset);
  }
  const Sequence* tryJoinQuerySequences(const Sequence* sequence1, const Sequence* sequence2, int offset) {  // :287-319
    int suffixStartIndex = sequence1->getLength() - offset;
    if (suffixStartIndex < 0) return nullptr;
    int match2IndexEnd = std::min(sequence2->getLength(), sequence1->getLength() - offset);
    for (int match2Index = 0; match2Index < match2IndexEnd; match2Index++) {
      int match1Index = match2Index + offset;
      if (sequence1->encodedCharAt(match1Index) != sequence2->encodedCharAt(match2Index)) return nullptr;
    }
    std::unique_ptr<Sequence> joined(new Sequence());
    joined->name = "joined";
    joined->codes = sequence1->codes;
    int endIndex = sequence2->getLength();
    // sequence2.getRange(suffixStartIndex, endIndex - suffixStartIndex): a negative length throws in Java [inferred]
    if (endIndex - suffixStartIndex < 0) throw std::runtime_error("getRange with negative length");
    for (int i = suffixStartIndex; i < endIndex; i++) joined->codes.push_back(sequence2->codes[(size_t)i]);
    // [inferred] a sequence built by SequenceBuilder is never a reverse complement
    joinedSequences.push_back(std::move(joined));
    return joinedSequences.back().get();
  }
  SequenceAlignmentP computeJoinedAlignment(const Sequence* joined, const QueryMatch& originalMatch) {  // :321-330
    int joinedOffset = std::min(originalMatch.getComponent(0).getOffset(), originalMatch.getComponent(1).getOffset());
    SequenceMatch joinedMatch(joined, originalMatch.getComponent(0).getSequenceB(), joinedOffset);
    AlignmentParameters subParameters = parameters;
    subParameters.MaxErrorRate = jnextUp(subParameters.MaxErrorRate);
    return alignMatch(joinedMatch, subParameters);
  }
  bool splitAlignment(const SequenceAlignmentP& joinedAlignment, const QueryMatch& queryMatch, std::vector<SequenceAlignmentP>& out) {  // :332-360
    if (!joinedAlignment) return false;
    const SequenceMatch& match1 = queryMatch.getComponent(0);
    const Sequence* sequence1 = match1.getSequenceA();
    const SequenceMatch& match2 = queryMatch.getComponent(1);
    const Sequence* sequence2 = match2.getSequenceA();
    int offset = match2.getOffset() - match1.getOffset();
    SequenceAlignmentP alignment1, alignment2;
    if (offset >= 0) {
      alignment1 = extract(*joinedAlignment, 0, sequence1->getLength(), sequence1, match1.getReversed());
      alignment2 = extract(*joinedAlignment, offset, sequence2->getLength() + offset, sequence2, match2.getReversed());
    } else {
      alignment2 = extract(*joinedAlignment, 0, sequence2->getLength(), sequence2, match2.getReversed());
      alignment1 = extract(*joinedAlignment, -offset, sequence1->getLength() - offset, sequence1, match1.getReversed());
    }
    if (!alignment1 || !alignment2) return false;
    out.clear();
    out.push_back(alignment1);
    out.push_back(alignment2);
    return true;
  }
  SequenceAlignmentP extract(const SequenceAlignment& joinedAlignment, int queryStart, int queryEnd, const Sequence* querySeq, bool reverse) {  // :362-405
    bool referenceReversed = joinedAlignment.isReferenceReversed() != reverse;
    const Sequence* reference = joinedAlignment.getSequenceB();
    std::vector<AlignedBlock> blocks;
    for (const AlignedBlock& block : joinedAlignment.getSections()) {
      if (block.getStartIndexA() >= queryEnd) break;
      if (block.getEndIndexA() <= queryStart) continue;
      int selectionStart = std::max(block.getStartIndexA(), queryStart);
      int selectionEnd = std::min(block.getEndIndexA(), queryEnd);
      int querySelectionLength = selectionEnd - selectionStart;
      int referenceSelectionLength, referenceStart;
      if (block.getLengthA() == block.getLengthB()) { referenceSelectionLength = querySelectionLength; referenceStart = selectionStart + block.getOffset(); }
      else if (block.getLengthA() > block.getLengthB()) { referenceSelectionLength = 0; referenceStart = block.getStartIndexB(); }
      else { referenceSelectionLength = block.getLengthB(); referenceStart = selectionStart + block.getOffset(); }
      blocks.push_back(AlignedBlock(querySeq, reference, selectionStart - queryStart, referenceStart, querySelectionLength, referenceSelectionLength));
    }
    if (blocks.empty()) return nullptr;
    return parameters.newSequenceAlignment(blocks, referenceReversed);
  }

  SequenceAlignmentP alignMatch(const SequenceMatch& sequenceMatch, const AlignmentParameters& params) {  // :412-462 (fromHashblockMatch is always true)
    SequenceSection querySection(sequenceMatch.getSequenceA(), sequenceMatch.getStartIndexA(), sequenceMatch.getEndIndexA());
    double maxInterestingPenalty = querySection.getLength() * params.MaxErrorRate;
    int maxIndelLength = j2i(std::max((double)0, (double)(maxInterestingPenalty - params.DeletionStart_Penalty) / params.DeletionExtension_Penalty));
    int maxShift = maxIndelLength;
    int bestOffset = sequenceMatch.getOffset();
    SequenceSection referenceSection(sequenceMatch.getSequenceB(), std::max(0, sequenceMatch.getStartIndexB() - maxShift), std::min(sequenceMatch.getEndIndexB() + maxShift, sequenceMatch.getSequenceB()->getLength()));
    AlignmentAnalysis alignmentAnalysis;
    alignmentAnalysis.maxInsertionExtensionPenalty = maxInterestingPenalty - params.InsertionStart_Penalty;
    alignmentAnalysis.maxDeletionExtensionPenalty = maxInterestingPenalty - params.DeletionStart_Penalty;
    alignmentAnalysis.predictedBestOffset = bestOffset;
    alignmentAnalysis.confidentAboutBestOffset = sequenceMatch.fromHashblockMatch;
    return aligner->align(querySection, referenceSection, params, alignmentAnalysis);
  }

  double multiplyPenaltyForOverlap(const std::vector<SequenceAlignmentP>& components, double totalPenalty) const {  // :464-504
    if (components.size() < 2) return totalPenalty;
    const SequenceAlignment& first = *components[0];
    const SequenceAlignment& second = *components[1];
    double overlappingLengthB = std::min(first.getEndIndexB(), second.getEndIndexB()) - std::max(first.getStartIndexB(), second.getStartIndexB());
    if (overlappingLengthB <= 0) return totalPenalty;
    int uniqueLengthA;
    if (first.getStartIndexB() <= second.getStartIndexB())
      uniqueLengthA = first.getLengthABefore(second.getStartIndexB()) + second.getLengthA() + first.getLengthAAfter(second.getEndIndexB());
    else
      uniqueLengthA = second.getLengthABefore(first.getStartIndexB()) + first.getLengthA() + second.getLengthAAfter(first.getEndIndexB());
    double deletion = std::min(first.getInsertAOrBLength(), second.getInsertAOrBLength());
    uniqueLengthA = j2i((double)uniqueLengthA - deletion);  // Java: int -= double
    if (uniqueLengthA <= 0) return totalPenalty;
    int totalLengthA = first.getLengthA() + second.getLengthA();
    return divideRoundUp(totalPenalty, uniqueLengthA) * totalLengthA;
  }
  double computeDuplicationBonus(const std::vector<SequenceAlignmentP>& components) const {  // :506-520
    if (components.size() < 2) return 0;
    const SequenceAlignment& a = *components[0];
    const SequenceAlignment& b = *components[1];
    double overlappingLength = std::min(a.getEndIndexB(), b.getEndIndexB()) - std::max(a.getStartIndexB(), b.getStartIndexB());
    if (overlappingLength < 0) return 0;
    return (parameters.getPenalty(a, b.getStartIndexB(), b.getEndIndexB()) + parameters.getPenalty(b, a.getStartIndexB(), a.getEndIndexB())) / 2;
  }
};

}  // namespace xmo
