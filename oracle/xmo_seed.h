// ORACLE — test infrastructure only (see xmo_types.h).
// Restates the seed walk + voting: M/HashBlockPath.java, M/Counting_HashBlockPath.java,
// M/HashBlockMatch_Counter.java, M/HashBlockPaths_Counter.java, M/HashBlock_Match.java,
// M/SequenceMatch.java, M/QueryMatch.java.
#pragma once
#include "xmo_index.h"

namespace xmo {

struct SequenceMatch {  // M/SequenceMatch.java
  const Sequence* sequenceA; const Sequence* sequenceB; int offset;
  bool fromHashblockMatch = true;
  SequenceMatch(const Sequence* a, const Sequence* b, int offset) : sequenceA(a), sequenceB(b), offset(offset) {}
  const Sequence* getSequenceA() const { return sequenceA; }
  const Sequence* getSequenceB() const { return sequenceB; }
  int getStartIndexB() const { return std::max(0, offset); }
  int getEndIndexB() const { return std::min(offset + sequenceA->getLength(), sequenceB->getLength()); }
  int getStartIndexA() const { return getStartIndexB() - offset; }
  int getEndIndexA() const { return getEndIndexB() - offset; }
  int getLength() const { return getEndIndexB() - getStartIndexB(); }
  int getOffset() const { return offset; }
  bool equals(const SequenceMatch& o) const { return offset == o.offset && sequenceA == o.sequenceA && sequenceB == o.sequenceB; }
  bool getReversed() const { return sequenceA->getComplementedFrom() != nullptr; }
};
typedef std::shared_ptr<SequenceMatch> SequenceMatchP;

struct QueryMatch {  // M/QueryMatch.java
  std::vector<SequenceMatchP> components;
  int priority = 0;
  bool hintCheckComponentsInForwardOrder = false;
  QueryMatch(const SequenceMatchP& c, int priority) : components(1, c), priority(priority) {}
  QueryMatch(const std::vector<SequenceMatchP>& c, int priority, bool hint) : components(c), priority(priority), hintCheckComponentsInForwardOrder(hint) {}
  const SequenceMatch& getComponent(int i) const { return *components[(size_t)i]; }
  int getNumSequences() const { return (int)components.size(); }
  int getPriority() const { return priority; }
  int getQueryTotalLength() const { int t = 0; for (auto& m : components) t += m->sequenceA->getLength(); return t; }
  bool getReversed() const { return components[0]->getReversed(); }
  int getStartIndexB() const { return std::min(components.front()->getStartIndexB(), components.back()->getStartIndexB()); }  // :48-52
  int getEndIndexB() const { return std::max(components.front()->getStartIndexB(), components.back()->getStartIndexB()); }    // :54-58 (sic: max of the two *start* indices)
  int getTotalDistanceAcross() const {  // :60-67
    if (getReversed()) return components.front()->getEndIndexB() - components.back()->getStartIndexB();
    return components.back()->getEndIndexB() - components.front()->getStartIndexB();
  }
  int getDistance(const SequenceMatch& a, const SequenceMatch& b) const {  // :123-132
    if (a.sequenceB != b.sequenceB) return INT32_MAX;
    if (getReversed()) return a.getStartIndexB() - b.getEndIndexB();
    return b.getStartIndexB() - a.getEndIndexB();
  }
  int getTotalDistanceBetweenComponents() const {  // :70-79
    int totalDistance = 0;
    const SequenceMatch* previous = components[0].get();
    for (size_t i = 1; i < components.size(); i++) {
      totalDistance = jadd(totalDistance, getDistance(*previous, *components[i]));
      previous = components[i].get();
    }
    return totalDistance;
  }
  bool samePosition(const QueryMatch& o) const {  // :81-93
    if (components.size() != o.components.size()) return false;
    for (size_t i = 0; i < components.size(); i++) if (!components[i]->equals(*o.components[i])) return false;
    return true;
  }
};
typedef std::shared_ptr<QueryMatch> QueryMatchP;
typedef std::shared_ptr<std::vector<QueryMatchP>> QueryMatchListP;

// ---------------------------------------------------------------- HashBlockPath (M/HashBlockPath.java)
struct HashBlockPath {
  HashBlock_Pyramid* pyramid;
  Readable_HashBlock_Database* database;
  const Sequence* query;
  int batchIndex = -1;
  MultiBlockP currentBlock;
  const HashBlock* currentGapmer = nullptr;
  bool currentGapmerComputed = false;
  const HashBlock* previousInterestingBlock = nullptr;
  const HashBlock* previousPreviousInterestingBlock = nullptr;
  std::deque<HashBlock> pool;  // owns the gapmers (Java: heap objects with identity)

  HashBlockPath(HashBlock_Pyramid* pyramid, Readable_HashBlock_Database* database, const Sequence* query)
      : pyramid(pyramid), database(database), query(query), currentBlock(new MultiBlock()) {
    currentBlock->single = HashBlock(0, 0);  // :20
  }

  const HashBlock* getNextInterestingBlock() {  // :27-50
    if (!currentBlock) return nullptr;
    while (true) {
      const HashBlock* result = getNextBlockWithGoodNumberOfMatches();
      if (!result) return nullptr;
      if (recentlySeen(result)) continue;
      // "previousBlock" is never assigned in the reference => the overlap skip at :43 is dead code
      return result;
    }
  }
  bool recentlySeen(const HashBlock* block) {  // :52-65
    bool result = false;
    if (previousInterestingBlock && block->getForwardHash() == previousInterestingBlock->getForwardHash()) result = true;
    else if (previousPreviousInterestingBlock && block->getForwardHash() == previousPreviousInterestingBlock->getForwardHash()) result = true;
    previousPreviousInterestingBlock = previousInterestingBlock;
    previousInterestingBlock = block;
    return result;
  }
  const HashBlock* getNextBlockWithGoodNumberOfMatches() {  // :68-96
    while (true) {
      const HashBlock* next = advanceToNextPosition();
      if (!next) return nullptr;
      const HashBlock* extended = withGap();
      if (!extended) continue;
      if (!hasFewEnoughMatches(*extended)) continue;
      return extended;
    }
  }
  void moveDown() {  // :99-108
    batchIndex--;
    currentBlock = pyramid->get(batchIndex)->getAfter(currentBlock->getStartIndex());
    resetGapmer();
  }
  void moveUpOrRight() {  // :111-122
    const HashBlock* left = currentBlock->getSingle();
    MultiBlockP up = pyramid->get(batchIndex + 1)->get(left->getStartIndex());
    if (up && up->getStartIndex() <= left->getStartIndex()) {
      batchIndex++;
      currentBlock = up;
      resetGapmer();
    } else {
      moveRight();
    }
  }
  void moveRight() {  // :125-128
    currentBlock = pyramid->get(batchIndex)->getAfter(currentBlock->getStartIndex());
    resetGapmer();
  }
  void skipMultiblocks() {  // :130-140
    while (true) {
      if (!currentBlock || currentBlock->getSingle()) return;
      if (batchIndex > 0) moveDown(); else moveRight();
    }
  }
  const HashBlock* advanceToNextPosition() {  // :143-195
    const HashBlock* single = currentBlock->getSingle();
    if (HashBlock::getMaxGapmerNumBasepairsUsed(single->getLength()) < database->getMinInterestingSize() && database->getEnableGapmers()) {
      moveUpOrRight();
    } else {
      const HashBlock* extended = withGap();
      if (extended) {
        int numMatches = database->getNumMatchesLowerBound(*extended);
        if (numMatches < 6) {
          if (batchIndex > 0) moveDown(); else moveRight();
        } else {
          if (numMatches > getMaxNumMatchesAllowed(*extended)) moveUpOrRight();
          else moveRight();
        }
      } else {
        int typicalGapmerNumBasepairsUsed = single->getLength() * 3 / 2;
        if (typicalGapmerNumBasepairsUsed <= database->getMinInterestingSize() && database->getEnableGapmers()) {
          moveUpOrRight();
        } else {
          if (batchIndex > 0) moveDown(); else moveRight();
        }
      }
    }
    skipMultiblocks();
    if (!currentBlock) return nullptr;
    return currentBlock->getSingle();
  }
  void resetGapmer() { currentGapmer = nullptr; currentGapmerComputed = false; }
  const HashBlock* withGap() {  // :197-203
    const HashBlock* single = currentBlock->getSingle();
    if (!database->getEnableGapmers()) return single;
    if (!currentGapmerComputed) {
      // Java caches only non-null results (re-evaluating a null is idempotent)
      HashBlock g;
      int r = single->withGapAndExtension(*query, g);
      if (r == 0) currentGapmer = nullptr;
      else if (r == 1) currentGapmer = single;
      else { pool.push_back(g); currentGapmer = &pool.back(); }
      currentGapmerComputed = true;
    }
    return currentGapmer;
  }
  int getMaxNumMatchesAllowed(const HashBlock& block) {  // :205-219
    if (block.getLength() >= query->getLength() / 6) return database->getMaxNumMatchesAllowed(block);
    if (block.requestMergeRight) return 5;
    return block.getNumBasepairsUsed() + 1;
  }
  bool hasFewEnoughMatches(const HashBlock& block) { return database->getNumMatchesLowerBound(block) <= getMaxNumMatchesAllowed(block); }  // :221-223
};

// ---------------------------------------------------------------- HashBlockMatch_Counter (M/HashBlockMatch_Counter.java)
struct HashBlockMatch_Counter {
  int numMatches = 0, numDistinctMismatches, lastMismatchedPosition;
  const HashBlock* lastMatchedBlock = nullptr;
  SequenceMatchP match;
  const std::vector<const HashBlock*>* matchHistory;
  int historyProcessed_index;
  bool good = false;
  HashBlockMatch_Counter* nextCounter = nullptr;
  HashBlockMatch_Counter* previousCounter = nullptr;
  int priority = 0;
  HashBlockMatch_Counter(const SequenceMatchP& match, const std::vector<const HashBlock*>* matchHistory, int initialNumDistinctMismatches, int lastMismatchedPosition)
      : numDistinctMismatches(initialNumDistinctMismatches), lastMismatchedPosition(lastMismatchedPosition), match(match), matchHistory(matchHistory),
        historyProcessed_index((int)matchHistory->size() - 1) {}
  int getNumMatches() const { return numMatches; }
  int getNumDistinctMismatches() { update(); return numDistinctMismatches; }
  void addMatch(const HashBlock* block) { numMatches++; lastMatchedBlock = block; }
  void update() {  // :41-46
    while (historyProcessed_index < (int)matchHistory->size()) {
      update((*matchHistory)[(size_t)historyProcessed_index]);
      historyProcessed_index++;
    }
  }
  void setGood() { good = true; priority = getNumDistinctMismatches(); }
  bool isGood() const { return good; }
  int getPriority() const { return priority; }
  const SequenceMatchP& getMatch() const { return match; }
 private:
  void update(const HashBlock* block) {  // :74-88
    if (block != lastMatchedBlock) {
      int blockStart = block->getStartIndex();
      int blockEnd = block->getEndIndex();
      if (blockStart >= lastMismatchedPosition) {
        if (match->getOffset() + blockEnd <= match->getSequenceB()->getLength()) {
          numDistinctMismatches++;
          lastMismatchedPosition = blockEnd;
        }
      }
    }
  }
};
typedef std::shared_ptr<HashBlockMatch_Counter> CounterP;
typedef std::shared_ptr<std::vector<CounterP>> CounterListP;

// ---------------------------------------------------------------- Counting_HashBlockPath (M/Counting_HashBlockPath.java)
struct Counting_HashBlockPath {
  static constexpr int usualNumberOfMatchesRequiredBeforeInvestigating = 1;  // :18
  HashBlock_Pyramid pyramid;
  HashBlockPath path;
  Readable_HashBlock_Database* database;
  const SequenceDatabase* sequenceDatabase;
  const Sequence* query;
  const Sequence* reverseComplementQuery;
  Counters* counters;
  // [parity unpinned] the reference keys these by Sequence in a HashMap; iteration here is by contig index
  std::map<int, std::map<int, CounterP>> forwardMatchCounters, reverseMatchCounters;
  CounterListP goodMatchCounters;
  bool foundGoodMatchCounter = false;
  std::vector<const HashBlock*> interestingMatch_history;
  int numBlocksMatchingAnywhere = 0, numMatchCounters = 0;
  int maxNonoverlappingBlockVisited = 0, numNonoverlappingBlocksVisited = 0;
  int minNumDistinctMismatches = -1;
  bool done = false;
  int maxIndelLengthToConsider;
  std::deque<const HashBlock*> pendingBlocks;
  CounterListP previousHighPriorityMatchCounters, previousAllPositions;

  Counting_HashBlockPath(Readable_HashBlock_Database* database, const SequenceDatabase* sequenceDatabase, const Sequence* query,
                         const Sequence* reverseComplementQuery, const AlignmentParameters& p, Counters* counters)  // :20-37
      : pyramid(query, false, nullptr), path(&pyramid, database, query), database(database), sequenceDatabase(sequenceDatabase), query(query),
        reverseComplementQuery(reverseComplementQuery), counters(counters), goodMatchCounters(new std::vector<CounterP>()) {
    int maxPossibleIndel = j2i((query->getLength() * p.MaxErrorRate - p.DeletionStart_Penalty) / p.DeletionExtension_Penalty);
    maxIndelLengthToConsider = maxPossibleIndel / 2;
  }
  const Sequence* getQuerySequence() const { return query; }
  int getNumBlocks() const { return numBlocksMatchingAnywhere; }

  bool step() {  // :40-179
    if (done) return false;
    const HashBlock* queryBlock = nullptr;
    std::vector<SequencePosition> matches;
    if (!getNextInterestingMatch(queryBlock, matches)) {
      done = true;
      if (numBlocksMatchingAnywhere < usualNumberOfMatchesRequiredBeforeInvestigating) tryEnsureGoodMatchCounter();
      return false;
    }
    interestingMatch_history.push_back(queryBlock);
    int queryBlockNumMatches = (int)matches.size();
    for (const SequencePosition& referenceBlock : matches) {
      const Sequence* currentMatchedSequence = referenceBlock.sequence;
      int numMismatchedItems = 0, numMatchedItems = 0;
      if (counters) counters->flankChecks++;
      for (int distance = 1; distance < 20; distance++) {
        int checkOffset = -distance;
        int queryIndex = queryBlock->getStartIndex() + checkOffset;
        if (queryIndex >= 0 && queryIndex < query->getLength()) {
          int referenceIndex = referenceBlock.startIndex + checkOffset;
          if (referenceIndex >= 0 && referenceIndex < currentMatchedSequence->getLength()) {
            if (!Basepairs::canMatch(query->encodedCharAt(queryIndex), currentMatchedSequence->encodedCharAt(referenceIndex))) numMismatchedItems++;
            else numMatchedItems++;
          }
        }
        checkOffset = queryBlock->getLength() - 1 + distance;
        queryIndex = queryBlock->getStartIndex() + checkOffset;
        if (queryIndex >= 0 && queryIndex < query->getLength()) {
          int referenceIndex = referenceBlock.startIndex + checkOffset;
          if (referenceIndex >= 0 && referenceIndex < currentMatchedSequence->getLength()) {
            if (!Basepairs::canMatch(query->encodedCharAt(queryIndex), currentMatchedSequence->encodedCharAt(referenceIndex))) numMismatchedItems++;
            else numMatchedItems++;
          }
        }
        if (numMatchedItems < numMismatchedItems) break;
        if (numMatchedItems >= numMismatchedItems + queryBlock->getNumBasepairsUsed()) break;
      }
      if (numMismatchedItems > numMatchedItems) continue;
      SequenceMatchP fullMatch;
      if (currentMatchedSequence->getComplementedFrom() != nullptr) {
        const Sequence* forwardRef = currentMatchedSequence->getComplementedFrom();
        int reverseQueryBlockStart = query->getLength() - queryBlock->getEndIndex();
        int reverseReferenceBlockStart = currentMatchedSequence->getLength() - (referenceBlock.startIndex + queryBlock->getLength());
        int reverseLocalOffset = reverseReferenceBlockStart - reverseQueryBlockStart;
        fullMatch.reset(new SequenceMatch(reverseComplementQuery, forwardRef, reverseLocalOffset));
      } else {
        int currentLocalOffset = referenceBlock.startIndex - queryBlock->getStartIndex();
        fullMatch.reset(new SequenceMatch(query, currentMatchedSequence, currentLocalOffset));
      }
      updateMatches(fullMatch, queryBlock, queryBlockNumMatches);
    }
    if (queryBlock->getStartIndex() >= maxNonoverlappingBlockVisited) {
      maxNonoverlappingBlockVisited = queryBlock->getEndIndex();
      numNonoverlappingBlocksVisited++;
    }
    numBlocksMatchingAnywhere++;
    minNumDistinctMismatches = -1;
    return true;
  }

  void updateMatches(const SequenceMatchP& sequenceMatch, const HashBlock* queryBlock, int queryBlockNumMatches) {  // :193-252
    int offset = sequenceMatch->getOffset();
    auto& allMatchCounters = sequenceMatch->getReversed() ? forwardMatchCounters : reverseMatchCounters;  // (sic) :197-200
    std::map<int, CounterP>& matchesOnSequence = allMatchCounters[sequenceMatch->getSequenceB()->contigIndex];
    CounterP currentCounter;
    auto found = matchesOnSequence.find(offset);
    if (found != matchesOnSequence.end()) currentCounter = found->second;
    if (!currentCounter) {
      currentCounter.reset(new HashBlockMatch_Counter(sequenceMatch, &interestingMatch_history, numNonoverlappingBlocksVisited, queryBlock->getStartIndex()));
      auto it = matchesOnSequence.insert(std::make_pair(offset, currentCounter)).first;
      numMatchCounters++;
      if (it != matchesOnSequence.begin()) {  // lowerEntry
        auto prev = it; --prev;
        if (std::abs(prev->first - offset) <= maxIndelLengthToConsider) {
          currentCounter->previousCounter = prev->second.get();
          prev->second->nextCounter = currentCounter.get();
        }
      }
      auto next = it; ++next;  // higherEntry
      if (next != matchesOnSequence.end()) {
        if (std::abs(next->first - offset) <= maxIndelLengthToConsider) {
          currentCounter->nextCounter = next->second.get();
          next->second->previousCounter = currentCounter.get();
        }
      }
    }
    HashBlockMatch_Counter* previousCounter = currentCounter->previousCounter;
    if (previousCounter) addMatch(*sequenceMatch, queryBlock, findShared(previousCounter), queryBlockNumMatches);
    HashBlockMatch_Counter* nextCounter = currentCounter->nextCounter;
    if (nextCounter) addMatch(*sequenceMatch, queryBlock, findShared(nextCounter), queryBlockNumMatches);
    bool updateThisOne = true;
    if ((previousCounter && previousCounter->isGood()) || (nextCounter && nextCounter->isGood())) {
      if (!currentCounter->isGood()) updateThisOne = false;
    }
    if (updateThisOne) addMatch(*sequenceMatch, queryBlock, currentCounter, queryBlockNumMatches);
  }
  CounterP findShared(HashBlockMatch_Counter* raw) {
    bool rev = raw->match->getReversed();
    auto& m = (rev ? forwardMatchCounters : reverseMatchCounters)[raw->match->getSequenceB()->contigIndex];
    return m[raw->match->getOffset()];
  }

  void addMatch(const SequenceMatch& fullMatch, const HashBlock* queryBlock, const CounterP& counter, int queryBlockNumMatches) {  // :254-277
    counter->addMatch(queryBlock);
    counter->update();
    if (counter->getNumMatches() <= usualNumberOfMatchesRequiredBeforeInvestigating) {
      if (counter->getNumMatches() == usualNumberOfMatchesRequiredBeforeInvestigating) {
        foundGoodMatchCounter = true;
        declareGood(counter);
      } else {
        if (queryBlockNumMatches <= queryBlock->getLength()) {
          int distanceFromStart = fullMatch.getOffset();
          int distanceFromEnd = fullMatch.getSequenceB()->getLength() - (fullMatch.getOffset() + fullMatch.getSequenceA()->getLength());
          if (std::min(distanceFromStart, distanceFromEnd) < 0) declareGood(counter);
        }
      }
    }
  }
  void declareGood(const CounterP& counter) {  // :280-285
    if (!counter->isGood()) {
      // goodMatchCounters is handed out by reference to callers that compare sizes; copy-on-write keeps
      // previously returned lists stable like the Java code (which never hands out this list itself)
      goodMatchCounters->push_back(counter);
      counter->setGood();
    }
  }
  void tryEnsureGoodMatchCounter() {  // :291-308
    if (!foundGoodMatchCounter && numMatchCounters <= query->getLength()) {
      for (auto& s : forwardMatchCounters) for (auto& e : s.second) declareGood(e.second);
      for (auto& s : reverseMatchCounters) for (auto& e : s.second) declareGood(e.second);
      foundGoodMatchCounter = true;
    }
  }
  const HashBlock* getNextInterestingBlock() {  // :344-368
    previousAllPositions.reset();
    while (true) {
      const HashBlock* block = path.getNextInterestingBlock();
      if (!block) {
        if (pendingBlocks.empty()) return nullptr;
        const HashBlock* b = pendingBlocks.front();
        pendingBlocks.pop_front();
        return b;
      }
      bool overlap = block->getStartIndex() < maxNonoverlappingBlockVisited;
      if (overlap) { pendingBlocks.push_back(block); continue; }
      return block;
    }
  }
  bool getNextInterestingMatch(const HashBlock*& blockOut, std::vector<SequencePosition>& matches) {  // :371-384
    while (true) {
      const HashBlock* block = getNextInterestingBlock();
      if (!block) return false;
      if (!database->matchBlock(*block, matches)) continue;
      blockOut = block;
      return true;
    }
  }
  CounterListP findGoodPositionsHavingPriorityUpTo(int priority) {  // :406-433
    while (true) {
      if (numNonoverlappingBlocksVisited >= jadd(priority, usualNumberOfMatchesRequiredBeforeInvestigating)) break;
      if (!step()) break;
    }
    if (previousHighPriorityMatchCounters && previousHighPriorityMatchCounters->size() == goodMatchCounters->size()) return previousHighPriorityMatchCounters;
    CounterListP matches(new std::vector<CounterP>());
    for (auto& counter : *goodMatchCounters) if (counter->getPriority() <= priority) matches->push_back(counter);
    previousHighPriorityMatchCounters = matches;
    return matches;
  }
  CounterListP getAllPositions() {  // :435-451
    if (!previousAllPositions) {
      CounterListP results(new std::vector<CounterP>());
      for (auto& s : forwardMatchCounters) for (auto& e : s.second) results->push_back(e.second);
      for (auto& s : reverseMatchCounters) for (auto& e : s.second) results->push_back(e.second);
      previousAllPositions = results;
    }
    return previousAllPositions;
  }
  int getNumGoodDistinctMismatches() {  // :457-469
    if (minNumDistinctMismatches < 0) {
      int min = numNonoverlappingBlocksVisited - 1;
      for (auto& counter : *goodMatchCounters) {
        int count = counter->getNumDistinctMismatches();
        if (min >= count) min = count;
      }
      minNumDistinctMismatches = min;
    }
    return minNumDistinctMismatches;
  }
  CounterListP getBestMatches() {  // :471-493
    CounterListP best(new std::vector<CounterP>());
    if (numBlocksMatchingAnywhere < usualNumberOfMatchesRequiredBeforeInvestigating) return best;
    int min = getNumGoodDistinctMismatches();
    for (auto& counter : *goodMatchCounters) {
      int count = counter->getNumDistinctMismatches();
      if (count <= min) best->push_back(counter);
    }
    return best;
  }
};

// ---------------------------------------------------------------- HashBlockPaths_Counter (M/HashBlockPaths_Counter.java)
struct HashBlockPaths_Counter {
  std::vector<Counting_HashBlockPath*> components;
  int maxOffsetBetweenComponents;
  QueryMatchListP previousAssembledMatches;
  std::vector<CounterListP> previousMatchComponents;
  bool havePrevious = false;
  bool foundNonemptyResult = false;

  HashBlockPaths_Counter(const std::vector<Counting_HashBlockPath*>& components, int expectedInnerDistance, int maxInnerDistanceBetweenComponents)  // :13-18
      : components(components) {
    (void)expectedInnerDistance;
    maxOffsetBetweenComponents = jadd(maxInnerDistanceBetweenComponents, components[0]->getQuerySequence()->getLength());
  }

  QueryMatchListP findGoodPositionsHavingPriority(int numMismatches) {  // :21-24
    QueryMatchListP allMatches = findGoodPositionsWithPriorityUpTo(numMismatches);
    return filterMatchesHavingPriority(*allMatches, numMismatches);
  }
  QueryMatchListP findPartiallyGoodPositions() {  // :26-49
    QueryMatchListP empty(new std::vector<QueryMatchP>());
    if (components.size() != 2) return empty;
    if (!foundNonemptyResult) return empty;
    std::vector<CounterListP> pieces;
    bool foundGoodPosition = false, foundBadPosition = false;
    for (Counting_HashBlockPath* component : components) {
      CounterListP matchesHere = component->findGoodPositionsHavingPriorityUpTo(INT32_MAX);
      if (matchesHere->empty()) { foundBadPosition = true; matchesHere = component->getAllPositions(); }
      else foundGoodPosition = true;
      pieces.push_back(matchesHere);
    }
    if (foundGoodPosition && foundBadPosition) return match(pieces);
    return empty;
  }
  QueryMatchListP findGoodPositionsWithPriorityUpTo(int numMismatches) {  // :51-81
    std::vector<CounterListP> pieces;
    for (Counting_HashBlockPath* component : components) {
      CounterListP matchesHere = component->findGoodPositionsHavingPriorityUpTo(numMismatches);
      if (!matchesHere->empty()) foundNonemptyResult = true;
      pieces.push_back(matchesHere);
    }
    return match(pieces);
  }
  QueryMatchListP optimisticGetBestMatches() {  // :84-98
    std::vector<CounterListP> pieces;
    for (Counting_HashBlockPath* component : components) {
      while (true) {
        CounterListP best = component->getBestMatches();
        if (best->size() == 1 || !component->step()) { pieces.push_back(best); break; }
      }
    }
    QueryMatchListP allMatches = match(pieces);
    return filterMatchesHavingMinPriority(*allMatches);
  }
  std::vector<SequenceMatchP> findGoodComponentMatches(int sequenceIndex, int maxPriority) {  // :102-106
    CounterListP componentMatches = components[(size_t)sequenceIndex]->findGoodPositionsHavingPriorityUpTo(maxPriority);
    std::vector<SequenceMatchP> r;
    for (auto& c : *componentMatches) r.push_back(c->getMatch());
    return r;
  }
  int getNumBlocks() const { int t = 0; for (auto c : components) t += c->getNumBlocks(); return t; }  // :108-114

  QueryMatchListP match(const std::vector<CounterListP>& comps) {  // :116-133 (cache keyed on list identity)
    bool same = havePrevious;
    if (same) for (size_t i = 0; i < comps.size(); i++) if (previousMatchComponents[i] != comps[i]) { same = false; break; }
    if (!same) {
      previousAssembledMatches = matchWithoutCache(comps);
      previousMatchComponents = comps;
      havePrevious = true;
    }
    return previousAssembledMatches;
  }

  QueryMatchListP matchWithoutCache(const std::vector<CounterListP>& comps) {  // :136-247
    if (comps.size() > 2) throw std::runtime_error("only 2 query ends supported");
    QueryMatchListP results(new std::vector<QueryMatchP>());
    if (comps.size() == 1) {
      for (auto& possibility : *comps[0]) results->push_back(QueryMatchP(new QueryMatch(possibility->getMatch(), possibility->getPriority())));
      return results;
    }
    // LinkedHashMap<Sequence, TreeMap<offset, counter>>: only per-sequence lookups are performed, never iteration
    std::map<const Sequence*, std::map<int, CounterP>> forwardMatchingComponents, reverseMatchingComponents;
    std::vector<std::vector<CounterP>> matchedCounters;
    bool lastComponentIsLargest = comps.size() <= 1 || comps[0]->size() <= comps[1]->size();
    for (size_t i = 0; i < comps.size(); i++) {
      size_t componentIndex = lastComponentIsLargest ? i : 1 - i;
      for (const CounterP& counter : *comps[componentIndex]) {
        const SequenceMatch& m = *counter->getMatch();
        const Sequence* referenceSequence = m.getSequenceB();
        int querySequenceLength = m.getSequenceA()->getLength();
        int maxReverseOffset = querySequenceLength / 2;
        bool sequenceMatchReversed = m.getReversed();
        bool queryMatchReversed = (sequenceMatchReversed == (componentIndex % 2 == 0));
        auto& matchingComponents = queryMatchReversed ? reverseMatchingComponents : forwardMatchingComponents;
        std::map<int, CounterP>& matchesOnThisSequence = matchingComponents[referenceSequence];
        int offset = m.getOffset();
        if (i == 0) {
          matchesOnThisSequence[offset] = counter;
        } else {
          int searchStart, searchEnd;
          bool otherSequenceExpectEarlier = (queryMatchReversed == lastComponentIsLargest);
          if (otherSequenceExpectEarlier) { searchStart = offset - maxReverseOffset; searchEnd = jadd(offset, maxOffsetBetweenComponents); }
          else { searchStart = offset - maxOffsetBetweenComponents; searchEnd = offset + maxReverseOffset; }
          std::vector<CounterP> nearby;
          if (searchStart <= searchEnd) {
            for (auto it = matchesOnThisSequence.lower_bound(searchStart); it != matchesOnThisSequence.end() && it->first <= searchEnd; ++it) nearby.push_back(it->second);
          } else {
            throw std::runtime_error("subMap: fromKey > toKey");  // TreeMap.subMap would throw IllegalArgumentException
          }
          if (queryMatchReversed && nearby.size() > 1) std::reverse(nearby.begin(), nearby.end());
          for (const CounterP& other : nearby) {
            std::vector<CounterP> matchingCounters;
            if (lastComponentIsLargest) { matchingCounters.push_back(other); matchingCounters.push_back(counter); }
            else { matchingCounters.push_back(counter); matchingCounters.push_back(other); }
            matchedCounters.push_back(matchingCounters);
          }
        }
      }
    }
    for (auto& group : matchedCounters) {  // assembleQueryMatches :249-265
      std::vector<SequenceMatchP> sequenceMatches;
      for (auto& c : group) sequenceMatches.push_back(c->getMatch());
      bool hintSearchForward = group.size() > 1 ? group[0]->getNumDistinctMismatches() < group[1]->getNumDistinctMismatches() : true;
      int numMismatches = countPriority(group);
      results->push_back(QueryMatchP(new QueryMatch(sequenceMatches, numMismatches, hintSearchForward)));
    }
    return results;
  }
  static QueryMatchListP filterMatchesHavingPriority(const std::vector<QueryMatchP>& matches, int numDistinctMismatches) {  // :267-294
    QueryMatchListP results(new std::vector<QueryMatchP>());
    for (auto& m : matches) if (m->getPriority() == numDistinctMismatches) results->push_back(m);
    return results;
  }
  static QueryMatchListP filterMatchesHavingMinPriority(const std::vector<QueryMatchP>& matches) {  // :296-304 (sic: selects the max)
    int min = -1;
    for (auto& m : matches) if (min < 0 || min < m->getPriority()) min = m->getPriority();
    return filterMatchesHavingPriority(matches, min);
  }
  static int countPriority(const std::vector<CounterP>& counters) {  // :314-334
    if (counters.size() == 2) {
      const SequenceMatch& match1 = *counters[0]->getMatch();
      const SequenceMatch& match2 = *counters[1]->getMatch();
      if (match1.getStartIndexB() < match2.getEndIndexB() && match1.getEndIndexB() > match2.getStartIndexB()) {
        int max = 0;
        for (auto& c : counters) max = std::max(max, c->getPriority());
        return max;
      }
    }
    int total = 0;
    for (auto& c : counters) total += c->getPriority();
    return total;
  }
};

}  // namespace xmo
