// ORACLE — test infrastructure only (see xmo_types.h).
// Restates M/DuplicationDetector.java, M/Readable_DuplicationDetector.java, M/Duplication.java
// (reference-only precompute that alignToAncestralReference consults once per read).
#pragma once
#include "xmo_index.h"
#include <unordered_map>
#include <set>

namespace xmo {

struct Duplication {  // M/Duplication.java
  int length;
  std::vector<SequencePosition> startPositions;
  explicit Duplication(int length) : length(length) {}
  void addPosition(const SequencePosition& p) { startPositions.push_back(p); }
  void removeDuplicatePositions() {  // :23-25 (HashSet => set semantics; order unobservable)
    std::vector<SequencePosition> out;
    for (auto& p : startPositions) {
      bool dup = false;
      for (auto& q : out) if (q.sequence == p.sequence && q.startIndex == p.startIndex) { dup = true; break; }
      if (!dup) out.push_back(p);
    }
    startPositions.swap(out);
  }
  int getLength() const { return length; }
  int getNumInstances() const { return (int)startPositions.size(); }
};
typedef std::shared_ptr<Duplication> DuplicationP;

struct DuplicationDetector {
  HashBlock_Database* hashblockDatabase;
  bool enableGapmers;
  int minSizeToProcess, maxSizeToProcess, minNumInterestingCopies, windowSize;
  bool detected = false;
  std::map<const Sequence*, std::map<int, DuplicationP>> duplicationsBySequence;

  DuplicationDetector(HashBlock_Database* db, int minDuplicationLength, int maxDuplicationLength, int minNumInterestingCopies, int windowSize)  // :38-53
      : hashblockDatabase(db), enableGapmers(db->getEnableGapmers()), minSizeToProcess(minDuplicationLength), maxSizeToProcess(maxDuplicationLength),
        minNumInterestingCopies(minNumInterestingCopies), windowSize(windowSize) {}

  double getDetectionGranularity() const {  // :67-77  (integer arithmetic before widening, as written)
    if (enableGapmers) return (double)(minSizeToProcess * 5 / 8);
    return (double)minSizeToProcess;
  }
  int getWindowNumber(int index) const { return index / windowSize; }  // :438-440

  void detect() {  // :97-104
    if (detected) return;
    for (int size = minSizeToProcess; size <= maxSizeToProcess; size++) process(size);
    detected = true;
  }

  static bool isAmbiguousText(const std::string& t) {
    for (char c : t) if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return true;
    return false;
  }

  void process(int blockLength) {  // :129-250 (cache paths omitted)
    Readable_HashBlock_Database readable(hashblockDatabase);
    readable.ensureHashed(minSizeToProcess + 1);  // :118-123
    int numBlocks = readable.getNumHashKeys(blockLength);
    std::map<const Sequence*, std::map<int, DuplicationP>> blocks;
    std::vector<SequencePosition> matches;
    for (int hashcode = 0; hashcode < numBlocks; hashcode++) {
      if (readable.lookupByForwardHash(blockLength, hashcode, matches)) {
        int numForwardMatches = (int)matches.size() / 2;
        if (numForwardMatches >= minNumInterestingCopies) {
          std::map<std::string, DuplicationP> positionsByText;
          for (size_t i = 0; i < matches.size(); i++) {
            const SequencePosition& position = matches[i];
            int prefixLength = (blockLength + 3) / 4;
            std::string prefix = position.sequence->getRange(position.startIndex, prefixLength);
            std::string suffix = position.sequence->getRange(position.startIndex + blockLength - prefixLength, prefixLength);
            std::string text = prefix + suffix;
            if (!isAmbiguousText(text)) {
              DuplicationP& mp = positionsByText[text];
              if (!mp) mp.reset(new Duplication(blockLength));
              mp->addPosition(position);
            }
          }
          for (auto& e : positionsByText) e.second->removeDuplicatePositions();
          for (auto& e : positionsByText) groupDuplicationBySequence(e.second, blocks);
        }
      }
      if (hashcode % 10000 == 9999 || hashcode == numBlocks - 1) {
        saveDuplications(blocks);
        blocks.clear();
      }
    }
  }

  void groupDuplicationBySequence(const DuplicationP& group, std::map<const Sequence*, std::map<int, DuplicationP>>& blocks) {  // :252-269
    if (group->getNumInstances() >= minNumInterestingCopies) {
      for (const SequencePosition& position : group->startPositions) blocks[position.sequence][position.startIndex] = group;
    }
  }

  void saveDuplications(std::map<const Sequence*, std::map<int, DuplicationP>>& blocks) {  // :332-400
    for (auto& entry : blocks) {
      std::map<int, DuplicationP>& all = duplicationsBySequence[entry.first];
      for (auto& positions : entry.second) {
        int duplicationStart = positions.first;
        const DuplicationP& newDuplication = positions.second;
        bool insert = true;
        while (true) {
          auto it = all.upper_bound(duplicationStart);  // floorEntry
          if (it != all.begin()) {
            --it;
            int comparison = compareDuplications(duplicationStart, *newDuplication, it->first, *it->second);
            if (comparison > 0) { insert = false; break; }
            if (comparison < 0) { all.erase(it); continue; }
          }
          break;
        }
        while (true) {
          auto it = all.lower_bound(duplicationStart);  // ceilingEntry
          if (it != all.end()) {
            int comparison = compareDuplications(duplicationStart, *newDuplication, it->first, *it->second);
            if (comparison > 0) { insert = false; break; }
            if (comparison < 0) { all.erase(it); continue; }
          }
          break;
        }
        if (insert) all[duplicationStart] = newDuplication;
      }
    }
  }

  int compareDuplications(int start1, const Duplication& d1, int start2, const Duplication& d2) const {  // :406-436
    if (windowSize > 1) {
      if (getWindowNumber(start1) != getWindowNumber(start2)) return 0;
    }
    int end1 = start1 + d1.getLength();
    int end2 = start2 + d2.getLength();
    if (start1 <= start2 && end1 >= end2) return 1;
    if (start1 >= start2 && end1 <= end2) return -1;
    if (windowSize > 1) {
      int countDifference = d1.getNumInstances() - d2.getNumInstances();
      if (countDifference != 0) return countDifference;
      if (start1 != start2) return start1 - start2;
    }
    return 0;
  }

  // M/Readable_DuplicationDetector.java:28-47.  returns -1 for Java null, else the key found
  bool mayContainDuplicationInRange(const Sequence* sequence, int startIndex, int endIndex) {
    detect();
    int windowStart = getWindowNumber(startIndex);
    int windowEnd = getWindowNumber(endIndex);
    auto sit = duplicationsBySequence.find(sequence);
    if (sit == duplicationsBySequence.end()) return false;
    const std::map<int, DuplicationP>& entriesHere = sit->second;
    auto it = entriesHere.upper_bound(endIndex);  // floorEntry(endIndex)
    if (it != entriesHere.begin()) {
      auto prev = it; --prev;
      int previousWindow = getWindowNumber(prev->first);
      if (previousWindow >= windowStart && previousWindow <= windowEnd) return true;
    }
    auto next = entriesHere.lower_bound(startIndex);  // ceilingEntry(startIndex)
    if (next != entriesHere.end()) {
      int nextWindow = getWindowNumber(next->first);
      if (nextWindow >= windowStart && nextWindow <= windowEnd) return true;
    }
    return false;
  }

  // sorted duplication start keys of one forward contig (the only thing a read ever consults)
  std::vector<int> keysOnSequence(const Sequence* sequence) {
    detect();
    std::vector<int> keys;
    auto sit = duplicationsBySequence.find(sequence);
    if (sit != duplicationsBySequence.end()) for (auto& e : sit->second) keys.push_back(e.first);
    return keys;
  }
};

}  // namespace xmo
