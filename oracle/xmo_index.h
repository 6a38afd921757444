// ORACLE — test infrastructure only (see xmo_types.h).
// Restates the seed index: M/PackedMap.java, M/HashBlock_Buffer.java, M/HashJob.java,
// M/HashBlock_Database.java, M/Readable_HashBlock_Database.java and the inferred ByteKeyStore
// (QuickVariants).  Single-threaded: the reference's cooperative helpers (helpLoad/helpHash/helpPack)
// collapse to plain loops; the on-disk cache (--cache-dir) is out of scope.
#pragma once
#include "xmo_hashblock.h"
#include <deque>

namespace xmo {

// ---------------------------------------------------------------- PackedMap (M/PackedMap.java)
// [inferred] ByteKeyStore(numKeys, maxBytesPerKey = encodedLength(maxCount), bits): a key holds at most
// maxInterestingCountPerKey positions; one more append => knowsAllMatches(key) = false.
// [inferred, parity unpinned] pack() orders a bucket by ascending encoded position.
struct PackedMap {
  int maxInterestingCountPerKey;
  int keyCapacity;
  const SequenceDatabase* sequenceDatabase;
  int id;
  int64_t numItemsAdded = 0;
  std::vector<std::vector<int64_t>> buckets;
  std::vector<uint8_t> overfull;

  PackedMap(int maxInterestingCountPerKey, int keyCapacity, const SequenceDatabase* db, int id)  // :19-32
      : maxInterestingCountPerKey(maxInterestingCountPerKey), sequenceDatabase(db), id(id) {
    if (keyCapacity < 1) keyCapacity = 1;
    int64_t maxArrayLength = INT32_MAX / 2;
    if ((int64_t)keyCapacity > maxArrayLength) keyCapacity = (int)maxArrayLength;
    this->keyCapacity = keyCapacity;
    buckets.resize((size_t)keyCapacity);
    overfull.assign((size_t)keyCapacity, 0);
  }
  int getCapacity() const { return keyCapacity; }
  int getMaxInterestingCountPerKey() const { return maxInterestingCountPerKey; }
  int getPackedKey(int32_t originalKey) const {  // :210-215
    int result = originalKey % keyCapacity;
    if (result < 0) result += keyCapacity;
    return result;
  }
  bool knowsAllMatches(int32_t key) const { return !overfull[(size_t)getPackedKey(key)]; }  // :174-180
  int getNumMatchesLowerBound(int32_t key) const {  // :228-236
    int packedKey = getPackedKey(key);
    if (overfull[(size_t)packedKey]) return INT32_MAX;
    return (int)buckets[(size_t)packedKey].size();
  }
  // :160-172.  returns false for Java null ("too many matches")
  bool get(int32_t key, int maxInterestingCount, std::vector<SequencePosition>& out) const {
    out.clear();
    int count = getNumMatchesLowerBound(key);
    if (count > maxInterestingCount || count > maxInterestingCountPerKey) return false;
    for (int64_t enc : buckets[(size_t)getPackedKey(key)]) out.push_back(sequenceDatabase->decodePosition(enc));
    return true;
  }
  void addOne(int32_t key, const Sequence* sequence, int startIndex, bool preventDuplicates) {  // :124-153
    bool duplicate = false;
    if (preventDuplicates) {
      std::vector<SequencePosition> existing;
      if (get(key, INT32_MAX, existing)) {
        for (auto& e : existing) if (e.sequence == sequence && e.startIndex == startIndex) { duplicate = true; break; }
      }
    }
    if (!duplicate) {
      size_t packedKey = (size_t)getPackedKey(key);
      if (!overfull[packedKey]) {
        if ((int)buckets[packedKey].size() >= maxInterestingCountPerKey) { overfull[packedKey] = 1; buckets[packedKey].clear(); buckets[packedKey].shrink_to_fit(); }
        else buckets[packedKey].push_back(sequenceDatabase->encodePosition(sequence, startIndex));
      }
    }
    numItemsAdded++;
  }
  void add(const Sequence* sequence, const std::vector<HashBlock>& blocks, bool preventDuplicates) {  // :54-122 (process)
    const Sequence* reverseSequence = sequenceDatabase->getReverseComplement(sequence);
    for (const HashBlock& block : blocks) {
      if (block.isPrimaryPolarity())
        addOne(block.getForwardHash(), sequence, block.getStartIndex(), preventDuplicates);
      if (block.isSecondaryPolarity())
        addOne(block.getReverseHash(), reverseSequence, reverseSequence->getLength() - block.getEndIndex(), preventDuplicates);
    }
  }
  void pack() { for (auto& b : buckets) std::sort(b.begin(), b.end()); }  // :239-243
};
typedef std::shared_ptr<PackedMap> PackedMapP;

struct HashJob { const Sequence* sequence; int minStartIndex, maxStartIndexExclusive; };  // M/HashJob.java

struct HashBlock_Database;

struct HashBlock_Buffer : BlockListener {  // M/HashBlock_Buffer.java
  HashJob section; HashBlock_Database* database; int minInterestingSize;
  std::vector<MultiBlockP> multiBlocks, singleBlocks;
  HashBlock_Buffer(const HashJob& s, HashBlock_Database* db, int minInterestingSize) : section(s), database(db), minInterestingSize(minInterestingSize) {}
  void addHashblock(const MultiBlockP& block) override;
  void flush();
};

// ---------------------------------------------------------------- HashBlock_Database (M/HashBlock_Database.java)
struct HashBlock_Database {
  const SequenceDatabase* sequenceDatabase;
  std::vector<PackedMapP> hashedBlocks;
  int maxFullySetUpSize = 0, maxInterestingSize = 0, minInterestingSize = 0;
  int maxNumShortMatches = 5;
  int64_t totalForwardSize = 0;
  bool enableGapmers = true;
  bool hashInReverseOrder = false;
  std::deque<HashJob> sectionsLeftToHash;

  static int chooseMinDuplicationLength(const SequenceDatabase& db) { return SequenceDatabase::log2RoundUp(db.getTotalForwardSize()); }  // M/DuplicationDetector.java:17-31
  static int chooseMaxDuplicationLength(const SequenceDatabase& db) { return chooseMinDuplicationLength(db) * 2; }                      // :34-36

  HashBlock_Database(const SequenceDatabase* sequences, int minInterestingSize = -1, int hintMaxInterestingSize = -1,
                     int maxNumShortMatches = -1, bool enableGapmers = true, bool hashInReverseOrder = false) {  // :41-91
    this->enableGapmers = enableGapmers;
    this->sequenceDatabase = sequences;
    this->totalForwardSize = sequences->getTotalForwardSize();
    if (minInterestingSize <= 0)
      this->minInterestingSize = j2i(std::max((std::log((double)(totalForwardSize + 1)) / std::log(4.0)) - 2, 1.0));
    else
      this->minInterestingSize = minInterestingSize;
    if (hintMaxInterestingSize > 0) {
      if (sequences->getTotalForwardSize() > 1000000000LL) this->maxInterestingSize = (hintMaxInterestingSize + 1) / 2;
      else this->maxInterestingSize = hintMaxInterestingSize;
    } else {
      this->maxInterestingSize = -1;
    }
    this->maxNumShortMatches = maxNumShortMatches < 0 ? 5 : maxNumShortMatches;
    this->hashInReverseOrder = hashInReverseOrder;
    chooseNextHashSize(0);
  }

  int getMinInterestingSize() const { return minInterestingSize; }
  bool getEnableGapmers() const { return enableGapmers; }

  void requireSetUpThroughSize(int size) {  // :148-173
    while (true) {
      if (maxFullySetUpSize >= size) return;
      if (maxFullySetUpSize >= maxInterestingSize) chooseNextHashSize(size);
      helpSetUp();
    }
  }

  void chooseNextHashSize(int requestSize) {  // :183-215
    if (maxFullySetUpSize < 1) {
      if (maxInterestingSize < 0) {
        int initialSize = chooseMaxDuplicationLength(*sequenceDatabase);
        maxInterestingSize = std::max(initialSize, requestSize);
      }
    } else {
      maxInterestingSize = requestSize * 2;
    }
    split_hashJobs();
  }

  void split_hashJobs() {  // :218-235
    int targetJobSize = 50000;
    sectionsLeftToHash.clear();
    std::vector<const Sequence*> sequences;
    for (int c = 0; c < sequenceDatabase->numContigs(); c++) sequences.push_back(sequenceDatabase->forward(c));
    if (hashInReverseOrder) std::reverse(sequences.begin(), sequences.end());
    for (const Sequence* sequence : sequences) {
      int numJobsForThisSequence = (sequence->getLength() + targetJobSize - 1) / targetJobSize;
      int previousStartIndex = 0;
      for (int i = 1; i <= numJobsForThisSequence; i++) {
        int startIndex = (int)((int64_t)sequence->getLength() * (int64_t)i / (int64_t)numJobsForThisSequence);
        sectionsLeftToHash.push_back(HashJob{sequence, previousStartIndex, startIndex});
        previousStartIndex = startIndex;
      }
    }
  }

  void helpSetUp() {  // :238-242 + helpHash :337-403 + helpPack :405-458
    int size = maxInterestingSize;
    bool hashedAny = !sectionsLeftToHash.empty();
    while (!sectionsLeftToHash.empty()) {
      HashJob job = sectionsLeftToHash.front();
      sectionsLeftToHash.pop_front();
      hashSequenceThroughSize(job, size);
    }
    if (hashedAny || maxFullySetUpSize < maxInterestingSize) {
      while (size >= (int)hashedBlocks.size()) hashedBlocks.push_back(nullptr);
      for (int i = 0; i <= size; i++) {
        if (!hashedBlocks[(size_t)i]) hashedBlocks[(size_t)i].reset(new PackedMap(1, 1, sequenceDatabase, i));  // :387-393
      }
      for (int i = maxFullySetUpSize + 1; i <= size; i++) hashedBlocks[(size_t)i]->pack();
      maxFullySetUpSize = maxInterestingSize;
    }
  }

  void hashSequenceThroughSize(const HashJob& section, int size) {  // :490-528
    HashBlock_Buffer buffer(section, this, minInterestingSize);
    HashBlock_Pyramid pyramid(section.sequence, true, &buffer);
    int startIndex = section.minStartIndex;
    int maxStartIndex = section.maxStartIndexExclusive;
    int level = 0;
    int offset = startIndex - 1;
    while (true) {
      HashBlock_Row* batch = pyramid.get(level);
      batch->skipTo(startIndex - 1);
      MultiBlockP block = batch->getAfter(offset);
      if (!block || block->getStartIndex() > maxStartIndex) break;
      if (block->getMinLength() <= size) {
        level++;
        offset = block->getStartIndex() - 1;
      } else {
        level--;
        offset = block->getStartIndex();
        batch->garbageCollect(block->getStartIndex());
      }
    }
    buffer.flush();
  }

  void addToBlocksBySize(const HashBlock& blockIn, const Sequence* sequence, std::map<int, std::vector<HashBlock>>& blocksBySize) {  // :594-616
    HashBlock block = blockIn;
    if (enableGapmers) {
      HashBlock g;
      int r = blockIn.withGapAndExtension(*sequence, g);
      if (r == 0) return;
      if (r == 2) block = g;
    }
    int length = block.getNumBasepairsUsed();
    if (length < minInterestingSize) return;
    if (length <= maxFullySetUpSize) return;
    if (length > maxInterestingSize) return;
    blocksBySize[length].push_back(block);
  }

  void addHashblocks(const Sequence* sequence, const std::vector<MultiBlockP>& blocks) {  // :530-590
    bool containsAmbiguousPosition = false;
    std::map<int, std::vector<HashBlock>> blocksBySize;
    for (const MultiBlockP& multiblock : blocks) {
      const HashBlock* block = multiblock->getSingle();
      if (block) {
        addToBlocksBySize(*block, sequence, blocksBySize);
      } else {
        containsAmbiguousPosition = true;
        for (const ConditionalHashBlock& possibility : multiblock->possibilities)
          if (possibility.hasBlock) addToBlocksBySize(possibility.block, sequence, blocksBySize);
      }
    }
    for (auto& entry : blocksBySize) {
      int numBasepairsUsed = entry.first;
      while ((int)hashedBlocks.size() <= numBasepairsUsed) hashedBlocks.push_back(nullptr);
      PackedMapP& blocksOfThisSize = hashedBlocks[(size_t)numBasepairsUsed];
      if (!blocksOfThisSize) {
        int estimatedCapacity = estimateRequiredCapacity(numBasepairsUsed);
        int maxNumInterestingMatches = numBasepairsUsed * numBasepairsUsed;
        if (maxNumInterestingMatches < maxNumShortMatches) maxNumInterestingMatches = maxNumShortMatches;
        if (maxNumInterestingMatches > 32766) maxNumInterestingMatches = 32766;
        if (maxNumInterestingMatches < 1) maxNumInterestingMatches = 1;
        blocksOfThisSize.reset(new PackedMap(maxNumInterestingMatches, estimatedCapacity, sequenceDatabase, numBasepairsUsed));
      }
      blocksOfThisSize->add(sequence, entry.second, containsAmbiguousPosition);
    }
  }

  int estimateRequiredCapacity(int numPositionsPerBlock) const {  // :620-665
    int anchorBlockSize = enableGapmers ? numPositionsPerBlock * 2 / 3 : numPositionsPerBlock;
    double sizeProbability = std::min(1.0, 2.0 / anchorBlockSize);
    double offsetProbability = std::min(1.0, 2.0 / anchorBlockSize);
    double blockPossibilityProbability = sizeProbability * offsetProbability;
    int64_t maxNumSequencesOfThisLength = (numPositionsPerBlock <= 16) ? ((int64_t)1 << (numPositionsPerBlock * 2)) : ((int64_t)1 << 32);
    int64_t maxNumStoredSequencesOfThisLength = maxNumSequencesOfThisLength / 2;
    int64_t maxNumExistentHashcodes = j2l((double)maxNumStoredSequencesOfThisLength * blockPossibilityProbability);
    int64_t effectiveSize = totalForwardSize;
    int64_t numBlocksOfThisSize = j2l((double)effectiveSize * blockPossibilityProbability);
    double existenceFraction = 1 - std::pow((double)((double)maxNumExistentHashcodes - 1.0) / (double)maxNumExistentHashcodes, (double)numBlocksOfThisSize);
    int uniqueCount = j2i((double)maxNumExistentHashcodes * existenceFraction);
    int result = uniqueCount;
    if (result % 2 == 0) result++;
    return result;
  }
};

inline void HashBlock_Buffer::addHashblock(const MultiBlockP& block) {  // M/HashBlock_Buffer.java:14-33
  int startIndex = block->getStartIndex();
  if (startIndex < section.minStartIndex) return;
  if (startIndex >= section.maxStartIndexExclusive) return;
  if (!block->getSingle()) {
    multiBlocks.push_back(block);
    if (multiBlocks.size() >= 65536) flush();
  } else {
    singleBlocks.push_back(block);
    if (singleBlocks.size() >= 8096) flush();
  }
}
inline void HashBlock_Buffer::flush() {  // :35-40
  database->addHashblocks(section.sequence, multiBlocks);
  multiBlocks.clear();
  database->addHashblocks(section.sequence, singleBlocks);
  singleBlocks.clear();
}

// ---------------------------------------------------------------- Readable_HashBlock_Database (M/Readable_HashBlock_Database.java)
struct Readable_HashBlock_Database {
  HashBlock_Database* database;
  int minInterestingSize;
  int maxHashedLength = -1;
  Counters* counters = nullptr;
  explicit Readable_HashBlock_Database(HashBlock_Database* db) : database(db), minInterestingSize(db->getMinInterestingSize()) {}

  PackedMap* getContainingMap(int length) {  // :108-113
    if (maxHashedLength < length) {
      database->requireSetUpThroughSize(length);
      maxHashedLength = database->maxFullySetUpSize;
    }
    return database->hashedBlocks[(size_t)length].get();
  }
  SequencePosition reverseComplement(const SequencePosition& position, int blockLength) const {  // :55-59
    const Sequence* rc = database->sequenceDatabase->getReverseComplement(position.sequence);
    return SequencePosition{rc, rc->getLength() - position.startIndex - blockLength};
  }
  // :22-38.  returns false for Java null
  bool matchBlock(const HashBlock& block, std::vector<SequencePosition>& results) {
    results.clear();
    if (block.getNumBasepairsUsed() < minInterestingSize) return false;
    PackedMap* m = getContainingMap(block.getNumBasepairsUsed());
    if (!m) return true;
    int32_t key2 = block.getLookupKey();
    bool invert = !block.isPrimaryPolarity();
    if (counters) counters->bucketFetches++;
    if (!m->get(key2, INT32_MAX, results)) return false;
    if (counters) counters->hitsFetched += (int64_t)results.size();
    if (invert) for (auto& r : results) r = reverseComplement(r, block.getLength());
    return true;
  }
  bool lookupByForwardHash(int blockLength, int32_t hashKey, std::vector<SequencePosition>& allMatches) {  // :41-52
    PackedMap* m = getContainingMap(blockLength);
    std::vector<SequencePosition> forwardMatches;
    allMatches.clear();
    if (!m->get(hashKey, INT32_MAX, forwardMatches)) return false;
    allMatches.resize(forwardMatches.size() * 2);
    for (size_t i = 0; i < forwardMatches.size(); i++) {
      allMatches[i] = forwardMatches[i];
      allMatches[i + forwardMatches.size()] = reverseComplement(forwardMatches[i], blockLength);
    }
    return true;
  }
  void ensureHashed(int blockLength) { getContainingMap(blockLength); }
  int getNumHashKeys(int blockLength) { PackedMap* m = getContainingMap(blockLength); return m ? m->getCapacity() : 0; }
  int getNumMatchesLowerBound(const HashBlock& block) {  // :72-80
    if (block.getNumBasepairsUsed() < minInterestingSize) return INT32_MAX;
    PackedMap* m = getContainingMap(block.getNumBasepairsUsed());
    if (!m) return INT32_MAX;
    if (counters) counters->headerProbes++;
    return m->getNumMatchesLowerBound(block.getLookupKey());
  }
  int getMaxNumMatchesAllowed(const HashBlock& block) {  // :82-90
    if (block.getNumBasepairsUsed() < minInterestingSize) return -1;
    PackedMap* m = getContainingMap(block.getNumBasepairsUsed());
    if (!m) return 0;
    return m->getMaxInterestingCountPerKey();
  }
  void prepare() { database->requireSetUpThroughSize(1); maxHashedLength = database->maxFullySetUpSize; }  // :127-130
  int getHashedLength() const { return maxHashedLength; }
  bool getEnableGapmers() const { return database->getEnableGapmers(); }
  int getMinInterestingSize() const { return minInterestingSize; }
};

}  // namespace xmo
