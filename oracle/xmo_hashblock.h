// ORACLE — test infrastructure only (see xmo_types.h).
// Restates the content-defined HashBlock pyramid:
//   M/HashBlock.java, M/Gapped_HashBlock.java, M/IMultiHashBlock.java, M/MultiHashBlock.java,
//   M/ConditionalHashBlock.java, M/SequenceCondition.java, M/HashBlock_Row.java, M/HashBlock_BaseRow.java,
//   M/HashBlock_ParentRow.java, M/HashBlock_Stream.java, M/HashBlock_Pyramid.java
// (M/ = /root/reference/src/main/java/mapper/).  HashBlock_Compiler is a memoiser whose output the
// reference pins as identical to HashBlock_ParentRow (T/HashBlockCompiler_Test.java:26-35); not restated.
#pragma once
#include "xmo_types.h"
#include <unordered_map>
#include <deque>

namespace xmo {

struct HashBlock;
typedef std::shared_ptr<HashBlock> HashBlockP;

struct HashBlock {  // M/HashBlock.java:385-397
  int startIndex = 0, length = 0, numBasepairsUsed = 0;
  int32_t forwardHash = 0, reverseHash = 0;
  int gapDirection = 0, extraGapmerLength = 0;
  bool requestMergeLeft = false, requestMergeRight = false, nextRequestMergeLeft = false, nextRequestMergeRight = false;

  static int getMaxGapmerNumBasepairsUsed(int startingLength) { return startingLength + startingLength * 9 / 8 + 1; }  // :11-13
  static int getMaxGapmerLength(int startingLength) { return startingLength + startingLength * 9 / 4 + 1; }          // :15-17

  HashBlock() {}
  HashBlock(int startIndex, int length) : startIndex(startIndex), length(length), numBasepairsUsed(length) {}  // :46-50
  HashBlock(int startIndex, int length, int32_t f, int32_t r)                                                   // :52-58
      : startIndex(startIndex), length(length), numBasepairsUsed(length), forwardHash(f), reverseHash(r) {}
  HashBlock(char itemHere, int index) : startIndex(index), length(1), numBasepairsUsed(1) { hashChar(itemHere); }  // :60-65

  // :20-44 merge of two parents
  HashBlock(int startIndex, int length, const HashBlock& leftParent, const HashBlock& rightParent)
      : startIndex(startIndex), length(length), numBasepairsUsed(length) {
    mergeHashes(leftParent, rightParent);
    if (requestMergeLeft != requestMergeRight) {
      gapDirection = requestMergeLeft ? 1 : -1;
    } else if (leftParent.forwardHash != rightParent.reverseHash) {
      gapDirection = (leftParent.forwardHash > rightParent.reverseHash) ? 1 : -1;
    }
    extraGapmerLength = (leftParent.length + rightParent.length - this->length) / 4;
  }

  int getStartIndex() const { return startIndex; }
  int getEndIndex() const { return startIndex + length; }
  int getLength() const { return length; }
  int getNumBasepairsUsed() const { return numBasepairsUsed; }
  int32_t getForwardHash() const { return forwardHash; }
  int32_t getReverseHash() const { return reverseHash; }
  bool isPrimaryPolarity() const {  // :329-334
    if (requestMergeLeft != requestMergeRight) return requestMergeLeft;
    return forwardHash >= reverseHash;
  }
  bool isSecondaryPolarity() const {  // :336-340
    if (requestMergeLeft != requestMergeRight) return requestMergeRight;
    return forwardHash <= reverseHash;
  }
  int32_t getLookupKey() const { return isPrimaryPolarity() ? forwardHash : reverseHash; }  // :322-326

  static int charToInt(char c) {  // :152-169
    if (c == 'A') return 1;
    if (c == 'C') return 2;
    if (c == 'G') return 3;
    if (c == 'T') return 4;
    return 0;
  }

  // :67-150.  Returns: 0 = null (no space), 1 = `this` unchanged (gapDirection 0), 2 = new gapmer in `out`.
  int withGapAndExtension(const Sequence& sequence, HashBlock& out) const {
    int32_t extensionHash = 0;
    int targetExtraLength = this->length;
    targetExtraLength += jabs(std::max(forwardHash, reverseHash)) % 3;
    targetExtraLength += extraGapmerLength;
    int gapLength = this->length / 2;
    int extensionLength = targetExtraLength - gapLength;
    if (gapDirection == 0) return 1;
    bool leftGap = gapDirection < 0;
    if (leftGap) {
      int extensionEnd = startIndex - gapLength;
      int extensionStart = extensionEnd - extensionLength;
      if (extensionStart < 0) return 0;
      for (int i = extensionEnd - 1; i >= extensionStart; i--) {
        extensionHash = jmul(extensionHash, 7654337);
        extensionHash = jadd(extensionHash, charToInt(sequence.charAt(i)));
      }
      out = HashBlock(extensionStart, extensionLength + gapLength + this->length);  // Gapped_HashBlock(:122)
    } else {
      int extensionStart = getEndIndex() + gapLength;
      int extensionEnd = extensionStart + extensionLength;
      if (extensionEnd > sequence.getLength()) return 0;
      for (int i = extensionStart; i < extensionEnd; i++) {
        extensionHash = jmul(extensionHash, 7654337);
        char c = Basepairs::decode(Basepairs::complement(sequence.encodedCharAt(i)));
        extensionHash = jadd(extensionHash, charToInt(c));
      }
      out = HashBlock(startIndex, this->length + gapLength + extensionLength);  // Gapped_HashBlock(:138)
    }
    out.forwardHash = jadd(forwardHash, extensionHash);
    out.reverseHash = jadd(reverseHash, extensionHash);
    out.numBasepairsUsed = this->length + extensionLength;
    if (out.numBasepairsUsed > getMaxGapmerNumBasepairsUsed(this->length)) throw std::runtime_error("gapmer uses more basepairs than expected");
    if (out.length > getMaxGapmerLength(this->length)) throw std::runtime_error("gapmer longer than expected");
    // NOTE: a Gapped_HashBlock is built by the (start,length) constructor, so its merge flags are all
    // false and gapDirection/extraGapmerLength are 0 (M/Gapped_HashBlock.java:7-11).
    return 2;
  }

  HashBlock shifted(int shift) const {  // :369-383
    HashBlock r = *this;
    r.startIndex = startIndex + shift;
    return r;
  }

 private:
  void hashChar(char itemHere) {  // :171-188
    if ('A' == itemHere) forwardHash = 0;
    else if ('C' == itemHere) forwardHash = 1;
    else if ('G' == itemHere) forwardHash = 2;
    else forwardHash = 3;
    if (forwardHash / 2 == 0) requestMergeLeft = true;
    requestMergeRight = !requestMergeLeft;
    if (forwardHash % 2 == 0) nextRequestMergeLeft = true;
    nextRequestMergeRight = !nextRequestMergeLeft;
    reverseHash = 3 - forwardHash;
  }
  static int32_t mergeHashes(int leftLength, int32_t leftContentHash, int rightLength, int32_t rightContentHash) {  // :261-269
    int64_t rotatedLeft = ((int64_t)leftContentHash + 1) * (54323 + 323 * (int64_t)rightLength);
    int64_t rotatedRight = (int64_t)jadd(rightContentHash, 1) * (int64_t)leftLength;
    int64_t longTopBits = (int64_t)((uint64_t)rotatedLeft + (uint64_t)rotatedRight);
    return jadd((int32_t)(uint32_t)(uint64_t)longTopBits, (int32_t)(uint32_t)((uint64_t)(longTopBits >> 32)));
  }
  void mergeHashes(const HashBlock& leftParent, const HashBlock& rightParent) {  // :192-259
    forwardHash = mergeHashes(leftParent.length, leftParent.forwardHash, rightParent.length, rightParent.forwardHash);
    reverseHash = mergeHashes(rightParent.length, rightParent.reverseHash, leftParent.length, leftParent.reverseHash);
    requestMergeLeft = requestMergeRight = true;
    nextRequestMergeLeft = nextRequestMergeRight = true;
    const HashBlock* anchorParent = nullptr;
    const HashBlock* otherParent = nullptr;
    if (leftParent.forwardHash != rightParent.reverseHash) {
      if (leftParent.forwardHash > rightParent.reverseHash) { anchorParent = &rightParent; otherParent = &leftParent; }
      else { anchorParent = &leftParent; otherParent = &rightParent; }
    }
    if (anchorParent != nullptr) {
      if (forwardHash != reverseHash) {
        bool isReverse = forwardHash < reverseHash;
        bool invert = isReverse == (anchorParent == &rightParent);
        bool anchorNextLeft = anchorParent->nextRequestMergeLeft;
        bool anchorNextRight = anchorParent->nextRequestMergeRight;
        if (anchorNextLeft && anchorNextRight) {
          if (anchorParent == &rightParent) anchorNextRight = false; else anchorNextLeft = false;
        }
        bool otherNextLeft = otherParent->nextRequestMergeLeft;
        bool otherNextRight = otherParent->nextRequestMergeRight;
        if (otherNextLeft && otherNextRight) {
          if (otherParent == &rightParent) otherNextLeft = false; else otherNextRight = false;
        }
        requestMergeLeft = anchorNextLeft != invert;
        requestMergeRight = anchorNextRight != invert;
        nextRequestMergeLeft = otherNextLeft != invert;
        nextRequestMergeRight = otherNextRight != invert;
      }
    }
    if (leftParent.length != rightParent.length) {
      requestMergeLeft = (leftParent.length > rightParent.length);
      requestMergeRight = !requestMergeLeft;
      nextRequestMergeLeft = !requestMergeLeft;
      nextRequestMergeRight = !nextRequestMergeLeft;
    }
    if (forwardHash != reverseHash) {
      if (requestMergeLeft && requestMergeRight) {
        requestMergeLeft = (forwardHash > reverseHash);
        requestMergeRight = !requestMergeLeft;
      }
      if (nextRequestMergeLeft && nextRequestMergeRight) {
        nextRequestMergeLeft = requestMergeLeft;
        nextRequestMergeRight = !nextRequestMergeLeft;
      }
    }
  }
};

// ---------------------------------------------------------------- SequenceCondition (M/SequenceCondition.java)
struct SequenceCondition {
  std::vector<int> keys;
  std::vector<char> values;
  SequenceCondition() {}
  SequenceCondition(int position, char value) : keys(1, position), values(1, value) {}
  // returns false on conflict (Java: null)
  static bool intersect(const SequenceCondition& a, const SequenceCondition& b, SequenceCondition& out) {  // :22-94
    if (b.values.empty()) { out = a; return true; }
    if (a.values.empty()) { out = b; return true; }
    size_t i = 0, j = 0, numMatchingKeys = 0;
    while (i < a.keys.size() && j < b.keys.size()) {
      if (a.keys[i] < b.keys[j]) i++;
      else if (b.keys[j] < a.keys[i]) j++;
      else {
        if (a.values[i] != b.values[j]) return false;
        numMatchingKeys++; i++; j++;
      }
    }
    if (numMatchingKeys == a.keys.size()) { out = b; return true; }
    if (numMatchingKeys == b.keys.size()) { out = a; return true; }
    SequenceCondition m;
    i = j = 0;
    while (i < a.keys.size() && j < b.keys.size()) {
      if (a.keys[i] < b.keys[j]) { m.keys.push_back(a.keys[i]); m.values.push_back(a.values[i]); i++; }
      else if (b.keys[j] < a.keys[i]) { m.keys.push_back(b.keys[j]); m.values.push_back(b.values[j]); j++; }
      else { m.keys.push_back(a.keys[i]); m.values.push_back(a.values[i]); i++; j++; }
    }
    while (i < a.keys.size()) { m.keys.push_back(a.keys[i]); m.values.push_back(a.values[i]); i++; }
    while (j < b.keys.size()) { m.keys.push_back(b.keys[j]); m.values.push_back(b.values[j]); j++; }
    out = m;
    return true;
  }
};

struct ConditionalHashBlock {  // M/ConditionalHashBlock.java
  bool hasBlock = false;
  HashBlock block;
  SequenceCondition condition;
  ConditionalHashBlock() {}
  ConditionalHashBlock(const HashBlock& b, const SequenceCondition& c) : hasBlock(true), block(b), condition(c) {}
  explicit ConditionalHashBlock(const SequenceCondition& c) : hasBlock(false), condition(c) {}
};

// IMultiHashBlock: either a single HashBlock or a MultiHashBlock (list of conditional possibilities)
struct MultiBlock {
  bool isSingle = true;
  HashBlock single;
  std::vector<ConditionalHashBlock> possibilities;  // only when !isSingle
  uint32_t id = 0;                                   // unique identity (Java object identity)

  const HashBlock* getSingle() const { return isSingle ? &single : nullptr; }
  int getStartIndex() const {  // M/MultiHashBlock.java:17-28
    if (isSingle) return single.startIndex;
    int min = -1;
    for (auto& p : possibilities) if (p.hasBlock) { int v = p.block.startIndex; if (min < 0 || min > v) min = v; }
    return min;
  }
  int getEndIndex() const {  // :29-40
    if (isSingle) return single.getEndIndex();
    int max = -1;
    for (auto& p : possibilities) if (p.hasBlock) { int v = p.block.getEndIndex(); if (max < v) max = v; }
    return max;
  }
  int getMinLength() const {  // :41-52
    if (isSingle) return single.length;
    int min = -1;
    for (auto& p : possibilities) if (p.hasBlock) { int v = p.block.length; if (min < 0 || min > v) min = v; }
    return min;
  }
  // getPossibilities() of a single HashBlock = [(this, ALWAYS)]  (M/HashBlock.java:352-356)
  std::vector<ConditionalHashBlock> getPossibilities() const {
    if (!isSingle) return possibilities;
    return std::vector<ConditionalHashBlock>(1, ConditionalHashBlock(single, SequenceCondition()));
  }
};
typedef std::shared_ptr<MultiBlock> MultiBlockP;

struct BlockListener {  // the part of HashBlock_Buffer a row talks to
  virtual void addHashblock(const MultiBlockP& block) = 0;
  virtual ~BlockListener() {}
};

struct HashBlock_Row {  // M/HashBlock_Row.java
  virtual MultiBlockP get(int index) = 0;
  virtual MultiBlockP getAfter(int index) = 0;
  virtual const Sequence* getSequence() const = 0;
  virtual void garbageCollect(int index) = 0;
  virtual int getLevel() const = 0;
  virtual void skipTo(int index) = 0;
  virtual ~HashBlock_Row() {}
};

struct HashBlock_BaseRow : HashBlock_Row {  // M/HashBlock_BaseRow.java
  const Sequence* sequence;
  BlockListener* blockListener;
  std::unordered_map<int, MultiBlockP> blocks;
  HashBlock_BaseRow(const Sequence* s, BlockListener* l) : sequence(s), blockListener(l) {}
  MultiBlockP get(int index) override {  // :20-49
    if (index >= sequence->getLength()) return nullptr;
    auto it = blocks.find(index);
    if (it != blocks.end()) return it->second;
    MultiBlockP block(new MultiBlock());
    uint8_t encodedItemHere = sequence->encodedCharAt(index);
    if (Basepairs::isAmbiguous(encodedItemHere)) {
      block->isSingle = false;
      static const uint8_t encodedChars[4] = {1, 2, 4, 8};
      for (uint8_t encodedOptionHere : encodedChars) {
        if (Basepairs::canMatch(encodedItemHere, encodedOptionHere)) {
          char optionHere = Basepairs::decode(encodedOptionHere);
          block->possibilities.push_back(ConditionalHashBlock(HashBlock(optionHere, index), SequenceCondition(index, optionHere)));
        }
      }
    } else {
      block->single = HashBlock(Basepairs::decode(encodedItemHere), index);
    }
    if (blockListener) blockListener->addHashblock(block);
    blocks[index] = block;
    return block;
  }
  void skipTo(int) override {}
  MultiBlockP getAfter(int index) override { return get(index + 1); }
  const Sequence* getSequence() const override { return sequence; }
  void garbageCollect(int index) override { blocks.erase(index); }
  int getLevel() const override { return 0; }
};

struct HashBlock_ParentRow : HashBlock_Row {  // M/HashBlock_ParentRow.java
  static constexpr int maxNumCombinationsToExpand = 64;
  std::shared_ptr<HashBlock_Row> previousBatch;
  const Sequence* sequence;
  int maxPositionChecked = -1;
  bool assumeOnlyUsedOnce;
  BlockListener* blockListener;
  std::vector<MultiBlockP> blockList;
  int level;

  HashBlock_ParentRow(std::shared_ptr<HashBlock_Row> prev, bool assumeOnlyUsedOnce, BlockListener* l)
      : previousBatch(prev), sequence(prev->getSequence()), assumeOnlyUsedOnce(assumeOnlyUsedOnce), blockListener(l), level(prev->getLevel() + 1) {}

  MultiBlockP get(int index) override {  // :20-25
    MultiBlockP next = getAfter(index - 1);
    if (next && next->getStartIndex() == index) return next;
    return nullptr;
  }
  MultiBlockP getAfter(int position) override {  // :27-60
    if (position < maxPositionChecked) {
      MultiBlockP prev;
      for (int i = (int)blockList.size() - 1; i >= 0; i--) {
        const MultiBlockP& block = blockList[(size_t)i];
        if (block->getStartIndex() > position) prev = block; else break;
      }
      if (prev) return prev;
    }
    while (true) {
      if (maxPositionChecked >= sequence->getLength()) break;
      if (!blockList.empty()) {
        const MultiBlockP& lastBlock = blockList.back();
        if (lastBlock->getStartIndex() > position) return lastBlock;
      }
      maybeMakeBlock();
    }
    return nullptr;
  }
  void skipTo(int index) override {  // :62-67
    if (maxPositionChecked < index && assumeOnlyUsedOnce) {
      maxPositionChecked = index;
      blockList.clear();
    }
  }
  const Sequence* getSequence() const override { return sequence; }
  void garbageCollect(int index) override {  // :221-228
    for (size_t i = 0; i < blockList.size(); i++) {
      if (blockList[i]->getStartIndex() == index) { blockList.erase(blockList.begin() + (long)i); return; }
    }
  }
  int getLevel() const override { return level; }

 private:
  void maybeMakeBlock() {  // :69-127
    int afterIndex = maxPositionChecked;
    MultiBlockP leftBlock = previousBatch->getAfter(afterIndex);
    if (!leftBlock) { maxPositionChecked = sequence->getLength(); return; }
    int index = leftBlock->getStartIndex();
    maxPositionChecked = index;
    MultiBlockP rightBlock = previousBatch->getAfter(index);
    if (rightBlock) {
      const HashBlock* leftSingle = leftBlock->getSingle();
      const HashBlock* rightSingle = rightBlock->getSingle();
      if (leftSingle && rightSingle) {
        if (shouldMergeBlocks(*leftSingle, *rightSingle)) {
          MultiBlockP merged(new MultiBlock());
          merged->single = mergeBlocks(*leftSingle, *rightSingle);
          putBlock(merged);
        }
      } else {
        std::vector<ConditionalHashBlock> mergeOptions;
        for (const ConditionalHashBlock& leftOption : leftBlock->getPossibilities()) {
          if (leftOption.hasBlock) expand(leftOption.block, leftOption.condition, index, mergeOptions);
          else mergeOptions.push_back(ConditionalHashBlock(leftOption.condition));
        }
        if (!mergeOptions.empty() && (int)mergeOptions.size() <= maxNumCombinationsToExpand) {
          bool hasNonEmpty = false;
          for (auto& c : mergeOptions) if (c.hasBlock) hasNonEmpty = true;
          if (hasNonEmpty) {
            MultiBlockP multi(new MultiBlock());
            multi->isSingle = false;
            multi->possibilities = mergeOptions;
            putBlock(multi);
          }
        }
      }
    }
    if (assumeOnlyUsedOnce) previousBatch->garbageCollect(index);
  }
  void putBlock(const MultiBlockP& block) {  // :129-134
    blockList.push_back(block);
    if (blockListener) blockListener->addHashblock(block);
  }
  void expand(const HashBlock& leftBlock, const SequenceCondition& startingCondition, int startIndex, std::vector<ConditionalHashBlock>& results) {  // :137-191
    MultiBlockP next = previousBatch->getAfter(startIndex);
    if (!next) return;
    bool foundAnIntersection = false;
    for (const ConditionalHashBlock& rightOption : next->getPossibilities()) {
      SequenceCondition intersectionCondition;
      if (!SequenceCondition::intersect(startingCondition, rightOption.condition, intersectionCondition)) {
        if (foundAnIntersection) break;
        continue;
      }
      foundAnIntersection = true;
      if ((int)results.size() > maxNumCombinationsToExpand) return;
      if (!rightOption.hasBlock) {
        expand(leftBlock, intersectionCondition, next->getStartIndex(), results);
        continue;
      }
      if (shouldMergeBlocks(leftBlock, rightOption.block))
        results.push_back(ConditionalHashBlock(mergeBlocks(leftBlock, rightOption.block), intersectionCondition));
      else
        results.push_back(ConditionalHashBlock(intersectionCondition));
    }
  }
  static bool shouldMergeBlocks(const HashBlock& left, const HashBlock& right) {  // :200-208
    if (left.getEndIndex() < right.getStartIndex()) return false;
    if (left.requestMergeRight) return true;
    if (right.requestMergeLeft) return true;
    return false;
  }
  static HashBlock mergeBlocks(const HashBlock& left, const HashBlock& right) {  // :211-215
    int startIndex = left.getStartIndex();
    int endIndex = right.getEndIndex();
    return HashBlock(startIndex, endIndex - startIndex, left, right);
  }
};

struct HashBlock_Stream {  // M/HashBlock_Stream.java (compiler wrapping omitted, see header)
  std::shared_ptr<HashBlock_Row> blocks;
  bool emittedCurrentBlocks = false;
  bool assumeOnlyUsedOnce;
  BlockListener* blockListener;
  HashBlock_Stream(const Sequence* sequence, bool assumeOnlyUsedOnce, BlockListener* l)
      : blocks(new HashBlock_BaseRow(sequence, l)), assumeOnlyUsedOnce(assumeOnlyUsedOnce), blockListener(l) {}
  std::shared_ptr<HashBlock_Row> getNextBatch() {  // :21-36
    if (emittedCurrentBlocks) {
      blocks.reset(new HashBlock_ParentRow(blocks, assumeOnlyUsedOnce, blockListener));
      emittedCurrentBlocks = false;
    }
    emittedCurrentBlocks = true;
    return blocks;
  }
};

struct HashBlock_Pyramid {  // M/HashBlock_Pyramid.java
  HashBlock_Stream stream;
  std::vector<std::shared_ptr<HashBlock_Row>> rows;
  HashBlock_Pyramid(const Sequence* sequence, bool assumeOnlyUsedOnce, BlockListener* l) : stream(sequence, assumeOnlyUsedOnce, l) {}
  HashBlock_Row* get(int index) {  // :15-24
    while ((int)rows.size() <= index) rows.push_back(stream.getNextBatch());
    return rows[(size_t)index].get();
  }
};

}  // namespace xmo
