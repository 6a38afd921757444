// xmapper-hip: host-side construction of the seed index and the duplication map.
// Replaces M/HashBlock_Database.java:41-91,490-665 (hashing the reference), M/PackedMap.java:54-153 (bucket fill),
// M/HashBlock_Buffer.java, M/DuplicationDetector.java:97-436 (reference-only precompute).  In the reference this is
// CPU work done once per run (threads cooperate through helpHash/helpPack); here it is plain host C++ that emits the
// flat CSR layout the GPU probes.  A device-side build is a later row of SURVEY.md §8(f).
//
// Design (differs from the reference's lazy, garbage-collected row objects): a pyramid level is a pure function of the
// level below, so each contig is hashed level by level over flat arrays; every gapmer becomes a (table, bucket,
// position) record; each table is then sorted by (bucket, position) and cut into CSR form, a bucket that received
// more than maxInterestingCountPerKey records being marked overfull (its content is never observable: PackedMap.get
// returns null for it).
#pragma once
#include "xm_seed.h"
#include <vector>
#include <map>
#include <string>
#include <algorithm>
#include <stdexcept>
#include <cmath>

namespace xm {

struct HBlock {  // a reference-side block (contig coordinates need 32 bits)
  int32_t start, len;
  int32_t fwd, rev;
  uint8_t flags;
  int8_t gapDir;
  int16_t extraGap;
};

struct HostIndex {
  // reference
  std::vector<std::string> names;
  std::vector<int64_t> contigStart;
  std::vector<int32_t> contigLen;
  std::vector<int64_t> seqCumStart;  // 2*n + 1
  std::vector<uint8_t> refCodes;
  int64_t totalForwardSize = 0;
  // tables
  int32_t minInterestingSize = 0, maxHashedLength = 0, enableGapmers = 1, maxNumShortMatches = 5;
  std::vector<Table> tables;        // [maxHashedLength + 1]
  std::vector<uint32_t> bucketOff;  // concatenated (capacity + 1 per table)
  std::vector<uint64_t> positions;  // concatenated encoded positions
  // duplications
  int32_t dupWindow = 1000, dupMinCopies = 2, dupMinLength = 0, dupMaxLength = 0;
  std::vector<int64_t> dupKeyStart;
  std::vector<int32_t> dupKeys;
  bool dupDone = false;

  int numContigs() const { return (int)contigLen.size(); }
  SeqView contigView(int c, bool rc) const {
    SeqView v;
    v.base = refCodes.data() + contigStart[(size_t)c];
    v.len = contigLen[(size_t)c];
    v.rc = rc ? 1 : 0;
    v.id = 0;
    return v;
  }
  static int log2RoundUp(int64_t v) { int bits = 0; int64_t p = 1; while (p < v) { p <<= 1; bits++; } return bits; }
  int chooseMinDuplicationLength() const { return log2RoundUp(totalForwardSize); }  // M/DuplicationDetector.java:17-31
  int chooseMaxDuplicationLength() const { return chooseMinDuplicationLength() * 2; }  // :34-36
  double dupGranularity() const { return enableGapmers ? (double)(dupMinLength * 5 / 8) : (double)dupMinLength; }  // :67-77

  void setReference(int n, const char* const* nm, const uint8_t* const* codes, const int64_t* lengths) {
    int64_t total = 0;
    for (int i = 0; i < n; i++) {
      if (lengths[i] < 1 || lengths[i] > 0x7FFFFFF0LL) throw std::runtime_error("contig length out of range");
      total += lengths[i];
    }
    refCodes.resize((size_t)total);
    int64_t off = 0;
    for (int i = 0; i < n; i++) {
      names.push_back(nm && nm[i] ? nm[i] : ("contig" + std::to_string(i)));
      contigStart.push_back(off);
      contigLen.push_back((int32_t)lengths[i]);
      seqCumStart.push_back(2 * off);
      seqCumStart.push_back(2 * off + lengths[i]);
      for (int64_t k = 0; k < lengths[i]; k++) {
        uint8_t c = codes[i][k];
        if (c != 1 && c != 2 && c != 4 && c != 8)
          throw std::runtime_error("reference contig " + names.back() + " has a non-ACGT base at " + std::to_string(k) +
                                   ": ambiguous reference bases (MultiHashBlock, M/MultiHashBlock.java) are not supported by this build");
        refCodes[(size_t)(off + k)] = c;
      }
      off += lengths[i];
    }
    seqCumStart.push_back(2 * off);
    totalForwardSize = total;
  }

  int64_t encodePosition(int contig, bool rc, int start) const { return seqCumStart[(size_t)contig * 2 + (rc ? 1 : 0)] + start; }

  int estimateRequiredCapacity(int numPositionsPerBlock) const {  // M/HashBlock_Database.java:620-665
    int anchorBlockSize = enableGapmers ? numPositionsPerBlock * 2 / 3 : numPositionsPerBlock;
    double sizeProbability = std::min(1.0, 2.0 / anchorBlockSize);
    double offsetProbability = std::min(1.0, 2.0 / anchorBlockSize);
    double blockPossibilityProbability = sizeProbability * offsetProbability;
    int64_t maxNumSequencesOfThisLength = numPositionsPerBlock <= 16 ? ((int64_t)1 << (numPositionsPerBlock * 2)) : ((int64_t)1 << 32);
    int64_t maxNumStoredSequencesOfThisLength = maxNumSequencesOfThisLength / 2;
    int64_t maxNumExistentHashcodes = (int64_t)((double)maxNumStoredSequencesOfThisLength * blockPossibilityProbability);
    int64_t numBlocksOfThisSize = (int64_t)((double)totalForwardSize * blockPossibilityProbability);
    double existenceFraction = 1 - std::pow((double)((double)maxNumExistentHashcodes - 1.0) / (double)maxNumExistentHashcodes, (double)numBlocksOfThisSize);
    int uniqueCount = j2i((double)maxNumExistentHashcodes * existenceFraction);
    int result = uniqueCount;
    if (result % 2 == 0) result++;
    return result;
  }

  struct Rec { uint32_t bucket; uint64_t pos; };

  // Hash every gapmer with minLen <= used <= maxLen and append tables [minLen..maxLen].  Tables below minInterestingSize and
  // tables that receive no record are the reference's PackedMap(1, 1) placeholders (M/HashBlock_Database.java:387-393).
  void hashLengths(int minLen, int maxLen) {
    std::vector<int> capacity((size_t)maxLen + 1, 0), maxCount((size_t)maxLen + 1, 0);
    for (int L = std::max(minLen, minInterestingSize); L <= maxLen; L++) {
      int cap = estimateRequiredCapacity(L);
      if (cap < 1) cap = 1;
      if (cap > INT32_MAX / 2) cap = INT32_MAX / 2;  // M/PackedMap.java:22-25
      capacity[(size_t)L] = cap;
      int mx = L * L;  // M/HashBlock_Database.java:569-576
      if (mx < maxNumShortMatches) mx = maxNumShortMatches;
      if (mx > 32766) mx = 32766;
      if (mx < 1) mx = 1;
      maxCount[(size_t)L] = mx;
    }
    std::vector<std::vector<Rec>> recs((size_t)maxLen + 1);
    int lo = std::max(minLen, minInterestingSize);
    for (int c = 0; c < numContigs(); c++) {
      SeqView seq = contigView(c, false);
      int n = seq.len;
      std::vector<HBlock> cur((size_t)n), next;
      for (int i = 0; i < n; i++) {
        PBlock b0 = level0Block(seq.base[i], 0);
        HBlock h;
        h.start = i; h.len = 1; h.fwd = b0.fwd; h.rev = b0.rev; h.flags = b0.flags; h.gapDir = 0; h.extraGap = 0;
        cur[(size_t)i] = h;
      }
      while (!cur.empty()) {
        bool anyShort = false;
        for (const HBlock& b : cur) {
          if (b.len > maxLen) continue;  // its gapmer uses at least b.len bases
          anyShort = true;
          QBlock g;
          int st;
          if (enableGapmers) {
            st = withGapAndExtension(b, seq, g);
            if (st == 0) continue;
          } else {
            g.start = b.start; g.len = b.len; g.used = b.len; g.fwd = b.fwd; g.rev = b.rev; g.flags = b.flags;
          }
          int used = g.used;
          if (used < lo || used > maxLen) continue;
          bool rml = (g.flags & F_RML) != 0, rmr = (g.flags & F_RMR) != 0;
          bool primary = (rml != rmr) ? rml : (g.fwd >= g.rev);     // M/HashBlock.java:329-334
          bool secondary = (rml != rmr) ? rmr : (g.fwd <= g.rev);   // :336-340
          int cap = capacity[(size_t)used];
          if (primary) {  // M/PackedMap.java:107-112
            int32_t r = g.fwd % cap; if (r < 0) r += cap;
            recs[(size_t)used].push_back(Rec{(uint32_t)r, (uint64_t)encodePosition(c, false, g.start)});
          }
          if (secondary) {  // :113-118
            int32_t r = g.rev % cap; if (r < 0) r += cap;
            recs[(size_t)used].push_back(Rec{(uint32_t)r, (uint64_t)encodePosition(c, true, n - (g.start + g.len))});
          }
        }
        if (!anyShort) break;
        next.clear();
        for (size_t i = 0; i + 1 < cur.size(); i++) {
          if (shouldMergeBlocks(cur[i], cur[i + 1])) next.push_back(mergeBlocks(cur[i], cur[i + 1]));
        }
        cur.swap(next);
      }
    }
    if ((int)tables.size() < maxLen + 1) tables.resize((size_t)maxLen + 1);
    for (int L = minLen; L <= maxLen; L++) {
      Table t;
      std::vector<Rec>& v = recs[(size_t)L];
      if (v.empty()) { t.capacity = 1; t.maxCount = 1; }
      else { t.capacity = capacity[(size_t)L]; t.maxCount = maxCount[(size_t)L]; }
      t.offBase = (int64_t)bucketOff.size();
      t.posBase = (int64_t)positions.size();
      std::sort(v.begin(), v.end(), [](const Rec& a, const Rec& b) { return a.bucket != b.bucket ? a.bucket < b.bucket : a.pos < b.pos; });
      size_t i = 0;
      uint64_t stored = 0;
      for (int k = 0; k < t.capacity; k++) {
        size_t j = i;
        while (j < v.size() && v[j].bucket == (uint32_t)k) j++;
        size_t cnt = j - i;
        if (stored > 0x7FFFFFFFull) throw std::runtime_error("table too large for 31-bit bucket offsets");
        if ((int64_t)cnt > (int64_t)t.maxCount) {
          bucketOff.push_back((uint32_t)stored | XM_OVERFULL);
        } else {
          bucketOff.push_back((uint32_t)stored);
          for (size_t x = i; x < j; x++) positions.push_back(v[x].pos);
          stored += cnt;
        }
        i = j;
      }
      bucketOff.push_back((uint32_t)stored);
      tables[(size_t)L] = t;
      std::vector<Rec>().swap(v);
    }
  }

  void build(int enableGapmers_, int minInteresting, int maxHashed, int dupWindow_, int dupMinCopies_, int dupMinLen, int dupMaxLen) {
    enableGapmers = enableGapmers_ ? 1 : 0;
    if (minInteresting <= 0) minInterestingSize = j2i(std::max((std::log((double)(totalForwardSize + 1)) / std::log(4.0)) - 2, 1.0));  // M/HashBlock_Database.java:52
    else minInterestingSize = minInteresting;
    dupWindow = dupWindow_ > 0 ? dupWindow_ : 1000;
    dupMinCopies = dupMinCopies_ > 0 ? dupMinCopies_ : 2;
    dupMinLength = dupMinLen > 0 ? dupMinLen : chooseMinDuplicationLength();
    dupMaxLength = dupMaxLen > 0 ? dupMaxLen : chooseMaxDuplicationLength();
    int want = maxHashed > 0 ? maxHashed : chooseMaxDuplicationLength();
    want = std::max(want, std::max(dupMaxLength, dupMinLength + 1));
    want = std::max(want, 1);
    hashLengths(0, want);
    maxHashedLength = want;
    detectDuplications();
  }
  void ensureLength(int length) {
    if (length <= maxHashedLength) return;
    hashLengths(maxHashedLength + 1, length);
    maxHashedLength = length;
  }

  // ---- PackedMap.get on the host tables (used by the duplication pass and by the inspection API)
  // returns -1 for "null" (overfull or more than maxCount), else count; first = index into positions
  int bucketGet(int L, uint32_t bucket, int64_t& first) const {
    const Table& t = tables[(size_t)L];
    uint32_t o0 = bucketOff[(size_t)(t.offBase + bucket)], o1 = bucketOff[(size_t)(t.offBase + bucket + 1)];
    if (o0 & XM_OVERFULL) return -1;
    int cnt = (int)((o1 & ~XM_OVERFULL) - (o0 & ~XM_OVERFULL));
    if (cnt > t.maxCount) return -1;
    first = t.posBase + (int64_t)(o0 & ~XM_OVERFULL);
    return cnt;
  }
  void decode(int64_t enc, int& contig, bool& rc, int& start) const {
    size_t idx = (size_t)(std::upper_bound(seqCumStart.begin(), seqCumStart.end(), enc) - seqCumStart.begin()) - 1;
    if (idx >= (size_t)numContigs() * 2) idx = (size_t)numContigs() * 2 - 1;
    contig = (int)(idx >> 1);
    rc = (idx & 1) != 0;
    start = (int)(enc - seqCumStart[idx]);
  }

  // ---- duplication map: M/DuplicationDetector.java:97-436.  Keys are kept per (contig, strand) like the reference's
  // Map<Sequence, TreeMap<Integer, Duplication>>; only the forward-strand keys are ever consulted by a read.
  struct Dup { int length; int copies; };
  int windowNumber(int index) const { return index / dupWindow; }
  int compareDuplications(int start1, const Dup& d1, int start2, const Dup& d2) const {  // :406-436
    if (dupWindow > 1 && windowNumber(start1) != windowNumber(start2)) return 0;
    int end1 = start1 + d1.length, end2 = start2 + d2.length;
    if (start1 <= start2 && end1 >= end2) return 1;
    if (start1 >= start2 && end1 <= end2) return -1;
    if (dupWindow > 1) {
      int countDifference = d1.copies - d2.copies;
      if (countDifference != 0) return countDifference;
      if (start1 != start2) return start1 - start2;
    }
    return 0;
  }
  void detectDuplications() {
    typedef std::map<int, Dup> KeyMap;
    std::vector<KeyMap> all((size_t)numContigs() * 2);      // duplicationsBySequence
    for (int L = dupMinLength; L <= dupMaxLength; L++) {
      if (L > maxHashedLength) break;
      const Table& t = tables[(size_t)L];
      std::map<size_t, KeyMap> pending;                       // `blocks` of process(): flushed every 10000 hashcodes
      int prefixLength = (L + 3) / 4;
      for (int hashcode = 0; hashcode < t.capacity; hashcode++) {
        int64_t first = 0;
        int cnt = bucketGet(L, (uint32_t)hashcode, first);   // lookupByForwardHash: packed key of `hashcode` is itself
        if (cnt >= dupMinCopies) {                            // numForwardMatches = matches.length / 2
          struct P { int contig; bool rc; int start; };
          std::vector<P> matches;
          for (int i = 0; i < cnt; i++) { P p; decode((int64_t)positions[(size_t)(first + i)], p.contig, p.rc, p.start); matches.push_back(p); }
          for (int i = 0; i < cnt; i++) {                     // + reverseComplement(position, blockLength)  (sic: used length, not span)
            P p = matches[(size_t)i];
            p.start = contigLen[(size_t)p.contig] - p.start - L;
            p.rc = !p.rc;
            matches.push_back(p);
          }
          std::map<std::string, std::vector<P>> byText;       // group by prefix + suffix text to skip hash collisions
          for (const P& p : matches) {
            SeqView v = contigView(p.contig, p.rc);
            std::string text;
            for (int i = 0; i < prefixLength; i++) text.push_back((char)v.at(p.start + i));
            for (int i = 0; i < prefixLength; i++) text.push_back((char)v.at(p.start + L - prefixLength + i));
            std::vector<P>& g = byText[text];
            bool dupPos = false;                               // removeDuplicatePositions
            for (const P& q : g) if (q.contig == p.contig && q.rc == p.rc && q.start == p.start) { dupPos = true; break; }
            if (!dupPos) g.push_back(p);
          }
          for (auto& e : byText) {
            if ((int)e.second.size() < dupMinCopies) continue;
            Dup d{L, (int)e.second.size()};
            for (const P& p : e.second) pending[(size_t)p.contig * 2 + (p.rc ? 1 : 0)][p.start] = d;
          }
        }
        if (hashcode % 10000 == 9999 || hashcode == t.capacity - 1) {  // saveDuplications :332-400
          for (auto& seqEntry : pending) {
            KeyMap& m = all[seqEntry.first];
            for (auto& kv : seqEntry.second) {
              int start = kv.first;
              const Dup& nd = kv.second;
              bool insert = true;
              while (true) {
                auto it = m.upper_bound(start);
                if (it != m.begin()) {
                  --it;
                  int cmp = compareDuplications(start, nd, it->first, it->second);
                  if (cmp > 0) { insert = false; break; }
                  if (cmp < 0) { m.erase(it); continue; }
                }
                break;
              }
              while (true) {
                auto it = m.lower_bound(start);
                if (it != m.end()) {
                  int cmp = compareDuplications(start, nd, it->first, it->second);
                  if (cmp > 0) { insert = false; break; }
                  if (cmp < 0) { m.erase(it); continue; }
                }
                break;
              }
              if (insert) m[start] = nd;
            }
          }
          pending.clear();
        }
      }
    }
    dupKeyStart.assign(1, 0);
    dupKeys.clear();
    for (int c = 0; c < numContigs(); c++) {
      for (auto& kv : all[(size_t)c * 2]) dupKeys.push_back(kv.first);
      dupKeyStart.push_back((int64_t)dupKeys.size());
    }
    dupDone = true;
  }
};

}  // namespace xm
