// xmapper-hip: host-side construction of the seed index and the duplication map.
// Replaces M/HashBlock_Database.java:41-91,490-665 (hashing the reference), M/PackedMap.java:54-153 (bucket fill),
// M/HashBlock_Buffer.java, M/DuplicationDetector.java:97-436 (reference-only precompute).  In the reference this is
// CPU work done once per run (threads cooperate through helpHash/helpPack); here it is plain host C++ that emits the
// flat CSR layout the GPU probes.  With a GPU present, references without ambiguity codes are hashed there instead
// (xm_index_device.hip, same tables); this file is the builder for the rest and the oracle-checked statement of the layout.
//
// Design (differs from the reference's lazy, garbage-collected row objects): a pyramid level is a pure function of the
// level below, so each contig is hashed level by level over flat arrays; every gapmer becomes a (table, bucket,
// position) record; each table is then sorted by (bucket, position) and cut into CSR form, a bucket that received
// more than maxInterestingCountPerKey records being marked overfull (its content is never observable: PackedMap.get
// returns null for it).  Contigs with ambiguous bases go through a vector-based restatement of the multi-block rule
// (possibilities under conditions), see nextLevelMulti below.
#pragma once
#include "xm_seed.h"
#include <vector>
#include <map>
#include <string>
#include <algorithm>
#include <stdexcept>
#include <cmath>
#include <thread>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <functional>
#include <unistd.h>

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
        if (c < 1 || c > 15) throw std::runtime_error("reference contig " + names.back() + " has an invalid base code at " + std::to_string(k));
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

  // host threads for the build (the reference's workers cooperate through helpHash/helpPack, M/HashBlock_Database.java:237-242)
  static int buildThreads(size_t work) {
    if (work < 200000) return 1;
    unsigned hc = std::thread::hardware_concurrency();
    int t = hc ? (int)hc : 1;
    const char* e = getenv("XM_BUILD_THREADS");
    if (e && *e) t = atoi(e);
    return t < 1 ? 1 : (t > 64 ? 64 : t);
  }
  // fn(part, begin, end) over [0, n) cut into nThreads consecutive parts
  static void parallelParts(size_t n, int nThreads, const std::function<void(int, size_t, size_t)>& fn) {
    if (nThreads <= 1 || n < 2) { fn(0, 0, n); return; }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err((size_t)nThreads);
    for (int t = 0; t < nThreads; t++) {
      size_t b = n * (size_t)t / (size_t)nThreads, e = n * (size_t)(t + 1) / (size_t)nThreads;
      th.emplace_back([&, t, b, e]() { try { fn(t, b, e); } catch (...) { err[(size_t)t] = std::current_exception(); } });
    }
    for (auto& x : th) x.join();
    for (auto& x : err) if (x) std::rethrow_exception(x);
  }

  struct Rec { uint32_t bucket; uint64_t pos; };  // pos bit 63: the record comes from a possibility of a multi block
  static constexpr uint64_t REC_MULTI = 1ull << 63;

  // ---- contigs with ambiguous bases: blocks over such a base are lists of conditional possibilities (M/MultiHashBlock.java,
  // M/ConditionalHashBlock.java, M/SequenceCondition.java, the multi branch of M/HashBlock_ParentRow.java:69-191 and
  // M/HashBlock_BaseRow.java:20-49), restated over vectors; xm_seed.h has the fixed-capacity read-side version of the same rule.
  typedef std::vector<std::pair<int32_t, uint8_t>> HCond;  // sorted (position, base code)
  static bool hcondIntersect(const HCond& a, const HCond& b, HCond& out) {  // SequenceCondition.intersect :22-94; false = null
    if (b.empty()) { out = a; return true; }
    if (a.empty()) { out = b; return true; }
    size_t i = 0, j = 0, same = 0;
    while (i < a.size() && j < b.size()) {
      if (a[i].first < b[j].first) i++;
      else if (b[j].first < a[i].first) j++;
      else { if (a[i].second != b[j].second) return false; same++; i++; j++; }
    }
    if (same == a.size()) { out = b; return true; }
    if (same == b.size()) { out = a; return true; }
    HCond m;
    i = j = 0;
    while (i < a.size() && j < b.size()) {
      if (a[i].first < b[j].first) m.push_back(a[i++]);
      else if (b[j].first < a[i].first) m.push_back(b[j++]);
      else { m.push_back(a[i]); i++; j++; }
    }
    while (i < a.size()) m.push_back(a[i++]);
    while (j < b.size()) m.push_back(b[j++]);
    out.swap(m);
    return true;
  }
  struct HPoss { HBlock block; bool hasBlock; HCond cond; };
  struct HEntry {
    bool multi = false;
    HBlock single;            // !multi
    std::vector<HPoss> poss;  // multi
    int start() const {
      if (!multi) return single.start;
      int mn = -1;
      for (const HPoss& p : poss) if (p.hasBlock && (mn < 0 || p.block.start < mn)) mn = p.block.start;
      return mn;
    }
    int minLength() const {
      if (!multi) return single.len;
      int mn = -1;
      for (const HPoss& p : poss) if (p.hasBlock && (mn < 0 || p.block.len < mn)) mn = p.block.len;
      return mn;
    }
  };
  static HBlock hblock0(uint8_t code, int index) {
    PBlock b0 = level0Block(code, 0);
    HBlock h;
    h.start = index; h.len = 1; h.fwd = b0.fwd; h.rev = b0.rev; h.flags = b0.flags; h.gapDir = 0; h.extraGap = 0;
    return h;
  }
  static std::vector<HPoss> possibilitiesOf(const HEntry& e) {  // getPossibilities(): a single block is [(this, ALWAYS)]
    if (e.multi) return e.poss;
    return std::vector<HPoss>(1, HPoss{e.single, true, HCond()});
  }
  static void expandMulti(const std::vector<HEntry>& prev, const HBlock& leftBlock, const HCond& startingCondition, size_t j, std::vector<HPoss>& results) {  // :137-191
    if (j + 1 >= prev.size()) return;
    const HEntry& next = prev[j + 1];
    bool foundAnIntersection = false;
    for (const HPoss& rightOption : possibilitiesOf(next)) {
      HCond ic;
      if (!hcondIntersect(startingCondition, rightOption.cond, ic)) {
        if (foundAnIntersection) break;
        continue;
      }
      foundAnIntersection = true;
      if ((int)results.size() > XM_MAX_COMBINATIONS) return;
      if (!rightOption.hasBlock) { expandMulti(prev, leftBlock, ic, j + 1, results); continue; }
      if (shouldMergeBlocks(leftBlock, rightOption.block)) results.push_back(HPoss{mergeBlocks(leftBlock, rightOption.block), true, ic});
      else results.push_back(HPoss{leftBlock, false, ic});
    }
  }
  static std::vector<HEntry> nextLevelMulti(const std::vector<HEntry>& prev) {  // maybeMakeBlock :69-127 for every block of the level below
    std::vector<HEntry> out;
    for (size_t i = 0; i + 1 < prev.size(); i++) {
      const HEntry& L = prev[i];
      const HEntry& R = prev[i + 1];
      if (!L.multi && !R.multi) {
        if (shouldMergeBlocks(L.single, R.single)) { HEntry e; e.single = mergeBlocks(L.single, R.single); out.push_back(e); }
        continue;
      }
      std::vector<HPoss> mergeOptions;
      for (const HPoss& leftOption : possibilitiesOf(L)) {
        if (leftOption.hasBlock) expandMulti(prev, leftOption.block, leftOption.cond, i, mergeOptions);
        else mergeOptions.push_back(HPoss{leftOption.block, false, leftOption.cond});
      }
      if (!mergeOptions.empty() && (int)mergeOptions.size() <= XM_MAX_COMBINATIONS) {
        bool hasNonEmpty = false;
        for (const HPoss& p : mergeOptions) if (p.hasBlock) hasNonEmpty = true;
        if (hasNonEmpty) { HEntry e; e.multi = true; e.poss.swap(mergeOptions); out.push_back(e); }
      }
    }
    return out;
  }

  // one block (a single block, or one possibility of a multi block) -> its gapmer -> records
  void emitBlockRecords(const SeqView& seq, int c, const HBlock& b, uint32_t fromMulti, int lo, int maxLen, const std::vector<int>& capacity, std::vector<std::vector<Rec>>& recs) const {
    const int n = seq.len;
    QBlock g;
    int st;
    if (enableGapmers) {
      st = withGapAndExtension(b, seq, g);
      if (st == 0) return;
    } else {
      g.start = b.start; g.len = b.len; g.used = b.len; g.fwd = b.fwd; g.rev = b.rev; g.flags = b.flags;
    }
    int used = g.used;
    if (used < lo || used > maxLen) return;
    bool rml = (g.flags & F_RML) != 0, rmr = (g.flags & F_RMR) != 0;
    bool primary = (rml != rmr) ? rml : (g.fwd >= g.rev);     // M/HashBlock.java:329-334
    bool secondary = (rml != rmr) ? rmr : (g.fwd <= g.rev);   // :336-340
    int cap = capacity[(size_t)used];
    if (primary) {  // M/PackedMap.java:107-112
      int32_t r = g.fwd % cap; if (r < 0) r += cap;
      recs[(size_t)used].push_back(Rec{(uint32_t)r, (uint64_t)encodePosition(c, false, g.start) | (fromMulti ? REC_MULTI : 0)});
    }
    if (secondary) {  // :113-118
      int32_t r = g.rev % cap; if (r < 0) r += cap;
      recs[(size_t)used].push_back(Rec{(uint32_t)r, (uint64_t)encodePosition(c, true, n - (g.start + g.len)) | (fromMulti ? REC_MULTI : 0)});
    }
  }
  // The pyramid with conditional multi blocks over positions [ws, we) of contig c, level by level, emitting the records of every block with
  // len <= maxLen (onlyMulti: only those of multi blocks, i.e. of blocks over an ambiguous base).  A block depends on the bases of its own
  // span only (a level is an ordered list whose block ends never decrease, and a block is merged from two neighbours of that list), so a
  // window computes every block whose span lies inside it exactly as the whole contig would.
  //
  // Runs of N (GRCh38 has runs of megabases; 2 048 and longer are treated this way): with ws2 >= 0 the window is [ws, we) + [ws2, we2) with the middle of the run left out.
  // Inside a run every position looks the same, and its blocks stop existing after a few levels (the possibilities multiply past
  // HashBlock_ParentRow.maxNumCombinationsToExpand); from that level on the last block before the run's interior and the first one behind it are
  // neighbours in the level's list (which matters: merges and the combination limit see that neighbour).  The two halves are therefore built
  // level by level on their own until nothing is left in the inner quarter of either, then joined into one list.  The blocks next to the cut of
  // the first half miss their right neighbours; they lie in the part of the run that has died by then.  Returns false (nothing emitted) when the
  // run's interior would emit records of its own (only with a small minInterestingSize): the caller then takes the window whole.
  bool hashAmbiguousWindow(int c, int ws, int we, bool onlyMulti, int lo, int maxLen, const std::vector<int>& capacity, std::vector<std::vector<Rec>>& recs, int ws2 = -1, int we2 = -1, int halfN = 0) const {
    SeqView seq = contigView(c, false);
    auto level0 = [&](int from, int to, std::vector<HEntry>& cur) {
      cur.assign((size_t)(to - from), HEntry());
      for (int i = from; i < to; i++) {
        uint8_t code = seq.base[i];
        HEntry& e = cur[(size_t)(i - from)];
        if (bpIsAmbiguous(code)) {  // HashBlock_BaseRow.get: one possibility per base the code can stand for, A C G T order
          e.multi = true;
          for (int bit = 0; bit < 4; bit++) if (code & (1 << bit)) e.poss.push_back(HPoss{hblock0((uint8_t)(1 << bit), i), true, HCond(1, std::make_pair((int32_t)i, (uint8_t)(1 << bit)))});
        } else {
          e.single = hblock0(code, i);
        }
      }
    };
    std::vector<HEntry> cur, curB;
    level0(ws, we, cur);
    bool joined = ws2 < 0;
    if (!joined) level0(ws2, we2, curB);
    // the first half ends and the second begins with halfN positions of the run.  [we - halfN/2, we - halfN/4) of the first half is far from the
    // run's start and from the cut: what lives there lives at every position of the left-out middle (the blocks between it and the cut miss right
    // neighbours and may outlive it: they are dropped at the join); the same for [ws2, ws2 + halfN/2) of the second half, whose blocks are all exact
    const int deepA = we - halfN / 2, probeEndA = we - halfN / 4, deepEndB = ws2 + halfN / 2;
    std::vector<std::vector<Rec>> mine((size_t)maxLen + 1);  // (kept apart until the window is known to be usable)
    auto emitList = [&](const std::vector<HEntry>& list, bool& anyShort, int from, int to) {  // entries starting in [from, to)
      for (const HEntry& e : list) {
        const int st = e.start();
        if (st < from || st >= to) continue;
        if (e.minLength() > maxLen) continue;
        anyShort = true;
        if (!e.multi) { if (!onlyMulti && e.single.len <= maxLen) emitBlockRecords(seq, c, e.single, 0, lo, maxLen, capacity, mine); }
        else for (const HPoss& p : e.poss) if (p.hasBlock && p.block.len <= maxLen) emitBlockRecords(seq, c, p.block, 1, lo, maxLen, capacity, mine);
      }
    };
    auto startsIn = [&](const std::vector<HEntry>& list, int from, int to) { for (const HEntry& e : list) { const int st = e.start(); if (st >= from && st < to) return true; } return false; };
    auto numRecords = [&]() { size_t n = 0; for (auto& v : mine) n += v.size(); return n; };
    while (!cur.empty() || !curB.empty()) {
      bool anyShort = false;
      if (!joined && !startsIn(cur, deepA, probeEndA) && !startsIn(curB, ws2, deepEndB)) {
        std::vector<HEntry> both;
        for (const HEntry& e : cur) if (e.start() < deepA) both.push_back(e);
        both.insert(both.end(), curB.begin(), curB.end());
        cur.swap(both);
        curB.clear();
        joined = true;
      }
      if (joined) emitList(cur, anyShort, INT32_MIN, INT32_MAX);
      else {
        emitList(cur, anyShort, INT32_MIN, deepA);
        emitList(curB, anyShort, deepEndB, INT32_MAX);
        // the deep parts stand for every position of the run's middle: records from there would have to be repeated for each of them
        const size_t before = numRecords();
        bool deepShort = false;
        emitList(cur, deepShort, deepA, probeEndA);
        emitList(curB, deepShort, ws2, deepEndB);
        if (numRecords() != before) return false;
        anyShort = true;  // (the middle of the run is still alive: the halves go on)
      }
      if (!anyShort) break;
      std::vector<HEntry> next = nextLevelMulti(cur);
      cur.swap(next);
      if (!joined) { std::vector<HEntry> nextB = nextLevelMulti(curB); curB.swap(nextB); }
    }
    for (int L = 0; L <= maxLen; L++) recs[(size_t)L].insert(recs[(size_t)L].end(), mine[(size_t)L].begin(), mine[(size_t)L].end());
    return true;
  }
  // Hybrid build (xm_index_device.hip): the GPU hashes every block that lies clear of the ambiguous bases; the blocks over an ambiguous base are
  // the multi blocks of the windows around them (a block that is emitted is at most maxLen long, so maxLen + 2 bases on either side of a run of
  // ambiguous bases hold every such block whole).  Ambiguous bases less than two margins apart share a window; all host threads work on the
  // windows.  recs[L] += the multi records.
  void multiRecordsNearAmbiguity(int lo, int maxLen, const std::vector<int>& capacity, std::vector<std::vector<Rec>>& recs) const {
    struct Win { int c, ws, we, ws2, we2; };
    std::vector<Win> wins;
    const int margin = maxLen + 2;
    // a run of N longer than this is not held whole: 2 * half positions of it stay in the window (XM_BUILD_SPLICE_MIN: test hook)
    int spliceMin = 2048;
    if (const char* e = getenv("XM_BUILD_SPLICE_MIN")) { if (*e) spliceMin = std::max(64, atoi(e)); }
    const int half = std::max(32, spliceMin / 4);
    for (int c = 0; c < numContigs(); c++) {
      SeqView seq = contigView(c, false);
      const int n = seq.len;
      int i = 0;
      while (i < n) {
        if (!bpIsAmbiguous(seq.base[i])) { i++; continue; }
        // the cluster [i, last]: ambiguous bases less than two margins apart; the longest run of plain N inside it
        int last = i, runStart = -1, runLen = 0, bestStart = -1, bestLen = 0;
        for (int j = i; j < n && j - last <= 2 * margin; j++) {
          if (bpIsAmbiguous(seq.base[j])) last = j;
          if (seq.base[j] == 15) { if (runLen == 0) runStart = j; runLen++; if (runLen > bestLen) { bestLen = runLen; bestStart = runStart; } }
          else runLen = 0;
        }
        const int from = std::max(0, i - margin), to = std::min(n, last + 1 + margin);
        if (bestLen >= spliceMin) wins.push_back(Win{c, from, bestStart + half, bestStart + bestLen - half, to});
        else wins.push_back(Win{c, from, to, -1, -1});
        i = last + 1;
      }
    }
    const int nT = buildThreads(wins.size() * 4096);
    std::vector<std::vector<std::vector<Rec>>> part((size_t)nT, std::vector<std::vector<Rec>>((size_t)maxLen + 1));
    std::vector<size_t> nextJob(1, 0);
    std::mutex jobMu;
    std::atomic<int> nSplit{0}, nWholeAfterAll{0};
    parallelParts((size_t)nT, nT, [&](int t, size_t, size_t) {
      while (true) {
        size_t job;
        { std::lock_guard<std::mutex> lock(jobMu); job = nextJob[0]++; }
        if (job >= wins.size()) break;
        const Win& w = wins[job];
        if (w.ws2 >= 0 && hashAmbiguousWindow(w.c, w.ws, w.we, true, lo, maxLen, capacity, part[(size_t)t], w.ws2, w.we2, half)) { nSplit++; continue; }
        if (w.ws2 >= 0) nWholeAfterAll++;
        hashAmbiguousWindow(w.c, w.ws, w.ws2 < 0 ? w.we : w.we2, true, lo, maxLen, capacity, part[(size_t)t]);
      }
    });
    for (int t = 0; t < nT; t++)
      for (int L = 0; L <= maxLen; L++) {
        recs[(size_t)L].insert(recs[(size_t)L].end(), part[(size_t)t][(size_t)L].begin(), part[(size_t)t][(size_t)L].end());
        std::vector<Rec>().swap(part[(size_t)t][(size_t)L]);
      }
    lastSplitWindows = nSplit.load(); lastWholeWindows = (int)wins.size() - nSplit.load(); lastSplitRefused = nWholeAfterAll.load();
    if (getenv("XM_TRACE_BUILD")) fprintf(stderr, "[xm] multi blocks near ambiguous bases: %zu windows (%d with the middle of a long run of N left out, %d such runs taken whole after all)\n",
                                          wins.size(), nSplit.load(), nWholeAfterAll.load());
  }
  mutable int lastSplitWindows = 0, lastWholeWindows = 0, lastSplitRefused = 0;

  // Hash every gapmer with minLen <= used <= maxLen and append tables [minLen..maxLen].  Tables below minInterestingSize and
  // tables that receive no record are the reference's PackedMap(1, 1) placeholders (M/HashBlock_Database.java:387-393).
  // set by the library when a GPU is there (xm_index_device.hip): the same tables, hashed, sorted and cut into CSR form on the device
  bool (*deviceHasher)(HostIndex&, int, int, int) = nullptr;
  int deviceForBuild = -1;
  int hasAmbiguity = -1;  // (cached) a contig holds a base code other than A C G T
  bool builtOnDevice = false;
  std::map<int, std::vector<int>> dupCandidates;  // (GPU build) per table: the hashcodes that can hold a duplication, ascending
  double hashSeconds = 0, dupSeconds = 0;
  bool referenceIsAmbiguous() {
    if (hasAmbiguity < 0) {
      hasAmbiguity = 0;
      for (uint8_t c : refCodes) if (bpIsAmbiguous(c)) { hasAmbiguity = 1; break; }
    }
    return hasAmbiguity != 0;
  }
  void hashLengths(int minLen, int maxLen) {
    if (deviceHasher && deviceForBuild >= 0) {  // (references with ambiguity codes too: the GPU hashes what lies clear of them, multiRecordsNearAmbiguity the rest)
      const char* e = getenv("XM_DEVICE_BUILD");  // 0: hash on the host even though a GPU is there
      if (!(e && *e && atoi(e) == 0) && deviceHasher(*this, minLen, maxLen, deviceForBuild)) { builtOnDevice = true; return; }
    }
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
    const char* hy = getenv("XM_BUILD_HYBRID_ON_HOST");
    const bool hybridOnHost = hy && *hy && atoi(hy) != 0;
    bool anyHybrid = false;
    for (int c = 0; c < numContigs(); c++) {
      SeqView seq = contigView(c, false);
      int n = seq.len;
      auto emit = [&](const HBlock& b, uint32_t fromMulti, std::vector<std::vector<Rec>>& recs) { emitBlockRecords(seq, c, b, fromMulti, lo, maxLen, capacity, recs); };
      bool ambiguous = false;
      for (int i = 0; i < n && !ambiguous; i++) if (bpIsAmbiguous(seq.base[i])) ambiguous = true;
      // XM_BUILD_HYBRID_ON_HOST=1 (test hook): the composition the GPU build uses for references with ambiguity codes, on the host - the plain
      // rule over the whole contig for the blocks that lie clear of the ambiguous bases + the multi blocks of the windows around them
      if (ambiguous && !hybridOnHost) {
        hashAmbiguousWindow(c, 0, n, false, lo, maxLen, capacity, recs);
        continue;
      }
      std::vector<uint32_t> ambPrefix;
      if (ambiguous) {
        ambPrefix.assign((size_t)n + 1, 0);
        for (int i = 0; i < n; i++) ambPrefix[(size_t)i + 1] = ambPrefix[(size_t)i] + (bpIsAmbiguous(seq.base[i]) ? 1u : 0u);
        anyHybrid = true;
      }
      // plain ACGT contig: every level is cut into consecutive parts, one per host thread; a part emits the records of its blocks and
      // merges its pairs (the pair that straddles two parts belongs to the left one); parts are concatenated in order
      std::vector<HBlock> cur((size_t)n), next;
      for (int i = 0; i < n; i++) cur[(size_t)i] = hblock0(seq.base[i], i);
      const int nT = buildThreads((size_t)n);
      std::vector<std::vector<std::vector<Rec>>> partRecs((size_t)nT, std::vector<std::vector<Rec>>((size_t)maxLen + 1));
      std::vector<std::vector<HBlock>> partNext((size_t)nT);
      std::vector<char> partShort((size_t)nT);
      while (!cur.empty()) {
        parallelParts(cur.size(), nT, [&](int t, size_t b, size_t e) {
          bool anyShort = false;
          std::vector<HBlock>& nx = partNext[(size_t)t];
          nx.clear();
          for (size_t i = b; i < e; i++) {
            const HBlock& blk = cur[i];
            if (blk.len <= maxLen) {  // (a longer block's gapmer uses at least blk.len bases)
              anyShort = true;
              if (ambPrefix.empty() || ambPrefix[(size_t)(blk.start + blk.len)] == ambPrefix[(size_t)blk.start]) emit(blk, 0, partRecs[(size_t)t]);
            }
            if (i + 1 < cur.size() && shouldMergeBlocks(blk, cur[i + 1])) nx.push_back(mergeBlocks(blk, cur[i + 1]));
          }
          partShort[(size_t)t] = anyShort ? 1 : 0;
        });
        bool anyShort = false;
        for (int t = 0; t < nT; t++) if (partShort[(size_t)t]) anyShort = true;
        if (!anyShort) break;  // (the level after a level without short blocks is never looked at)
        next.clear();
        for (int t = 0; t < nT; t++) next.insert(next.end(), partNext[(size_t)t].begin(), partNext[(size_t)t].end());
        cur.swap(next);
      }
      for (int t = 0; t < nT; t++)
        for (int L = 0; L <= maxLen; L++) {
          std::vector<Rec>& src = partRecs[(size_t)t][(size_t)L];
          recs[(size_t)L].insert(recs[(size_t)L].end(), src.begin(), src.end());
          std::vector<Rec>().swap(src);
        }
    }
    if (anyHybrid) multiRecordsNearAmbiguity(lo, maxLen, capacity, recs);
    if ((int)tables.size() < maxLen + 1) tables.resize((size_t)maxLen + 1);
    // every table on its own (sort, duplicate suppression, CSR), tables in parallel; then concatenated in order of L
    const int nTables = maxLen - minLen + 1;
    std::vector<std::vector<uint32_t>> tOff((size_t)nTables);
    std::vector<std::vector<uint64_t>> tPos((size_t)nTables);
    std::vector<Table> tHdr((size_t)nTables);
    size_t totalRecs = 0;
    for (int L = minLen; L <= maxLen; L++) totalRecs += recs[(size_t)L].size();
    const int nT2 = std::min(buildThreads(totalRecs), nTables);
    std::vector<int> order((size_t)nTables);
    for (int k = 0; k < nTables; k++) order[(size_t)k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return recs[(size_t)(minLen + a)].size() > recs[(size_t)(minLen + b)].size(); });  // large tables first
    std::vector<size_t> nextJob(1, 0);
    std::mutex jobMu;
    parallelParts((size_t)nT2, nT2, [&](int, size_t, size_t) {
      while (true) {
        size_t job;
        { std::lock_guard<std::mutex> lock(jobMu); job = nextJob[0]++; }
        if (job >= (size_t)nTables) break;
        const int k = order[job];
        const int L = minLen + k;
        Table t;
        std::vector<Rec>& v = recs[(size_t)L];
        if (v.empty()) { t.capacity = 1; t.maxCount = 1; }
        else { t.capacity = capacity[(size_t)L]; t.maxCount = maxCount[(size_t)L]; }
        // (bucket, position, single before multi): the flag is the top bit of pos, so plain order on (pos << 1 | flag)
        std::sort(v.begin(), v.end(), [](const Rec& a, const Rec& b) {
          if (a.bucket != b.bucket) return a.bucket < b.bucket;
          return ((a.pos << 1) | (a.pos >> 63)) < ((b.pos << 1) | (b.pos >> 63));
        });
        {  // PackedMap.add with preventDuplicates (:124-153): a record that comes from a multi block is not added when its bucket already holds
           // that position ([approximation, see DESIGN.md] "already" is taken as: among all single-block records and the earlier multi records)
          size_t w = 0;
          for (size_t r = 0; r < v.size(); r++) {
            const bool multi = (v[r].pos & REC_MULTI) != 0;
            const uint64_t pos = v[r].pos & ~REC_MULTI;
            if (multi && w > 0 && v[w - 1].bucket == v[r].bucket && v[w - 1].pos == pos) continue;
            v[w].bucket = v[r].bucket; v[w].pos = pos;
            w++;
          }
          v.resize(w);
        }
        std::vector<uint32_t>& off = tOff[(size_t)k];
        std::vector<uint64_t>& pos = tPos[(size_t)k];
        off.reserve((size_t)t.capacity + 1);
        size_t i = 0;
        uint64_t stored = 0;
        for (int b = 0; b < t.capacity; b++) {
          size_t j = i;
          while (j < v.size() && v[j].bucket == (uint32_t)b) j++;
          size_t cnt = j - i;
          if (stored + cnt > 0x7FFFFFFFull) throw std::runtime_error("table too large for 31-bit bucket offsets");
          if ((int64_t)cnt > (int64_t)t.maxCount) {
            off.push_back((uint32_t)stored | XM_OVERFULL);
          } else {
            off.push_back((uint32_t)stored);
            for (size_t x = i; x < j; x++) pos.push_back(v[x].pos);
            stored += cnt;
          }
          i = j;
        }
        off.push_back((uint32_t)stored);
        tHdr[(size_t)k] = t;
        std::vector<Rec>().swap(v);
      }
    });
    for (int k = 0; k < nTables; k++) {
      Table t = tHdr[(size_t)k];
      t.offBase = (int64_t)bucketOff.size();
      t.posBase = (int64_t)positions.size();
      bucketOff.insert(bucketOff.end(), tOff[(size_t)k].begin(), tOff[(size_t)k].end());
      positions.insert(positions.end(), tPos[(size_t)k].begin(), tPos[(size_t)k].end());
      tables[(size_t)(minLen + k)] = t;
      std::vector<uint32_t>().swap(tOff[(size_t)k]);
      std::vector<uint64_t>().swap(tPos[(size_t)k]);
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
    const bool trace = getenv("XM_TRACE_BUILD") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    hashLengths(0, want);
    maxHashedLength = want;
    auto t1 = std::chrono::steady_clock::now();
    detectDuplications();
    auto t2 = std::chrono::steady_clock::now();
    hashSeconds += std::chrono::duration<double>(t1 - t0).count();
    dupSeconds += std::chrono::duration<double>(t2 - t1).count();
    if (trace) fprintf(stderr, "[xm] index build: hashing + tables %.3f s, duplication map %.3f s\n", std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
  }
  void ensureLength(int length) {
    if (length <= maxHashedLength) return;
    auto t0 = std::chrono::steady_clock::now();
    hashLengths(maxHashedLength + 1, length);
    maxHashedLength = length;
    hashSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }

  // ---- binary cache (in the spirit of --cache-dir: M/DirCache.java:19-60, M/HashBlock_Database.java:106-114,477-487, M/PackedMap.java:249-279).
  // One file holds the reference, every table hashed so far and the duplication map; the reference's own cache keys (enableGapmers,
  // minInterestingSize, maxNumShortMatches, formatVersion) and the duplication settings are in its header and must match on load.
  static constexpr uint64_t CACHE_MAGIC = 0x3158444958504D58ull;  // "XMPXIDX1"
  static constexpr uint32_t CACHE_FORMAT = 1;
  struct CacheHeader {
    uint64_t magic; uint32_t format; int32_t enableGapmers, minInterestingSize, maxNumShortMatches, maxHashedLength;
    int32_t dupWindow, dupMinCopies, dupMinLength, dupMaxLength, dupDone; int64_t totalForwardSize; uint64_t referenceDigest;
  };
  static uint64_t fnv(const void* data, size_t n, uint64_t h = 0xCBF29CE484222325ull) {
    const uint8_t* b = (const uint8_t*)data;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001B3ull; }
    return h;
  }
  uint64_t referenceDigest() const {
    uint64_t h = fnv(refCodes.data(), refCodes.size());
    for (size_t c = 0; c < names.size(); c++) { h = fnv(names[c].data(), names[c].size() + 1, h); h = fnv(&contigLen[c], sizeof(int32_t), h); }
    return h;
  }
  struct CacheWriter {
    FILE* f;
    void raw(const void* p, size_t n) { if (n && fwrite(p, 1, n, f) != n) throw std::runtime_error("index cache: write failed"); }
    template <typename T> void vec(const std::vector<T>& v) { uint64_t n = v.size(); raw(&n, 8); raw(v.data(), v.size() * sizeof(T)); }
  };
  struct CacheReader {
    FILE* f;
    void raw(void* p, size_t n) { if (n && fread(p, 1, n, f) != n) throw std::runtime_error("index cache: file is truncated"); }
    template <typename T> void vec(std::vector<T>& v, uint64_t limit) {
      uint64_t n = 0; raw(&n, 8);
      if (n > limit) throw std::runtime_error("index cache: file is corrupt");
      v.resize((size_t)n); raw(v.data(), (size_t)n * sizeof(T));
    }
  };
  void save(const std::string& path) const {
    // written beside the target and renamed, so that ranks that build the same index at once never read a half-written file
    const std::string tmp = path + ".tmp." + std::to_string((long long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) throw std::runtime_error("index cache: cannot create " + tmp);
    try {
      CacheWriter w{f};
      CacheHeader h{CACHE_MAGIC, CACHE_FORMAT, enableGapmers, minInterestingSize, maxNumShortMatches, maxHashedLength, dupWindow, dupMinCopies, dupMinLength, dupMaxLength,
                    dupDone ? 1 : 0, totalForwardSize, referenceDigest()};
      w.raw(&h, sizeof(h));
      uint64_t n = names.size(); w.raw(&n, 8);
      for (const std::string& s : names) { uint64_t l = s.size(); w.raw(&l, 8); w.raw(s.data(), s.size()); }
      w.vec(contigStart); w.vec(contigLen); w.vec(seqCumStart); w.vec(refCodes); w.vec(tables); w.vec(bucketOff); w.vec(positions); w.vec(dupKeyStart); w.vec(dupKeys);
      const uint64_t tail[2] = {CACHE_MAGIC, (uint64_t)ftell(f)};
      w.raw(tail, sizeof(tail));
      if (fclose(f) != 0) { f = nullptr; throw std::runtime_error("index cache: write failed"); }
      f = nullptr;
      if (rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("index cache: cannot rename to " + path);
    } catch (...) {
      if (f) fclose(f);
      remove(tmp.c_str());
      throw;
    }
  }
  void load(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("index cache: cannot open " + path);
    try {
      fseek(f, 0, SEEK_END);
      const uint64_t fileBytes = (uint64_t)ftell(f);
      fseek(f, 0, SEEK_SET);
      CacheReader r{f};
      CacheHeader h;
      r.raw(&h, sizeof(h));
      if (h.magic != CACHE_MAGIC) throw std::runtime_error("index cache: not an index file");
      if (h.format != CACHE_FORMAT) throw std::runtime_error("index cache: format version " + std::to_string(h.format) + " (this library reads " + std::to_string(CACHE_FORMAT) + ")");
      enableGapmers = h.enableGapmers; minInterestingSize = h.minInterestingSize; maxNumShortMatches = h.maxNumShortMatches; maxHashedLength = h.maxHashedLength;
      dupWindow = h.dupWindow; dupMinCopies = h.dupMinCopies; dupMinLength = h.dupMinLength; dupMaxLength = h.dupMaxLength; dupDone = h.dupDone != 0; totalForwardSize = h.totalForwardSize;
      uint64_t n = 0; r.raw(&n, 8);
      if (n > fileBytes) throw std::runtime_error("index cache: file is corrupt");
      names.resize((size_t)n);
      for (std::string& s : names) { uint64_t l = 0; r.raw(&l, 8); if (l > fileBytes) throw std::runtime_error("index cache: file is corrupt"); s.resize((size_t)l); r.raw(&s[0], (size_t)l); }
      r.vec(contigStart, fileBytes); r.vec(contigLen, fileBytes); r.vec(seqCumStart, fileBytes); r.vec(refCodes, fileBytes); r.vec(tables, fileBytes); r.vec(bucketOff, fileBytes);
      r.vec(positions, fileBytes); r.vec(dupKeyStart, fileBytes); r.vec(dupKeys, fileBytes);
      uint64_t tail[2] = {0, 0};
      const uint64_t at = (uint64_t)ftell(f);
      r.raw(tail, sizeof(tail));
      if (tail[0] != CACHE_MAGIC || tail[1] != at) throw std::runtime_error("index cache: file is truncated");
      fclose(f); f = nullptr;
      if (contigLen.size() != names.size() || contigStart.size() != names.size() || seqCumStart.size() != names.size() * 2 + 1 || (int)tables.size() != maxHashedLength + 1 ||
          h.referenceDigest != referenceDigest())
        throw std::runtime_error("index cache: file is corrupt");
      for (const Table& t : tables)
        if (t.capacity < 1 || t.offBase < 0 || t.posBase < 0 || (uint64_t)t.offBase + (uint64_t)t.capacity + 1 > bucketOff.size() || (uint64_t)t.posBase > positions.size())
          throw std::runtime_error("index cache: file is corrupt");
      validateTables();
    } catch (...) {
      if (f) fclose(f);
      throw;
    }
  }
  // Everything the kernels index with values taken from the file is bounds-checked here, once, on load: a damaged file (or a foreign one
  // over the same reference) must fail in xm_index_load, not read outside positions[] / refCodes[] on the GPU.
  void validateTables() const {
    const auto bad = [](const char* what) { throw std::runtime_error(std::string("index cache: file is corrupt (") + what + ")"); };
    if (seqCumStart.empty() || seqCumStart[0] != 0) bad("sequence starts");
    for (size_t i = 0; i + 1 < seqCumStart.size(); i++)
      if (seqCumStart[i + 1] - seqCumStart[i] != (int64_t)contigLen[i >> 1]) bad("sequence starts");
    uint64_t refTotal = 0;
    for (size_t c = 0; c < contigLen.size(); c++) {
      if (contigLen[c] < 0 || (uint64_t)contigStart[c] != refTotal) bad("contig starts");
      refTotal += (uint64_t)contigLen[c];
    }
    if (refTotal != refCodes.size() || totalForwardSize != (int64_t)refTotal) bad("contig lengths");
    const uint64_t encodedEnd = (uint64_t)seqCumStart.back();
    const int nT = buildThreads(positions.size() + bucketOff.size());
    for (size_t L = 0; L < tables.size(); L++) {
      const Table& t = tables[L];
      if (t.maxCount < 0) bad("table header");
      const uint64_t stored = bucketOff[(size_t)(t.offBase + t.capacity)] & ~XM_OVERFULL;
      if ((bucketOff[(size_t)(t.offBase + t.capacity)] & XM_OVERFULL) || (uint64_t)t.posBase + stored > positions.size()) bad("table extent");
      std::vector<int> wrong((size_t)nT, 0);
      parallelParts((size_t)t.capacity, nT, [&](int ti, size_t b, size_t e) {
        for (size_t k = b; k < e; k++) {
          const uint32_t o0 = bucketOff[(size_t)t.offBase + k] & ~XM_OVERFULL, o1 = bucketOff[(size_t)t.offBase + k + 1] & ~XM_OVERFULL;
          if (o1 < o0 || o1 > stored) { wrong[(size_t)ti] = 1; return; }
        }
      });
      for (int w : wrong) if (w) bad("bucket offsets");
    }
    {
      std::vector<int> wrong((size_t)nT, 0);
      parallelParts(positions.size(), nT, [&](int ti, size_t b, size_t e) {
        for (size_t k = b; k < e; k++) if (positions[k] >= encodedEnd) { wrong[(size_t)ti] = 1; return; }
      });
      for (int w : wrong) if (w) bad("positions");
    }
    if (dupDone) {
      if (dupKeyStart.size() != names.size() + 1 || dupKeyStart[0] != 0 || (uint64_t)dupKeyStart.back() != dupKeys.size()) bad("duplication keys");
      for (size_t c = 0; c + 1 < dupKeyStart.size(); c++) {
        if (dupKeyStart[c + 1] < dupKeyStart[c]) bad("duplication keys");
        for (int64_t k = dupKeyStart[c]; k < dupKeyStart[c + 1]; k++)
          if (dupKeys[(size_t)k] < 0 || dupKeys[(size_t)k] >= contigLen[c] || (k > dupKeyStart[c] && dupKeys[(size_t)k] <= dupKeys[(size_t)k - 1])) bad("duplication keys");
      }
    } else if (!dupKeys.empty()) bad("duplication keys");
  }
  // does this (loaded) index answer a build request for `other`'s reference with these settings?  (same resolution of defaults as build())
  bool matchesRequest(const HostIndex& other, int enableGapmers_, int minInteresting, int dupWindow_, int dupMinCopies_, int dupMinLen, int dupMaxLen) const {
    if (names != other.names || contigLen != other.contigLen || refCodes != other.refCodes) return false;
    HostIndex probe;
    probe.totalForwardSize = other.totalForwardSize;
    const int wantMin = minInteresting > 0 ? minInteresting : j2i(std::max((std::log((double)(other.totalForwardSize + 1)) / std::log(4.0)) - 2, 1.0));
    return enableGapmers == (enableGapmers_ ? 1 : 0) && minInterestingSize == wantMin && dupWindow == (dupWindow_ > 0 ? dupWindow_ : 1000) &&
           dupMinCopies == (dupMinCopies_ > 0 ? dupMinCopies_ : 2) && dupMinLength == (dupMinLen > 0 ? dupMinLen : probe.chooseMinDuplicationLength()) &&
           dupMaxLength == (dupMaxLen > 0 ? dupMaxLen : probe.chooseMaxDuplicationLength());
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
      // the reference walks every hashcode of the table in order; nearly all of them hold fewer than dupMinCopies positions, so the walk
      // itself is done by all host threads (each over a slice, slices concatenated in order) and only the hashcodes that can hold a
      // duplication go through the ordered part below.  saveDuplications runs after every 10000th hashcode and after the last one: with
      // nothing pending it does nothing, so it is enough to run it whenever the next candidate lies past such a boundary.
      std::vector<int> candidates;
      auto given = dupCandidates.find(L);
      if (given != dupCandidates.end()) {
        candidates.swap(given->second);  // the GPU build has looked at every bucket of this table already (xmb_dup_candidates_kernel)
      } else {
        const int nT = buildThreads((size_t)t.capacity);
        std::vector<std::vector<int>> part((size_t)nT);
        parallelParts((size_t)t.capacity, nT, [&](int ti, size_t b, size_t e) {
          for (size_t hc = b; hc < e; hc++) {
            int64_t first = 0;
            if (bucketGet(L, (uint32_t)hc, first) >= dupMinCopies) part[(size_t)ti].push_back((int)hc);
          }
        });
        for (auto& v : part) candidates.insert(candidates.end(), v.begin(), v.end());
      }
      // what a candidate hashcode contributes (its positions grouped by text) does not depend on the others: computed by all threads;
      // the contributions are then applied in hashcode order
      struct Found { size_t ci; size_t seq; int start; Dup d; };
      std::vector<Found> found;
      {
        const int nT = buildThreads(candidates.size() * 16);
        std::vector<std::vector<Found>> part((size_t)nT);
        parallelParts(candidates.size(), nT, [&](int ti, size_t cb, size_t ce) {
          struct P { int contig; bool rc; int start; };
          std::vector<P> matches;
          for (size_t ci = cb; ci < ce; ci++) {
            int64_t first = 0;
            const int cnt = bucketGet(L, (uint32_t)candidates[ci], first);  // lookupByForwardHash: packed key of `hashcode` is itself; numForwardMatches = matches.length / 2
            matches.clear();
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
              bool ambiguousText = false;  // :75 isAmbiguousText: positions whose prefix or suffix has a non-ACGT base are left out
              for (char ch : text) if (bpIsAmbiguous((uint8_t)ch)) ambiguousText = true;
              if (ambiguousText) continue;
              std::vector<P>& g = byText[text];
              bool dupPos = false;                               // removeDuplicatePositions
              for (const P& q : g) if (q.contig == p.contig && q.rc == p.rc && q.start == p.start) { dupPos = true; break; }
              if (!dupPos) g.push_back(p);
            }
            for (auto& e : byText) {
              if ((int)e.second.size() < dupMinCopies) continue;
              Dup d{L, (int)e.second.size()};
              for (const P& p : e.second) part[(size_t)ti].push_back(Found{ci, (size_t)p.contig * 2 + (p.rc ? 1 : 0), p.start, d});
            }
          }
        });
        for (auto& v : part) found.insert(found.end(), v.begin(), v.end());
      }
      size_t fi = 0;
      for (size_t ci = 0; ci < candidates.size(); ci++) {
        const int hashcode = candidates[ci];
        for (; fi < found.size() && found[fi].ci == ci; fi++) pending[found[fi].seq][found[fi].start] = found[fi].d;
        if (ci + 1 == candidates.size() || candidates[ci + 1] / 10000 != hashcode / 10000) {  // saveDuplications :332-400 (every 10000 hashcodes, and at the end)
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
    dupCandidates.clear();
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
