// ORACLE — test infrastructure only (see xmo_types.h).  C entry points for tests/ (ctypes), smoke() and the
// cpu_baseline leg of bench.py.  The result streams use the same layout as include/xmapper_hip.h so that
// parity tests compare the product's output with the oracle's byte for byte.
#include "xmo_worker.h"
#include <thread>
#include <atomic>
#include <cstdlib>
#include <cstdio>

using namespace xmo;

namespace {

struct ParamsIn {  // mirrors xm_params in include/xmapper_hip.h
  double MutationPenalty, InsertionStart_Penalty, InsertionExtension_Penalty, DeletionStart_Penalty, DeletionExtension_Penalty,
      MaxErrorRate, UnalignedPenalty, AmbiguityPenalty, Max_PenaltySpan;
  int32_t MaxNumMatches;
  int32_t reserved;
};

AlignmentParameters toParams(const ParamsIn* p) {
  AlignmentParameters a;
  a.MutationPenalty = p->MutationPenalty;
  a.InsertionStart_Penalty = p->InsertionStart_Penalty;
  a.InsertionExtension_Penalty = p->InsertionExtension_Penalty;
  a.DeletionStart_Penalty = p->DeletionStart_Penalty;
  a.DeletionExtension_Penalty = p->DeletionExtension_Penalty;
  a.MaxErrorRate = p->MaxErrorRate;
  a.UnalignedPenalty = p->UnalignedPenalty;
  a.AmbiguityPenalty = p->AmbiguityPenalty;
  a.Max_PenaltySpan = p->Max_PenaltySpan;
  a.MaxNumMatches = p->MaxNumMatches;
  return a;
}

struct ResultStreams {
  std::vector<int32_t> ints;
  std::vector<double> dbls;
  std::vector<int64_t> intOff, dblOff;  // nq + 1
};

void appendResult(ResultStreams& rs, const QueryAlignments& qa) {
  rs.ints.push_back((int32_t)qa.components.size());
  for (auto& comp : qa.components) {
    rs.ints.push_back((int32_t)comp.size());
    for (auto& al : comp) {
      rs.ints.push_back(al->innerDistance);
      rs.ints.push_back((int32_t)al->components.size());
      rs.dbls.push_back(al->spacingPenalty);
      rs.dbls.push_back(al->overlapMultiplier);
      rs.dbls.push_back(al->duplicationBonus);
      rs.dbls.push_back(al->totalPenalty);
      for (auto& sa : al->components) {
        rs.ints.push_back(sa->getSequenceB()->contigIndex);
        rs.ints.push_back(sa->referenceReversed ? 1 : 0);
        rs.ints.push_back((int32_t)sa->sections.size());
        for (auto& b : sa->sections) {
          rs.ints.push_back(b.startIndexA); rs.ints.push_back(b.startIndexB); rs.ints.push_back(b.lengthA); rs.ints.push_back(b.lengthB);
        }
        rs.dbls.push_back(sa->totalPenalty);
        rs.dbls.push_back(sa->alignedPenalty);
      }
    }
  }
}

struct BatchIn {
  int64_t nq;
  const int32_t* mateCount;     // [nq] 1 or 2
  const int64_t* mateOffset;    // [nq*2] offset of each mate into codes
  const int32_t* mateLength;    // [nq*2]
  const uint8_t* codes;         // 4-bit IUPAC codes, one per byte
  const double* expectedInner;  // [nq]
  const double* deviation;      // [nq]
};

QueryAlignments alignOne(AlignerWorker& w, const BatchIn& in, int64_t q) {
  AlignerWorker::QueryContext ctx;
  for (int m = 0; m < in.mateCount[q]; m++) {
    int64_t off = in.mateOffset[q * 2 + m];
    int32_t len = in.mateLength[q * 2 + m];
    std::vector<uint8_t> codes(in.codes + off, in.codes + off + len);
    ctx.mates.emplace_back(new QuerySequence("q", codes));
    ctx.query.sequences.push_back(ctx.mates.back()->fwd.get());
  }
  if (in.mateCount[q] > 1) {
    ctx.query.expectedInnerDistance = in.expectedInner[q];
    ctx.query.spacingDeviationPerUnitPenalty = in.deviation[q];
  }
  return w.alignToAncestralReference(ctx);
}

void addCounters(Counters& a, const Counters& b) {
  a.reads += b.reads; a.headerProbes += b.headerProbes; a.bucketFetches += b.bucketFetches; a.hitsFetched += b.hitsFetched;
  a.flankChecks += b.flankChecks; a.candidatesExtended += b.candidatesExtended; a.pathAlignerCalls += b.pathAlignerCalls;
  a.pathAlignerNodes += b.pathAlignerNodes; a.quickAccepts += b.quickAccepts;
  a.pathNullSearches += b.pathNullSearches; a.pathNullNodes += b.pathNullNodes; a.pathBoundRejects += b.pathBoundRejects;
  a.pathBoundRejectNodes += b.pathBoundRejectNodes; a.pathBoundChecks += b.pathBoundChecks;
  a.pieceChecks += b.pieceChecks; a.pieceRejects += b.pieceRejects; a.skippedCalls += b.skippedCalls; a.skippedNodes += b.skippedNodes;
}

thread_local std::string g_error;

}  // namespace

extern "C" {

const char* xmo_last_error() { return g_error.c_str(); }

void* xmo_ref_new() { return new ReferenceDatabase(); }
void xmo_ref_free(void* r) { delete (ReferenceDatabase*)r; }
int xmo_ref_add_contig(void* r, const char* name, const char* text) {
  ((ReferenceDatabase*)r)->sequences.addForward(makeSequence(name, text));
  return 0;
}
int xmo_ref_add_contig_codes(void* r, const char* name, const uint8_t* codes, int64_t len) {
  std::unique_ptr<Sequence> s(new Sequence());
  s->name = name;
  s->codes.assign(codes, codes + len);
  ((ReferenceDatabase*)r)->sequences.addForward(std::move(s));
  return 0;
}
// mode 0: Mapper.run assembly, mode 1: Api.newDatabase assembly
int xmo_ref_finish(void* r, int mode, int enableGapmers) {
  try { ((ReferenceDatabase*)r)->finish(mode == 1, enableGapmers != 0); return 0; } catch (std::exception& e) { g_error = e.what(); return 1; }
}
int xmo_ref_finish_min_interesting(void* r, int mode, int enableGapmers, int minInterestingSize) {
  try { ((ReferenceDatabase*)r)->finish(mode == 1, enableGapmers != 0, minInterestingSize); return 0; } catch (std::exception& e) { g_error = e.what(); return 1; }
}
int xmo_ref_finish_custom_dup(void* r, int minDup, int maxDup, int copies, int window) {
  try { ((ReferenceDatabase*)r)->finishCustomDup(minDup, maxDup, copies, window); return 0; } catch (std::exception& e) { g_error = e.what(); return 1; }
}
int xmo_ref_require_size(void* r, int size) {
  try { ((ReferenceDatabase*)r)->hashblockDatabase->requireSetUpThroughSize(size); return 0; } catch (std::exception& e) { g_error = e.what(); return 1; }
}

// ---- index inspection (parity of the product's index builder)
int xmo_index_info(void* r, int* minInteresting, int* maxFullySetUp) {
  HashBlock_Database* db = ((ReferenceDatabase*)r)->hashblockDatabase.get();
  *minInteresting = db->minInterestingSize;
  *maxFullySetUp = db->maxFullySetUpSize;
  return 0;
}
int xmo_index_table_info(void* r, int L, int* capacity, int* maxCountPerKey, int64_t* numStored, int64_t* numOverfull) {
  HashBlock_Database* db = ((ReferenceDatabase*)r)->hashblockDatabase.get();
  if (L < 0 || L >= (int)db->hashedBlocks.size() || !db->hashedBlocks[(size_t)L]) return 1;
  PackedMap* m = db->hashedBlocks[(size_t)L].get();
  *capacity = m->keyCapacity;
  *maxCountPerKey = m->maxInterestingCountPerKey;
  int64_t n = 0, o = 0;
  for (int k = 0; k < m->keyCapacity; k++) { if (m->overfull[(size_t)k]) o++; else n += (int64_t)m->buckets[(size_t)k].size(); }
  *numStored = n;
  *numOverfull = o;
  return 0;
}
// counts[capacity]: number of positions per bucket or -1 when overfull; positions[numStored]: concatenated in bucket order
int xmo_index_table_dump(void* r, int L, int32_t* counts, int64_t* positions) {
  HashBlock_Database* db = ((ReferenceDatabase*)r)->hashblockDatabase.get();
  PackedMap* m = db->hashedBlocks[(size_t)L].get();
  int64_t w = 0;
  for (int k = 0; k < m->keyCapacity; k++) {
    if (m->overfull[(size_t)k]) { counts[k] = -1; continue; }
    counts[k] = (int32_t)m->buckets[(size_t)k].size();
    for (int64_t e : m->buckets[(size_t)k]) positions[w++] = e;
  }
  return 0;
}
int64_t xmo_dup_keys(void* r, int contig, int32_t* out, int64_t cap) {
  ReferenceDatabase* ref = (ReferenceDatabase*)r;
  std::vector<int> keys = ref->duplicationDetector->keysOnSequence(ref->sequences.forward(contig));
  for (int64_t i = 0; i < (int64_t)keys.size() && i < cap; i++) out[i] = keys[(size_t)i];
  return (int64_t)keys.size();
}
double xmo_dup_granularity(void* r) { return ((ReferenceDatabase*)r)->duplicationDetector->getDetectionGranularity(); }

// ---- alignment
// the observer of the product's rejection filter (xmo_extend.h PathAligner::boundObserve): 1 = every PathAligner search is also put to the filter's bound
// (counters 9-17 of a result: null searches, their nodes, searches the filter rejects, the reference's nodes in those, searches the filter takes - all outside rejected
// pieces - then pieces the piece-level filter takes, pieces it rejects, PathAligner calls and nodes the reference spent inside rejected pieces)
void xmo_observe_bound(int on) { PathAligner::boundObserver() = on; }

struct xmo_result {
  int64_t nq, nInts, nDbls;
  int32_t* ints; double* dbls; int64_t* intOff; int64_t* dblOff;
  int64_t counters[24];
};

void xmo_result_free(xmo_result* res) {
  if (!res) return;
  free(res->ints); free(res->dbls); free(res->intOff); free(res->dblOff);
  delete res;
}

static xmo_result* packResult(std::vector<ResultStreams>& parts, int64_t nq, const Counters& c) {
  xmo_result* res = new xmo_result();
  res->nq = nq;
  int64_t ni = 0, nd = 0;
  for (auto& p : parts) { ni += (int64_t)p.ints.size(); nd += (int64_t)p.dbls.size(); }
  res->nInts = ni; res->nDbls = nd;
  res->ints = (int32_t*)malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(ni, 1));
  res->dbls = (double*)malloc(sizeof(double) * (size_t)std::max<int64_t>(nd, 1));
  res->intOff = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nq + 1));
  res->dblOff = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nq + 1));
  int64_t wi = 0, wd = 0, q = 0;
  for (auto& p : parts) {
    for (size_t k = 0; k + 1 < p.intOff.size(); k++) {
      res->intOff[q] = wi + p.intOff[k];
      res->dblOff[q] = wd + p.dblOff[k];
      q++;
    }
    if (!p.ints.empty()) memcpy(res->ints + wi, p.ints.data(), p.ints.size() * sizeof(int32_t));
    if (!p.dbls.empty()) memcpy(res->dbls + wd, p.dbls.data(), p.dbls.size() * sizeof(double));
    wi += (int64_t)p.ints.size();
    wd += (int64_t)p.dbls.size();
  }
  res->intOff[nq] = wi;
  res->dblOff[nq] = wd;
  memset(res->counters, 0, sizeof(res->counters));
  res->counters[0] = c.reads; res->counters[1] = c.headerProbes; res->counters[2] = c.bucketFetches; res->counters[3] = c.hitsFetched;
  res->counters[4] = c.flankChecks; res->counters[5] = c.candidatesExtended; res->counters[6] = c.pathAlignerCalls;
  res->counters[7] = c.pathAlignerNodes; res->counters[8] = c.quickAccepts;
  res->counters[9] = c.pathNullSearches; res->counters[10] = c.pathNullNodes; res->counters[11] = c.pathBoundRejects;
  res->counters[12] = c.pathBoundRejectNodes; res->counters[13] = c.pathBoundChecks;
  res->counters[14] = c.pieceChecks; res->counters[15] = c.pieceRejects; res->counters[16] = c.skippedCalls; res->counters[17] = c.skippedNodes;
  return res;
}

// Aligns nq queries with `threads` worker threads (each owns a Readable view, like one AlignerWorker per
// JVM thread, M/Mapper.java:1026-1040).  Queries are handed out in jobs of <= 50,000 bases (M/Mapper.java:926).
xmo_result* xmo_align_batch(void* r, const void* paramsIn, int threads, int64_t nq, const int32_t* mateCount, const int64_t* mateOffset,
                            const int32_t* mateLength, const uint8_t* codes, const double* expectedInner, const double* deviation) {
  ReferenceDatabase* ref = (ReferenceDatabase*)r;
  AlignmentParameters params = toParams((const ParamsIn*)paramsIn);
  BatchIn in{nq, mateCount, mateOffset, mateLength, codes, expectedInner, deviation};
  try {
    // jobs
    std::vector<std::pair<int64_t, int64_t>> jobs;
    {
      int64_t start = 0, bases = 0;
      for (int64_t q = 0; q < nq; q++) {
        for (int m = 0; m < mateCount[q]; m++) bases += mateLength[q * 2 + m];
        if (bases >= 50000) { jobs.push_back(std::make_pair(start, q + 1)); start = q + 1; bases = 0; }
      }
      if (start < nq) jobs.push_back(std::make_pair(start, nq));
    }
    if (threads < 1) threads = 1;
    if (threads > 1) {
      // the lazily growing index is not thread-safe in this restatement: grow it up front (tables are a pure
      // function of the reference, so this cannot change results)
      int maxLen = 1;
      for (int64_t q = 0; q < nq; q++) for (int m = 0; m < mateCount[q]; m++) maxLen = std::max(maxLen, mateLength[q * 2 + m]);
      ref->duplicationDetector->detect();
      ref->hashblockDatabase->requireSetUpThroughSize(maxLen);
    }
    std::vector<ResultStreams> parts(jobs.size());
    std::atomic<size_t> nextJob(0);
    std::vector<Counters> ctrs((size_t)threads);
    std::string err;
    std::atomic<bool> failed(false);
    auto work = [&](int t) {
      try {
        AlignerWorker w(ref, params);
        while (true) {
          size_t j = nextJob.fetch_add(1);
          if (j >= jobs.size() || failed.load()) break;
          ResultStreams& rs = parts[j];
          for (int64_t q = jobs[j].first; q < jobs[j].second; q++) {
            rs.intOff.push_back((int64_t)rs.ints.size());
            rs.dblOff.push_back((int64_t)rs.dbls.size());
            appendResult(rs, alignOne(w, in, q));
          }
          rs.intOff.push_back((int64_t)rs.ints.size());
          rs.dblOff.push_back((int64_t)rs.dbls.size());
        }
        ctrs[(size_t)t] = w.counters;
      } catch (std::exception& e) {
        if (!failed.exchange(true)) err = e.what();
      }
    };
    if (threads == 1) work(0);
    else {
      std::vector<std::thread> pool;
      for (int t = 0; t < threads; t++) pool.emplace_back(work, t);
      for (auto& th : pool) th.join();
    }
    if (failed.load()) { g_error = "Failed to align: " + err; return nullptr; }
    Counters total;
    for (auto& c : ctrs) addCounters(total, c);
    return packResult(parts, nq, total);
  } catch (std::exception& e) {
    g_error = e.what();
    return nullptr;
  }
}

// ---- component-level known-answer entry points (T/PathAligner_Test.java, T/HashBlockAligner_Test.java, ...)
// chain 0: PathAligner alone; chain 1: HashBlock_Aligner -> StraightAligner -> PathAligner_Runner
int xmo_kat_local_align(int chain, const char* queryText, const char* refText, const void* paramsIn, double maxInsExt, double maxDelExt,
                        char* alignedA, char* alignedB, int cap, double* penalty) {
  try {
    AlignmentParameters params = toParams((const ParamsIn*)paramsIn);
    std::unique_ptr<Sequence> a = makeSequence("a", queryText), b = makeSequence("b", refText);
    AlignmentAnalysis analysis;
    analysis.maxInsertionExtensionPenalty = maxInsExt;
    analysis.maxDeletionExtensionPenalty = maxDelExt;
    SequenceSection qs(a.get(), 0, a->getLength()), rs(b.get(), 0, b->getLength());
    SequenceAlignmentP result;
    if (chain == 0) { PathAligner pa; result = pa.align(qs, rs, params, analysis); }
    else { PathAligner_Runner r0; StraightAligner s(&r0); HashBlock_Aligner h(&s); result = h.align(qs, rs, params, analysis); }
    if (!result) return 1;
    std::string ta = result->getAlignedTextA(), tb = result->getAlignedTextB();
    if ((int)ta.size() + 1 > cap || (int)tb.size() + 1 > cap) return 2;
    strcpy(alignedA, ta.c_str());
    strcpy(alignedB, tb.c_str());
    *penalty = result->getPenalty();
    return 0;
  } catch (std::exception& e) { g_error = e.what(); return 3; }
}

// The observer of the product's rejection filter on one problem: PathAligner.align on query[startA, endA) - of the reverse complement of the query when queryRc -
// against reference[startB, endB), with the observer's verdict beside it.  out4: verdict (0 not taken, 1 taken, 2 rejected), 1 if the search returned an
// alignment, nodes the search put, searches that returned null.  Returns non-zero when the search or the observer threw (a rejected search that aligned).
int xmo_kat_bound(const void* paramsIn, const uint8_t* queryCodes, int queryLength, int queryRc, int startA, int endA, const uint8_t* refCodes, int refLength, int startB, int endB,
                  int predictedBestOffset, int64_t* out4) {
  try {
    AlignmentParameters params = toParams((const ParamsIn*)paramsIn);
    std::unique_ptr<Sequence> fwd(new Sequence()), b(new Sequence());
    fwd->name = "q"; fwd->codes.assign(queryCodes, queryCodes + queryLength);
    b->name = "r"; b->codes.assign(refCodes, refCodes + refLength);
    std::unique_ptr<Sequence> rc = makeReverseComplement(*fwd);
    const Sequence* a = queryRc ? rc.get() : fwd.get();
    AlignmentAnalysis analysis;
    analysis.predictedBestOffset = predictedBestOffset;
    // (the limits the chain would have derived: with the constructor's 1e6 a window at a contig end gets two million start nodes, PathAligner.java:144)
    analysis.maxInsertionExtensionPenalty = analysis.maxDeletionExtensionPenalty = std::max(1.0, (endA - startA) * params.MaxErrorRate);
    SequenceSection qs(a, startA, endA), rs(b.get(), startB, endB);
    Counters c;
    PathAligner pa;
    pa.counters = &c;
    const int was = PathAligner::boundObserver();
    PathAligner::boundObserver() = 1;
    SequenceAlignmentP result;
    try { result = pa.align(qs, rs, params, analysis); } catch (...) { PathAligner::boundObserver() = was; throw; }
    PathAligner::boundObserver() = was;
    out4[0] = c.pathBoundRejects ? 2 : (c.pathBoundChecks ? 1 : 0);
    out4[1] = result ? 1 : 0;
    out4[2] = c.pathAlignerNodes;
    out4[3] = c.pathNullSearches;
    return 0;
  } catch (std::exception& e) { g_error = e.what(); return 3; }
}

// T/BasepairsTest.java:9-45: AlignmentParameters.getPenalty(byte, byte) on IUPAC letters
double xmo_kat_base_penalty(char a, char b, double mutationPenalty, double ambiguityPenalty) {
  AlignmentParameters p;
  p.MutationPenalty = mutationPenalty;
  p.AmbiguityPenalty = ambiguityPenalty;
  return p.getPenalty(Basepairs::encode(a), Basepairs::encode(b));
}

// T/MultiHashBlock_Test.java:84-133 (checkExpandingAmbiguitiesInto): hashing `ambiguous` must offer, among the possibilities of the blocks that start
// at 0 and end at the end of the text, the block that `text` hashes to (same start, end and forward hash).  0 = it does; 1 = it does not;
// 2 = `text` itself has no single block over its whole length (the reference's test then skips, or fails with "Not a hashblock")
static void katHashString(const Sequence* sequence, std::vector<HashBlock>& results) {  // hashString :135-171
  HashBlock_Stream stream(sequence, true, nullptr);
  while (true) {
    std::shared_ptr<HashBlock_Row> row = stream.getNextBatch();
    if (!row) break;
    MultiBlockP block = row->get(0);
    if (!block) break;
    for (auto& c : block->getPossibilities())
      if (c.hasBlock && c.block.getEndIndex() == sequence->getLength()) results.push_back(c.block);
  }
}
int xmo_kat_multi_contains(const char* text, const char* ambiguous) {
  try {
    std::unique_ptr<Sequence> a = makeSequence("q", text), b = makeSequence("q", ambiguous);
    std::vector<HashBlock> options, expanded;
    katHashString(a.get(), options);
    if (options.size() != 1) return 2;
    katHashString(b.get(), expanded);
    for (const HashBlock& p : expanded)
      if (p.getStartIndex() == options[0].getStartIndex() && p.getEndIndex() == options[0].getEndIndex() && p.getForwardHash() == options[0].getForwardHash()) return 0;
    return 1;
  } catch (std::exception& e) { g_error = e.what(); return 3; }
}

// T/SequenceDatabase_Test.java:16-42,118-132: encodePosition / decodePosition round trips on numSequences repeating sequences of
// sequenceLength - i bases (positions 0, 100, length - 100, length - 1 of every sequence).  0 = all round trips hold.
int xmo_kat_position_codec(int numSequences, int sequenceLength) {
  try {
    SequenceDatabase db;
    for (int i = 0; i < numSequences; i++) {
      std::unique_ptr<Sequence> s(new Sequence());
      s->name = "seq" + std::to_string(i);
      s->repeatedLength = sequenceLength - i;
      db.addForward(std::move(s));
    }
    for (const Sequence* s : db.all) {
      const int n = s->getLength();
      for (int position : {0, 100, n - 100, n - 1}) {
        if (position < 0 || position >= n) continue;
        const SequencePosition d = db.decodePosition(db.encodePosition(s, position));
        if (d.sequence != s || d.startIndex != position) return 1;
      }
    }
    return 0;
  } catch (std::exception& e) { g_error = e.what(); return 3; }
}

// T/PackedMap_Test.java:13-49 testLargeReferenceSize: eight repeating sequences of (int)Math.pow(2, 31) = Integer.MAX_VALUE bases (+ their reverse
// complements: 2^35 encoded positions), PackedMap(5, 10, db, 1), twenty blocks HashBlock(i, 1, i % 10, -(i % 10) - 1) of the first sequence:
// get(i) must return the two positions i and i + 10.  0 = it does.
int xmo_kat_packed_map_large() {
  try {
    SequenceDatabase db;
    for (int i = 0; i < 8; i++) {
      std::unique_ptr<Sequence> s(new Sequence());
      s->name = std::to_string(i);
      s->repeatedLength = INT32_MAX;
      db.addForward(std::move(s));
    }
    const int keyCapacity = 10;
    PackedMap map(5, keyCapacity, &db, 1);
    std::vector<HashBlock> blocks;
    for (int i = 0; i < keyCapacity * 2; i++) {
      blocks.push_back(HashBlock(i, 1, i % keyCapacity, -(i % keyCapacity) - 1));
    }
    map.add(db.forward(0), blocks, false);
    // and, beyond the reference's test, the same twenty blocks on the LAST sequence: its encoded positions need more than 34 bits
    PackedMap last(5, keyCapacity, &db, 2);
    last.add(db.forward(7), blocks, false);
    for (int i = 0; i < keyCapacity; i++) {
      for (PackedMap* m : {&map, &last}) {
        std::vector<SequencePosition> got;
        if (!m->get(i, INT32_MAX, got) || got.size() != 2) return 1;
        int a0 = got[0].startIndex, a1 = got[1].startIndex;
        if (a1 < a0) std::swap(a0, a1);
        if (a0 != i || a1 != i + keyCapacity) return 2;
        if (got[0].sequence != db.forward(m == &map ? 0 : 7) || got[1].sequence != got[0].sequence) return 3;
      }
    }
    return 0;
  } catch (std::exception& e) { g_error = e.what(); return 3; }
}

// T/HashBlock_Test.java:30-92 checkSymmetry; returns 0 when every block is symmetric, else a failure code
int xmo_kat_hash_symmetry(const char* text) {
  try {
    std::unique_ptr<Sequence> seq = makeSequence("q", text);
    std::unique_ptr<Sequence> rev = makeReverseComplement(*seq);
    int n = seq->getLength();
    // hashSequence(reverseSequence, startIndex, endIndex): first row whose block at startIndex ends at endIndex
    auto hashSequence = [&](int startIndex, int endIndex, HashBlock& out) -> bool {
      HashBlock_Stream stream(rev.get(), true, nullptr);
      while (true) {
        std::shared_ptr<HashBlock_Row> row = stream.getNextBatch();
        MultiBlockP block = row->get(startIndex);
        if (!block) return false;
        for (auto& c : block->getPossibilities()) if (c.hasBlock && c.block.getEndIndex() == endIndex) { out = c.block; return true; }
      }
    };
    HashBlock_Stream stream(seq.get(), true, nullptr);
    while (true) {
      std::shared_ptr<HashBlock_Row> row = stream.getNextBatch();
      if (!row->getAfter(-1)) return 0;
      int i = -1;
      while (true) {
        MultiBlockP mb = row->getAfter(i);
        if (!mb) break;
        const HashBlock* block = mb->getSingle();
        if (!block) return 100;  // no ambiguity in these KATs
        HashBlock reverseBlock;
        if (!hashSequence(n - block->getEndIndex(), n - block->getStartIndex(), reverseBlock)) return 1;
        if (reverseBlock.forwardHash != block->reverseHash) return 2;
        if (reverseBlock.reverseHash != block->forwardHash) return 3;
        if (block->requestMergeLeft != reverseBlock.requestMergeRight) return 4;
        if (block->requestMergeRight != reverseBlock.requestMergeLeft) return 5;
        if (block->nextRequestMergeLeft != reverseBlock.nextRequestMergeRight) return 6;
        if (block->nextRequestMergeRight != reverseBlock.nextRequestMergeLeft) return 7;
        if (!block->isPrimaryPolarity() && !block->isSecondaryPolarity()) return 8;
        HashBlock e, re;
        int r1 = block->withGapAndExtension(*seq, e);
        int r2 = reverseBlock.withGapAndExtension(*rev, re);
        if ((r1 == 0) != (r2 == 0)) return 9;
        if (r1 != 0) {
          const HashBlock& ee = (r1 == 1) ? *block : e;
          const HashBlock& rr = (r2 == 1) ? reverseBlock : re;
          if (rr.forwardHash != ee.reverseHash) return 10;
          if (rr.reverseHash != ee.forwardHash) return 11;
        }
        i = mb->getStartIndex();
      }
    }
  } catch (std::exception& e) { g_error = e.what(); return 99; }
}

// T/Counting_HashBlockPath_Test.java makePath + findGoodPositionsHavingPriorityUpTo(priority): returns offsets of good counters
int xmo_kat_counting_path(const char* queryText, const char* refText, double deletionExtensionPenalty, int priority, int32_t* offsets, int cap) {
  try {
    ReferenceDatabase ref;
    ref.sequences.addForward(makeSequence("reference", refText));
    ref.hashblockDatabase.reset(new HashBlock_Database(&ref.sequences));
    Readable_HashBlock_Database view(ref.hashblockDatabase.get());
    std::unique_ptr<Sequence> q = makeSequence("query", queryText);
    std::unique_ptr<Sequence> qr = makeReverseComplement(*q);
    AlignmentParameters p;  // all zero except:
    p.DeletionExtension_Penalty = deletionExtensionPenalty;
    Counting_HashBlockPath path(&view, &ref.sequences, q.get(), qr.get(), p, nullptr);
    CounterListP counters = path.findGoodPositionsHavingPriorityUpTo(priority);
    int n = 0;
    for (auto& c : *counters) { if (n < cap) offsets[n] = c->getMatch()->getOffset(); n++; }
    return n;
  } catch (std::exception& e) { g_error = e.what(); return -1; }
}

// T/HashBlockPaths_Counter_Test.java getMatches: returns number of QueryMatches of priority 0; fills inner/outer distances
int xmo_kat_paths_counter(const char* refText, const char* seq1Text, const char* seq2TextForward, int expectedInnerDistance, int maxInnerDistance,
                          int32_t* inner, int32_t* outer, int cap) {
  try {
    ReferenceDatabase ref;
    ref.sequences.addForward(makeSequence("ref", refText));
    ref.hashblockDatabase.reset(new HashBlock_Database(&ref.sequences));
    Readable_HashBlock_Database view1(ref.hashblockDatabase.get()), view2(ref.hashblockDatabase.get());
    std::unique_ptr<Sequence> q1 = makeSequence("seq1", seq1Text);
    std::unique_ptr<Sequence> q1r = makeReverseComplement(*q1);
    // seq2Text = reverseComplement(seq2Text); query2 = new Sequence(seq2Text)  (a *fresh* sequence: not flagged as a complement)
    std::unique_ptr<Sequence> tmp = makeSequence("tmp", seq2TextForward);
    std::unique_ptr<Sequence> tmpRc = makeReverseComplement(*tmp);
    std::unique_ptr<Sequence> q2(new Sequence());
    q2->name = "seq2";
    q2->codes = tmpRc->codes;
    std::unique_ptr<Sequence> q2r = makeReverseComplement(*q2);
    AlignmentParameters p;
    p.DeletionExtension_Penalty = 0.1;
    Counting_HashBlockPath c1(&view1, &ref.sequences, q1.get(), q1r.get(), p, nullptr);
    Counting_HashBlockPath c2(&view2, &ref.sequences, q2.get(), q2r.get(), p, nullptr);
    std::vector<Counting_HashBlockPath*> comps;
    comps.push_back(&c1);
    comps.push_back(&c2);
    HashBlockPaths_Counter counter(comps, expectedInnerDistance, maxInnerDistance);
    QueryMatchListP matches = counter.findGoodPositionsHavingPriority(0);
    int n = 0;
    for (auto& m : *matches) {
      if (n < cap) { inner[n] = m->getTotalDistanceBetweenComponents(); outer[n] = m->getTotalDistanceAcross(); }
      n++;
    }
    return n;
  } catch (std::exception& e) { g_error = e.what(); return -1; }
}

// T/HashBlockDatabase_Test.java: index built in reverse job order must equal the forward-built one. returns 0 if equal.
int xmo_kat_db_order_independent(int ncontigs, const char** texts, int throughSize) {
  try {
    SequenceDatabase sdb;
    for (int i = 0; i < ncontigs; i++) sdb.addForward(makeSequence("c" + std::to_string(i), texts[i]));
    HashBlock_Database a(&sdb, -1, -1, -1, true, false), b(&sdb, -1, -1, -1, true, true);
    a.requireSetUpThroughSize(throughSize);
    b.requireSetUpThroughSize(throughSize);
    if (a.maxFullySetUpSize != b.maxFullySetUpSize) return 1;
    for (int L = a.minInterestingSize; L <= a.maxFullySetUpSize; L++) {
      PackedMap *x = a.hashedBlocks[(size_t)L].get(), *y = b.hashedBlocks[(size_t)L].get();
      if (x->keyCapacity != y->keyCapacity || x->maxInterestingCountPerKey != y->maxInterestingCountPerKey) return 2;
      if (x->overfull != y->overfull) return 3;
      if (x->buckets != y->buckets) return 4;
      if (x->numItemsAdded != y->numItemsAdded) return 5;
    }
    return 0;
  } catch (std::exception& e) { g_error = e.what(); return 99; }
}

// pyramid dump for parity of the product's read-side pyramid: for every level and start, the block tuple
// out rows: level, start, length, fwd, rev, flags(bit0 rml, bit1 rmr, bit2 nrml, bit3 nrmr), gapDirection, extraGapmerLength,
//           gapmer status (0 null,1 same,2 new), gStart, gLength, gUsed, gFwd, gRev
int64_t xmo_pyramid_dump(const uint8_t* codes, int len, int32_t* out, int64_t capRows) {
  std::unique_ptr<Sequence> s(new Sequence());
  s->codes.assign(codes, codes + len);
  HashBlock_Pyramid pyr(s.get(), false, nullptr);
  int64_t n = 0;
  for (int level = 0;; level++) {
    HashBlock_Row* row = pyr.get(level);
    MultiBlockP b = row->getAfter(-1);
    if (!b) break;
    while (b) {
      const HashBlock* h = b->getSingle();
      if (h) {
        if (n < capRows) {
          int32_t* o = out + n * 14;
          o[0] = level; o[1] = h->startIndex; o[2] = h->length; o[3] = h->forwardHash; o[4] = h->reverseHash;
          o[5] = (h->requestMergeLeft ? 1 : 0) | (h->requestMergeRight ? 2 : 0) | (h->nextRequestMergeLeft ? 4 : 0) | (h->nextRequestMergeRight ? 8 : 0);
          o[6] = h->gapDirection; o[7] = h->extraGapmerLength;
          HashBlock g;
          int st = h->withGapAndExtension(*s, g);
          const HashBlock& gg = (st == 2) ? g : *h;
          o[8] = st; o[9] = st ? gg.startIndex : 0; o[10] = st ? gg.length : 0; o[11] = st ? gg.numBasepairsUsed : 0; o[12] = st ? gg.forwardHash : 0; o[13] = st ? gg.reverseHash : 0;
        }
        n++;
      }
      b = row->getAfter(b->getStartIndex());
    }
  }
  return n;
}

// every block of the read-side pyramid, multi blocks included (reads with ambiguous bases): one row per possibility, 12 ints:
//   level, block start, block end - start, number of possibilities (0: a single block), possibility index, has block, its start, length, fwd, rev,
//   flags (as above), a hash of its condition (entries (position, base index A C G T = 0..3) in order).  Stops at the first level without blocks.
int64_t xmo_pyramid_dump_multi(const uint8_t* codes, int len, int32_t* out, int64_t capRows) {
  std::unique_ptr<Sequence> s(new Sequence());
  s->codes.assign(codes, codes + len);
  HashBlock_Pyramid pyr(s.get(), false, nullptr);
  int64_t n = 0;
  auto row = [&](int level, int bs, int bl, int np, int k, bool has, const HashBlock* h, const SequenceCondition* c) {
    if (n < capRows) {
      int32_t* o = out + n * 12;
      o[0] = level; o[1] = bs; o[2] = bl; o[3] = np; o[4] = k; o[5] = has ? 1 : 0;
      o[6] = has ? h->startIndex : 0; o[7] = has ? h->length : 0; o[8] = has ? h->forwardHash : 0; o[9] = has ? h->reverseHash : 0;
      o[10] = has ? ((h->requestMergeLeft ? 1 : 0) | (h->requestMergeRight ? 2 : 0) | (h->nextRequestMergeLeft ? 4 : 0) | (h->nextRequestMergeRight ? 8 : 0)) : 0;
      uint32_t hash = c ? (uint32_t)c->keys.size() : 0u;
      if (c) for (size_t i = 0; i < c->keys.size(); i++) {
        const char v = c->values[i];
        hash = hash * 1000003u + ((uint32_t)c->keys[i] * 4u + (v == 'A' ? 0u : v == 'C' ? 1u : v == 'G' ? 2u : 3u));
      }
      o[11] = (int32_t)hash;
    }
    n++;
  };
  for (int level = 0;; level++) {
    HashBlock_Row* r = pyr.get(level);
    MultiBlockP b = r->getAfter(-1);
    if (!b) break;
    while (b) {
      const HashBlock* h = b->getSingle();
      if (h) row(level, h->startIndex, h->length, 0, 0, true, h, nullptr);
      else {
        const int bs = b->getStartIndex(), be = b->getEndIndex();
        for (size_t k = 0; k < b->possibilities.size(); k++) {
          const ConditionalHashBlock& p = b->possibilities[k];
          row(level, bs, be - bs, (int)b->possibilities.size(), (int)k, p.hasBlock, &p.block, &p.condition);
        }
      }
      b = r->getAfter(b->getStartIndex());
    }
  }
  return n;
}

}  // extern "C"
