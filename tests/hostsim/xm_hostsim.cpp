// HOST SIMULATION OF THE DEVICE KERNEL — TEST HARNESS ONLY.
//
// There is no GPU in the development container, so the kernel sources (mapper_amd/csrc/xm_*.h) are also compiled for the
// host here and driven read by read, exactly as a lane of xm_align_kernel would, to let the CPU-only test tier compare the
// kernel logic with the oracle.  This file is built by tests/ into tests/_build/libxm_hostsim.so; it is never linked into
// or loaded by libxmapper_hip.so / the mapper_amd package, which has no CPU path.
#include "../../include/xmapper_hip.h"
#include "../../mapper_amd/csrc/xm_worker.h"
#include "../../mapper_amd/csrc/xm_wsearch.h"
#include "../../mapper_amd/csrc/xm_wave.h"
#include "../../mapper_amd/csrc/xm_index_host.h"
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace xm;

// XM_ARENA_TRACE builds (scripts/light_pass_footprint.py): the allocations of the arenas, in order
#if defined(XM_ARENA_TRACE)
struct ArenaAlloc { const void* base; size_t offset, bytes; };
static std::vector<ArenaAlloc> g_allocs;
static bool g_traceOn = false;
namespace xm { void xm_arena_trace(const void* base, size_t offset, size_t bytes) { if (g_traceOn) g_allocs.push_back(ArenaAlloc{base, offset, bytes}); } }
#endif

static thread_local std::string g_err;
static int g_waveMode = -1;  // -1: from XMSIM_WAVE (default 0)
static long long g_waveStatus[16];
static long long g_waveWhy[64];
static long long g_markHist[6][128];  // high-water marks of the wave form per read: chunks, counters, history, pending, query matches, alignments
static void markDump() { const char* names[6] = {"chunks", "counters", "history", "pending", "qmatches", "alignments"}; for (int k = 0; k < 6; k++) { fprintf(stderr, "[wave marks] %s:", names[k]); for (int i = 0; i < 128; i++) if (g_markHist[k][i]) fprintf(stderr, " %d:%lld", i, g_markHist[k][i]); fprintf(stderr, "\n"); } }

struct SimIndex {
  HostIndex host;
  IndexView view;
  int32_t baSteps[24];
  std::vector<uint32_t> p32, l32;
  std::vector<uint64_t> l64;
  void refresh() {
    bool is64 = host.seqCumStart.back() > 0xFFFFFFFFll;
    p32.clear();
    if (!is64) { p32.resize(host.positions.size()); for (size_t i = 0; i < p32.size(); i++) p32[i] = (uint32_t)host.positions[i]; }
    view.numContigs = host.numContigs(); view.minInterestingSize = host.minInterestingSize; view.maxHashedLength = host.maxHashedLength;
    view.enableGapmers = host.enableGapmers; view.posIs64 = is64 ? 1 : 0; view.dupWindow = host.dupWindow; view.dupGranularity = host.dupGranularity();
    view.totalForwardAndReverseSize = host.totalForwardSize * 2;
    view.contigStart = host.contigStart.data(); view.contigLen = host.contigLen.data(); view.seqCumStart = host.seqCumStart.data(); view.refCodes = host.refCodes.data();
    view.tables = host.tables.data(); view.bucketOff = host.bucketOff.data(); view.positions32 = p32.data(); view.positions64 = (const uint64_t*)host.positions.data();
    view.dupKeyStart = host.dupKeyStart.data(); view.dupKeys = host.dupKeys.data();
    view.conf = nullptr; view.confMask = 0; view.confMiss = nullptr;   // (the host simulation evaluates the confidence term in place: xm_worker.h)
    blockAlignerLogSteps(baSteps, 24);
    view.baLogStep = baSteps;
    // bucket lines (IndexView::lines32 / lines64), filled by the function the device kernel uses; XMSIM_LINES=0: CSR probes only
    view.lines32 = nullptr; view.lines64 = nullptr;
    l32.clear(); l64.clear();
    const char* le = getenv("XMSIM_LINES");
    if (!(le && *le && atoi(le) == 0)) {
      if (is64) l64.assign(host.bucketOff.size() * 8, 0); else l32.assign(host.bucketOff.size() * 8, 0);
      for (const Table& t : host.tables)
        for (int64_t k = 0; k < t.capacity; k++) {
          const uint32_t* off = host.bucketOff.data() + t.offBase + k;
          if (is64) xmFillLine(l64.data() + (t.offBase + k) * 8, off[0], off[1], (const uint64_t*)host.positions.data() + t.posBase);
          else xmFillLine(l32.data() + (t.offBase + k) * 8, off[0], off[1], p32.data() + t.posBase);
        }
      view.lines32 = is64 ? nullptr : l32.data(); view.lines64 = is64 ? l64.data() : nullptr;
    }
  }
};

extern "C" {

const char* xmsim_last_error() { return g_err.c_str(); }

void* xmsim_index_build(const xm_ref* ref, const xm_build_opts* o) {
  try {
    SimIndex* s = new SimIndex();
    s->host.setReference(ref->num_contigs, ref->names, ref->codes, ref->lengths);
    s->host.build(o->enable_gapmers, o->min_interesting_size, o->max_hashed_length, o->dup_window, o->dup_min_copies, o->dup_min_length, o->dup_max_length);
    s->refresh();
    return s;
  } catch (std::exception& e) { g_err = e.what(); return nullptr; }
}
void xmsim_index_free(void* p) { delete (SimIndex*)p; }

// same result layout as xm_align_batch; counters[11] = reads rerun at a larger scale; returns non-zero on failure
int xmsim_align_batch(void* idxp, const xm_params* p, const xm_query_batch* b, xm_result** out) {
  SimIndex* idx = (SimIndex*)idxp;
  try {
    int maxLen = 1;
    for (int64_t q = 0; q < b->num_queries; q++) for (int m = 0; m < b->mate_count[q]; m++) if (b->mate_length[q * 2 + m] > maxLen) maxLen = b->mate_length[q * 2 + m];
    if (maxLen > idx->host.maxHashedLength) { idx->host.ensureLength(maxLen); idx->refresh(); }
    Params params;
    params.MutationPenalty = p->MutationPenalty; params.InsertionStart_Penalty = p->InsertionStart_Penalty; params.InsertionExtension_Penalty = p->InsertionExtension_Penalty;
    params.DeletionStart_Penalty = p->DeletionStart_Penalty; params.DeletionExtension_Penalty = p->DeletionExtension_Penalty; params.MaxErrorRate = p->MaxErrorRate;
    params.UnalignedPenalty = p->UnalignedPenalty; params.AmbiguityPenalty = p->AmbiguityPenalty; params.Max_PenaltySpan = p->Max_PenaltySpan;
    params.MaxNumMatches = p->MaxNumMatches; params.StartingInsertionStartFree = 0;
    const int64_t nq = b->num_queries;
    std::vector<int32_t> ints;
    std::vector<double> dbls;
    xm_result* res = (xm_result*)calloc(1, sizeof(xm_result));
    res->num_queries = nq;
    res->int_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nq + 1));
    res->dbl_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nq + 1));
    DevCounters dc;
    memset(&dc, 0, sizeof(dc));
    int64_t rerun = 0;
    std::vector<uint8_t> arena;
    static ReadCtx cx;  // large; keep it off the stack
    for (int64_t q = 0; q < nq; q++) {
      ReadIn in;
      in.nMates = b->mate_count[q];
      for (int m = 0; m < 2; m++) { in.mate[m] = b->codes + b->mate_offset[q * 2 + m]; in.mateLen[m] = m < in.nMates ? b->mate_length[q * 2 + m] : 0; }
      in.expectedInner = in.nMates > 1 ? b->expected_inner[q] : 0.0;
      in.deviation = in.nMates > 1 ? b->deviation[q] : 1.0;
      // the wave-per-read form first (XMSIM_WAVE: 1 = its light tier, 2 = light then heavy tier), exactly as the product's passes 0 / 0b run
      // it (mapper_amd/csrc/xm_wave.h; a WV_PAR region is a loop over the 64 lanes here); reads it does not finish go through the
      // lane-per-read sequence below
      const int waveMode = g_waveMode >= 0 ? g_waveMode : (getenv("XMSIM_WAVE") ? atoi(getenv("XMSIM_WAVE")) : 0);
      if (waveMode > 0) {
        typedef WaveLdsT<WCfgLightPE> LightLds;
        typedef WaveLdsT<WCfgMidPE> MidLds;
        typedef WaveLdsT<WCfgHeavy> HeavyLds;
        static LightLds ldsLight;
        static MidLds ldsMid;
        static HeavyLds ldsHeavy;
        static std::vector<uint8_t> waveArena((size_t)(288 * 1024 * 4 * 7 / 12 + 4096));
        static WMemo memo;
        memo.count = 0; memo.pending = 0;
        bool finished = false;
        int searchRounds = 0;
        // waveMode 1: light tier only; 2: + chain tier (with its search rounds); 3: + the chain tier with the largest capacities
        for (int tier = 0; tier < waveMode && tier < 3 && !finished; ) {
          DevCounters before = dc;
          WEnv e;
          e.ix = idx->view; e.params = params; e.dc = &dc; e.mateBase[0] = in.mate[0]; e.mateBase[1] = in.mate[1]; e.tier = tier;
          Caps caps = makeCaps(4);
          caps.heavyAllowed = 2;
          Arena tmp;
          tmp.init((void*)(((uintptr_t)waveArena.data() + 15) & ~(uintptr_t)15), waveArena.size() - 64);
          e.caps = &caps; e.tmp = &tmp;
          e.memo = tier == 0 ? nullptr : &memo;
          static std::vector<WSNode> inlineNodes(WSearchLdsInline::kNodes);
          static const bool inlineSearch = !(getenv("XMSIM_NO_INLINE_SEARCH") && atoi(getenv("XMSIM_NO_INLINE_SEARCH")) != 0);  // (0: every search through the memo and the search "kernel")
          e.searchNodes = tier == 0 || !inlineSearch ? nullptr : inlineNodes.data();
          WResult wr;
          int32_t st = 0, why = 0;
          int64_t ni = 0, nd = 0;
          auto run = [&](auto& Ld) {
            wAlignRead(&Ld, e, in, wr);
            st = Ld.status; why = Ld.why;
            if (st == XM_OK) wResultSize(&Ld, wr, ni, nd);
            if (getenv("XMSIM_WAVE_STATS")) {
              static bool reg = false; if (!reg) { reg = true; atexit(markDump); }
              auto clip = [](int v) { return v < 0 ? 0 : (v > 127 ? 127 : v); };
              g_markHist[0][clip(Ld.nChunksUsed)]++;
              for (int m = 0; m < in.nMates; m++) { g_markHist[1][clip(Ld.m[m].nCounters)]++; g_markHist[2][clip(Ld.m[m].nHistory)]++; g_markHist[3][clip(Ld.m[m].pendTail)]++; }
              g_markHist[4][clip(Ld.nAssembled)]++; g_markHist[5][clip(Ld.al[0].nGood)]++;
            }
          };
          auto write = [&](auto& Ld) { wResultWrite(&Ld, wr, ints.data() + res->int_off[q], dbls.data() + res->dbl_off[q], &dc); };
          if (tier == 0) run(ldsLight); else if (tier == 1) run(ldsMid); else run(ldsHeavy);
          g_waveStatus[st < 16 ? st : 15]++;
          if (st == XM_ST_WAVE_FALLBACK || (st == XM_ST_WAVE_GAPPED && tier >= 1)) g_waveWhy[why & 63]++;
          if (st == XM_OK) {
            res->int_off[q] = (int64_t)ints.size(); res->dbl_off[q] = (int64_t)dbls.size();
            ints.resize(ints.size() + (size_t)ni); dbls.resize(dbls.size() + (size_t)nd);
            if (tier == 0) write(ldsLight); else if (tier == 1) write(ldsMid); else write(ldsHeavy);
            finished = true;
          } else {
            dc = before;
            if (st == XM_ST_WAVE_SEARCH) {  // the search "kernel", then the same tier again
              const int mi = memo.req.seqAId >> 1;
              static WSearchLdsKernel searchLds;
              static std::vector<WSNode> searchNodes(WSearchLdsKernel::kNodes);
              wRunSearch(&searchLds, searchNodes.data(), idx->view, params, in.mate[mi], in.mateLen[mi], &memo);
              if (++searchRounds > 4 * WV_MEMO_MAX) throw std::runtime_error("search rounds do not end");
              continue;
            }
            if (st != XM_ST_WAVE_FALLBACK && st != XM_ST_WAVE_GAPPED) throw std::runtime_error("Failed to align query " + std::to_string(q) + " (wave status " + std::to_string(st) + ")");
            if (st == XM_ST_WAVE_FALLBACK) break;  // (ambiguity codes, long or overlapping mates: no tier of the wave form takes them)
            tier++;
          }
        }
        if (finished) continue;
      }
      // the product's pass sequence for one read: light pass at scale 1; a read that needs the gapped chain goes on at scale 4 from the state it
      // saved (hand-over), on another context object and other temporaries; scratch overflow -> reruns from the start at 16x, 64x...
      // XMSIM_INLINE=1: plain run at scale 1, 4, 16... (the first implementation's sequence)
      static const bool inlineOnly = getenv("XMSIM_INLINE") && atoi(getenv("XMSIM_INLINE")) != 0;
      // (the product sizes a batch by its longest mate: scratch scale 1 up to 320 bases, 4 up to 1280, 16 beyond, the gapped pass at four times that;
      // here every read by its own longest mate)
      int longestMate = 0;
      for (int m = 0; m < in.nMates; m++) longestMate = std::max(longestMate, (int)in.mateLen[m]);
      const int seedScale = longestMate <= 320 ? 1 : (longestMate <= 1280 ? 4 : 16);
      int scale = seedScale;
      int stage = inlineOnly ? 2 : 0;  // 0 light, 1 gapped, 2 rerun
      static const int lightLevel = getenv("XMSIM_LIGHT_LEVEL") ? atoi(getenv("XMSIM_LIGHT_LEVEL")) : 0;
      // light pass -> gapped pass hand-over (runReadRetaining / runReadResumed): the read's region outlives the light "lane"; the gapped pass
      // runs on another context object and another temporaries buffer, as it does on another lane of the GPU
      static const bool handOver = !(getenv("XMSIM_NO_HANDOVER") && atoi(getenv("XMSIM_NO_HANDOVER")) != 0);
      // the product's sizes: light-pass temporaries 48 KB, a read's region 72 KB (single-end) or 120 KB (paired) + its saved context
      const size_t lightTmpBytes = (size_t)48 * 1024 * (size_t)seedScale, regionBytes = ((size_t)(in.nMates > 1 ? 120 : 72) * 1024 * (size_t)seedScale) + ((sizeof(SavedRead) + 15) & ~(size_t)15);
      // temporaries of a gapped-pass lane: a fifth of 7/12 of the arena for short reads (their HBM-mode searches use the wave's buffer), all of it + the
      // node arrays of a long-read chain for long ones (xm_capi.hip, gappedTmpBytes)
      auto gappedTmp = [&](size_t bytes, int chainScale) -> size_t {
        const size_t whole = bytes - arenaPersistBytes(bytes);
        return ((seedScale == 1 ? whole / 5 : whole) & ~(size_t)15) + chainExtraTmpBytes(chainScale);
      };
      std::vector<double> regionBuf(regionBytes / 8 + 2);
      uint8_t* region = (uint8_t*)(((uintptr_t)regionBuf.data() + 15) & ~(uintptr_t)15);
      SavedRead* saved = nullptr;
      static ReadCtx cx2;
      std::vector<uint8_t> arena2;
      while (true) {
        size_t bytes = (size_t)288 * 1024 * (size_t)scale;
        ReadResult rr;
        DevCounters before = dc;
        // the rejection filter in front of PathAligner (xm_bound.h): the product turns it on for the gapped passes of batches of long reads (xm_capi.hip, boundFilter)
        static const bool boundFilterOn = !(getenv("XMSIM_BOUND_FILTER") && atoi(getenv("XMSIM_BOUND_FILTER")) == 0);
        xmSetBoundFilter(boundFilterOn && stage != 0 && seedScale >= 4 && scale >= 16 ? 1 : 0);
        if (stage == 1 && saved) {
          arena2.assign(bytes + chainExtraTmpBytes(scale) + 64, 0xAB);
          uint8_t* a2 = (uint8_t*)(((uintptr_t)arena2.data() + 15) & ~(uintptr_t)15);
          runReadResumed(cx2, saved, &idx->view, scale, a2, gappedTmp(bytes, scale), &dc, rr);
          cx.status = cx2.status;
          cx.persist = cx2.persist; cx.tmp = cx2.tmp;  // (for the overflow trace)
          saved = nullptr;
        } else {
          arena.resize(bytes + chainExtraTmpBytes(scale) + 64);
          uint8_t* a = (uint8_t*)(((uintptr_t)arena.data() + 15) & ~(uintptr_t)15);
          if (stage == 0 && handOver) {
            runReadRetaining(cx, &idx->view, params, in, scale, region, regionBytes, a, lightTmpBytes, &dc, rr, lightLevel);
            if (cx.status == XM_ST_NEED_HEAVY) {
              SavedRead* sv = savedReadOf(region, regionBytes);
              saved = sv->valid ? sv : nullptr;
              std::vector<uint8_t>().swap(arena);  // the light lane's temporaries are gone (a stale pointer into them would be caught by the sanitizer run)
            }
          } else if (stage == 1 && handOver) {
            // gapped pass, read without saved state (stopped where it cannot be resumed): seeded again in a region of light-pass size
            runReadRetaining(cx, &idx->view, params, in, seedScale, region, regionBytes, a, ((bytes - arenaPersistBytes(bytes)) & ~(size_t)15) + chainExtraTmpBytes(scale), &dc, rr, 2, scale);
          } else {
            runRead(cx, &idx->view, params, in, scale, a, bytes, &dc, rr, stage != 0 ? 2 : lightLevel);
          }
          if (cx.status == XM_ST_NEED_HEAVY && stage == 0) { dc = before; stage = 1; scale = seedScale * 4; continue; }
        }
        if (cx.status == XM_ST_OVERFLOW) {
          if (getenv("XMSIM_TRACE_OVERFLOW")) fprintf(stderr, "[xmsim] query %lld overflow at stage %d scale %d: persist %zu of %zu%s, tmp %zu of %zu%s\n", (long long)q, stage, scale, cx.persist.used, cx.persist.size, cx.persist.overflow ? " (overflow)" : "", cx.tmp.used, cx.tmp.size, cx.tmp.overflow ? " (overflow)" : "");
          dc = before; rerun++;
          saved = nullptr;
          if (stage == 0) { stage = 2; scale = seedScale * 4; } else { stage = 2; scale *= 4; }
          if (scale > 4096) throw std::runtime_error("scratch scale limit");
          continue;
        }
        if (cx.status != XM_OK) throw std::runtime_error("Failed to align query " + std::to_string(q) + " (status " + std::to_string(cx.status) + ")");
        int64_t ni, nd;
        resultSize(rr, ni, nd);
        res->int_off[q] = (int64_t)ints.size(); res->dbl_off[q] = (int64_t)dbls.size();
        ints.resize(ints.size() + (size_t)ni); dbls.resize(dbls.size() + (size_t)nd);
        OutWriter w;
        w.ints = ints.data() + res->int_off[q]; w.dbls = dbls.data() + res->dbl_off[q]; w.ni = 0; w.nd = 0;
        resultWrite(rr, w, &dc);
        break;
      }
    }
    res->int_off[nq] = (int64_t)ints.size(); res->dbl_off[nq] = (int64_t)dbls.size();
    res->num_ints = (int64_t)ints.size(); res->num_dbls = (int64_t)dbls.size();
    res->ints = (int32_t*)malloc(sizeof(int32_t) * (ints.size() + 1));
    res->dbls = (double*)malloc(sizeof(double) * (dbls.size() + 1));
    if (!ints.empty()) memcpy(res->ints, ints.data(), ints.size() * sizeof(int32_t));
    if (!dbls.empty()) memcpy(res->dbls, dbls.data(), dbls.size() * sizeof(double));
    res->counters[0] = (int64_t)dc.reads; res->counters[1] = (int64_t)dc.headerProbes; res->counters[2] = (int64_t)dc.bucketFetches; res->counters[3] = (int64_t)dc.hitsFetched;
    res->counters[4] = (int64_t)dc.candidatesExtended; res->counters[5] = (int64_t)dc.pathAlignerCalls; res->counters[6] = (int64_t)dc.pathAlignerNodes;
    res->counters[7] = (int64_t)dc.quickAccepts; res->counters[8] = (int64_t)dc.alignmentsOut; res->counters[9] = (int64_t)dc.refWindowBytes; res->counters[10] = (int64_t)dc.readBytes;
    res->counters[11] = rerun;
    res->extra[0] = (int64_t)dc.boundChecks; res->extra[1] = (int64_t)dc.boundRejects; res->extra[2] = (int64_t)dc.boundCells; res->extra[3] = (dc.boundChecks | dc.boundPieceChecks) > 0 ? 1 : 0; res->extra[4] = (int64_t)dc.boundPieceChecks; res->extra[5] = (int64_t)dc.boundPieceRejects;
    *out = res;
    return 0;
  } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// the rejection filter of xm_bound.h alone on one problem (what xm_test_bound runs on the GPU); out3: taken, rejected, cells
int xmsim_test_bound(const xm_params* p, const uint8_t* query, int queryLength, int queryRc, int startA, int endA, const uint8_t* reference, int referenceLength, int startB, int endB,
                     int predictedBestOffset, int64_t* out3) {
  BoundProblem bp;
  bp.qBase = query; bp.qLen = queryLength; bp.qRc = queryRc != 0; bp.rBase = reference; bp.referenceLen = referenceLength;
  bp.startA = startA; bp.endA = endA; bp.startB = startB; bp.endB = endB; bp.predictedBestOffset = predictedBestOffset;
  bp.mutation = p->MutationPenalty; bp.insStart = p->InsertionStart_Penalty; bp.insExt = p->InsertionExtension_Penalty; bp.delStart = p->DeletionStart_Penalty;
  bp.delExt = p->DeletionExtension_Penalty; bp.maxErrorRate = p->MaxErrorRate; bp.ambiguity = p->AmbiguityPenalty;
  bp.budget = (endA - startA) * p->MaxErrorRate; bp.piece = 0;
  bool taken = false;
  unsigned long long cells = 0;
  static thread_local std::vector<uint8_t> arena(64 * 1024 + 64);
  Arena tmp;
  tmp.init((void*)(((uintptr_t)arena.data() + 15) & ~(uintptr_t)15), 64 * 1024);
  const bool rejected = boundRejects(bp, false, tmp, taken, cells);
  out3[0] = taken ? 1 : 0; out3[1] = rejected ? 1 : 0; out3[2] = (int64_t)cells;
  return 0;
}

// 0: lane-per-read sequence only, 1: the wave form's light tier first, 2: light then heavy tier first (the product's sequence)
void xmsim_set_wave_mode(int mode) { g_waveMode = mode; }
// how the wave form left the reads of all calls so far, by status (0 = finished there)
void xmsim_wave_status_counts(long long* out, int reset) { for (int i = 0; i < 16; i++) { out[i] = g_waveStatus[i]; if (reset) g_waveStatus[i] = 0; } }
void xmsim_wave_why_counts(long long* out) { for (int i = 0; i < 64; i++) out[i] = g_waveWhy[i]; }

#if defined(XM_ARENA_TRACE)
// The light pass's footprint in a lane's scratch, structure by structure: every read of the batch through runReadRetaining at the light pass's
// settings (scale 1, region + 48 KB of temporaries, heavyAllowed 0) over memory filled with a pattern; per allocation of the two arenas (in
// allocation order: the order is the same for every single-end read without ambiguity codes) the bytes that no longer hold the pattern.
// out[3 * i + 0] = arena (0 region, 1 temporaries), [1] = capacity in bytes, [2] = bytes written, summed over the reads; returns the number of rows.
int64_t xmsim_light_footprint(void* idxp, const xm_params* p, const xm_query_batch* b, int64_t* out, int64_t capRows, int64_t* readsDone) {
  SimIndex* idx = (SimIndex*)idxp;
  Params params;
  params.MutationPenalty = p->MutationPenalty; params.InsertionStart_Penalty = p->InsertionStart_Penalty; params.InsertionExtension_Penalty = p->InsertionExtension_Penalty;
  params.DeletionStart_Penalty = p->DeletionStart_Penalty; params.DeletionExtension_Penalty = p->DeletionExtension_Penalty; params.MaxErrorRate = p->MaxErrorRate;
  params.UnalignedPenalty = p->UnalignedPenalty; params.AmbiguityPenalty = p->AmbiguityPenalty; params.Max_PenaltySpan = p->Max_PenaltySpan;
  params.MaxNumMatches = p->MaxNumMatches; params.StartingInsertionStartFree = 0;
  {
    int maxLen = 1;
    for (int64_t q = 0; q < b->num_queries; q++) for (int m = 0; m < b->mate_count[q]; m++) if (b->mate_length[q * 2 + m] > maxLen) maxLen = b->mate_length[q * 2 + m];
    if (maxLen > idx->host.maxHashedLength) { idx->host.ensureLength(maxLen); idx->refresh(); }
  }
  const size_t regionBytes = (size_t)72 * 1024 + ((sizeof(SavedRead) + 15) & ~(size_t)15), tmpBytes = (size_t)48 * 1024;
  std::vector<uint8_t> region(regionBytes + 64), tmp(tmpBytes + 64);
  uint8_t* rg = (uint8_t*)(((uintptr_t)region.data() + 15) & ~(uintptr_t)15);
  uint8_t* tp = (uint8_t*)(((uintptr_t)tmp.data() + 15) & ~(uintptr_t)15);
  static ReadCtx cx;
  int64_t rows = 0;
  *readsDone = 0;
  for (int64_t q = 0; q < b->num_queries; q++) {
    if (b->mate_count[q] != 1) continue;
    ReadIn in;
    in.nMates = 1;
    for (int m = 0; m < 2; m++) { in.mate[m] = b->codes + b->mate_offset[q * 2 + m]; in.mateLen[m] = m < 1 ? b->mate_length[q * 2 + m] : 0; }
    in.expectedInner = 0.0; in.deviation = 1.0;
    memset(rg, 0xCD, regionBytes); memset(tp, 0xCD, tmpBytes);
    g_allocs.clear();
    g_traceOn = true;
    DevCounters dc; memset(&dc, 0, sizeof(dc));
    ReadResult rr;
    runReadRetaining(cx, &idx->view, params, in, 1, rg, regionBytes, tp, tmpBytes, &dc, rr, 0);
    g_traceOn = false;
    if ((int64_t)g_allocs.size() > capRows) return -1;
    if ((int64_t)g_allocs.size() > rows) { for (int64_t i = rows; i < (int64_t)g_allocs.size(); i++) { out[3 * i] = 0; out[3 * i + 1] = 0; out[3 * i + 2] = 0; } rows = (int64_t)g_allocs.size(); }
    for (size_t i = 0; i < g_allocs.size(); i++) {
      const ArenaAlloc& a = g_allocs[i];
      const uint8_t* base = (const uint8_t*)a.base;
      const int arena = base == rg ? 0 : 1;
      int64_t dirty = 0;
      for (size_t k = 0; k < a.bytes; k++) if (base[a.offset + k] != 0xCD) dirty++;
      out[3 * i] = arena; if ((int64_t)a.bytes > out[3 * i + 1]) out[3 * i + 1] = (int64_t)a.bytes; out[3 * i + 2] += dirty;
    }
    (*readsDone)++;
  }
  return rows;
}
#endif


void xmsim_result_free(xm_result* r) {
  if (!r) return;
  free(r->ints); free(r->dbls); free(r->int_off); free(r->dbl_off); free(r);
}

int xmsim_table_info(void* idxp, int L, int32_t* capacity, int32_t* maxCount, int64_t* numStored) {
  SimIndex* idx = (SimIndex*)idxp;
  if (L < 0 || L > idx->host.maxHashedLength) return 1;
  const Table& t = idx->host.tables[(size_t)L];
  *capacity = t.capacity; *maxCount = t.maxCount;
  *numStored = (int64_t)(idx->host.bucketOff[(size_t)(t.offBase + t.capacity)] & ~XM_OVERFULL);
  return 0;
}
int xmsim_table_dump(void* idxp, int L, int32_t* counts, int64_t* positionsOut) {
  SimIndex* idx = (SimIndex*)idxp;
  const HostIndex& h = idx->host;
  const Table& t = h.tables[(size_t)L];
  int64_t w = 0;
  for (int k = 0; k < t.capacity; k++) {
    uint32_t o0 = h.bucketOff[(size_t)(t.offBase + k)], o1 = h.bucketOff[(size_t)(t.offBase + k + 1)];
    if (o0 & XM_OVERFULL) { counts[k] = -1; continue; }
    int c = (int)((o1 & ~XM_OVERFULL) - (o0 & ~XM_OVERFULL));
    counts[k] = c;
    for (int j = 0; j < c; j++) positionsOut[w++] = (int64_t)h.positions[(size_t)(t.posBase + (o0 & ~XM_OVERFULL) + j)];
  }
  return 0;
}
int xmsim_index_info(void* idxp, int32_t* minInteresting, int32_t* maxHashed) {
  SimIndex* idx = (SimIndex*)idxp;
  *minInteresting = idx->host.minInterestingSize; *maxHashed = idx->host.maxHashedLength;
  return 0;
}
int xmsim_ensure_length(void* idxp, int length) {
  SimIndex* idx = (SimIndex*)idxp;
  try { idx->host.ensureLength(length); idx->refresh(); return 0; } catch (std::exception& e) { g_err = e.what(); return 1; }
}
int64_t xmsim_dup_keys(void* idxp, int contig, int32_t* out, int64_t cap) {
  SimIndex* idx = (SimIndex*)idxp;
  const HostIndex& h = idx->host;
  int64_t a = h.dupKeyStart[(size_t)contig], bb = h.dupKeyStart[(size_t)contig + 1];
  for (int64_t i = a; i < bb && i - a < cap; i++) out[i - a] = h.dupKeys[(size_t)i];
  return bb - a;
}

// T/MultiHashBlock_Test.java:84-133 through the product's reference-side multi blocks (HostIndex::nextLevelMulti, the code that hashes the windows
// around ambiguous bases): the possibilities of the blocks that start at 0 and end at the end of `ambiguous` (4-bit codes) must hold the block the
// plain rule makes of `text`.  0 = they do; 1 = they do not; 2 = `text` has no single block over its whole length.
int xmsim_kat_multi_contains(const uint8_t* text, const uint8_t* ambiguous, int len) {
  // the plain pyramid of `text`: the block at 0 that ends at len, if a level has one
  bool have = false;
  HBlock want{};
  {
    std::vector<HBlock> cur((size_t)len), next;
    for (int i = 0; i < len; i++) cur[(size_t)i] = HostIndex::hblock0(text[i], i);
    while (!cur.empty()) {
      if (cur[0].start == 0 && cur[0].len == len) { have = true; want = cur[0]; }
      next.clear();
      for (size_t i = 0; i + 1 < cur.size(); i++) if (shouldMergeBlocks(cur[i], cur[i + 1])) next.push_back(mergeBlocks(cur[i], cur[i + 1]));
      cur.swap(next);
    }
  }
  if (!have) return 2;
  std::vector<HostIndex::HEntry> cur((size_t)len);
  for (int i = 0; i < len; i++) {
    HostIndex::HEntry& e = cur[(size_t)i];
    const uint8_t code = ambiguous[i];
    if (bpIsAmbiguous(code)) {
      e.multi = true;
      for (int bit = 0; bit < 4; bit++) if (code & (1 << bit)) e.poss.push_back(HostIndex::HPoss{HostIndex::hblock0((uint8_t)(1 << bit), i), true, HostIndex::HCond(1, std::make_pair((int32_t)i, (uint8_t)(1 << bit)))});
    } else e.single = HostIndex::hblock0(code, i);
  }
  while (!cur.empty()) {
    for (const HostIndex::HEntry& e : cur) {
      if (e.start() != 0) continue;
      for (const HostIndex::HPoss& p : HostIndex::possibilitiesOf(e))
        if (p.hasBlock && p.block.start == 0 && p.block.len == len && p.block.fwd == want.fwd) return 0;
    }
    std::vector<HostIndex::HEntry> nx = HostIndex::nextLevelMulti(cur);
    cur.swap(nx);
  }
  return 1;
}

// T/SequenceDatabase_Test.java:16-42,118-132 through the product's position codec: HostIndex::encodePosition / decode on the host and decodePosition
// (xm_seed.h, what the kernels run) over the same cumulative starts, for numSequences contigs of sequenceLength - i bases and their reverse
// complements; positions 0, 100, length - 100, length - 1.  0 = every round trip holds.
int xmsim_kat_position_codec(int numSequences, int sequenceLength) {
  HostIndex h;
  int64_t off = 0;
  for (int i = 0; i < numSequences; i++) {
    const int64_t len = (int64_t)sequenceLength - i;
    h.contigStart.push_back(off); h.contigLen.push_back((int32_t)len);
    h.seqCumStart.push_back(2 * off); h.seqCumStart.push_back(2 * off + len);
    off += len;
  }
  h.seqCumStart.push_back(2 * off);
  h.totalForwardSize = off;
  IndexView v;
  memset(&v, 0, sizeof(v));
  v.numContigs = numSequences; v.seqCumStart = h.seqCumStart.data(); v.contigLen = h.contigLen.data(); v.contigStart = h.contigStart.data();
  for (int c = 0; c < numSequences; c++)
    for (int rc = 0; rc < 2; rc++) {
      const int n = h.contigLen[(size_t)c];
      const int positions[4] = {0, 100, n - 100, n - 1};
      for (int k = 0; k < 4; k++) {
        const int position = positions[k];
        if (position < 0 || position >= n) continue;
        const int64_t enc = h.encodePosition(c, rc != 0, position);
        int dc = -1, ds = -1; bool drc = false;
        h.decode(enc, dc, drc, ds);
        if (dc != c || drc != (rc != 0) || ds != position) return 1;
        const RefPos p = decodePosition(v, enc);
        if (p.contig != c || (p.rc != 0) != (rc != 0) || p.start != position) return 2;
      }
    }
  return 0;
}

// read-side pyramid dump in the layout of oracle xmo_pyramid_dump (14 ints per block)
int64_t xmsim_pyramid_dump(const uint8_t* codes, int len, int32_t* out, int64_t capRows) {
  std::vector<PBlock> blocks((size_t)len * 64 + 64);
  std::vector<int32_t> ls((size_t)len + 8);
  int32_t status = 0;
  SeqView s;
  s.base = codes; s.len = len; s.rc = 0; s.id = 0;
  Pyramid pyr;
  pyr.init(s, blocks.data(), (int)blocks.size(), ls.data(), len + 4, &status);
  int64_t n = 0;
  for (int level = 0;; level++) {
    if (level > 0) { pyr.ensure(level); if (status) return -1; }
    int cnt = pyr.count(level);
    if (cnt == 0) break;
    for (int i = 0; i < cnt; i++) {
      PBlock h = pyr.blockAt(level, i);
      if (n < capRows) {
        int32_t* o = out + n * 14;
        o[0] = level; o[1] = h.start; o[2] = h.len; o[3] = h.fwd; o[4] = h.rev; o[5] = h.flags; o[6] = h.gapDir; o[7] = h.extraGap;
        QBlock g;
        int st = withGapAndExtension(h, s, g);
        o[8] = st; o[9] = st ? g.start : 0; o[10] = st ? g.len : 0; o[11] = st ? g.used : 0; o[12] = st ? g.fwd : 0; o[13] = st ? g.rev : 0;
      }
      n++;
    }
  }
  return n;
}

// the same with the multi blocks of a read with ambiguous bases, in the layout of the oracle's xmo_pyramid_dump_multi (12 ints per possibility);
// `scale` sizes the pools as compInit does.  -1: a capacity was exceeded (on the GPU: the read runs again with more scratch)
int64_t xmsim_pyramid_dump_multi(const uint8_t* codes, int len, int scale, int32_t* out, int64_t capRows) {
  std::vector<PBlock> blocks((size_t)len * 64 + 64);
  std::vector<int32_t> ls((size_t)len + 8);
  int32_t status = 0;
  SeqView s;
  s.base = codes; s.len = len; s.rc = 0; s.id = 0;
  Pyramid pyr;
  pyr.init(s, blocks.data(), (int)blocks.size(), ls.data(), len + 4, &status);
  MultiStore ms;
  const MultiCaps mc = multiCaps(scale);
  std::vector<Poss> pool((size_t)mc.pool);
  std::vector<CondEnt> conds((size_t)mc.conds), stack((size_t)mc.stack);
  std::vector<MFrame> frames((size_t)mc.frames);
  ms.pool = pool.data(); ms.poolUsed = 0; ms.poolCap = (int)pool.size();
  ms.conds = conds.data(); ms.condUsed = 0; ms.condCap = (int)conds.size();
  ms.stack = stack.data(); ms.stackCap = (int)stack.size();
  ms.frames = frames.data(); ms.framesCap = (int)frames.size();
  bool amb = false;
  for (int i = 0; i < len; i++) if (bpIsAmbiguous(codes[i])) amb = true;
  if (amb) pyr.ms = &ms;
  int64_t n = 0;
  for (int level = 0;; level++) {
    if (level > 0) { pyr.ensure(level); if (status) return -1; }
    int cnt = pyr.count(level);
    if (cnt == 0) break;
    for (int i = 0; i < cnt; i++) {
      const PBlock e = amb ? pyr.entryAt(level, i) : pyr.blockAt(level, i);
      const bool multi = (e.flags & F_MULTI) != 0;
      const int np = multi ? pyr.numPoss(level, e) : 1;
      for (int k = 0; k < np; k++) {
        PossView p;
        pyr.possAt(level, e, k, p);
        if (n < capRows) {
          int32_t* o = out + n * 12;
          o[0] = level; o[1] = e.start; o[2] = e.len; o[3] = multi ? np : 0; o[4] = k; o[5] = p.hasBlock;
          o[6] = p.hasBlock ? p.block.start : 0; o[7] = p.hasBlock ? p.block.len : 0; o[8] = p.hasBlock ? p.block.fwd : 0; o[9] = p.hasBlock ? p.block.rev : 0;
          o[10] = p.hasBlock ? (p.block.flags & 15) : 0;
          uint32_t hash = multi ? (uint32_t)p.condLen : 0u;
          if (multi) for (int t = 0; t < p.condLen; t++) hash = hash * 1000003u + ((p.cond[t] >> 2) * 4u + (p.cond[t] & 3u));
          o[11] = (int32_t)hash;
        }
        n++;
      }
    }
  }
  return n;
}

}  // extern "C"
