// xmapper-hip device core, wave-per-read form: PathAligner's best-first search, ONE WAVEFRONT PER SEARCH (included by xm_wave.h).
//
// Exact emulation of M/PathAligner.java:55-293 (bucket per exact double key, insertion order inside a bucket, stale re-exploration, `==`
// tie-breaks in the traceback, right-justify), as pathSearchT of xm_extend.h is, but with the steps of one explored list entry spread over
// the lanes of the wave instead of run by one lane:
//   - explore(x, y) (:722-729) updates (x+d, y), (x, y+d), (x+d, y+d).  The first two read disjoint cells and neither reads what the other
//     puts: lane 0 and lane 1 compute one each, every lookup (three cells, their node payloads, six bases) in flight at once, and lane 2
//     looks the third cell up meanwhile.  The outcomes are broadcast; the nodes are put in the reference's order (:446-473) by wave-uniform
//     code; the third update then takes its left and upper neighbours from the two outcomes (what the map holds after the two puts).
//   - the smallest live key (PriorityQueue.poll) is a wave-wide minimum over the bucket keys, one key per lane;
//   - clearing the cell hash, copying the two texts into LDS, chooseSearchReverse (:17-53) and the penalties of the final blocks are
//     lane-parallel loops.
// Lookup structures (cell hash, list entries, buckets, both texts) and the payloads of the most recent nodes live in the wave's LDS
// (WSearchLdsT); every node payload also goes to the wave's buffer in HBM, which is read only for a node that has left the LDS cache.
#pragma once

namespace xm {

constexpr int WS_BUCKETS = 128, WS_BHASH = 256;
constexpr int WS_TEXTA = 126, WS_TEXTB = 509;  // x in 7 bits, y in 9 bits (grid = text + 2)

// NODES: nodes (= list entries) of one search; HASH_BITS: log2 of the cell hash (at most CELLS cells); CACHE: node payloads kept in LDS
template <int NODES, int HASH_BITS, int CELLS, int CACHE>
struct WSearchLdsT {
  static constexpr int kNodes = NODES, kHashBits = HASH_BITS, kHash = 1 << HASH_BITS, kCells = CELLS, kCache = CACHE;
  uint32_t hash[1 << HASH_BITS];  // (x << 9 | y) << 16 | node index + 1 (the latest node of the cell); 0 = empty
  uint16_t xy[NODES];             // x << 9 | y of node i (= list entry i)
  uint16_t next[NODES];           // next list entry of the same bucket, 0xFFFF = none
  double bkey[WS_BUCKETS];        // exact key of bucket b; +inf once removed
  uint16_t bhead[WS_BUCKETS], btail[WS_BUCKETS];
  uint8_t bhash[WS_BHASH];        // key -> bucket + 1
  uint8_t textA[WS_TEXTA + 2], textB[WS_TEXTB + 3];
  // the payloads of the most recent nodes (direct-mapped by node index): a search looks at the nodes of the last layer or two
  double cPen[CACHE], cInsX[CACHE], cInsY[CACHE];
  uint16_t cIdx[CACHE];
  uint8_t cFl[CACHE];
};
typedef WSearchLdsT<1024, 10, 768, 256> WSearchLdsInline;   // a search run by the read's own wave (over the pyramid's chunk pool)
typedef WSearchLdsT<4096, 13, 6000, 256> WSearchLdsKernel;  // the search kernel (searches that outgrow the inline capacities)

struct WSNode { double pen, insX, insY; int32_t fl, pad; };  // node payload in HBM (32 bytes)

struct WSState {  // wave-uniform scalars of a running search
  Params p;
  int32_t textALength, textBLength, startIndexA, startIndexB;
  int32_t startX, startY, goalX, goalY, diagonal, stepDelta;
  int32_t nNodes, nCells, nBuckets, liveBuckets, lastBucket, lastTail;
  double lastKey, activePenalty, maxInterestingPenalty, maxInsExt, maxDelExt;
  bool confident, mayQueryExtendPastEndOfReference, searchReverse, overflow;
  unsigned long long nodesPut;
};

XM_INL uint32_t wsCellKey(int x, int y) { return ((uint32_t)x << 9) | (uint32_t)y; }
#define WSL_T XM_LDSP(SL)*
// XM_WAVE_PROFILE builds: shader-clock ticks of the phases of a search into DevCounters::t[1] (whole search), [7] set-up and first nodes,
// [8] smallest-key scans, [9] the lane-parallel part of explore, [15] puts and the third update
#if defined(XM_WAVE_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
struct WSTimer {
  DevCounters* dc; int slot; unsigned long long t0;
  XM_INL WSTimer(DevCounters* d, int s) : dc(d), slot(s), t0(clock64()) {}
  XM_INL ~WSTimer() { if (dc) dc->t[slot] += clock64() - t0; }
};
#define WS_TIMER(dc, slot) WSTimer wstimer_##slot(dc, slot)
#define WS_TIC(var) const unsigned long long var = clock64()
#define WS_TOC(dc, slot, var) do { if (dc) (dc)->t[slot] += clock64() - var; } while (0)
#else
#define WS_TIMER(dc, slot) do { } while (0)
#define WS_TIC(var) do { } while (0)
#define WS_TOC(dc, slot, var) do { } while (0)
#endif

// cell (x, y) -> slot of the cell in the hash (its own, or the empty one that ends its run) and the node index stored there (-1 = none)
template <class SL>
XM_INL int wsFind(WSL_T S, int x, int y, int& slot) {
  const uint32_t key = wsCellKey(x, y);
  uint32_t h = (key * 2654435761u) >> (32 - SL::kHashBits);
  uint32_t v = S->hash[h];
  while (v != 0 && (v >> 16) != key) { h = (h + 1) & (SL::kHash - 1); v = S->hash[h]; }
  slot = (int)h;
  return (int)(v & 0xFFFFu) - 1;
}

// estimateOverallPenalty :475-521
XM_INL double wsEstimate(const WSState& st, int x, int y, double pen, double insX, double insY, int fl) {
  const double disallowed = 1000000.0;
  if (!st.confident) return pen;
  const int sd = x - y - st.diagonal;
  const Params& q = st.p;
  if (fl & 1) {
    if (sd * st.stepDelta > 0) {
      const double ext = fabs(sd * q.InsertionExtension_Penalty);
      if (ext > st.maxInsExt) return disallowed;
    } else {
      const double ext = fabs(sd * q.DeletionExtension_Penalty);
      if (ext > st.maxDelExt) return disallowed;
    }
    if (fl & 2) return pen;
    const double indelPenalty = dmin(q.InsertionStart_Penalty + q.InsertionExtension_Penalty, q.DeletionStart_Penalty + q.DeletionExtension_Penalty);
    return pen + indelPenalty;
  }
  if (sd * st.stepDelta < 0) {
    const double ext = fabs(sd * q.InsertionExtension_Penalty);
    if (ext > st.maxInsExt) return disallowed;
    const double startP = dmin(q.InsertionStart_Penalty, insX - pen);
    return pen + startP + ext;
  } else {
    const double ext = fabs(sd * q.DeletionExtension_Penalty);
    if (ext > st.maxDelExt) return disallowed;
    const double startP = dmin(q.DeletionStart_Penalty, insY - pen);
    return pen + startP + ext;
  }
}

XM_INL uint32_t wsMixKey(double key) {
  uint64_t kb;
  __builtin_memcpy(&kb, &key, 8);
  return (uint32_t)((kb ^ (kb >> 29)) * 0x9E3779B97F4A7C15ull >> 40);
}

// node payload: from the LDS cache when it is there (it is, for the nodes a search step looks at), else from the wave's buffer in HBM
template <class SL>
XM_INL WSNode wsLoadNode(WSL_T S, const WSNode* nodes, int idx) {
  WSNode n;
  const int c = idx & (SL::kCache - 1);
  if ((int)S->cIdx[c] == idx) { n.pen = S->cPen[c]; n.insX = S->cInsX[c]; n.insY = S->cInsY[c]; n.fl = S->cFl[c]; n.pad = 0; }
  else n = nodes[idx];
  return n;
}

// putNode :446-473 (wave-uniform).  est: estimateOverallPenalty of the node (computed by the caller); slot: the cell's hash slot when the
// caller has just looked the cell up (-1: look it up here)
template <class SL>
XM_INL void wsPutNode(WSL_T S, WSNode* nodes, WSState& st, int x, int y, double pen, double insX, double insY, int fl, double est, int slot) {
  if (est < st.activePenalty) est = st.activePenalty;
  if (st.nNodes >= SL::kNodes || (x | y) < 0 || x > 127 || y > 511) { st.overflow = true; return; }
  int b = -1, tail = -1;
  if (st.lastBucket >= 0 && est == st.lastKey) { b = st.lastBucket; tail = st.lastTail; }
  else {
    uint32_t h = wsMixKey(est) & (WS_BHASH - 1);
    while (true) {
      const uint32_t v = S->bhash[h];
      if (v == 0) break;
      if (S->bkey[v - 1] == est) { b = (int)v - 1; break; }
      h = (h + 1) & (WS_BHASH - 1);
    }
    if (b < 0) {
      if (st.nBuckets >= WS_BUCKETS) { st.overflow = true; return; }
      b = st.nBuckets++;
      S->bkey[b] = est; S->bhead[b] = 0xFFFF; S->btail[b] = 0xFFFF; S->bhash[h] = (uint8_t)(b + 1);
      st.liveBuckets++;
    } else {
      const uint16_t t = S->btail[b];
      tail = t == 0xFFFF ? -1 : (int)t;
    }
    st.lastBucket = b; st.lastKey = est;
  }
  // the cell: its slot in the hash
  uint32_t h;
  bool taken;
  const uint32_t key = wsCellKey(x, y);
  if (slot >= 0) { h = (uint32_t)slot; const uint32_t v = S->hash[h]; taken = v != 0 && (v >> 16) == key; if (v != 0 && !taken) { int s2; wsFind(S, x, y, s2); h = (uint32_t)s2; taken = S->hash[h] != 0; } }
  else { int s2; taken = wsFind(S, x, y, s2) >= 0; h = (uint32_t)s2; }
  if (!taken) {
    if (st.nCells >= SL::kCells) { st.overflow = true; return; }
    st.nCells++;
  }
  const int idx = st.nNodes++;  // = list entry
  S->xy[idx] = (uint16_t)wsCellKey(x, y);
  S->next[idx] = 0xFFFF;
  if (tail >= 0) S->next[tail] = (uint16_t)idx; else S->bhead[b] = (uint16_t)idx;
  S->btail[b] = (uint16_t)idx;
  st.lastTail = idx;
  WV_LANE0 {  // (HBM copy: read again only long after, when the node has left the LDS cache)
    WSNode n;
    n.pen = pen; n.insX = insX; n.insY = insY; n.fl = fl; n.pad = 0;
    nodes[idx] = n;
  }
  {
    const int c = idx & (SL::kCache - 1);
    S->cPen[c] = pen; S->cInsX[c] = insX; S->cInsY[c] = insY; S->cFl[c] = (uint8_t)fl; S->cIdx[c] = (uint16_t)idx;
  }
  S->hash[h] = (key << 16) | (uint32_t)(idx + 1);
  st.nodesPut++;
  wvFence();  // (wavefront scope: orders the LDS writes for the other lanes, does not wait for the HBM copy)
}

// what an update decided (computeUpdated :573-719): the node to put, or nothing
struct WSUpdate { int32_t put, fl, slot, existing; double pen, insX, insY, est; };

// computeUpdated for target (x, y) given its four neighbours as (exists, payload) pairs and the six bases around it
XM_INL void wsCompute(const WSState& st, int x, int y, bool hasE, const WSNode& nE, bool hasL, const WSNode& nL, bool hasU, const WSNode& nU, bool hasD, const WSNode& nD,
                      uint8_t a0, uint8_t b0, uint8_t aPrev, uint8_t aNext, uint8_t bPrev, uint8_t bNext, WSUpdate& out) {
  const double disallowed = 1000000.0;
  const Params& q = st.p;
  const int d = st.stepDelta;
  double insertXPenalty = disallowed, insertYPenalty = disallowed, overlayPenalty = disallowed;
  if (hasD) overlayPenalty = nD.pen + q.getPenalty(a0, b0);
  if (hasL) {
    if (y == st.goalY && st.mayQueryExtendPastEndOfReference) {
      insertXPenalty = nL.pen + q.UnalignedPenalty;
    } else {
      bool allowed = true;
      const int prevA = x - 1 - d, prevB = y - 1;
      if (prevA >= 0 && prevA < st.textALength && prevB >= 0 && prevB < st.textBLength) {
        if (!bpCanMatch(aPrev, b0)) allowed = false;
      }
      if (allowed) {
        const int nextA = x - 1, nextB = y - 1 + d;
        if (nextA >= 0 && nextA < st.textALength && nextB >= 0 && nextB < st.textBLength) {
          if (q.getPenalty(a0, bNext) == 0) allowed = false;
          else if (bpIsFullyAmbiguous(a0) || bpIsFullyAmbiguous(bNext)) allowed = false;
        }
      }
      const double newInsertX = allowed ? nL.pen + q.InsertionStart_Penalty + q.InsertionExtension_Penalty : disallowed;
      const double extendInsertX = nL.insX + q.InsertionExtension_Penalty;
      insertXPenalty = dmin(extendInsertX, newInsertX);
    }
  }
  if (hasU) {
    bool allowed = true;
    const int prevA = x - 1, prevB = y - 1 - d;
    if (prevA >= 0 && prevA < st.textALength && prevB >= 0 && prevB < st.textBLength) {
      if (!bpCanMatch(a0, bPrev)) allowed = false;
    }
    if (allowed) {
      const int nextA = x - 1 + d, nextB = y - 1;
      if (nextA >= 0 && nextA < st.textALength && nextB >= 0 && nextB < st.textBLength) {
        if (q.getPenalty(aNext, b0) == 0) allowed = false;
        else if (bpIsFullyAmbiguous(aNext) || bpIsFullyAmbiguous(b0)) allowed = false;
      }
    }
    const double newInsertY = allowed ? nU.pen + q.DeletionStart_Penalty + q.DeletionExtension_Penalty : disallowed;
    const double extendInsertY = nU.insY + q.DeletionExtension_Penalty;
    insertYPenalty = dmin(extendInsertY, newInsertY);
  }
  const double bestPenalty = dmin(dmin(overlayPenalty, insertXPenalty), insertYPenalty);
  out.put = 0;
  if (!hasE || bestPenalty < nE.pen || insertXPenalty < nE.insX || insertYPenalty < nE.insY) {
    int fl = 0;
    if (bestPenalty != disallowed) {
      if (bestPenalty == overlayPenalty) fl = nD.fl;
      else if (bestPenalty == insertXPenalty) fl = nL.fl;
      else fl = nU.fl;
      if (x - y - st.diagonal == 0) fl |= 1; else fl |= 2;
    }
    out.put = 1; out.pen = bestPenalty; out.insX = insertXPenalty; out.insY = insertYPenalty; out.fl = fl;
    out.est = wsEstimate(st, x, y, bestPenalty, insertXPenalty, insertYPenalty, fl);
  }
}

template <class SL>
XM_INL uint8_t wsCharA(WSL_T S, const WSState& st, int i) { return S->textA[iclamp(i, 0, st.textALength > 0 ? st.textALength - 1 : 0)]; }
template <class SL>
XM_INL uint8_t wsCharB(WSL_T S, const WSState& st, int i) { return S->textB[iclamp(i, 0, st.textBLength > 0 ? st.textBLength - 1 : 0)]; }

#if defined(__HIP_DEVICE_COMPILE__)
XM_INL double wvBcastD(double v, int lane) {
  uint64_t b;
  __builtin_memcpy(&b, &v, 8);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
  b = ((uint64_t)hi << 32) | lo;
  __builtin_memcpy(&v, &b, 8);
  return v;
}
#define WV_BCAST_D(name, lane) wvBcastD(name, lane)
XM_INL int wvShflI(int v, int lane) { return __shfl(v, lane); }
XM_INL double wvShflD(double v, int lane) { return __shfl(v, lane); }
XM_INL int wvOf(int v, int lane) { return wvShflI(v, lane); }
XM_INL double wvOf(double v, int lane) { return wvShflD(v, lane); }
#define WV_OF(name, lane) wvOf(name, lane)
#else
#define WV_BCAST_D(name, lane) ((name)[lane])
#define WV_OF(name, lane) ((name)[lane])
#endif

// PathAligner.align :55-293 up to and including justify and the penalties of the result.  ok: 1 alignment (blocks, penalties), 0 null,
// -1 failed (status).  Texts: the query section [qsStart, qsEnd) of the mate view (reverse complement when qRc) and the reference section.
template <class SL>
XM_INL void wPathSearch(WSL_T S, WSNode* nodes, const IndexView& ix, const Params& paramsIn, const uint8_t* mateBase, int mateLen, const WSearchReq& r, WSearchResult& res, DevCounters* dc = nullptr) {
  WS_TIMER(dc, 1);
  const double disallowed = 1000000.0;
  res.ok = 0; res.nb = 0; res.status = 0; res.nodesPut = 0; res.totalPenalty = 0; res.alignedPenalty = 0;
  WS_TIC(tSetup);
  WSState st;
  st.p = paramsIn;
  st.p.MaxErrorRate = r.maxErrorRate;
  st.p.StartingInsertionStartFree = r.startingInsertionStartFree;
  const Params& params = st.p;
  const bool qRc = (r.seqAId & 1) != 0;
  const int referenceLen = ix.contigLen[r.contig];
  XM_GLOBAL(const uint8_t)* const qg = (XM_GLOBAL(const uint8_t)*)mateBase;
  XM_GLOBAL(const uint8_t)* const rg = (XM_GLOBAL(const uint8_t)*)(ix.refCodes + ix.contigStart[r.contig]);
  st.startIndexA = r.qsStart; st.startIndexB = r.rsStart;
  st.textALength = r.qsEnd - r.qsStart; st.textBLength = r.rsEnd - r.rsStart;
  const int endIndexA = r.qsEnd, endIndexB = r.rsEnd;
  if (st.textALength > WS_TEXTA || st.textBLength > WS_TEXTB || st.textALength < 0 || st.textBLength < 0) { res.ok = -1; res.status = XM_ST_OVERFLOW; return; }
  st.confident = r.confident != 0; st.maxInsExt = r.maxIns; st.maxDelExt = r.maxDel;
  st.overflow = false; st.nodesPut = 0; st.lastBucket = -1; st.lastTail = -1; st.lastKey = 0;
  st.nNodes = 0; st.nCells = 0; st.nBuckets = 0; st.liveBuckets = 0; st.activePenalty = 0;
  // set-up, lane-parallel: clear the hashes, copy the texts
  WV_PAR
    for (int i = wl; i < SL::kHash; i += 64) S->hash[i] = 0;
    for (int i = wl; i < SL::kCache; i += 64) S->cIdx[i] = 0xFFFF;
    for (int i = wl; i < WS_BHASH; i += 64) S->bhash[i] = 0;
    for (int i = wl; i < st.textALength; i += 64) { const int k = st.startIndexA + i; S->textA[i] = qRc ? bpComplement(qg[mateLen - 1 - k]) : qg[k]; }
    for (int i = wl; i < st.textBLength; i += 64) S->textB[i] = rg[st.startIndexB + i];
  WV_ENDPAR
  st.maxInterestingPenalty = st.textALength * params.MaxErrorRate;
  st.diagonal = st.startIndexB - (st.startIndexA + r.predictedBestOffset);
  // chooseSearchReverse :17-53 (integer sums: any order)
  {
    const int s0 = imax(st.startIndexA, st.startIndexB - r.predictedBestOffset);
    const int t0 = imin(endIndexA, endIndexB - r.predictedBestOffset);
    const int length = t0 - s0;
    int sumMis = 0, numMis = 0, sumMatch = 0, numMatch = 0;
    for (int r0 = 0; r0 < length; r0 += 64) {
      WV_VAR(int, cls);
      WV_PAR
        WV(cls) = 0;
        const int i = r0 + wl, j = i - st.diagonal;
        if (i < length && j >= 0 && j < st.textBLength) WV(cls) = bpCanMatch(wsCharA(S, st, i), S->textB[j]) ? 1 : 2;
      WV_ENDPAR
      WV_VAR(int, isMis);
      WV_VAR(int, isMatch);
      WV_PAR
        WV(isMis) = WV(cls) == 2; WV(isMatch) = WV(cls) == 1;
      WV_ENDPAR
      unsigned long long mm = WV_BALLOT(isMis), ma = WV_BALLOT(isMatch);
      numMis += __builtin_popcountll(mm); numMatch += __builtin_popcountll(ma);
      while (mm) { sumMis += r0 + __builtin_ctzll(mm); mm &= mm - 1; }
      while (ma) { sumMatch += r0 + __builtin_ctzll(ma); ma &= ma - 1; }
    }
    st.searchReverse = (numMis > 1 && numMatch > 1) ? (sumMis / numMis) > (sumMatch / numMatch) : true;
  }
  if (st.searchReverse) { st.stepDelta = -1; st.mayQueryExtendPastEndOfReference = st.startIndexB == 0; }
  else { st.stepDelta = 1; st.mayQueryExtendPastEndOfReference = endIndexB == referenceLen; }
  const int width = st.textALength + 2, height = st.textBLength + 2;
  if (st.searchReverse) { st.startX = width - 1; st.startY = height - 1; st.goalX = 1; st.goalY = 1; }
  else { st.startX = 0; st.startY = 0; st.goalX = width - 2; st.goalY = height - 2; }
  const int d = st.stepDelta;
  // the first nodes :104-149
  if (st.textBLength >= st.textALength) {
    double startingInsertionStartPenalty = params.getStartingInsertionStartPenalty();
    if (!st.mayQueryExtendPastEndOfReference) startingInsertionStartPenalty = disallowed;
    const int initialDeletionCount = imax(0, st.textBLength - st.textALength) + 1;
    for (int i = 0; i < initialDeletionCount && !st.overflow; i++) {
      const int x = st.startX, y = st.startY + i * d;
      wsPutNode(S, nodes, st, x, y, 0, startingInsertionStartPenalty, disallowed, 0, wsEstimate(st, x, y, 0, startingInsertionStartPenalty, disallowed, 0), -1);
    }
  } else {
    const int initialInsertionCount = imax(0, st.textALength - st.textBLength) + 1;
    for (int i = 0; i < initialInsertionCount && !st.overflow; i++) {
      const int x = st.startX + i * d, y = st.startY;
      wsPutNode(S, nodes, st, x, y, 0, disallowed, disallowed, 0, wsEstimate(st, x, y, 0, disallowed, disallowed, 0), -1);
    }
  }
  if (st.mayQueryExtendPastEndOfReference) {
    const int initialInsertionCount = j2i(st.maxInsExt / params.DeletionExtension_Penalty);
    for (int i = 1; i < initialInsertionCount && !st.overflow; i++) {
      const int x = st.startX + i * d, y = st.startY;
      const double pen = i * params.UnalignedPenalty;
      wsPutNode(S, nodes, st, x, y, pen, disallowed, disallowed, 0, wsEstimate(st, x, y, pen, disallowed, disallowed, 0), -1);
    }
  }
  WS_TOC(dc, 7, tSetup);
  bool haveLast = false, failed = false;
  int lastX = 0, lastY = 0;
  while (!haveLast && !failed) {
    if (st.overflow) break;
    if (st.liveBuckets < 1) { res.ok = -1; res.status = XM_ST_INTERNAL; return; }  // Java: NullPointerException
    // priorities.poll(): the live bucket with the smallest key, one or two keys per lane
    int b;
    WS_TIC(tPop);
    {
      WV_VAR(double, myKey);
      WV_VAR(int, myB);
      WV_PAR
        double best = HUGE_VAL; int bi = -1;
        for (int k = wl; k < st.nBuckets; k += 64) { const double key = S->bkey[k]; if (key < best) { best = key; bi = k; } }
        WV(myKey) = best; WV(myB) = bi;
      WV_ENDPAR
      double best = HUGE_VAL; b = -1;
      const int nl = st.nBuckets < 64 ? st.nBuckets : 64;
      for (int l = 0; l < nl; l++) { const double key = WV_BCAST_D(myKey, l); if (key < best) { best = key; b = WV_BCAST_I(myB, l); } }
    }
    WS_TOC(dc, 8, tPop);
    st.activePenalty = S->bkey[b];
    int li = S->bhead[b] == 0xFFFF ? -1 : (int)S->bhead[b];
    while (li >= 0) {
      const int xyv = S->xy[li];
      const int x = xyv >> 9, y = xyv & 511;
      if (st.activePenalty > st.maxInterestingPenalty + 0.000001) { failed = true; break; }
      if (x == st.goalX) { haveLast = true; lastX = x; lastY = y; break; }
      // explore :722-729.  The eight cells the three updates look at, one per lane: the explored cell N, the three targets A = (x+d, y),
      // B = (x, y+d), C = (x+d, y+d), and A's upper / diagonal and B's left / diagonal neighbours
      const int ax = x + d, ay = y, bx = x, by = y + d, cx2 = x + d, cy2 = y + d;
      const bool inA = !(ax <= 0 || ax > st.textALength || ay <= 0 || ay > st.textBLength);
      const bool inB = !(bx <= 0 || bx > st.textALength || by <= 0 || by > st.textBLength);
      const bool inC = !(cx2 <= 0 || cx2 > st.textALength || cy2 <= 0 || cy2 > st.textBLength);
      WS_TIC(tRegion);
      WV_VAR(int, cHas); WV_VAR(int, cSlotV); WV_VAR(int, cFlV);
      WV_VAR(double, cPenV); WV_VAR(double, cInsXV); WV_VAR(double, cInsYV);
      WV_PAR
        WV(cHas) = 0; WV(cSlotV) = -1; WV(cFlV) = 0; WV(cPenV) = 0; WV(cInsXV) = 0; WV(cInsYV) = 0;
        if (wl > 7) continue;
        // lane k: 0 N, 1 A, 2 B, 3 C, 4 A.up (x+d, y-d), 5 A.diag (x, y-d), 6 B.left (x-d, y+d), 7 B.diag (x-d, y)
        const int kx = (wl == 1 || wl == 3 || wl == 4) ? x + d : ((wl == 6 || wl == 7) ? x - d : x);
        const int ky = (wl == 2 || wl == 3 || wl == 6) ? y + d : ((wl == 4 || wl == 5) ? y - d : y);
        // (the cells of A and B are also the upper and left neighbours of C)
        const bool need = wl == 0 ? true : (wl == 1 ? (inA || inC) : (wl == 2 ? (inB || inC) : ((wl == 4 || wl == 5) ? inA : ((wl == 6 || wl == 7) ? inB : inC))));
        if (!need || kx < 0 || ky < 0 || kx > 127 || ky > 511) continue;
        int slot;
        const int idx = wsFind(S, kx, ky, slot);
        WV(cSlotV) = slot;
        if (idx >= 0) { const WSNode n = wsLoadNode(S, nodes, idx); WV(cHas) = 1; WV(cPenV) = n.pen; WV(cInsXV) = n.insX; WV(cInsYV) = n.insY; WV(cFlV) = n.fl; }
      WV_ENDPAR
      // lanes 0 and 1 compute the updates of A and B side by side, each taking its four neighbours from the lanes that looked them up
      WV_VAR(int, uPut); WV_VAR(int, uFl);
      WV_VAR(double, uPen); WV_VAR(double, uInsX); WV_VAR(double, uInsY); WV_VAR(double, uEst);
      WV_PAR
        // (A: existing = lane 1, left = N (lane 0), up = lane 4, diag = lane 5;  B: existing = lane 2, left = lane 6, up = N (lane 0), diag = lane 7)
        const int sE = wl == 0 ? 1 : 2, sL = wl == 0 ? 0 : 6, sU = wl == 0 ? 4 : 0, sD = wl == 0 ? 5 : 7;
        WSNode nE, nL, nU, nD;
        const int hasE = WV_OF(cHas, sE), hasL = WV_OF(cHas, sL), hasU = WV_OF(cHas, sU), hasD = WV_OF(cHas, sD);
        nE.pen = WV_OF(cPenV, sE); nE.insX = WV_OF(cInsXV, sE); nE.insY = WV_OF(cInsYV, sE); nE.fl = WV_OF(cFlV, sE); nE.pad = 0;
        nL.pen = WV_OF(cPenV, sL); nL.insX = WV_OF(cInsXV, sL); nL.insY = WV_OF(cInsYV, sL); nL.fl = WV_OF(cFlV, sL); nL.pad = 0;
        nU.pen = WV_OF(cPenV, sU); nU.insX = WV_OF(cInsXV, sU); nU.insY = WV_OF(cInsYV, sU); nU.fl = WV_OF(cFlV, sU); nU.pad = 0;
        nD.pen = WV_OF(cPenV, sD); nD.insX = WV_OF(cInsXV, sD); nD.insY = WV_OF(cInsYV, sD); nD.fl = WV_OF(cFlV, sD); nD.pad = 0;
        WV(uPut) = 0; WV(uFl) = 0; WV(uPen) = 0; WV(uInsX) = 0; WV(uInsY) = 0; WV(uEst) = 0;
        if (wl > 1) continue;
        const int tx = wl == 0 ? ax : bx, ty = wl == 0 ? ay : by;
        if (!(wl == 0 ? inA : inB)) continue;
        const int ia = tx - 1, ib = ty - 1;
        WSUpdate o;
        wsCompute(st, tx, ty, hasE != 0, nE, hasL != 0, nL, hasU != 0, nU, hasD != 0, nD, wsCharA(S, st, ia), wsCharB(S, st, ib), wsCharA(S, st, ia - d), wsCharA(S, st, ia + d),
                  wsCharB(S, st, ib - d), wsCharB(S, st, ib + d), o);
        if (o.put) { WV(uPut) = 1; WV(uFl) = o.fl; WV(uPen) = o.pen; WV(uInsX) = o.insX; WV(uInsY) = o.insY; WV(uEst) = o.est; }
      WV_ENDPAR
      WS_TOC(dc, 9, tRegion);
      WS_TIC(tPuts);
      // the two outcomes, wave-uniform; the nodes are put in the reference's order
      const int aPut = WV_BCAST_I(uPut, 0), bPut = WV_BCAST_I(uPut, 1);
      const int aFl = WV_BCAST_I(uFl, 0), bFl = WV_BCAST_I(uFl, 1);
      const int aSlot = WV_BCAST_I(cSlotV, 1), bSlot = WV_BCAST_I(cSlotV, 2), cSlot = WV_BCAST_I(cSlotV, 3);
      const int aHad = WV_BCAST_I(cHas, 1), bHad = WV_BCAST_I(cHas, 2), cHad = WV_BCAST_I(cHas, 3), nHad = WV_BCAST_I(cHas, 0);
      const double aPen = WV_BCAST_D(uPen, 0), aInsX = WV_BCAST_D(uInsX, 0), aInsY = WV_BCAST_D(uInsY, 0), aEst = WV_BCAST_D(uEst, 0);
      const double bPen = WV_BCAST_D(uPen, 1), bInsX = WV_BCAST_D(uInsX, 1), bInsY = WV_BCAST_D(uInsY, 1), bEst = WV_BCAST_D(uEst, 1);
      if (aPut) wsPutNode(S, nodes, st, ax, ay, aPen, aInsX, aInsY, aFl, aEst, aSlot);
      if (st.overflow) break;
      if (bPut) wsPutNode(S, nodes, st, bx, by, bPen, bInsX, bInsY, bFl, bEst, bSlot);
      if (st.overflow) break;
      if (inC) {
        // (x+d, y+d): left = what cell B holds after the two puts, up = what cell A holds, diag = the explored cell N (wave-uniform values)
        WSNode nE, nL, nU, nD;
        nE.pen = WV_BCAST_D(cPenV, 3); nE.insX = WV_BCAST_D(cInsXV, 3); nE.insY = WV_BCAST_D(cInsYV, 3); nE.fl = WV_BCAST_I(cFlV, 3); nE.pad = 0;
        nD.pen = WV_BCAST_D(cPenV, 0); nD.insX = WV_BCAST_D(cInsXV, 0); nD.insY = WV_BCAST_D(cInsYV, 0); nD.fl = WV_BCAST_I(cFlV, 0); nD.pad = 0;
        bool hasL = bHad != 0, hasU = aHad != 0;
        if (bPut) { nL.pen = bPen; nL.insX = bInsX; nL.insY = bInsY; nL.fl = bFl; hasL = true; }
        else { nL.pen = WV_BCAST_D(cPenV, 2); nL.insX = WV_BCAST_D(cInsXV, 2); nL.insY = WV_BCAST_D(cInsYV, 2); nL.fl = WV_BCAST_I(cFlV, 2); }
        if (aPut) { nU.pen = aPen; nU.insX = aInsX; nU.insY = aInsY; nU.fl = aFl; hasU = true; }
        else { nU.pen = WV_BCAST_D(cPenV, 1); nU.insX = WV_BCAST_D(cInsXV, 1); nU.insY = WV_BCAST_D(cInsYV, 1); nU.fl = WV_BCAST_I(cFlV, 1); }
        nL.pad = 0; nU.pad = 0;
        const int ia = cx2 - 1, ib = cy2 - 1;
        WSUpdate o;
        wsCompute(st, cx2, cy2, cHad != 0, nE, hasL, nL, hasU, nU, nHad != 0, nD, wsCharA(S, st, ia), wsCharB(S, st, ib), wsCharA(S, st, ia - d), wsCharA(S, st, ia + d),
                  wsCharB(S, st, ib - d), wsCharB(S, st, ib + d), o);
        // (the first two puts may have taken the empty slot this cell's lookup ended at: wsPutNode checks the slot)
        if (o.put) wsPutNode(S, nodes, st, cx2, cy2, o.pen, o.insX, o.insY, o.fl, o.est, cSlot);
        if (st.overflow) break;
      }
      WS_TOC(dc, 15, tPuts);
      li = S->next[li] == 0xFFFF ? -1 : (int)S->next[li];
    }
    if (st.overflow || failed || haveLast) break;
    // prioritizedNodes.remove(activePenalty)
    S->bkey[b] = HUGE_VAL;
    st.liveBuckets--;
    if (st.lastBucket == b) st.lastBucket = -1;
    wvFence();
  }
  res.nodesPut = (int32_t)st.nodesPut;
  if (st.overflow) { res.ok = -1; res.status = XM_ST_OVERFLOW; return; }
  if (failed || !haveLast) { res.ok = 0; return; }
  // traceback :195-264
  int i = lastX, j = lastY;
  int nb = 0;
  const int sA = st.startIndexA, sB = st.startIndexB;
  ABlock blocks[WV_MAXBLOCKS + 1];
  int sTmp;
  while (i != st.startX && j != st.startY) {
    if (nb >= WV_MAXBLOCKS) { res.ok = -1; res.status = XM_ST_OVERFLOW; return; }
    const int node = wsFind(S, i, j, sTmp);
    const WSNode n = wsLoadNode(S, nodes, node);
    ABlock blk;
    if (n.pen == n.insX) {
      const int oldI = i;
      i -= d;
      while (i != st.startX) {
        const WSNode o = wsLoadNode(S, nodes, wsFind(S, i, j, sTmp));
        const double otherNew = o.pen + params.InsertionStart_Penalty + params.InsertionExtension_Penalty;
        const double otherExtend = o.insX + params.InsertionExtension_Penalty;
        if (otherNew < otherExtend) break;
        i -= d;
      }
      if (st.searchReverse) blk = ABlock{sA + oldI - 1, sB + j - 1, i - oldI, 0};
      else blk = ABlock{sA + i, sB + j, oldI - i, 0};
    } else if (n.pen == n.insY) {
      const int oldJ = j;
      j -= d;
      while (j != st.startY) {
        const WSNode o = wsLoadNode(S, nodes, wsFind(S, i, j, sTmp));
        const double otherNew = o.pen + params.DeletionStart_Penalty + params.DeletionExtension_Penalty;
        const double otherExtend = o.insY + params.DeletionExtension_Penalty;
        if (otherNew < otherExtend) break;
        j -= d;
      }
      if (st.searchReverse) blk = ABlock{sA + i - 1, sB + oldJ - 1, 0, j - oldJ};
      else blk = ABlock{sA + i, sB + j, 0, oldJ - j};
    } else {
      const int oldI = i, oldJ = j;
      i -= d;
      j -= d;
      while (i != st.startX && j != st.startY) {
        const WSNode o = wsLoadNode(S, nodes, wsFind(S, i, j, sTmp));
        if (o.pen == o.insX || o.pen == o.insY) break;
        i -= d;
        j -= d;
      }
      if (st.searchReverse) blk = ABlock{sA + oldI - 1, sB + oldJ - 1, i - oldI, j - oldJ};
      else blk = ABlock{sA + i, sB + j, oldI - i, oldJ - j};
    }
    blocks[nb++] = blk;
  }
  if (!st.searchReverse) for (int a = 0, b2 = nb - 1; a < b2; a++, b2--) { const ABlock t = blocks[a]; blocks[a] = blocks[b2]; blocks[b2] = t; }
  if (nb < 1) { res.ok = 0; return; }
  // justify :307-352 (query / reference bases by absolute index: the sections of both texts are in LDS)
  auto qAt = [&](int k) -> uint8_t { return S->textA[k - sA]; };
  auto rAt = [&](int k) -> uint8_t { return S->textB[k - sB]; };
  for (int k = 1; k < nb - 1; k++) {
    while (true) {
      const ABlock left = blocks[k - 1], middle = blocks[k], right = blocks[k + 1];
      if ((middle.lenA > 0) == (middle.lenB > 0)) break;
      if (left.lenA == 0 || left.lenB == 0) break;
      if (right.lenA == 0 || right.lenB == 0) break;
      if (middle.lenA > 0) { if (qAt(abEndA(left) - 1) != qAt(abEndA(middle) - 1)) break; }
      else { if (rAt(abEndB(left) - 1) != rAt(abEndB(middle) - 1)) break; }
      blocks[k - 1] = ABlock{left.startA, left.startB, left.lenA - 1, left.lenB - 1};
      blocks[k] = ABlock{middle.startA - 1, middle.startB - 1, middle.lenA, middle.lenB};
      blocks[k + 1] = ABlock{right.startA - 1, right.startB - 1, right.lenA + 1, right.lenB + 1};
    }
  }
  int drop = 0;
  while (drop < nb && paCanRemoveSection(blocks[drop])) drop++;
  if (drop >= nb) { res.ok = -1; res.status = XM_ST_INTERNAL; return; }  // Java: IndexOutOfBoundsException
  if (drop > 0) { for (int k = drop; k < nb; k++) blocks[k - drop] = blocks[k]; nb -= drop; }
  // newSequenceAlignment :73-95: block penalties in block order, each block's bases in index order
  int alignedQueryLength = 0;
  double totalPenalty = 0;
  for (int k = 0; k < nb; k++) {
    const ABlock bl = blocks[k];
    double penalty = 0;
    if (bl.lenA == bl.lenB) {
      for (int r0 = 0; r0 < bl.lenA; r0 += 64) {
        WV_VAR(int, isMis);
        WV_VAR(int, isAmb);
        WV_PAR
          WV(isMis) = 0; WV(isAmb) = 0;
          const int t = r0 + wl;
          if (t >= bl.lenA) continue;
          const uint8_t a = qAt(bl.startA + t), c = rAt(bl.startB + t);
          if (!bpCanMatch(c, a)) WV(isMis) = 1;
          else if (bpPop((uint8_t)(a | c)) > 1) WV(isAmb) = 1;
        WV_ENDPAR
        const unsigned long long mMask = WV_BALLOT(isMis), aMask = WV_BALLOT(isAmb);
        unsigned long long both = mMask | aMask;
        while (both) {
          const int bit = __builtin_ctzll(both);
          both &= both - 1;
          if ((mMask >> bit) & 1ull) penalty += params.MutationPenalty;
          else penalty += params.AmbiguityPenalty * bpFalseNegativeRate((uint8_t)(qAt(bl.startA + r0 + bit) | rAt(bl.startB + r0 + bit)));
        }
      }
    } else if (bl.lenA > 0) { penalty += params.InsertionStart_Penalty; penalty += params.InsertionExtension_Penalty * bl.lenA; }
    else { penalty += params.DeletionStart_Penalty; penalty += params.DeletionExtension_Penalty * bl.lenB; }
    totalPenalty += penalty;
    alignedQueryLength += bl.lenA;
  }
  if (nb > 0 && params.StartingInsertionStartFree && blocks[0].lenB == 0) totalPenalty -= params.InsertionStart_Penalty;
  const double alignedPenalty = totalPenalty;
  totalPenalty += (double)(mateLen - alignedQueryLength) * params.UnalignedPenalty;
  if (alignedPenalty > st.textALength * params.MaxErrorRate) { res.ok = 0; return; }  // PathAligner.align :289-291
  res.ok = 1; res.nb = nb; res.totalPenalty = totalPenalty; res.alignedPenalty = alignedPenalty;
  for (int k = 0; k < nb; k++) res.blocks[k] = blocks[k];
}

// runs the request waiting in `M` and appends its result to the memo
template <class SL>
XM_INL void wRunSearch(WSL_T S, WSNode* nodes, const IndexView& ix, const Params& params, const uint8_t* mateBase, int mateLen, WMemo* M, DevCounters* dc = nullptr) {
  const WSearchReq r = M->req;
  WSearchResult res;
  wPathSearch(S, nodes, ix, params, mateBase, mateLen, r, res, dc);
  WV_LANE0 {
    const int k = M->count;
    WSearchResult* dst = &M->res[k];
    dst->ok = res.ok; dst->nb = res.nb; dst->status = res.status; dst->nodesPut = res.nodesPut; dst->totalPenalty = res.totalPenalty; dst->alignedPenalty = res.alignedPenalty;
    for (int i = 0; i < res.nb; i++) dst->blocks[i] = res.blocks[i];
    M->count = k + 1;
    M->pending = 0;
  }
}

}  // namespace xm
