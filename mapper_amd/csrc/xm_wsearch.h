// xmapper-hip device core: PathAligner's best-first search (M/PathAligner.java:55-293, updates :555-719, putNode :446-473) in a form built for lane-private
// state in HBM: what the chains whose searches start in HBM mode run (scale 16 and up: long reads, reruns; pathSearchW).
//
// Such a search does not have the wave's LDS slot (10 KB, one per wave): its structures live in the lane's own temporaries in HBM, and what
// an explored entry costs is the number of DEPENDENT trips to that memory (~0.5-1 us each, however many lanes take them together).  The lane-per-read
// form (PathAlignerT<false>, xm_extend.h) makes about thirteen per entry: list entry, per update cell -> node index -> node payload, per put bucket hash ->
// key -> tail.  This form makes four:
//   * a cell of the (x, y) table IS the latest node of that cell (payload inline: the search never looks at any other node - explore() uses the
//     coordinates of a list entry only, lookups and the traceback go through the cell): one trip per lookup, and the four lookups of an update
//     and the six bases it can look at are issued together;
//   * which slots of a table are in use is a bit map (256 bytes for the cells, 64 for the keys): a search clears those and nothing else, the
//     tables themselves need no initialisation and never grow;
//   * a list entry is one 8-byte word (x, y, next);
//   * a slot of the key table holds the bucket's key and its TAIL; the bucket a put goes to is nearly always the active one or the one of the
//     previous put, which are kept in registers;
//   * the live buckets - key, slot, head in one 16-byte word each - are a dense array; priorities.poll() scans it eight words per trip (the keys
//     only ever grow, so the array stays short: a removed bucket's place is taken by the last one).
// A search can be suspended after any explored entry and continued later (WSearch lives in the arena; wsRun takes a number of steps).
// Exactly the reference's search: one bucket per exact double key with insertion order preserved, stale re-exploration, == tie-breaks; the nodes it
// puts are counted and compared with the oracle's (DevCounters::pathAlignerNodes).  What does not fit (more than 2048 nodes, 1536 cells or 256
// keys; coordinates beyond an int16) reports XM_ST_OVERFLOW and is run by the lane-per-read form in the wave's big buffer.
#pragma once
#include "xm_extend.h"

namespace xm {

// XM_PROFILE builds: where an explored entry spends its time, in shader-clock ticks of wave time (the lowest active lane of the wave counts):
// [0] poll (bucket scan), [1] list entry, [2] lookups (probes + bases), [3] update arithmetic, [4] puts, [5] state load / store, [6] traceback
#if defined(XM_PROFILE) && defined(__HIPCC__)
__device__ unsigned long long xm_ws_prof[8];
#endif
#if defined(XM_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
// (accumulated in the lane's registers while it is the wave's lowest active lane; wsRun adds them to the global counters when it returns)
#define WSP_TIC(var) unsigned long long var = clock64()
#define WSP_TOC(slot, var) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long n_ = clock64(); if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) wsAcc[slot] += n_ - var; var = n_; } while (0)
#define WSP_ACC_DECL unsigned long long wsAcc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define WS_ACC_PARAM , unsigned long long* wsAcc
#define WS_ACC_ARG , wsAcc
#define WSP_ACC_FLUSH do { for (int i_ = 0; i_ < 8; i_++) if (wsAcc[i_]) atomicAdd(&xm_ws_prof[i_], wsAcc[i_]); } while (0)
#else
#define WSP_TIC(var) do { } while (0)
#define WSP_TOC(slot, var) do { } while (0)
#define WSP_ACC_DECL do { } while (0)
#define WS_ACC_PARAM
#define WS_ACC_ARG
#define WSP_ACC_FLUSH do { } while (0)
#endif

struct alignas(16) WCell { double pen, insX, insY; uint16_t x, y; uint8_t fl, pad0; uint16_t pad; };  // 32 bytes = two 16-byte loads
struct alignas(8) WList { uint32_t xy; int32_t next; };
struct alignas(16) WBkt { double key; int32_t tail, pad; };
struct alignas(16) WLive { double key; int32_t slot, head; };  // a live bucket: its key, its slot in the key table, its first list entry

// Capacities of a search's tables, chosen when it begins: the small set holds the searches of 150 bp reads and of the pieces BlockAligner cuts
// (2048 nodes on 1536 cells, 256 keys: 90 KB, of which a search of 300 entries touches a tenth); batches of long reads give their lanes a second,
// large set (the chain's capacities: tens of thousands of nodes) that a search starts over in when it outgrows the small one.
struct WSizes {
  int32_t maxNodes, maxCells, maxBuckets, maxBlocks;
  int32_t cellSlots, cellShift, bktSlots;       // cellSlots: a multiple of 128, slot of a cell = (hash x cellSlots) >> 32; bktSlots: a power of two
  uint32_t offCellBits, offBktBits, offCells, offList, offBkt, offLive, offBlocks, bytes;
};
constexpr uint32_t WS_OFF_TABLES = 512;  // (the WSearch header lives in front)
XM_INL int wsLog2Ceil(long long v) { int k = 0; while ((1ll << k) < v) k++; return k; }
XM_INL WSizes wsSizes(int maxNodes, int maxBuckets, int maxBlocks) {
  WSizes z;
  z.maxNodes = maxNodes;
  z.maxCells = maxNodes - maxNodes / 4;
  z.maxBuckets = maxBuckets;
  z.maxBlocks = maxBlocks;
  z.cellSlots = (z.maxCells + z.maxCells / 3 + 255) & ~127;  // load <= 3/4; a multiple of 128: the bit map is cleared in 16-byte words (slot = hash x slots >> 32)
  z.cellShift = 0;
  z.bktSlots = 1 << wsLog2Ceil(2ll * (maxBuckets < 64 ? 64 : maxBuckets));
  uint32_t o = WS_OFF_TABLES;
  z.offCellBits = o; o += (uint32_t)z.cellSlots / 8;
  z.offBktBits = o; o += (uint32_t)z.bktSlots / 8;
  o = (o + 63u) & ~63u;
  z.offCells = o; o += (uint32_t)z.cellSlots * 32u;
  z.offList = o; o += (uint32_t)z.maxNodes * 8u;
  z.offBkt = o; o += (uint32_t)z.bktSlots * 16u;
  z.offLive = o; o += (uint32_t)(z.maxBuckets + 8) * 16u;
  z.offBlocks = o; o += (uint32_t)maxBlocks * (uint32_t)sizeof(ABlock);
  z.bytes = (o + 255u) & ~255u;
  return z;
}
constexpr int WS_SMALL_NODES = 2048, WS_SMALL_BUCKETS = 256;
XM_INL WSizes wsSmallSizes(int maxBlocks) { return wsSizes(WS_SMALL_NODES, WS_SMALL_BUCKETS, maxBlocks); }
// bytes of a lane's search arena: the small set, or - bigNodes > 0 - the large one (the small set's tables lie inside it)
XM_INL size_t wsArenaBytes(int maxBlocks, int bigNodes = 0, int bigBuckets = 0) {
  const WSizes sm = wsSmallSizes(maxBlocks);
  if (bigNodes <= WS_SMALL_NODES) return sm.bytes;
  const WSizes bg = wsSizes(bigNodes, bigBuckets, maxBlocks);
  return bg.bytes > sm.bytes ? bg.bytes : sm.bytes;
}

// The search's state between two runs of wsRun.  (While it runs, the scalars are in registers.)
struct WSearch {
  // the problem
  PaProblem pr;
  WSizes z;
  int32_t maxBlocks;
  int32_t startIndexA, startIndexB, textALength, textBLength, diagonal, stepDelta, startX, startY, goalX, goalY, gridW, gridH;
  int32_t searchReverse, mayExtend;
  double maxInterestingPenalty;
  // the search
  int32_t nNodes, nCells, nBuckets, nLive;
  int32_t curPos, curSlot, li, pad1;  // bucket being explored (its place among the live ones, -1: none; its slot in the key table) and the list entry to explore next
  double activePenalty;
  unsigned long long nodesPut;
  // the outcome
  int32_t done, found, status, nb;
  int32_t lastSteps, pad;  // entries explored by the last wsRun (diagnostics)
};
static_assert(sizeof(WSearch) <= WS_OFF_TABLES, "the header must fit in front of the tables");

XM_INL uint32_t wsCellHash(uint32_t key) { return key * 2654435761u; }  // (the top bits are the slot)
XM_INL uint32_t wsKeyHash(double key) {
  uint64_t kb;
  __builtin_memcpy(&kb, &key, 8);
  return (uint32_t)((kb ^ (kb >> 29)) * 0x9E3779B97F4A7C15ull >> 32);
}

// The running search: WSearch's scalars + the table pointers in locals (every method force-inlined into wsRun).
struct WRun {
  uint8_t* arena;
  WCell* cells; uint32_t* cellBits; uint32_t cellSlots;
  WList* list;
  WBkt* bkt; uint32_t* bktBits; uint32_t bktMask;
  WLive* live;
  int32_t maxNodes, maxCells, maxBuckets, maxBlocks;
  const uint8_t* qBase; int32_t qLen; bool qRc; const uint8_t* rBase;
  Params P;
  bool confident; double maxInsExt, maxDelExt;
  int32_t startIndexA, startIndexB, textALength, textBLength, diagonal, stepDelta, goalY;
  bool mayExtend;
  int32_t nNodes, nCells, nBuckets, nLive, curPos;
  double activePenalty;
  unsigned long long nodesPut;
  bool overflow;
  // the two buckets puts nearly always go to: the active one and the one of the previous put (slot in the key table, tail; tail -2: not read yet)
  int32_t actSlot, actTail, lastSlot, lastTail; double lastKey;
  int32_t firstAppendToActive;  // this step's first put into the active bucket (the explored entry's successor if it was the tail)

  static constexpr double disallowed = 1000000.0;

  XM_INL void bind(uint8_t* a, const WSizes& z) {
    arena = a;
    cells = (WCell*)(a + z.offCells); cellBits = (uint32_t*)(a + z.offCellBits); cellSlots = (uint32_t)z.cellSlots;
    list = (WList*)(a + z.offList);
    bkt = (WBkt*)(a + z.offBkt); bktBits = (uint32_t*)(a + z.offBktBits); bktMask = (uint32_t)z.bktSlots - 1u;
    live = (WLive*)(a + z.offLive);
    maxNodes = z.maxNodes; maxCells = z.maxCells; maxBuckets = z.maxBuckets;
  }
  XM_INL uint8_t charA(int i) const {
    XM_GLOBAL(const uint8_t)* const g = (XM_GLOBAL(const uint8_t)*)qBase;
    const int k = startIndexA + i;
    return qRc ? bpComplement(g[qLen - 1 - k]) : g[k];
  }
  XM_INL uint8_t charB(int i) const { return ((XM_GLOBAL(const uint8_t)*)rBase)[startIndexB + i]; }
  XM_INL int signedDist(int x, int y) const { return x - y - diagonal; }

  // A lookup in two halves: the first probe's loads (bit-map word and cell: issued for all the cells an update looks at before any is waited
  // for), then the walk to the cell's slot, or to the empty slot that ends its run (-1 - slot), which only goes on when the first probe collided.
  XM_INL void probeIssue(int x, int y, uint32_t& h, uint32_t& bw, WCell& c) const {
    h = (uint32_t)(((uint64_t)wsCellHash(((uint32_t)x << 16) | (uint32_t)y) * cellSlots) >> 32);
    bw = cellBits[h >> 5];
    c = cells[h];
  }
  XM_INL int probeResolve(int x, int y, uint32_t h, uint32_t bw, WCell& c) const {
    while (true) {
      if (!((bw >> (h & 31u)) & 1u)) return -1 - (int)h;
      if (c.x == (uint16_t)x && c.y == (uint16_t)y) return (int)h;
      h = h + 1 == cellSlots ? 0u : h + 1;
      bw = cellBits[h >> 5];
      c = cells[h];
    }
  }
  XM_INL int findCell(int x, int y, WCell& c) const {
    uint32_t h, bw;
    probeIssue(x, y, h, bw, c);
    return probeResolve(x, y, h, bw, c);
  }
  XM_INL double estimateOverallPenalty(int x, int y, double pen, double insX, double insY, uint8_t fl) const {  // :475-521
    if (!confident) return pen;
    const int sd = signedDist(x, y);
    if (fl & 1) {
      if (sd * stepDelta > 0) {
        double ext = fabs(sd * P.InsertionExtension_Penalty);
        if (ext > maxInsExt) return disallowed;
      } else {
        double ext = fabs(sd * P.DeletionExtension_Penalty);
        if (ext > maxDelExt) return disallowed;
      }
      if (fl & 2) return pen;
      double indelPenalty = dmin(P.InsertionStart_Penalty + P.InsertionExtension_Penalty, P.DeletionStart_Penalty + P.DeletionExtension_Penalty);
      return pen + indelPenalty;
    }
    if (sd * stepDelta < 0) {
      double ext = fabs(sd * P.InsertionExtension_Penalty);
      if (ext > maxInsExt) return disallowed;
      double startP = dmin(P.InsertionStart_Penalty, insX - pen);
      return pen + startP + ext;
    } else {
      double ext = fabs(sd * P.DeletionExtension_Penalty);
      if (ext > maxDelExt) return disallowed;
      double startP = dmin(P.DeletionStart_Penalty, insY - pen);
      return pen + startP + ext;
    }
  }
  // putNode :446-473.  cellSlot: where the lookup left the cell (>= 0: its slot, < 0: -1 - the empty slot), or INT32_MIN: not looked up
  XM_INL void putNode(int x, int y, double pen, double insX, double insY, uint8_t fl, int cellSlot) {
    double est = estimateOverallPenalty(x, y, pen, insX, insY, fl);
    if (est < activePenalty) est = activePenalty;
    if (nNodes >= maxNodes || x < -32768 || x > 32767 || y < -32768 || y > 32767) { overflow = true; return; }  // (a list entry packs x and y as two int16: start nodes left of / above the grid have negative coordinates)
    const int idx = nNodes;
    int slot, tail;
    if (actSlot >= 0 && est == activePenalty) {
      slot = actSlot;
      if (actTail == -2) actTail = bkt[slot].tail;  // (first put into this bucket since it became the active one)
      tail = actTail;
    } else if (lastSlot >= 0 && est == lastKey) { slot = lastSlot; tail = lastTail; }
    else {
      uint32_t h = wsKeyHash(est) & bktMask;
      uint32_t bw = bktBits[h >> 5];
      WBkt b = bkt[h];
      bool have = false;
      while (true) {
        if (!((bw >> (h & 31u)) & 1u)) break;
        if (b.key == est) { have = true; break; }
        h = (h + 1) & bktMask;
        bw = bktBits[h >> 5];
        b = bkt[h];
      }
      slot = (int)h;
      tail = have ? b.tail : -1;
      if (!have) {  // prioritizedNodes.put(key, new list) + priorities.add(key)
        if (nBuckets >= maxBuckets) { overflow = true; return; }
        nBuckets++;
        bktBits[h >> 5] = bw | (1u << (h & 31u));
        WLive L; L.key = est; L.slot = slot; L.head = idx;
        live[nLive++] = L;
      }
    }
    nNodes++;
    WList e; e.xy = ((uint32_t)(uint16_t)(int16_t)x << 16) | (uint32_t)(uint16_t)(int16_t)y; e.next = -1;
    list[idx] = e;
    if (tail >= 0) list[tail].next = idx;
    WBkt nb; nb.key = est; nb.tail = idx; nb.pad = 0;
    bkt[slot] = nb;
    if (slot == actSlot) { actTail = idx; if (firstAppendToActive < 0) firstAppendToActive = idx; }
    else { lastSlot = slot; lastTail = idx; lastKey = est; }
    // saveNode :523-539 (a cell outside the grid is never looked at)
    if (x >= 0 && y >= 0 && x < textALength + 2 && y < textBLength + 2) {
      WCell c;
      if (cellSlot == INT32_MIN) cellSlot = findCell(x, y, c);
      if (cellSlot < 0) {
        if (nCells >= maxCells) { overflow = true; return; }
        nCells++;
        cellSlot = -1 - cellSlot;
        cellBits[cellSlot >> 5] |= 1u << (cellSlot & 31);
      }
      c.pen = pen; c.insX = insX; c.insY = insY; c.x = (uint16_t)x; c.y = (uint16_t)y; c.fl = fl; c.pad0 = 0; c.pad = 0;
      cells[cellSlot] = c;
    }
    nodesPut++;
  }
  // update :555-571 + computeUpdated :573-719
  XM_INL void update(int x, int y WS_ACC_PARAM) {
    if (x <= 0 || x > textALength) return;
    if (y <= 0 || y > textBLength) return;
    // the four lookups (independent loads) and the six bases the three transitions can look at, together
    WSP_TIC(tU);
    WCell nE, nL, nU, nD;
    const int ia = x - 1, ib = y - 1;
    const uint8_t a0 = charA(ia), b0 = charB(ib);
    const uint8_t aPrev = charA(iclamp(ia - stepDelta, 0, textALength - 1)), aNext = charA(iclamp(ia + stepDelta, 0, textALength - 1));
    const uint8_t bPrev = charB(iclamp(ib - stepDelta, 0, textBLength - 1)), bNext = charB(iclamp(ib + stepDelta, 0, textBLength - 1));
    uint32_t hE, hL, hU, hD, wE, wL, wU, wD;
    probeIssue(x, y, hE, wE, nE); probeIssue(x - stepDelta, y, hL, wL, nL); probeIssue(x, y - stepDelta, hU, wU, nU); probeIssue(x - stepDelta, y - stepDelta, hD, wD, nD);
    const int sE = probeResolve(x, y, hE, wE, nE);
    const bool left = probeResolve(x - stepDelta, y, hL, wL, nL) >= 0, up = probeResolve(x, y - stepDelta, hU, wU, nU) >= 0, diag = probeResolve(x - stepDelta, y - stepDelta, hD, wD, nD) >= 0;
    const bool existing = sE >= 0;
    WSP_TOC(2, tU);
    double insertXPenalty = disallowed, insertYPenalty = disallowed, overlayPenalty = disallowed;
    if (diag) overlayPenalty = nD.pen + P.getPenalty(a0, b0);
    if (left) {
      if (y == goalY && mayExtend) {
        insertXPenalty = nL.pen + P.UnalignedPenalty;
      } else {
        bool allowed = true;
        int prevA = x - 1 - stepDelta, prevB = y - 1;
        if (prevA >= 0 && prevA < textALength && prevB >= 0 && prevB < textBLength) {
          if (!bpCanMatch(aPrev, b0)) allowed = false;
        }
        if (allowed) {
          int nextA = x - 1, nextB = y - 1 + stepDelta;
          if (nextA >= 0 && nextA < textALength && nextB >= 0 && nextB < textBLength) {
            uint8_t a = a0, b = bNext;
            if (P.getPenalty(a, b) == 0) allowed = false;
            else if (bpIsFullyAmbiguous(a) || bpIsFullyAmbiguous(b)) allowed = false;
          }
        }
        double newInsertX = allowed ? nL.pen + P.InsertionStart_Penalty + P.InsertionExtension_Penalty : disallowed;
        double extendInsertX = nL.insX + P.InsertionExtension_Penalty;
        insertXPenalty = dmin(extendInsertX, newInsertX);
      }
    }
    if (up) {
      bool allowed = true;
      int prevA = x - 1, prevB = y - 1 - stepDelta;
      if (prevA >= 0 && prevA < textALength && prevB >= 0 && prevB < textBLength) {
        if (!bpCanMatch(a0, bPrev)) allowed = false;
      }
      if (allowed) {
        int nextA = x - 1 + stepDelta, nextB = y - 1;
        if (nextA >= 0 && nextA < textALength && nextB >= 0 && nextB < textBLength) {
          uint8_t a = aNext, b = b0;
          if (P.getPenalty(a, b) == 0) allowed = false;
          else if (bpIsFullyAmbiguous(a) || bpIsFullyAmbiguous(b)) allowed = false;
        }
      }
      double newInsertY = allowed ? nU.pen + P.DeletionStart_Penalty + P.DeletionExtension_Penalty : disallowed;
      double extendInsertY = nU.insY + P.DeletionExtension_Penalty;
      insertYPenalty = dmin(extendInsertY, newInsertY);
    }
    double bestPenalty = dmin(dmin(overlayPenalty, insertXPenalty), insertYPenalty);
    if (!existing || bestPenalty < nE.pen || insertXPenalty < nE.insX || insertYPenalty < nE.insY) {
      uint8_t fl = 0;
      if (bestPenalty != disallowed) {
        if (bestPenalty == overlayPenalty) fl = nD.fl;
        else if (bestPenalty == insertXPenalty) fl = nL.fl;
        else fl = nU.fl;
        if (iabs(signedDist(x, y)) == 0) fl |= 1; else fl |= 2;
      }
      WSP_TOC(3, tU);
      putNode(x, y, bestPenalty, insertXPenalty, insertYPenalty, fl, sE);
      WSP_TOC(4, tU);
    } else {
      WSP_TOC(3, tU);
    }
  }
};

// PathAligner.align's set-up (:55-140): the problem into the arena's header, the bit maps cleared, the start nodes put
XM_NOINL void wsBegin(uint8_t* arena, const PaProblem& prIn, const WSizes& zIn) {
  WSearch* const S = (WSearch*)arena;
  const PaProblem pr = prIn;
  const WSizes z = zIn;
  S->pr = pr;
  S->z = z;
  S->maxBlocks = z.maxBlocks;  // (what the caller's block array holds - the rounding of z.bytes leaves a little more room here, which must not be used)
  S->done = 0; S->found = 0; S->status = XM_OK; S->nb = 0;
  S->nodesPut = 0; S->lastSteps = 0;
  const Section qs = pr.qs, rs = pr.rs;
  const Params params = pr.params;
  S->startIndexA = qs.start; S->startIndexB = rs.start;
  S->textALength = secLen(qs); S->textBLength = secLen(rs);
  S->gridW = S->textALength + 2; S->gridH = S->textBLength + 2;
  S->maxInterestingPenalty = secLen(qs) * params.MaxErrorRate;
  if (S->textALength < 0 || S->textBLength < 0 || S->textALength + 2 > 32000 || S->textBLength + 2 > 32000) { S->status = XM_ST_OVERFLOW; S->done = 1; return; }
  WRun w;
  w.bind(arena, z);
  w.qBase = pr.qBase; w.qLen = pr.qLen; w.qRc = pr.qRc; w.rBase = pr.rBase;
  w.P = params;
  w.confident = pr.confident; w.maxInsExt = pr.maxInsExt; w.maxDelExt = pr.maxDelExt;
  w.startIndexA = qs.start; w.startIndexB = rs.start; w.textALength = S->textALength; w.textBLength = S->textBLength;
  w.diagonal = w.startIndexB - (w.startIndexA + pr.predictedBestOffset);
  // chooseSearchReverse :17-53 (eight positions per round: their sixteen bases are read together)
  bool searchReverse = true;
  {
    int sumMis = 0, numMis = 0, sumMatch = 0, numMatch = 0;
    const int offset = pr.predictedBestOffset;
    const int s = imax(qs.start, rs.start - offset);
    const int t = imin(qs.end, rs.end - offset);
    const int length = t - s;
    for (int i0 = 0; i0 < length; i0 += 8) {
      uint8_t a[8], b[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int i = imin(i0 + k, length - 1), j = i - w.diagonal;
        a[k] = w.charA(iclamp(i, 0, imax(w.textALength - 1, 0)));
        b[k] = w.charB(iclamp(j, 0, imax(w.textBLength - 1, 0)));
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int i = i0 + k, j = i - w.diagonal;
        if (i < length && j >= 0 && j < w.textBLength) {
          if (!bpCanMatch(a[k], b[k])) { sumMis += i; numMis++; } else { sumMatch += i; numMatch++; }
        }
      }
    }
    if (numMis > 1 && numMatch > 1) searchReverse = (sumMis / numMis) > (sumMatch / numMatch);
  }
  if (searchReverse) { w.stepDelta = -1; w.mayExtend = w.startIndexB == 0; }
  else { w.stepDelta = 1; w.mayExtend = rs.end == pr.referenceLen; }
  const int width = w.textALength + 2, height = rs.end - rs.start + 2;
  int startX, startY, goalX, goalY;
  if (searchReverse) { startX = width - 1; startY = height - 1; goalX = 1; goalY = 1; }
  else { startX = 0; startY = 0; goalX = width - 2; goalY = height - 2; }
  w.goalY = goalY;
  w.nNodes = 0; w.nCells = 0; w.nBuckets = 0; w.nLive = 0; w.curPos = -1;
  w.activePenalty = 0; w.nodesPut = 0; w.overflow = false;
  w.actSlot = -1; w.actTail = -1; w.lastSlot = -1; w.lastTail = -1; w.lastKey = 0; w.firstAppendToActive = -1;
  {  // the two bit maps (320 bytes for the small set)
    struct alignas(16) Z { uint32_t w[4]; };
    Z zero; zero.w[0] = zero.w[1] = zero.w[2] = zero.w[3] = 0;
    Z* const bits = (Z*)(arena + z.offCellBits);
    const int n = (z.cellSlots + z.bktSlots) / 8 / 16;
    for (int i = 0; i < n; i++) bits[i] = zero;
  }
  const double disallowed = WRun::disallowed;
  if (w.textBLength >= w.textALength) {
    double startingInsertionStartPenalty = params.getStartingInsertionStartPenalty();
    if (!w.mayExtend) startingInsertionStartPenalty = disallowed;
    const int initialDeletionCount = imax(0, w.textBLength - w.textALength) + 1;
    for (int i = 0; i < initialDeletionCount && !w.overflow; i++) w.putNode(startX, startY + i * w.stepDelta, 0, startingInsertionStartPenalty, disallowed, 0, INT32_MIN);
  } else {
    const int initialInsertionCount = imax(0, w.textALength - w.textBLength) + 1;
    for (int i = 0; i < initialInsertionCount && !w.overflow; i++) w.putNode(startX + i * w.stepDelta, startY, 0, disallowed, disallowed, 0, INT32_MIN);
  }
  if (w.mayExtend) {
    const int initialInsertionCount = j2i(pr.maxInsExt / params.DeletionExtension_Penalty);
    for (int i = 1; i < initialInsertionCount && !w.overflow; i++) w.putNode(startX + i * w.stepDelta, startY, i * params.UnalignedPenalty, disallowed, disallowed, 0, INT32_MIN);
  }
  S->diagonal = w.diagonal; S->stepDelta = w.stepDelta; S->searchReverse = searchReverse ? 1 : 0; S->mayExtend = w.mayExtend ? 1 : 0;
  S->startX = startX; S->startY = startY; S->goalX = goalX; S->goalY = goalY;
  S->nNodes = w.nNodes; S->nCells = w.nCells; S->nBuckets = w.nBuckets; S->nLive = w.nLive;
  S->curPos = -1; S->curSlot = -1; S->li = -1; S->activePenalty = 0; S->nodesPut = w.nodesPut;
  if (w.overflow) { S->status = XM_ST_OVERFLOW; S->done = 1; }
}

// The main loop (:141-193) for at most maxSteps explored entries, then - when the goal is reached - the traceback (:195-264) and justify (:307-352).
// Returns true when the search is over (S->done): S->found, S->status, the blocks at WS_OFF_BLOCKS, S->nb.
XM_NOINL bool wsRun(uint8_t* arena, int maxSteps) {
  WSearch* const S = (WSearch*)arena;
  if (S->done) return true;
  WSP_ACC_DECL;
  WSP_TIC(tR);
  WRun w;
  const WSizes z = S->z;
  w.bind(arena, z);
  {
    const PaProblem pr = S->pr;
    w.qBase = pr.qBase; w.qLen = pr.qLen; w.qRc = pr.qRc; w.rBase = pr.rBase;
    w.P = pr.params;
    w.confident = pr.confident; w.maxInsExt = pr.maxInsExt; w.maxDelExt = pr.maxDelExt;
  }
  w.startIndexA = S->startIndexA; w.startIndexB = S->startIndexB; w.textALength = S->textALength; w.textBLength = S->textBLength;
  w.diagonal = S->diagonal; w.stepDelta = S->stepDelta; w.goalY = S->goalY; w.mayExtend = S->mayExtend != 0;
  w.nNodes = S->nNodes; w.nCells = S->nCells; w.nBuckets = S->nBuckets; w.nLive = S->nLive; w.curPos = S->curPos;
  w.activePenalty = S->activePenalty; w.nodesPut = S->nodesPut; w.overflow = false;
  w.actSlot = S->curSlot; w.actTail = -2; w.lastSlot = -1; w.lastTail = -1; w.lastKey = 0; w.firstAppendToActive = -1;
  const int goalX = S->goalX, startX = S->startX, startY = S->startY;
  const double maxInterestingPenalty = S->maxInterestingPenalty;
  int li = S->li;
  int steps = 0;
  bool haveLast = false, fail = false;
  int lastX = 0, lastY = 0;
  int32_t status = XM_OK;
  WSP_TOC(5, tR);
  // ONE loop whose every iteration explores one entry (and, in front of it, takes the next bucket when the previous one is exhausted): the searches
  // of a wave run side by side, and a loop nest - buckets outside, their entries inside - would keep the lanes whose bucket is exhausted waiting for
  // the lane with the longest one.
  while (true) {
    if (li < 0) {
      // prioritizedNodes.remove(activePenalty) and priorities.poll(): the live buckets are read eight per trip; the exhausted one's place is taken by
      // the last one, the one with the smallest key becomes the active one
      const int n = w.nLive, gone = w.curPos;
      if (n - (gone >= 0 ? 1 : 0) < 1) { status = XM_ST_INTERNAL; fail = true; break; }  // Java: NullPointerException
      int bestPos = -1;
      WLive best; best.key = 0; best.slot = -1; best.head = -1;
      WLive lastL; lastL.key = 0; lastL.slot = -1; lastL.head = -1;
      for (int i0 = 0; i0 < n; i0 += 8) {
        WLive L[8];
#pragma unroll
        for (int k = 0; k < 8; k++) L[k] = w.live[i0 + k];  // (the array has eight spare entries behind the last one)
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int i = i0 + k;
          if (i >= n) continue;
          if (i == n - 1) lastL = L[k];
          if (i == gone) continue;
          if (bestPos < 0 || L[k].key < best.key) { best = L[k]; bestPos = i; }
        }
      }
      if (gone >= 0) {
        if (gone != n - 1) { w.live[gone] = lastL; if (bestPos == n - 1) bestPos = gone; }
        w.nLive = n - 1;
      }
      w.curPos = bestPos;
      w.activePenalty = best.key;
      li = best.head;
      w.actSlot = best.slot; w.actTail = -2;
      if (w.lastSlot == best.slot) { w.actTail = w.lastTail; w.lastSlot = -1; }
      WSP_TOC(0, tR);
    }
    if (steps >= maxSteps) break;  // suspended in front of entry li
    WSP_TOC(7, tR);
    const WList e = w.list[li];
    const int x = (int)(int16_t)(uint16_t)(e.xy >> 16), y = (int)(int16_t)(uint16_t)(e.xy & 0xFFFFu);
    WSP_TOC(1, tR);
    if (w.activePenalty > maxInterestingPenalty + 0.000001) { fail = true; break; }
    if (x == goalX) { haveLast = true; lastX = x; lastY = y; break; }
    w.firstAppendToActive = -1;
    w.update(x + w.stepDelta, y WS_ACC_ARG);
    w.update(x, y + w.stepDelta WS_ACC_ARG);
    w.update(x + w.stepDelta, y + w.stepDelta WS_ACC_ARG);
    if (w.overflow) { status = XM_ST_OVERFLOW; fail = true; break; }
    // the entry's successor: what it had when it was read, or - it was the tail then - this step's first put into the active bucket
    li = e.next >= 0 ? e.next : w.firstAppendToActive;
    steps++;
    WSP_TOC(7, tR);
  }
  WSP_TOC(7, tR);
  S->nNodes = w.nNodes; S->nCells = w.nCells; S->nBuckets = w.nBuckets; S->nLive = w.nLive;
  S->curPos = w.curPos; S->curSlot = w.actSlot; S->li = li; S->activePenalty = w.activePenalty; S->nodesPut = w.nodesPut;
  S->lastSteps = steps;
  WSP_TOC(5, tR);
  WSP_ACC_FLUSH;
  if (!fail && !haveLast) return false;  // suspended
  S->done = 1;
  S->status = status;
  if (fail) { S->found = 0; return true; }
  // traceback :195-264
  const Params params = w.P;
  ABlock* const outBlocks = (ABlock*)(arena + z.offBlocks);
  const int maxBlocks = S->maxBlocks;
  int i = lastX, j = lastY;
  int nb = 0;
  const int sd = w.stepDelta;
  const int sA = w.startIndexA, sB = w.startIndexB;
  const bool searchReverse = S->searchReverse != 0;
  WCell node, other;
  while (i != startX && j != startY) {
    if (nb >= maxBlocks) { S->status = XM_ST_OVERFLOW; S->found = 0; return true; }
    w.findCell(i, j, node);
    const double bestPenalty = node.pen, insertXPenalty = node.insX, insertYPenalty = node.insY;
    ABlock blk;
    if (bestPenalty == insertXPenalty) {
      int oldI = i;
      i -= sd;
      while (i != startX) {
        w.findCell(i, j, other);
        double otherNew = other.pen + params.InsertionStart_Penalty + params.InsertionExtension_Penalty;
        double otherExtend = other.insX + params.InsertionExtension_Penalty;
        if (otherNew < otherExtend) break;
        i -= sd;
      }
      if (searchReverse) blk = ABlock{sA + oldI - 1, sB + j - 1, i - oldI, 0};
      else blk = ABlock{sA + i, sB + j, oldI - i, 0};
    } else if (bestPenalty == insertYPenalty) {
      int oldJ = j;
      j -= sd;
      while (j != startY) {
        w.findCell(i, j, other);
        double otherNew = other.pen + params.DeletionStart_Penalty + params.DeletionExtension_Penalty;
        double otherExtend = other.insY + params.DeletionExtension_Penalty;
        if (otherNew < otherExtend) break;
        j -= sd;
      }
      if (searchReverse) blk = ABlock{sA + i - 1, sB + oldJ - 1, 0, j - oldJ};
      else blk = ABlock{sA + i, sB + j, 0, oldJ - j};
    } else {
      int oldI = i, oldJ = j;
      i -= sd;
      j -= sd;
      while (i != startX && j != startY) {
        w.findCell(i, j, other);
        if (other.pen == other.insX || other.pen == other.insY) break;
        i -= sd;
        j -= sd;
      }
      if (searchReverse) blk = ABlock{sA + oldI - 1, sB + oldJ - 1, i - oldI, j - oldJ};
      else blk = ABlock{sA + i, sB + j, oldI - i, oldJ - j};
    }
    outBlocks[nb++] = blk;
  }
  if (!searchReverse) for (int a = 0, b2 = nb - 1; a < b2; a++, b2--) { ABlock t = outBlocks[a]; outBlocks[a] = outBlocks[b2]; outBlocks[b2] = t; }
  if (nb < 1) { S->found = 0; return true; }
  // justify :307-352
  ABlock* s = outBlocks;
  SeqView jq, jr;
  {
    const PaProblem pr = S->pr;
    jq.base = pr.qBase; jq.len = pr.qLen; jq.rc = pr.qRc ? 1 : 0; jq.id = 0;
    jr.base = pr.rBase; jr.len = pr.referenceLen; jr.rc = 0; jr.id = 0;
  }
  for (int k = 1; k < nb - 1; k++) {
    while (true) {
      ABlock left = s[k - 1], middle = s[k], right = s[k + 1];
      if ((middle.lenA > 0) == (middle.lenB > 0)) break;
      if (left.lenA == 0 || left.lenB == 0) break;
      if (right.lenA == 0 || right.lenB == 0) break;
      if (middle.lenA > 0) { if (jq.at(abEndA(left) - 1) != jq.at(abEndA(middle) - 1)) break; }
      else { if (jr.at(abEndB(left) - 1) != jr.at(abEndB(middle) - 1)) break; }
      s[k - 1] = ABlock{left.startA, left.startB, left.lenA - 1, left.lenB - 1};
      s[k] = ABlock{middle.startA - 1, middle.startB - 1, middle.lenA, middle.lenB};
      s[k + 1] = ABlock{right.startA - 1, right.startB - 1, right.lenA + 1, right.lenB + 1};
    }
  }
  int drop = 0;
  while (drop < nb && paCanRemoveSection(s[drop])) drop++;
  if (drop >= nb) { S->status = XM_ST_INTERNAL; S->found = 0; return true; }  // Java: IndexOutOfBoundsException
  if (drop > 0) { for (int k = drop; k < nb; k++) s[k - drop] = s[k]; nb -= drop; }
  S->nb = nb;
  S->found = 1;
  return true;
}

// pathSearchW (declared in xm_extend.h): the whole search in the caller's temporaries, with the chain's capacities
XM_NOINL bool pathSearchW(const PaProblem& pr, Arena& tmp, const Caps& caps, int32_t* status, DevCounters* dc, ABlock* outBlocks, int32_t& nb) {
  const size_t mark = tmp.used;
  const WSizes z = wsSizes(caps.maxNodes, caps.maxBuckets, caps.maxBlocks);
  uint8_t* const a = (uint8_t*)tmp.alloc(z.bytes);
  nb = 0;
  if (tmp.overflow) { *status = XM_ST_OVERFLOW; tmp.used = mark; return false; }
  wsBegin(a, pr, z);
  wsRun(a, 0x7FFFFFFF);
  const WSearch* const S = (const WSearch*)a;
  if (dc) { dc->pathAlignerCalls++; dc->pathAlignerNodes += S->nodesPut; }
  bool found = false;
  if (S->status != XM_OK) *status = S->status;
  else if (S->found) {
    found = true;
    nb = S->nb;
    const ABlock* const b = (const ABlock*)(a + z.offBlocks);
    for (int i = 0; i < nb; i++) outBlocks[i] = b[i];
  }
  tmp.used = mark;
  return found;
}

}  // namespace xm

