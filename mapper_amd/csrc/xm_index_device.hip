// xmapper-hip: the seed index hashed on the GPU.
// Replaces the same reference code as xm_index_host.h's hashLengths (M/HashBlock_Database.java:490-616 hashing the reference level by
// level, M/HashBlock.java:67-150 gapmers, M/PackedMap.java:99-153 bucket fill) and produces bit-identical tables; the duplication map
// (M/DuplicationDetector.java) stays on the host and reads the tables this file copies back.
//
// Design: a pyramid level is a pure function of the level below, and every block of a level is independent of the others, so a level is
// one launch over its blocks: a block with len <= maxLen emits the records of its gapmer, (table L, bucket, encoded position), and says
// whether it merges with its right neighbour; a prefix sum over those flags places the merged blocks of the next level in order.  The
// records of all contigs are sorted by (L, bucket, position) with two stable radix sorts (position first), counted per bucket, and a
// bucket that received more than maxInterestingCountPerKey records is marked overfull and stores nothing (PackedMap.get returns null for
// it); prefix sums over the buckets give the CSR offsets and each record's slot.  All of it is streaming HBM work (sort passes, scans).
// Records are made in groups of consecutive tables whose record count fits the memory budget (first a counting run over all levels, then
// one hashing run per group), which is the reference's multi-pass hashing for large genomes (M/HashBlock_Database.java:57-61,183-215).
// Contigs with ambiguity codes (GRCh38's N-runs, IUPAC codes): the GPU hashes every block that lies clear of the ambiguous bases (a block depends
// on the bases of its own span only, so those are the blocks of the plain rule), the conditional multi blocks over the ambiguous bases
// (M/MultiHashBlock.java, M/HashBlock_ParentRow.java:109-165) come from the host's windows around them (HostIndex::multiRecordsNearAmbiguity) and
// join the records before the sort; PackedMap.add's duplicate suppression for multi records (:124-153) is a pass over the sorted records.
#define XM_NOINL_LINKAGE inline  // the out-of-line functions of the shared headers are defined (strongly) by xm_capi.hip
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include "xm_index_host.h"

namespace xm {

#define XMB_CHECK(expr)                                                                                        \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP error in index build: ") + hipGetErrorString(e_) + " at " #expr); \
  } while (0)

template <typename T>
struct BuildBuf {
  T* p = nullptr;
  size_t n = 0;
  void ensure(size_t count) {
    if (count <= n && p) return;
    if (p) (void)hipFree(p);
    p = nullptr;
    n = count ? count : 1;
    XMB_CHECK(hipMalloc((void**)&p, n * sizeof(T)));
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  ~BuildBuf() { release(); }
};

#define XMB_LHIST 2048

struct LevelCtl {
  unsigned long long nRec;      // records emitted so far (all contigs of the group)
  unsigned long long nNext;     // blocks of the next level
  unsigned int anyShort;        // a block of this level had len <= maxLen
  unsigned int overflow;        // more records than the counting run announced (internal error)
};

struct LevelArgs {
  const HBlock* cur; long long n;
  const uint8_t* contig; int32_t contigLen;
  int enableGapmers, lo, maxLen, gLo, gHi, emit;
  const int32_t* capacity;       // [maxLen + 1]
  long long fwdBase, rcBase;     // encodePosition bases of this contig's two strands
  unsigned long long* hist;      // [maxLen + 1] records per table (counting run)
  unsigned long long* recKey; unsigned long long* recPos; unsigned long long recCap;
  uint32_t* mergeFlag;
  LevelCtl* ctl;
  const uint32_t* ambPrefix;     // null, or [contigLen + 1]: ambiguous bases before position i (a block over one emits nothing here)
  int posShift;                  // 1 (reference with ambiguity codes): positions are stored shifted left by one, bit 0 = the record comes from a multi block
};

__global__ void xmb_level0_kernel(const uint8_t* codes, long long n, HBlock* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  PBlock b0 = level0Block(codes[i], 0);
  HBlock h;
  h.start = (int32_t)i; h.len = 1; h.fwd = b0.fwd; h.rev = b0.rev; h.flags = b0.flags; h.gapDir = 0; h.extraGap = 0;
  out[i] = h;
}

// one block of a level: its records (or their count per table) and whether it merges with the block after it
__global__ void __launch_bounds__(256) xmb_level_kernel(LevelArgs a) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long k0 = 0, p0 = 0, k1 = 0, p1 = 0;
  int nOut = 0;
  // counting run: most blocks of a level use the same few lengths, so the counts are gathered per workgroup in LDS first
  __shared__ unsigned int lhist[XMB_LHIST];
  const bool ldsHist = !a.emit && a.maxLen < XMB_LHIST;
  if (ldsHist) {
    for (int t = (int)threadIdx.x; t <= a.maxLen; t += (int)blockDim.x) lhist[t] = 0;
    __syncthreads();
  }
  if (i < a.n) {
    const HBlock blk = a.cur[i];
    const bool clean = !a.ambPrefix || a.ambPrefix[blk.start + blk.len] == a.ambPrefix[blk.start];
    if (blk.len <= a.maxLen) {  // (a longer block's gapmer uses at least blk.len bases)
      if (a.ctl->anyShort == 0) a.ctl->anyShort = 1;
    }
    if (blk.len <= a.maxLen && clean) {
      SeqView seq;
      seq.base = a.contig; seq.len = a.contigLen; seq.rc = 0; seq.id = 0;
      QBlock g;
      int st = 1;
      if (a.enableGapmers) st = withGapAndExtension(blk, seq, g);
      else { g.start = blk.start; g.len = blk.len; g.used = blk.len; g.fwd = blk.fwd; g.rev = blk.rev; g.flags = blk.flags; }
      if (st != 0 && g.used >= a.lo && g.used <= a.maxLen) {
        const bool rml = (g.flags & F_RML) != 0, rmr = (g.flags & F_RMR) != 0;
        const bool primary = (rml != rmr) ? rml : (g.fwd >= g.rev);     // M/HashBlock.java:329-334
        const bool secondary = (rml != rmr) ? rmr : (g.fwd <= g.rev);   // :336-340
        if (!a.emit) {
          const unsigned int c = (primary ? 1u : 0u) + (secondary ? 1u : 0u);
          if (ldsHist) atomicAdd(&lhist[g.used], c); else atomicAdd(&a.hist[g.used], (unsigned long long)c);
        } else if (g.used >= a.gLo && g.used <= a.gHi) {
          const int cap = a.capacity[g.used];
          const unsigned long long table = (unsigned long long)(unsigned)(g.used - a.gLo) << 32;
          if (primary) {  // M/PackedMap.java:107-112
            int32_t r = g.fwd % cap; if (r < 0) r += cap;
            k0 = table | (unsigned)r; p0 = (unsigned long long)(a.fwdBase + g.start) << a.posShift;
            nOut = 1;
          }
          if (secondary) {  // :113-118
            int32_t r = g.rev % cap; if (r < 0) r += cap;
            const unsigned long long k = table | (unsigned)r, p = (unsigned long long)(a.rcBase + (a.contigLen - (g.start + g.len))) << a.posShift;
            if (nOut == 0) { k0 = k; p0 = p; } else { k1 = k; p1 = p; }
            nOut++;
          }
        }
      }
    }
    a.mergeFlag[i] = (i + 1 < a.n && shouldMergeBlocks(blk, a.cur[i + 1])) ? 1u : 0u;
  }
  if (ldsHist) {
    __syncthreads();
    for (int t = (int)threadIdx.x; t <= a.maxLen; t += (int)blockDim.x) { const unsigned int c = lhist[t]; if (c) atomicAdd(&a.hist[t], (unsigned long long)c); }
  }
  // one atomic per wave for the record slots
  if (a.emit) {
    const unsigned long long active = __ballot(1);
    unsigned long long base = 0;
    // inclusive prefix of nOut over the wave (64 lanes): shuffle scan
    int incl = nOut;
    for (int d = 1; d < 64; d <<= 1) { int v = __shfl_up(incl, d); if ((int)__lane_id() >= d) incl += v; }
    const int total = __shfl(incl, 63 - __builtin_clzll(active));
    if (total > 0) {
      const int leader = __ffsll((long long)active) - 1;
      if ((int)__lane_id() == leader) base = atomicAdd(&a.ctl->nRec, (unsigned long long)total);
      base = (unsigned long long)__shfl((long long)base, leader);
      const unsigned long long at = base + (unsigned long long)(incl - nOut);
      if (nOut > 0) {
        if (at + (unsigned long long)nOut > a.recCap) { a.ctl->overflow = 1; }
        else {
          a.recKey[at] = k0; a.recPos[at] = p0;
          if (nOut > 1) { a.recKey[at + 1] = k1; a.recPos[at + 1] = p1; }
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256) xmb_merge_kernel(const HBlock* cur, long long n, const uint32_t* flag, const uint32_t* offset, HBlock* next, LevelCtl* ctl) {
  xmLoadMergeRule();  // (every thread of the block: it ends with a barrier)
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (flag[i]) next[offset[i]] = mergeBlocks(cur[i], cur[i + 1]);
  if (i == n - 1) ctl->nNext = (unsigned long long)offset[i] + flag[i];
}

__global__ void xmb_amb_flag_kernel(const uint8_t* codes, long long n, uint32_t* flag) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = bpIsAmbiguous(codes[i]) ? 1u : 0u;
  if (i == n) flag[i] = 0u;
}
// sorted records of an ambiguous reference (position << 1 | multi): PackedMap.add with preventDuplicates (:124-153) - a record from a multi block
// is not added when its bucket already holds that position (= the record before it in (table, bucket, position, single-before-multi) order)
__global__ void xmb_keep_kernel(const unsigned long long* key, const unsigned long long* pos, unsigned long long n, uint32_t* keep) {
  unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const bool dup = (pos[r] & 1ull) != 0 && r > 0 && key[r - 1] == key[r] && (pos[r - 1] >> 1) == (pos[r] >> 1);
  keep[r] = dup ? 0u : 1u;
}
__global__ void xmb_compact_kernel(const unsigned long long* key, const unsigned long long* pos, unsigned long long n, const uint32_t* keep, const unsigned long long* slot,
                                   unsigned long long* outKey, unsigned long long* outPos) {
  unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || !keep[r]) return;
  outKey[slot[r]] = key[r];
  outPos[slot[r]] = pos[r] >> 1;
}
__global__ void xmb_widen_kernel(const uint32_t* in, unsigned long long n, unsigned long long* out) {
  unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) out[r] = in[r];
}

struct TableDesc { unsigned long long bucketBase; int32_t capacity, maxCount; };  // bucketBase: first of the table's capacity + 1 offset entries in the group

XM_INL int xmbTableOf(const TableDesc* t, int nTables, unsigned long long j) {  // table whose offset entries contain j
  int lo = 0, hi = nTables - 1;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (t[mid].bucketBase <= j) lo = mid; else hi = mid - 1; }
  return lo;
}

__global__ void xmb_count_kernel(const unsigned long long* key, unsigned long long n, const TableDesc* tables, unsigned long long* rawCount) {
  unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const unsigned long long k = key[r];
  atomicAdd(&rawCount[tables[k >> 32].bucketBase + (k & 0xFFFFFFFFull)], 1ull);
}

// per offset entry: what the bucket stores (nothing when it is overfull; the extra last entry of a table is no bucket)
__global__ void xmb_stored_kernel(const unsigned long long* rawCount, unsigned long long nEntries, const TableDesc* tables, int nTables, unsigned long long* stored) {
  unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nEntries) return;
  const TableDesc t = tables[xmbTableOf(tables, nTables, j)];
  const unsigned long long c = rawCount[j];
  stored[j] = (j - t.bucketBase >= (unsigned long long)t.capacity || c > (unsigned long long)t.maxCount) ? 0ull : c;
}

__global__ void xmb_offsets_kernel(const unsigned long long* rawCount, const unsigned long long* storedOff, unsigned long long nEntries, const TableDesc* tables, int nTables,
                                   uint32_t* bucketOff, unsigned long long* tableStored) {
  unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nEntries) return;
  const int ti = xmbTableOf(tables, nTables, j);
  const TableDesc t = tables[ti];
  const unsigned long long rel = storedOff[j] - storedOff[t.bucketBase];
  const bool last = j - t.bucketBase >= (unsigned long long)t.capacity;
  const bool overfull = !last && rawCount[j] > (unsigned long long)t.maxCount;
  bucketOff[j] = (uint32_t)rel | (overfull ? XM_OVERFULL : 0u);
  if (last) tableStored[ti] = rel;  // (64-bit: the host refuses a table that does not fit 31-bit offsets)
}

__global__ void xmb_place_kernel(const unsigned long long* key, const unsigned long long* pos, unsigned long long n, const TableDesc* tables, const unsigned long long* rawCount,
                                 const unsigned long long* rawOff, const unsigned long long* storedOff, unsigned long long* outPos) {
  unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const unsigned long long k = key[r];
  const TableDesc t = tables[k >> 32];
  const unsigned long long j = t.bucketBase + (k & 0xFFFFFFFFull);
  if (rawCount[j] > (unsigned long long)t.maxCount) return;
  outPos[storedOff[j] + (r - rawOff[j])] = pos[r];
}

// Duplication map, first step (M/DuplicationDetector.java:97-330): a hashcode holds a duplication only if two different positions among
// its matches and their reverse complements show the same unambiguous text (prefix + suffix of the block).  Nearly every bucket with two
// or more positions holds hash collisions only; this kernel tells the host which buckets are worth its ordered pass (buckets with more
// than eight positions are passed on unseen).  One lane per bucket of the tables in the duplication range.
struct DupArgs {
  const unsigned long long* stored; const unsigned long long* storedOff; const unsigned long long* positions;
  const TableDesc* tables; int nTables, gLo, dupMinLength, dupMaxLength, dupMinCopies;
  const uint8_t* codes; const long long* contigStart; const int32_t* contigLen; const long long* seqCumStart; int nSeq;  // nSeq = 2 * contigs
  unsigned long long* out; unsigned long long outCap; unsigned long long* outCount;
};
__global__ void __launch_bounds__(256) xmb_dup_candidates_kernel(DupArgs a, unsigned long long nEntries) {
  const unsigned long long j = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nEntries) return;
  const int ti = xmbTableOf(a.tables, a.nTables, j);
  const int L = a.gLo + ti;
  if (L < a.dupMinLength || L > a.dupMaxLength) return;
  const TableDesc t = a.tables[ti];
  if (j - t.bucketBase >= (unsigned long long)t.capacity) return;
  const int cnt = (int)a.stored[j];
  if (cnt < a.dupMinCopies) return;  // (dupMinCopies >= 2 here: the host only asks then)
  bool flag = cnt > 8;
  if (!flag) {
    int contig[16], start[16]; bool rc[16];
    for (int i = 0; i < cnt; i++) {
      const long long enc = (long long)a.positions[a.storedOff[j] + (unsigned long long)i];
      int lo = 0, hi = a.nSeq - 1;  // last sequence whose cumulative start is <= enc
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.seqCumStart[mid] <= enc) lo = mid; else hi = mid - 1; }
      contig[i] = lo >> 1; rc[i] = (lo & 1) != 0; start[i] = (int)(enc - a.seqCumStart[lo]);
      contig[cnt + i] = contig[i]; rc[cnt + i] = !rc[i]; start[cnt + i] = a.contigLen[contig[i]] - start[i] - L;  // reverseComplement(position, used length)
    }
    const int n = 2 * cnt, prefixLength = (L + 3) / 4;
    for (int x = 0; x < n && !flag; x++) {
      for (int y = x + 1; y < n && !flag; y++) {
        if (contig[x] == contig[y] && rc[x] == rc[y] && start[x] == start[y]) continue;  // (the same position twice is one position)
        SeqView vx, vy;
        vx.base = a.codes + a.contigStart[contig[x]]; vx.len = a.contigLen[contig[x]]; vx.rc = rc[x] ? 1 : 0; vx.id = 0;
        vy.base = a.codes + a.contigStart[contig[y]]; vy.len = a.contigLen[contig[y]]; vy.rc = rc[y] ? 1 : 0; vy.id = 0;
        if (start[x] < 0 || start[y] < 0 || start[x] + L > vx.len || start[y] + L > vy.len) { flag = true; break; }  // (never expected: the host decides)
        bool same = true;
        for (int i = 0; i < prefixLength && same; i++) {
          const uint8_t p = vx.at(start[x] + i), q = vy.at(start[y] + i), r = vx.at(start[x] + L - prefixLength + i), u = vy.at(start[y] + L - prefixLength + i);
          if (p != q || r != u || bpIsAmbiguous(p) || bpIsAmbiguous(r)) same = false;
        }
        if (same) flag = true;
      }
    }
  }
  if (flag) {
    const unsigned long long at = atomicAdd(a.outCount, 1ull);
    if (at < a.outCap) a.out[at] = ((unsigned long long)(unsigned)ti << 32) | (unsigned long long)(j - t.bucketBase);
  }
}

static inline unsigned gridFor(unsigned long long n, int block = 256) { return (unsigned)((n + (unsigned long long)block - 1) / (unsigned long long)block); }

static void scanU64(BuildBuf<uint8_t>& temp, const unsigned long long* in, unsigned long long* out, size_t n, hipStream_t s) {
  size_t bytes = 0;
  XMB_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, 0ull, n, rocprim::plus<unsigned long long>(), s));
  temp.ensure(bytes);
  XMB_CHECK(rocprim::exclusive_scan(temp.p, bytes, in, out, 0ull, n, rocprim::plus<unsigned long long>(), s));
}

// hashLengths(minLen, maxLen) of xm_index_host.h on the GPU.  false: not done (the caller hashes on the host).
bool deviceHashLengths(HostIndex& h, int minLen, int maxLen, int device) {
  const bool trace = getenv("XM_TRACE_BUILD") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  XMB_CHECK(hipSetDevice(device));
  hipStream_t s = nullptr;
  XMB_CHECK(hipStreamCreate(&s));
  struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{s};

  // capacities and per-key limits exactly as the host builder chooses them
  std::vector<int32_t> capacity((size_t)maxLen + 1, 0), maxCount((size_t)maxLen + 1, 0);
  const int lo = std::max(minLen, h.minInterestingSize);
  for (int L = lo; L <= maxLen; L++) {
    int cap = h.estimateRequiredCapacity(L);
    if (cap < 1) cap = 1;
    if (cap > INT32_MAX / 2) cap = INT32_MAX / 2;  // M/PackedMap.java:22-25
    capacity[(size_t)L] = cap;
    int mx = L * L;  // M/HashBlock_Database.java:569-576
    if (mx < h.maxNumShortMatches) mx = h.maxNumShortMatches;
    if (mx > 32766) mx = 32766;
    if (mx < 1) mx = 1;
    maxCount[(size_t)L] = mx;
  }

  BuildBuf<uint8_t> dCodes, dTemp;
  BuildBuf<int32_t> dCapacity;
  BuildBuf<HBlock> dCur, dNext;
  BuildBuf<uint32_t> dFlag, dOffset;
  BuildBuf<unsigned long long> dHist;
  BuildBuf<LevelCtl> dCtl;
  dCodes.ensure(h.refCodes.size());
  XMB_CHECK(hipMemcpyAsync(dCodes.p, h.refCodes.data(), h.refCodes.size(), hipMemcpyHostToDevice, s));
  dCapacity.ensure(capacity.size());
  XMB_CHECK(hipMemcpyAsync(dCapacity.p, capacity.data(), capacity.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
  dHist.ensure((size_t)maxLen + 1);
  XMB_CHECK(hipMemsetAsync(dHist.p, 0, ((size_t)maxLen + 1) * sizeof(unsigned long long), s));
  dCtl.ensure(1);
  XMB_CHECK(hipMemsetAsync(dCtl.p, 0, sizeof(LevelCtl), s));
  int32_t longest = 0;
  for (int c = 0; c < h.numContigs(); c++) longest = std::max(longest, h.contigLen[(size_t)c]);
  dCur.ensure((size_t)longest); dNext.ensure((size_t)longest); dFlag.ensure((size_t)longest); dOffset.ensure((size_t)longest);

  BuildBuf<unsigned long long> dKeyA, dPosA, dKeyB, dPosB;
  unsigned long long recCap = 0;

  // reference with ambiguity codes: the multi blocks' records come from the host (windows around the ambiguous bases), the GPU skips every block
  // over an ambiguous base (prefix counts of the ambiguous bases of the contig being hashed)
  const bool amb = h.referenceIsAmbiguous();
  std::vector<std::vector<HostIndex::Rec>> multiRecs((size_t)maxLen + 1);
  std::vector<char> contigAmb((size_t)h.numContigs(), 0);
  BuildBuf<uint32_t> dAmbFlag, dAmbPrefix;
  if (amb) {
    std::vector<int> capInt(capacity.begin(), capacity.end());
    h.multiRecordsNearAmbiguity(lo, maxLen, capInt, multiRecs);
    for (int c = 0; c < h.numContigs(); c++) {
      const uint8_t* b = h.refCodes.data() + h.contigStart[(size_t)c];
      for (int32_t i = 0; i < h.contigLen[(size_t)c]; i++) if (bpIsAmbiguous(b[i])) { contigAmb[(size_t)c] = 1; break; }
    }
    dAmbFlag.ensure((size_t)longest + 1); dAmbPrefix.ensure((size_t)longest + 1);
  }

  // every level of every contig: counting run (emit = 0) or the records of the tables [gLo, gHi]
  auto hashAll = [&](int emit, int gLo, int gHi) {
    for (int c = 0; c < h.numContigs(); c++) {
      long long n = h.contigLen[(size_t)c];
      const uint8_t* contig = dCodes.p + h.contigStart[(size_t)c];
      hipLaunchKernelGGL(xmb_level0_kernel, dim3(gridFor((unsigned long long)n)), dim3(256), 0, s, contig, n, dCur.p);
      const uint32_t* ambPrefix = nullptr;
      if (amb && contigAmb[(size_t)c]) {
        hipLaunchKernelGGL(xmb_amb_flag_kernel, dim3(gridFor((unsigned long long)n + 1)), dim3(256), 0, s, contig, n, dAmbFlag.p);
        size_t bytes = 0;
        XMB_CHECK(rocprim::exclusive_scan(nullptr, bytes, dAmbFlag.p, dAmbPrefix.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), s));
        dTemp.ensure(bytes);
        XMB_CHECK(rocprim::exclusive_scan(dTemp.p, bytes, dAmbFlag.p, dAmbPrefix.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), s));
        ambPrefix = dAmbPrefix.p;
      }
      HBlock* cur = dCur.p;
      HBlock* next = dNext.p;
      while (n > 0) {
        XMB_CHECK(hipMemsetAsync(&dCtl.p->nNext, 0, sizeof(unsigned long long) + sizeof(unsigned int), s));  // nNext, anyShort
        LevelArgs a;
        a.cur = cur; a.n = n; a.contig = contig; a.contigLen = h.contigLen[(size_t)c]; a.enableGapmers = h.enableGapmers; a.lo = lo; a.maxLen = maxLen;
        a.gLo = gLo; a.gHi = gHi; a.emit = emit; a.capacity = dCapacity.p;
        a.fwdBase = h.encodePosition(c, false, 0); a.rcBase = h.encodePosition(c, true, 0);
        a.hist = dHist.p; a.recKey = dKeyA.p; a.recPos = dPosA.p; a.recCap = recCap; a.mergeFlag = dFlag.p; a.ctl = dCtl.p;
        a.ambPrefix = ambPrefix; a.posShift = amb ? 1 : 0;
        hipLaunchKernelGGL(xmb_level_kernel, dim3(gridFor((unsigned long long)n)), dim3(256), 0, s, a);
        size_t bytes = 0;
        XMB_CHECK(rocprim::exclusive_scan(nullptr, bytes, dFlag.p, dOffset.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
        dTemp.ensure(bytes);
        XMB_CHECK(rocprim::exclusive_scan(dTemp.p, bytes, dFlag.p, dOffset.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
        hipLaunchKernelGGL(xmb_merge_kernel, dim3(gridFor((unsigned long long)n)), dim3(256), 0, s, cur, n, dFlag.p, dOffset.p, next, dCtl.p);
        XMB_CHECK(hipGetLastError());
        LevelCtl ctl;
        XMB_CHECK(hipMemcpyAsync(&ctl, dCtl.p, sizeof(ctl), hipMemcpyDeviceToHost, s));
        XMB_CHECK(hipStreamSynchronize(s));
        if (ctl.overflow) throw std::runtime_error("internal error: index build emitted more records than it counted");
        if (!ctl.anyShort) break;  // (the level after a level without short blocks is never looked at)
        n = (long long)ctl.nNext;
        std::swap(cur, next);
      }
    }
  };

  hashAll(0, 0, 0);
  std::vector<unsigned long long> hist((size_t)maxLen + 1);
  XMB_CHECK(hipMemcpy(hist.data(), dHist.p, hist.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (int L = 0; L <= maxLen; L++) hist[(size_t)L] += multiRecs[(size_t)L].size();
  auto t1 = std::chrono::steady_clock::now();

  // groups of consecutive tables whose records fit the budget (a sort needs both record arrays twice + its own scratch)
  size_t freeB = 0, totalB = 0;
  XMB_CHECK(hipMemGetInfo(&freeB, &totalB));
  unsigned long long budgetRecs = (unsigned long long)(freeB / 2) / 40;
  if (const char* e = getenv("XM_BUILD_GROUP_RECORDS")) { if (*e) budgetRecs = strtoull(e, nullptr, 10); }  // (testing: force several groups)
  if (budgetRecs < 1) budgetRecs = 1;

  if ((int)h.tables.size() < maxLen + 1) h.tables.resize((size_t)maxLen + 1);
  BuildBuf<TableDesc> dTables;
  BuildBuf<unsigned long long> dRaw, dRawOff, dStored, dStoredOff, dTableStored, dOutPos;
  BuildBuf<uint32_t> dBucketOff;
  int groups = 0;
  unsigned long long totalRecs = 0;
  double tHash = 0, tSort = 0, tCsr = 0, tDup = 0, tCopy = 0;  // (XM_TRACE_BUILD) seconds per stage; the stream is synchronised at the marks then
  auto mark = [&](double& acc, std::chrono::steady_clock::time_point& from) {
    if (!trace) return;
    (void)hipStreamSynchronize(s);
    auto now = std::chrono::steady_clock::now();
    acc += std::chrono::duration<double>(now - from).count();
    from = now;
  };
  for (int gLo = minLen; gLo <= maxLen;) {
    int gHi = gLo;
    unsigned long long nRecs = hist[(size_t)gLo];
    while (gHi + 1 <= maxLen && nRecs + hist[(size_t)(gHi + 1)] <= budgetRecs) { gHi++; nRecs += hist[(size_t)gHi]; }
    groups++;
    totalRecs += nRecs;
    const int nTables = gHi - gLo + 1;
    std::vector<TableDesc> tdesc((size_t)nTables);
    unsigned long long nEntries = 0;
    for (int k = 0; k < nTables; k++) {
      const int L = gLo + k;
      const bool empty = hist[(size_t)L] == 0;
      tdesc[(size_t)k].bucketBase = nEntries;
      tdesc[(size_t)k].capacity = empty ? 1 : capacity[(size_t)L];  // PackedMap(1, 1) placeholder (M/HashBlock_Database.java:387-393)
      tdesc[(size_t)k].maxCount = empty ? 1 : maxCount[(size_t)L];
      nEntries += (unsigned long long)tdesc[(size_t)k].capacity + 1;
    }
    dTables.ensure((size_t)nTables);
    XMB_CHECK(hipMemcpyAsync(dTables.p, tdesc.data(), sizeof(TableDesc) * (size_t)nTables, hipMemcpyHostToDevice, s));
    unsigned long long* sortedKey = nullptr;
    unsigned long long* sortedPos = nullptr;
    if (nRecs > 0) {
      auto tm = std::chrono::steady_clock::now();
      recCap = nRecs;
      dKeyA.ensure((size_t)nRecs); dPosA.ensure((size_t)nRecs); dKeyB.ensure((size_t)nRecs); dPosB.ensure((size_t)nRecs);
      XMB_CHECK(hipMemsetAsync(dCtl.p, 0, sizeof(LevelCtl), s));
      hashAll(1, gLo, gHi);
      LevelCtl ctl;
      XMB_CHECK(hipMemcpy(&ctl, dCtl.p, sizeof(ctl), hipMemcpyDeviceToHost));
      unsigned long long hostRecs = 0;
      for (int L = gLo; L <= gHi; L++) hostRecs += multiRecs[(size_t)L].size();
      if (ctl.nRec + hostRecs != nRecs) throw std::runtime_error("internal error: index build counted " + std::to_string(nRecs) + " records and emitted " + std::to_string(ctl.nRec + hostRecs));
      if (hostRecs > 0) {  // the multi blocks' records (host windows) behind the GPU's: key = (table, bucket), position << 1 | 1
        std::vector<unsigned long long> hk((size_t)hostRecs), hp((size_t)hostRecs);
        size_t at = 0;
        for (int L = gLo; L <= gHi; L++)
          for (const HostIndex::Rec& r : multiRecs[(size_t)L]) {
            hk[at] = ((unsigned long long)(unsigned)(L - gLo) << 32) | r.bucket;
            hp[at] = ((r.pos & ~HostIndex::REC_MULTI) << 1) | 1ull;
            at++;
          }
        XMB_CHECK(hipMemcpyAsync(dKeyA.p + ctl.nRec, hk.data(), sizeof(unsigned long long) * hk.size(), hipMemcpyHostToDevice, s));
        XMB_CHECK(hipMemcpyAsync(dPosA.p + ctl.nRec, hp.data(), sizeof(unsigned long long) * hp.size(), hipMemcpyHostToDevice, s));
        XMB_CHECK(hipStreamSynchronize(s));
      }
      mark(tHash, tm);
      // (L, bucket, position): stable sort by position, then by (L, bucket)
      unsigned posBits = 1;
      while (posBits < 64 && ((unsigned long long)h.seqCumStart.back() >> posBits) != 0) posBits++;
      if (amb) posBits++;  // (positions are shifted left by one, bit 0 = multi: single before multi at the same position)
      unsigned tableBits = 1;
      while ((1 << tableBits) < nTables) tableBits++;
      size_t bytes = 0;
      XMB_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, dPosA.p, dPosB.p, dKeyA.p, dKeyB.p, (size_t)nRecs, 0, posBits, s));
      dTemp.ensure(bytes);
      XMB_CHECK(rocprim::radix_sort_pairs(dTemp.p, bytes, dPosA.p, dPosB.p, dKeyA.p, dKeyB.p, (size_t)nRecs, 0, posBits, s));
      XMB_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, dKeyB.p, dKeyA.p, dPosB.p, dPosA.p, (size_t)nRecs, 0, 32 + tableBits, s));
      dTemp.ensure(bytes);
      XMB_CHECK(rocprim::radix_sort_pairs(dTemp.p, bytes, dKeyB.p, dKeyA.p, dPosB.p, dPosA.p, (size_t)nRecs, 0, 32 + tableBits, s));
      sortedKey = dKeyA.p; sortedPos = dPosA.p;
      if (amb) {  // duplicate suppression of the multi records, positions back to their plain form, records compacted into the other pair of arrays
        BuildBuf<uint32_t> dKeep;
        BuildBuf<unsigned long long> dKeep64, dSlot;
        dKeep.ensure((size_t)nRecs); dKeep64.ensure((size_t)nRecs); dSlot.ensure((size_t)nRecs);
        hipLaunchKernelGGL(xmb_keep_kernel, dim3(gridFor(nRecs)), dim3(256), 0, s, sortedKey, sortedPos, nRecs, dKeep.p);
        hipLaunchKernelGGL(xmb_widen_kernel, dim3(gridFor(nRecs)), dim3(256), 0, s, dKeep.p, nRecs, dKeep64.p);
        scanU64(dTemp, dKeep64.p, dSlot.p, (size_t)nRecs, s);
        hipLaunchKernelGGL(xmb_compact_kernel, dim3(gridFor(nRecs)), dim3(256), 0, s, sortedKey, sortedPos, nRecs, dKeep.p, dSlot.p, dKeyB.p, dPosB.p);
        XMB_CHECK(hipGetLastError());
        unsigned long long lastSlot = 0;
        uint32_t lastKeep = 0;
        XMB_CHECK(hipMemcpyAsync(&lastSlot, dSlot.p + (nRecs - 1), sizeof(lastSlot), hipMemcpyDeviceToHost, s));
        XMB_CHECK(hipMemcpyAsync(&lastKeep, dKeep.p + (nRecs - 1), sizeof(lastKeep), hipMemcpyDeviceToHost, s));
        XMB_CHECK(hipStreamSynchronize(s));
        totalRecs -= nRecs;
        nRecs = lastSlot + lastKeep;
        totalRecs += nRecs;
        sortedKey = dKeyB.p; sortedPos = dPosB.p;
      }
      mark(tSort, tm);
    }
    auto tm2 = std::chrono::steady_clock::now();
    // CSR: records per bucket, what each bucket stores, offsets, slots
    dRaw.ensure((size_t)nEntries); dRawOff.ensure((size_t)nEntries); dStored.ensure((size_t)nEntries); dStoredOff.ensure((size_t)nEntries + 1); dBucketOff.ensure((size_t)nEntries);
    dTableStored.ensure((size_t)nTables);
    XMB_CHECK(hipMemsetAsync(dRaw.p, 0, sizeof(unsigned long long) * (size_t)nEntries, s));
    if (nRecs > 0) hipLaunchKernelGGL(xmb_count_kernel, dim3(gridFor(nRecs)), dim3(256), 0, s, sortedKey, nRecs, dTables.p, dRaw.p);
    hipLaunchKernelGGL(xmb_stored_kernel, dim3(gridFor(nEntries)), dim3(256), 0, s, dRaw.p, nEntries, dTables.p, nTables, dStored.p);
    scanU64(dTemp, dRaw.p, dRawOff.p, (size_t)nEntries, s);
    scanU64(dTemp, dStored.p, dStoredOff.p, (size_t)nEntries, s);
    hipLaunchKernelGGL(xmb_offsets_kernel, dim3(gridFor(nEntries)), dim3(256), 0, s, dRaw.p, dStoredOff.p, nEntries, dTables.p, nTables, dBucketOff.p, dTableStored.p);
    XMB_CHECK(hipGetLastError());
    std::vector<unsigned long long> tableStored((size_t)nTables);
    XMB_CHECK(hipMemcpyAsync(tableStored.data(), dTableStored.p, sizeof(unsigned long long) * (size_t)nTables, hipMemcpyDeviceToHost, s));
    XMB_CHECK(hipStreamSynchronize(s));
    unsigned long long groupStored = 0;
    for (int k = 0; k < nTables; k++) {
      if (tableStored[(size_t)k] > 0x7FFFFFFFull) throw std::runtime_error("table too large for 31-bit bucket offsets");
      groupStored += tableStored[(size_t)k];
    }
    dOutPos.ensure((size_t)groupStored);
    if (nRecs > 0) hipLaunchKernelGGL(xmb_place_kernel, dim3(gridFor(nRecs)), dim3(256), 0, s, sortedKey, sortedPos, nRecs, dTables.p, dRaw.p, dRawOff.p, dStoredOff.p, dOutPos.p);
    XMB_CHECK(hipGetLastError());
    mark(tCsr, tm2);
    // duplication map: the buckets of this group's tables that are worth the host's ordered pass (first build only)
    if (!h.dupDone && h.dupMinCopies >= 2 && gHi >= h.dupMinLength && gLo <= h.dupMaxLength && groupStored > 0) {
      std::vector<long long> cs(h.contigStart.begin(), h.contigStart.end()), sc(h.seqCumStart.begin(), h.seqCumStart.end());
      BuildBuf<long long> dContigStart, dSeqCum;
      BuildBuf<int32_t> dContigLen;
      BuildBuf<unsigned long long> dFlagged, dFlagCount;
      dContigStart.ensure(cs.size()); dSeqCum.ensure(sc.size()); dContigLen.ensure(h.contigLen.size()); dFlagCount.ensure(1);
      XMB_CHECK(hipMemcpyAsync(dContigStart.p, cs.data(), cs.size() * sizeof(long long), hipMemcpyHostToDevice, s));
      XMB_CHECK(hipMemcpyAsync(dSeqCum.p, sc.data(), sc.size() * sizeof(long long), hipMemcpyHostToDevice, s));
      XMB_CHECK(hipMemcpyAsync(dContigLen.p, h.contigLen.data(), h.contigLen.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
      XMB_CHECK(hipMemsetAsync(dFlagCount.p, 0, sizeof(unsigned long long), s));
      const unsigned long long cap = std::min<unsigned long long>(nEntries, 1ull << 26);
      dFlagged.ensure((size_t)cap);
      DupArgs da;
      da.stored = dStored.p; da.storedOff = dStoredOff.p; da.positions = dOutPos.p; da.tables = dTables.p; da.nTables = nTables; da.gLo = gLo;
      da.dupMinLength = h.dupMinLength; da.dupMaxLength = h.dupMaxLength; da.dupMinCopies = h.dupMinCopies;
      da.codes = dCodes.p; da.contigStart = dContigStart.p; da.contigLen = dContigLen.p; da.seqCumStart = dSeqCum.p; da.nSeq = h.numContigs() * 2;
      da.out = dFlagged.p; da.outCap = cap; da.outCount = dFlagCount.p;
      hipLaunchKernelGGL(xmb_dup_candidates_kernel, dim3(gridFor(nEntries)), dim3(256), 0, s, da, nEntries);
      XMB_CHECK(hipGetLastError());
      unsigned long long nFlagged = 0;
      XMB_CHECK(hipMemcpyAsync(&nFlagged, dFlagCount.p, sizeof(nFlagged), hipMemcpyDeviceToHost, s));
      XMB_CHECK(hipStreamSynchronize(s));
      if (nFlagged <= cap) {  // (more than the list holds: the host walks these tables itself)
        std::vector<unsigned long long> flagged((size_t)nFlagged);
        if (nFlagged) XMB_CHECK(hipMemcpy(flagged.data(), dFlagged.p, sizeof(unsigned long long) * (size_t)nFlagged, hipMemcpyDeviceToHost));
        std::sort(flagged.begin(), flagged.end());
        for (int k = 0; k < nTables; k++) {
          const int L = gLo + k;
          if (L >= h.dupMinLength && L <= h.dupMaxLength) h.dupCandidates[L];  // (an empty list is an answer too)
        }
        for (unsigned long long v : flagged) h.dupCandidates[gLo + (int)(v >> 32)].push_back((int)(v & 0xFFFFFFFFull));
      }
    }
    mark(tDup, tm2);
    // append to the host index (the duplication pass, the inspection API and the cache read the tables there)
    const size_t offBase = h.bucketOff.size(), posBase = h.positions.size();
    h.bucketOff.resize(offBase + (size_t)nEntries);
    h.positions.resize(posBase + (size_t)groupStored);
    XMB_CHECK(hipMemcpyAsync(h.bucketOff.data() + offBase, dBucketOff.p, sizeof(uint32_t) * (size_t)nEntries, hipMemcpyDeviceToHost, s));
    if (groupStored) XMB_CHECK(hipMemcpyAsync(h.positions.data() + posBase, dOutPos.p, sizeof(uint64_t) * (size_t)groupStored, hipMemcpyDeviceToHost, s));
    XMB_CHECK(hipStreamSynchronize(s));
    mark(tCopy, tm2);
    unsigned long long posAt = 0;
    for (int k = 0; k < nTables; k++) {
      Table t;
      t.capacity = tdesc[(size_t)k].capacity; t.maxCount = tdesc[(size_t)k].maxCount;
      t.offBase = (int64_t)(offBase + (size_t)tdesc[(size_t)k].bucketBase);
      t.posBase = (int64_t)(posBase + (size_t)posAt);
      posAt += tableStored[(size_t)k];
      h.tables[(size_t)(gLo + k)] = t;
    }
    gLo = gHi + 1;
  }
  if (trace) {
    auto t2 = std::chrono::steady_clock::now();
    fprintf(stderr, "[xm] index build on the GPU: tables %d..%d, %llu records in %d group(s): counting run %.3f s, then %.3f s (hashing %.3f, sorts %.3f, CSR %.3f, "
            "duplication candidates %.3f, copy to the host %.3f)\n", minLen, maxLen, totalRecs, groups, std::chrono::duration<double>(t1 - t0).count(),
            std::chrono::duration<double>(t2 - t1).count(), tHash, tSort, tCsr, tDup, tCopy);
  }
  return true;
}

}  // namespace xm
