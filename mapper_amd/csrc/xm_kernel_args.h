// Argument structures shared by the kernels of xm_capi.hip (lane-per-read passes) and xm_wave_kernel.hip (wave-per-read passes).
#pragma once
#include <stdint.h>
#include "xm_defs.h"

namespace xm {

struct BatchView {  // the resident batch (xm_query_batch in HBM)
  int64_t nq;
  const int32_t* mateCount;
  const int64_t* mateOffset;
  const int32_t* mateLength;
  const uint8_t* codes;
  const double* expectedInner;
  const double* deviation;
};

struct OutView {
  int32_t* ints; double* dbls;          // result arenas
  unsigned long long intCap, dblCap;
  unsigned long long* cursor;           // [0] ints used, [1] dbls used
  int32_t* status;                      // [nq]
  int64_t* intOff; int64_t* dblOff;     // [nq] offsets into the arenas
  int32_t* intLen; int32_t* dblLen;     // [nq]
};

// One launch of the wave-per-read form (xm_wave_kernel.hip).  config: 0 / 1 light tier (single-end / with pairs), 2 / 3 chain tier,
// 4 chain tier with the largest capacities.
struct WMemo;
struct WaveLaunch {
  int config;
  int grid, block;
  IndexView ix;
  Params params;
  BatchView batch;
  const int64_t* todo;       // null: all reads
  long long nTodo;
  OutView out;
  unsigned long long* nextItem;
  DevCounters* counters;
  WMemo* memoBase;           // chain tiers: the reads' memos (search requests and results), memo of read q = memoBase[slotOf[q]]
  const int32_t* slotOf;
  void* waveNodes;           // chain tiers: per wave, the node payloads of its inline searches (xmWaveInlineNodeBytes() each)
  int itemsPerFetch;         // reads a wave takes from the work counter at a time
};
// The search kernel: one wavefront per waiting search request (PathAligner's best-first emulation of xm_extend.h).
struct SearchLaunch {
  int grid, block;
  IndexView ix;
  Params params;
  BatchView batch;
  const int64_t* list;       // reads with a waiting request
  long long n;
  WMemo* memoBase;
  const int32_t* slotOf;
  unsigned long long* nextItem;
  void* waveNodes;           // per wave: the node payloads of its search
  DevCounters* counters;     // (profile builds: phase timers)
};
// waves per workgroup, LDS bytes per workgroup and waves per SIMD a configuration is compiled for; the launches (return hipError_t as int)
int xmWaveInlineNodeBytes();
void xmWaveGeometry(int config, int* wavesPerBlock, int* ldsBytesPerBlock, int* wavesPerSimd);
int xmWaveLaunch(const WaveLaunch& a, void* stream);
void xmSearchGeometry(int* wavesPerBlock, int* ldsBytesPerBlock, int* wavesPerSimd, int* memoBytes, int* nodeBytesPerWave);
int xmSearchLaunch(const SearchLaunch& a, void* stream);
int xmMemoInitLaunch(WMemo* memoBase, long long n, void* stream);
// test entry (xm_test_local_align): one wave-cooperative search over the whole of two given texts
struct TestSearch {
  int big;                   // 1: the search kernel's capacities, 0: those of the chain tiers' inline searches
  IndexView ix;              // only refCodes / contigStart / contigLen of contig 0 are read
  Params params;
  const uint8_t* query;
  int queryLength, referenceLength, predictedBestOffset, confident, blockCap;
  double maxIns, maxDel;
  void* nodes;               // xmTestWaveSearchNodeBytes(big) bytes
  int32_t* outInts;          // ok, nb, status, nodesPut, then nb x (startA, startB, lenA, lenB)
  double* outDbls;           // total penalty, aligned penalty
};
int xmTestWaveSearchNodeBytes(int big);
int xmTestWaveSearchLaunch(const TestSearch& t, void* stream);

}  // namespace xm
