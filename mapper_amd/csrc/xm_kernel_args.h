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

// One launch of the wave-per-read form (xm_wave_kernel.hip).  config: 0 single-end light tier, 1 paired light tier, 2 heavy tier.
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
  uint8_t* arenas;           // heavy tier: one scratch arena per wave (the gapped chain's temporaries)
  unsigned long long arenaBytes;
  int chainScale;            // heavy tier: capacities of the gapped chain (makeCaps)
  void* waveNodes;           // heavy tier: per-wave node payloads of PathAligner's LDS-mode search (PNode[waves * XM_PAL_NODES])
  int itemsPerFetch;         // reads a wave takes from the work counter at a time
};
// waves per workgroup and LDS bytes per workgroup of a configuration, and the launch itself (returns hipError_t as int)
void xmWaveGeometry(int config, int* wavesPerBlock, int* ldsBytesPerBlock, int* wavesPerSimd);
int xmWaveLaunch(const WaveLaunch& a, void* stream);

}  // namespace xm
