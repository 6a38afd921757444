// Host-side evaluation of the two transcendental terms of the path whose results decide outputs (SURVEY.md section 7 hard part 4): they are
// computed here, by the host's libm - the one the CPU oracle uses - and handed to the kernels as tables (IndexView::conf, IndexView::baLogStep),
// so that the device's own libm never takes part in a decision.  Host functions (no __device__): nothing here is compiled into a kernel.
#pragma once
#include <cmath>
#include <cstdint>
namespace xm {
// AlignerWorker.quicklyConfidentInBestAlignment (AlignerWorker.java:532-549): totalLengthForHighConfidence from the alignment's penalty and the
// query's total length; the expression order is the reference's.
inline double confidenceLengthOnHost(double penalty, int32_t queryTotalLength, double maxPenaltySpan, double mutationPenalty, double granularity, int64_t totalForwardAndReverseSize) {
  double numberOfMutations = (penalty + maxPenaltySpan) / mutationPenalty;
  double existingMutationRate = numberOfMutations / queryTotalLength;
  double probabilityMutationInSection = 1 - std::pow(1 - existingMutationRate, granularity);
  double acceptableProbability = 1.0 / (double)totalForwardAndReverseSize;
  double numberOfUnmatchedBlocksForHighConfidence = std::log(acceptableProbability) / std::log(probabilityMutationInSection);
  return numberOfUnmatchedBlocksForHighConfidence * granularity;
}
// BlockAligner.java:48: (int)Math.log(refLen / Math.log(4.0)) as a step function of refLen >= 1: step[k] = the smallest refLen whose value is >= k
inline void blockAlignerLogSteps(int32_t* step, int n) {
  auto value = [](int64_t refLen) -> int { double v = std::log((double)refLen / std::log(4.0)); return v >= 2147483647.0 ? 2147483647 : (int)v; };
  step[0] = 1;
  for (int k = 1; k < n; k++) {
    int64_t lo = 1, hi = 2147483647;   // value() is monotone in refLen
    if (value(hi) < k) { step[k] = 2147483647; continue; }
    while (lo < hi) { int64_t mid = (lo + hi) / 2; if (value(mid) >= k) hi = mid; else lo = mid + 1; }
    step[k] = (int32_t)lo;
  }
}
}  // namespace xm
