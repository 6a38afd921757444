// xmapper-hip device core: the gapped pass as a wave-level scheduler.
//
// The gapped chain (HashBlock_Aligner -> BlockAligner -> PathAligner, M/QueryMatch_Aligner.java:18-29) calls PathAligner's best-first search
// (M/PathAligner.java:55-293) from the bottom of a deep, data-dependent call stack: the lanes of a wave reach their searches at different pieces of
// different candidates, so inside the chain a search runs with one lane of the wave active.  The scheduler turns that inside out.  Every lane of a
// wave holds one read; the wave alternates between two phases that all its lanes execute together:
//   chain phase  - every lane that has chain work advances its read until the read is finished or NEEDS A SEARCH: pathAlign leaves the request
//                  in the read's memo (in-lane form: the texts stay where they are) and the read unwinds with XM_ST_NEED_PATH;
//   search phase - every lane that holds a request runs its search, all of them in the same loop: one explored entry per lane per iteration.
// A read whose search is done re-enters alignRead at the candidate it stopped in (AlignReadState) and finds the results of its finished calls -
// searches, BlockAligner pieces, whole alignMatch calls - in its memo log (MemoHdr, xm_extend.h), so it is back at the point where it stopped after a
// short replay, with the lanes of the wave replaying side by side.  Lanes whose read is finished take the next read of the pass's list.
// The functions here are what the kernel (xm_capi.hip, xm_sched_kernel) and the host simulation of the tests (tests/hostsim) both run.
#pragma once
#include "xm_worker.h"
#include "xm_wsearch.h"

namespace xm {

// A lane's own search arrays (xm_wsearch.h): the small set holds the searches of 150 bp reads and pairs (under one in a hundred outgrows 1008 nodes, fewer
// 2048) and of the pieces BlockAligner cuts; what outgrows it starts over in a large set - a buffer of the launch's pool (batches of long reads: the chain's
// capacities; the lanes do not own one each: a lane's scratch is what limits the reads in flight there) - or runs in the wave's big buffer (SearchPool),
// one search at a time, in the lane-per-read form.
XM_INL size_t schedSearchArenaBytes(const Caps& chain) { return wsSmallSizes(chain.maxBlocks).bytes; }

// chain phase, a read that was parked at a search: back into alignRead at the candidate it stopped in (cx.ar.phase 1, 3 or 4)
XM_INL void schedReplay(ReadCtx& cx, ReadResult& rr) {
  cx.status = XM_OK;
  cx.memoCursor = 0;
  cx.tmp.used = 0; cx.tmp.overflow = false;
#if defined(XM_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
  if (cx.memo) cx.memo->pad2 = (int64_t)clock64();
#endif
  alignRead(cx, rr, true);
}
// after a chain phase: is the read parked at a search?  (Its candidate is counted by the run that finishes it.)
XM_INL bool schedParked(ReadCtx& cx) {
  if (cx.status != XM_ST_NEED_PATH) return false;
  if (cx.dc) { cx.dc->candidatesExtended = cx.ar.candidatesAtCall; cx.dc->refWindowBytes = cx.ar.refWindowBytesAtCall; }
  return true;
}
// the request a parked read left in its memo, as PathAligner.align's arguments
XM_INL void schedRequest(const MemoHdr* m, PaProblem& pr) {
  pr.qBase = m->qBase; pr.qLen = m->qLen; pr.qRc = m->qRc != 0; pr.rBase = m->rBase; pr.referenceLen = m->referenceLen;
  pr.qs = Section{m->qsStart, m->qsEnd}; pr.rs = Section{m->rsStart, m->rsEnd};
  pr.params = m->params;
  pr.confident = m->confident != 0; pr.maxInsExt = m->maxInsExt; pr.maxDelExt = m->maxDelExt; pr.predictedBestOffset = m->predictedBestOffset;
}
// the search's outcome goes to the end of the read's log, where the replay will look for it
XM_INL void schedLogResult(MemoHdr* m, bool found, ABlock* blocks, int nb, int32_t st) {
  SeqAl al;
  al.blocks = blocks; al.nb = 0; al.contig = 0; al.referenceReversed = 0; al.seqAId = 0; al.totalPenalty = 0; al.alignedPenalty = 0;
  int32_t cursor = 0;
  if (!memoPut(m, cursor, m->logBytes, MEMO_PATH, found, al, found ? nb : 0, st)) {
    // no room for the blocks: the replay must still find an entry, and it sends the read to a pass with more scratch
    m->logBytes = m->logBytes < m->logCap - (int)sizeof(MemoEntry) ? m->logBytes : m->logCap - (int)sizeof(MemoEntry);
    memoPut(m, cursor, m->logBytes, MEMO_PATH, false, al, 0, XM_ST_OVERFLOW);
  }
  m->hasRequest = 0;
}
// search phase.  schedSearchBegin: the parked read's request becomes a search in the lane's own arrays.  schedSearchRun: up to maxSteps explored
// entries of it; true when the search is over - its outcome is then in the read's log, or *big is set: it outgrew the lane's arrays (nothing logged,
// nothing counted) and the caller runs it again with the chain's capacities (schedSearchBig).
XM_INL void schedSearchBegin(MemoHdr* m, void* searchArena, const Caps& caps) {
  PaProblem pr;
  schedRequest(m, pr);
  wsBegin((uint8_t*)searchArena, pr, wsSmallSizes(caps.maxBlocks));
}
// -> 0: suspended (more steps to go); 1: over, outcome logged; 2: it outgrew the small set (nothing logged, nothing counted): the caller gives it a large
// set (schedSearchRestartBig) or runs it in the lane-per-read form (schedSearchBig); 3: it outgrew the large set too (the read runs again with more scratch)
XM_INL int schedSearchRun(MemoHdr* m, void* searchArena, int maxSteps, DevCounters* dc) {
  if (!wsRun((uint8_t*)searchArena, maxSteps)) return 0;
  const WSearch* const S = (const WSearch*)searchArena;
  if (S->status == XM_ST_OVERFLOW) return S->z.maxNodes <= WS_SMALL_NODES ? 2 : 3;
  if (dc) { dc->pathAlignerCalls++; dc->pathAlignerNodes += S->nodesPut; }
  schedLogResult(m, S->found != 0, (ABlock*)((uint8_t*)searchArena + S->z.offBlocks), S->nb, S->status);
  return 1;
}
// the search that outgrew the small set, once more from the start in a large set (batches of long reads: a buffer of the launch's pool, sized for the chain's capacities)
XM_INL size_t schedBigSetBytes(const Caps& caps) { return wsSizes(caps.maxNodes, caps.maxBuckets, caps.maxBlocks).bytes; }
XM_INL void schedSearchRestartBig(MemoHdr* m, void* bigArena, const Caps& caps) {
  PaProblem pr;
  schedRequest(m, pr);
  wsBegin((uint8_t*)bigArena, pr, wsSizes(caps.maxNodes, caps.maxBuckets, caps.maxBlocks));
}
// ... with the chain's full capacities in `big` (the wave's buffer, or a host buffer in the simulation); an overflow there is the read's
XM_INL void schedSearchBig(MemoHdr* m, Arena& big, const Caps& caps, DevCounters* dc) {
  PaProblem pr;
  schedRequest(m, pr);
  const size_t mark = big.used;
  ABlock* blocks = arenaArray<ABlock>(big, caps.maxBlocks);
  int32_t st = big.overflow ? (int32_t)XM_ST_OVERFLOW : (int32_t)XM_OK;
  int32_t nb = 0;
  bool found = false;
  if (!st) found = pathSearchHbm(pr, big, searchPoolCaps(caps), &st, dc, blocks, nb);  // (the wave's buffer: sized for searchPoolCaps)
  schedLogResult(m, found, blocks, nb, st);
  big.used = mark;
}

}  // namespace xm
