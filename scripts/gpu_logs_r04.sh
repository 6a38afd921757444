#!/bin/bash
# The experiment logs profiles/r04/NOTES.md cites, in one go (run on the GPU box): scheduler against lane-per-read gapped pass on configs[1] / [2] shapes,
# 1 kb queries, the repeat-rich reference; the determinism check of the lane-per-read kernel.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
{
echo "# scripts/gpu_sched_r04.py 1 (1 M reads of 150 bp, one context; best of 2)"
timeout 900 python scripts/gpu_sched_r04.py 1 2 "XM_SCHED=0,XM_SCHED=1,XM_SCHED=1+XM_SCHED_QUANTUM=32,XM_SCHED=1+XM_SCHED_QUANTUM=100000,XM_SCHED=1+XM_SCHED_LPW=16,XM_SCHED=1+XM_SCHED_LPW=64,XM_SCHED=1+XM_SCHED_GATE=1,XM_SCHED=1+XM_SCHED_GATE=24,XM_SCHED=1+XM_SCHED_GATE=64,XM_SCHED=2,XM_SCHED=0" 2>&1 | grep -v Warn
echo "# XM_TRACE_PASSES=1 scripts/gpu_sched_r04.py 1 1 XM_SCHED=2 (split: front reads through the scheduler kernel)"
XM_TRACE_PASSES=1 timeout 600 python scripts/gpu_sched_r04.py 1 1 "XM_SCHED=2,XM_SCHED=2+XM_HEAVY_HINT=32" 2>&1 | grep "split\|XM_SCHED"
echo "# scripts/gpu_sched_r04.py 2 (1 M pairs, one context)"
timeout 900 python scripts/gpu_sched_r04.py 2 1 "XM_SCHED=0,XM_SCHED=1,XM_SCHED=1+XM_SCHED_LPW=64+XM_SCHED_GATE=16" 2>&1 | grep -v Warn
} > $O/sched_sweeps.log 2>&1
{
echo "# scripts/gpu_sched_long_r04.py (150 k easy / 100 k hard 1 kb queries, one context)"
timeout 1500 python scripts/gpu_sched_long_r04.py 150000 100000 "XM_SCHED_LONG=0,XM_SCHED_LONG=1,XM_SCHED_LONG=1+XM_SCHED_LPW=16,XM_SCHED_LONG=1+XM_SCHED_BIGSET_PCT=10,XM_WSEARCH_FROM=1000000,XM_WSEARCH_FROM=16" 2>&1 | grep -v Warn
} > $O/long_reads.log 2>&1
{
echo "# scripts/gpu_rep_r04.py (1 M reads of 150 bp on synth.repeat_rich_reference, one context, pass trace)"
timeout 900 python scripts/gpu_rep_r04.py 1000000 "XM_SCHED=0,XM_SCHED=1,XM_WSEARCH_FROM=1000000" 2>&1 | grep -v Warn
echo "# scripts/gpu_diag_r04.py (configs[1], the lane-per-read passes twice, then the scheduler twice: same streams, same counters)"
timeout 600 python scripts/gpu_diag_r04.py 0 0 1 1 2>&1 | grep -v "^\[xm\]"
} > $O/repeat_rich.log 2>&1
for f in sched_sweeps long_reads repeat_rich; do tail -n 3 $O/$f.log; done
