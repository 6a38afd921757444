#!/bin/bash
# first GPU contact of the wave scheduler: smoke, the sweep on configs[1] and configs[2]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 900 python scripts/gpu_sched_r04.py 1 2 2>&1 | tee gpurun_out/r04/sched_sweep1.log | tail -20
timeout 900 python scripts/gpu_sched_r04.py 2 1 "XM_SCHED=0,XM_SCHED=1,XM_SCHED=1+XM_SCHED_LPW=64,XM_SCHED=1+XM_SCHED_LPW=16" 2>&1 | tee gpurun_out/r04/sched_sweep2.log | tail -20
