#!/bin/bash
# round 6: eight lanes per read in the passes with the rejection filter (its recurrence spread over them) - what each step prints goes to gpurun_out/group/
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
O=$R/gpurun_out/group; mkdir -p $O
export XM_TRACE_PASSES=1
c4() { tag=$1; shift; env "$@" timeout ${T:-120} python3 scripts/gpu_c4_small.py 0.02 ${N:-40000} 0 > $O/$tag.log 2>&1; echo "$tag rc=$?"; grep "pass 2\|pass 3\|step 0\|phase" $O/$tag.log | tail -3 | cut -c1-700; }
c4 redundant1 XM_GROUP_SWEEP=0
c4 redundant2 XM_GROUP_SWEEP=0
c4 coop1 A=1
c4 coop2 A=1
