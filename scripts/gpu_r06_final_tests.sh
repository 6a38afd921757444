#!/bin/bash
# round 6, final build: the whole GPU tier (log -> gpurun_out/final/gputests.log), then reads-per-wave and eight-against-two lanes on configs[4] at reduced size
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
O=$R/gpurun_out/final; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=12 > $O/gputests.log 2>&1; echo "gpu tier rc=$?"; tail -18 $O/gputests.log | cut -c1-300
export XM_TRACE_PASSES=1
c4() { tag=$1; shift; env "$@" timeout ${T:-150} python3 scripts/gpu_c4_small.py 0.02 ${N:-40000} 0 > $O/$tag.log 2>&1; echo "$tag rc=$?"; grep "pass 2\|step 0" $O/$tag.log | tail -2 | cut -c1-420; }
c4 lpw3 XM_FULL_LPW=3
c4 lpw8 XM_FULL_LPW=8
c4 mild_two C4_MILD=1 XM_GROUP_LANES=0
c4 mild_eight C4_MILD=1
