#!/bin/bash
# round 6, final build: configs[4] at reduced size (400 k queries: five reads per wave, eight lanes per read) with the first queries' result streams compared with the oracle's
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
O=$R/gpurun_out/final; mkdir -p $O
export XM_TRACE_PASSES=1
c4() { tag=$1; shift; env "$@" timeout ${T:-420} python3 scripts/gpu_c4_small.py 0.02 ${N:-40000} 0 > $O/$tag.log 2>&1; echo "$tag rc=$?"; grep "pass 2\|step 0\|equal" $O/$tag.log | tail -3 | cut -c1-420; }
c4 c4_check_as_stated C4_CHECK=${1:-20000}
c4 c4_check_mild C4_MILD=1 C4_CHECK=${2:-8000}
