#!/bin/bash
# round 6: look-ahead lookups of the hash-block analysis across the eight lanes of a read - the filter's GPU tests (chains of long reads against the oracle), then configs[4] at reduced size
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
O=$R/gpurun_out/ahead; mkdir -p $O
export XM_TRACE_PASSES=1
timeout 900 python3 -m pytest tests/test_gpu_bound.py tests/test_gpu_big_reference.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-400
c4() { tag=$1; shift; env "$@" timeout ${T:-150} python3 scripts/gpu_c4_small.py 0.02 ${N:-40000} 0 > $O/$tag.log 2>&1; echo "$tag rc=$?"; grep "pass 2\|pass 3\|step 0\|phase\|equal" $O/$tag.log | tail -4 | cut -c1-900; }
c4 eight A=1
c4 eight_mild C4_MILD=1
