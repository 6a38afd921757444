#!/bin/bash
# round 6: eight lanes per read - the filter's GPU tests, configs[4] at reduced size, then the phase timers of the gapped pass (profile build)
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
O=$R/gpurun_out/group; mkdir -p $O
export XM_TRACE_PASSES=1
timeout 900 python3 -m pytest tests/test_gpu_bound.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-400
c4() { tag=$1; shift; env "$@" timeout ${T:-150} python3 scripts/gpu_c4_small.py 0.02 ${N:-40000} 0 > $O/$tag.log 2>&1; echo "$tag rc=$?"; grep "pass 2\|pass 3\|step 0\|phase\|equal" $O/$tag.log | tail -4 | cut -c1-900; }
c4 eight A=1
c4 eight_mild C4_MILD=1
c4 eight_prof XM_PROF_GAPPED_ONLY=1 XM_LIB_PATH=$R/mapper_amd/_lib_variants/libxm_prof.so
