#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
timeout 600 python3 -m pytest tests/test_gpu_bound.py -x -q 2>&1 | tail -3
export XM_TRACE_PASSES=1 XM_PROF_GAPPED_ONLY=1 XM_LIB_PATH=$R/mapper_amd/_lib_variants/libxm_prof.so
run() { echo "== $1"; shift; env "$@" timeout 400 python3 scripts/gpu_c4_small.py 0.02 ${N:-40000} 0 2>&1 | grep -v "pair checks" | tail -4 | cut -c1-400; }
run "prof, filter on" A=1
