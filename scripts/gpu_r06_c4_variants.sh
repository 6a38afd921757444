#!/bin/bash
# round 6: configs[4] at reduced size (400 k queries, 62 Mb reference) - what the size of a gapped-pass lane's temporaries does to the pass
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
export XM_TRACE_PASSES=1
run() { echo "== $1"; shift; env "$@" timeout 300 python3 scripts/gpu_c4_small.py 0.02 40000 1 2>&1 | grep "pass 2\|pass 3\|step 1" | tail -3 | cut -c1-330; }
run "default" A=1
run "chain nodes x2" XM_LIB_PATH=$R/mapper_amd/_lib_variants/libxm_nodes2.so
run "chain nodes x1" XM_LIB_PATH=$R/mapper_amd/_lib_variants/libxm_nodes1.so
run "default, 80 GiB of scratch" XM_SCRATCH_GIB=80
run "default, 16 waves per SIMD worth of lanes" XM_FULL_WAVES=16
