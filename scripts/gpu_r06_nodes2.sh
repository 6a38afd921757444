#!/bin/bash
# round 6 experiment: half the search-node capacity per long-read lane (variant build -DXM_LONG_CHAIN_NODES=2 in mapper_amd/_lib_variants/libxm_nodes2.so: more reads per wave fit
# the scratch) - configs[4] at reduced size, product against variant
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
O=$R/gpurun_out/nodes2; mkdir -p $O
export XM_TRACE_PASSES=1
c4() { tag=$1; shift; env "$@" timeout ${T:-150} python3 scripts/gpu_c4_small.py 0.02 ${N:-40000} 0 > $O/$tag.log 2>&1; echo "$tag rc=$?"; grep "pass 2\|pass 3\|pass 4\|step 0" $O/$tag.log | tail -4 | cut -c1-420; }
c4 product A=1
c4 nodes2 XM_LIB_PATH=$R/mapper_amd/_lib_variants/libxm_nodes2.so
