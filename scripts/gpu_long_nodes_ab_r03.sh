#!/bin/bash
# round 3: search nodes of long-read chains (XM_LONG_CHAIN_NODES 4 = product, 1 = before) on configs[4] / 4mild against the 3.1 Gb reference, one context, passes traced
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for c in 4mild 4; do
  for lib in _lib _lib_m2; do
    XM_LIB_PATH=$GRAFT_REPO_ROOT/mapper_amd/$lib/libxmapper_hip.so XM_TRACE_PASSES=1 timeout 900 python3 scripts/gpu_big_one_r03.py $c 100000 2> gpurun_out/r03/nodes_${c}_$lib.err | tail -2
    grep "\[xm\] pass" gpurun_out/r03/nodes_${c}_$lib.err | tail -3
  done
done
