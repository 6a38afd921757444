#!/bin/bash
# round 6: the piece-level rejection (the filter's bound over a piece's whole inner chain) - parity, then configs[4] at reduced size as stated and mild
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
timeout 900 python3 -m pytest tests/test_gpu_bound.py -x -q 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "grch38_regime" 2>&1 | tail -3
export XM_TRACE_PASSES=1
timeout 300 python3 scripts/gpu_c4_small.py 0.02 40000 1 2>&1 | grep "pass 2\|pass 3\|step 1" | tail -2 | cut -c1-400
C4_MILD=1 timeout 300 python3 scripts/gpu_c4_small.py 0.02 20000 1 2>&1 | grep "pass 2\|pass 3\|step 1" | tail -2 | cut -c1-400
