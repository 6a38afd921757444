#!/bin/bash
# round 6: table-free matchers (window bases in LDS, sections scanned) - parity, then configs[4] at reduced size as stated and mild, and 1 kb queries on 5 Mb
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
timeout 900 python3 -m pytest tests/test_gpu_bound.py -x -q 2>&1 | tail -2
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "long or grch38_regime or many_read_lengths" 2>&1 | tail -2
export XM_TRACE_PASSES=1
timeout 300 python3 scripts/gpu_c4_small.py 0.02 40000 1 2>&1 | grep "pass 2\|pass 3\|step 1" | tail -2 | cut -c1-330
C4_MILD=1 timeout 300 python3 scripts/gpu_c4_small.py 0.02 20000 1 2>&1 | grep "pass 2\|pass 3\|step 1" | tail -3 | cut -c1-330
