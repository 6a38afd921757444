#!/bin/bash
# round 6: matcher tables indexed in two sweeps (no read-modify-write chain) - parity, then configs[4] at reduced size and the headline's passes
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
timeout 900 python3 -m pytest tests/test_gpu_bound.py tests/test_gpu_kat.py -x -q 2>&1 | tail -2
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "full_size_golden or long or ambiguous_reference or repeat_rich or pass_shapes" 2>&1 | tail -2
XM_TRACE_PASSES=1 timeout 300 python3 scripts/gpu_c4_small.py 0.02 40000 1 2>&1 | grep "pass 2\|step 1" | tail -2 | cut -c1-330
timeout 600 python3 bench.py --steps 12 --cpu-sample 0 --seed-probes 0 --wave-steps 0 --stream-batches 0 --end-to-end-reads 0 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], 'single', d['single_context']['value'], d['single_context']['kernel_ms_by_pass'], 'golden', d['golden'])"
