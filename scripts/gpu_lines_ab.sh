#!/bin/bash
# Bucket lines (IndexView::lines32) against CSR probes: bench of the headline configuration, alternating runs (run on the GPU box)
R=$GRAFT_REPO_ROOT
cd $R
make -j8 -C $R/mapper_amd/csrc > /dev/null || exit 1
for rep in 1 2 3; do
  for L in 0 1; do
    XM_INDEX_LINES=$L python3 bench.py --cpu-sample 0 --wave-steps 0 --contexts 1 --steps 5 ${1:-} 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); sp=d.get('seed_probe') or {}
print('lines=$L', 'Mreads/s', d['value'], 'ms', d['ms_per_step'], d['roofline']['kernel_ms_by_pass'], 'probe G/s', round((sp.get('probes_per_s') or 0)/1e9,2), 'with positions ms', (sp.get('with_positions') or {}).get('kernel_ms'))"
  done
done
