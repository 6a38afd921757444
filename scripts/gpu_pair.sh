#!/bin/bash
# two lanes per read in the gapped pass (XM_PAIR_LANES) on/off on the bench workload (alternating), then the GPU parity suite
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3; do
  for h in 0 1; do XM_PAIR_LANES=$h timeout 300 python scripts/gpu_prof.py P$h-$i 1000000 2>&1 | grep -E "kernel ms" | cut -c1-160; done
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "Error|passed|failed|FAILED" | head -20
