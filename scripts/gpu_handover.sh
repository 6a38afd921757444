#!/bin/bash
# hand-over on/off on the bench workload (alternating), then the GPU parity suite
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3; do
  for h in 0 1; do XM_HANDOVER=$h XM_TRACE_PASSES=$([ $i = 1 ] && echo 1 || echo 0) timeout 300 python scripts/gpu_prof.py H$h-$i 1000000 2>&1 | grep -E "kernel ms|\[xm\] pass" | cut -c1-160; done
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "Error|passed|failed|FAILED" | head -20
