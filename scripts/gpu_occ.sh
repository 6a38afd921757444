#!/bin/bash
# occupancy / scratch-budget experiment (run on the GPU box)
for W in 1 2 3 4; do
  make -B -C mapper_amd/csrc EXTRA="-DXM_WAVES_PER_SIMD=$W" > /dev/null 2>&1
  for cfg in "48 2 1" "200 2 1" "200 $((W>2?W:2)) $W" "200 $((2*W)) $W"; do
    set -- $cfg
    XM_SCRATCH_GIB=$1 XM_LIGHT_WAVES=$2 XM_FULL_WAVES=$3 timeout 300 python scripts/gpu_prof.py "W$W/gib$1/light$2/full$3" 2>&1 | grep "kernel ms"
  done
done
