#!/bin/bash
# occupancy / lanes-per-wave experiment on the 1M-read workload (run on the GPU box); prints pass times
run() { XM_SCRATCH_GIB=$1 XM_LIGHT_WAVES=$2 XM_FULL_WAVES=$3 XM_FULL_LPW=$4 timeout 300 python scripts/gpu_prof.py "W$W/gib$1/light$2/full$3/lpw$4" 1000000 2>&1 | grep "kernel ms"; }
for W in 1 4; do
  make -B -C mapper_amd/csrc EXTRA="-DXM_WAVES_PER_SIMD=$W" > /dev/null 2>&1
  if [ $W = 1 ]; then
    run 48 2 1 64; run 200 2 1 64; run 200 2 2 32; run 200 2 4 16
  else
    run 200 4 4 64; run 200 8 4 64; run 200 8 4 32; run 200 8 4 16; run 200 8 8 16; run 200 8 8 8
  fi
done
