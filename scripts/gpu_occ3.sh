#!/bin/bash
# more waves per SIMD with a smaller LDS slot per wave (run on the GPU box)
run() { XM_FULL_WAVES=$1 XM_LIGHT_WAVES=$2 timeout 300 python scripts/gpu_prof.py "$3" 1000000 2>&1 | grep "kernel ms" | cut -c1-125; }
make -B -C mapper_amd/csrc EXTRA="-DXM_WAVES_PER_SIMD=4" > /dev/null 2>&1; run 4 8 "W4/slot9.9K"; run 4 8 "W4/slot9.9K"
for W in 4 5 6 8; do
  make -B -C mapper_amd/csrc EXTRA="-DXM_PAL_SMALL -DXM_WAVES_PER_SIMD=$W" > /dev/null 2>&1
  run $W $((W*2)) "W$W/slot5K"; run $W $((W*2)) "W$W/slot5K"
done
