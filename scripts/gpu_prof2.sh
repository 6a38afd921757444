#!/bin/bash
W=${W:-4}
for M in 1 2; do
make -B -C mapper_amd/csrc EXTRA="-DXM_WAVES_PER_SIMD=$W -DXM_PROFILE=$M" > /dev/null 2>&1
XM_SCRATCH_GIB=200 XM_LIGHT_WAVES=8 XM_FULL_WAVES=4 XM_FULL_LPW=${LPW:-32} timeout 300 python scripts/gpu_prof.py "mode$M/W$W/lpw${LPW:-32}" 1000000 2>&1 | grep -v Warn
done
