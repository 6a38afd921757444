#!/bin/bash
# experiment: the light pass compiled without the gapped chain (-DXM_LIGHT_ONLY) at 4, 5, 6 and 8 waves per SIMD; only the first pass of such
# a build means anything (the host loop stops after it)
R=$GRAFT_REPO_ROOT
cd $R
for W in 4 5 6 8; do
  make -B -C mapper_amd/csrc EXTRA="-DXM_WAVES_PER_SIMD=$W -DXM_LIGHT_ONLY=1" > /dev/null 2>&1
  for LW in $W $((2*W)); do
    XM_LIGHT_WAVES=$LW timeout 300 python scripts/gpu_prof.py L 1000000 2>&1 | grep "light-only" | tail -1 | sed "s/^/W=$W lightWaves=$LW /"
  done
done
make -B -C mapper_amd/csrc > /dev/null 2>&1
XM_TRACE_PASSES=1 timeout 300 python scripts/gpu_prof.py L 1000000 2>&1 | grep "pass 1" | tail -1 | sed "s/^/normal build /"
