#!/bin/bash
# pass structures side by side on the bench workload: lane-per-read only, wave light tier + lane-per-read, all wave tiers
cd $GRAFT_REPO_ROOT
W=${1:-se}; N=${2:-1000000}
for mode in "XM_WAVE=0" "XM_WAVE_TIERS=1" "XM_WAVE_TIERS=3"; do
  echo "== $mode"
  env $mode python3 scripts/gpu_wave.py $W $N 0 2>&1 | grep "rep 1\|rep 2" | sed 's/probes.*//'
done
