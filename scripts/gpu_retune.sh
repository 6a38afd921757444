#!/bin/bash
# gapped-pass launch shape after the hand-over and the pair lanes: reads per wave and taper (scripts/README.md)
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for L in 32 28 24 20; do for T in 100 50 200; do
  XM_FULL_LPW=$L XM_TAPER_PCT=$T timeout 300 python scripts/gpu_prof.py "lpw$L/taper$T" 1000000 2>&1 | grep "kernel ms" | cut -c1-110
done; done; done
