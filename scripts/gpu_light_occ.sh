#!/bin/bash
# wave-per-read light tier: waves per SIMD (register budget) against spills - builds in mapper_amd/_lib, _lib_m3, _lib_m2 (WV_SE_MINWAVES)
cd $GRAFT_REPO_ROOT
export XM_WAVE=1 XM_WAVE_TIERS=1 XM_TRACE_PASSES=1
for L in _lib _lib_m3 _lib_m2; do
  echo "== $L"
  XM_LIB_PATH=$GRAFT_REPO_ROOT/mapper_amd/$L/libxmapper_hip.so python3 scripts/gpu_wave.py se 1000000 0 2>&1 | grep "wave tier 0" | tail -2
done
