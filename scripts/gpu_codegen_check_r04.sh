#!/bin/bash
# Round 4: the lane-per-read kernel's results against code generation (profiles/r04/NOTES.md 14).
#   build (here, no GPU):  scripts/gpu_codegen_check_r04.sh build   -> mapper_amd/_lib_variants/<name>/libxmapper_hip.so: xm_capi.hip compiled with the product's flags
#                          (as the Makefile has them: WWM registers allocated by the basic allocator) and with the default (greedy) allocator, each plain, with
#                          -ftrivial-auto-var-init=pattern and with -DXM_PROFILE=2; the other objects are the product's
#   run (on the GPU box):  gpurun -- 'bash scripts/gpu_codegen_check_r04.sh run'   -> per variant: the bench batch (1 M reads) three times and with 8 reads per wave;
#                          "OK" = 31 059 911 search nodes and 1 000 005 alignments, the oracle's counts
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
V=$R/mapper_amd/_lib_variants
if [ "${1:-build}" = build ]; then
  cd $R/mapper_amd/csrc
  BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-command-line-argument"
  one() { # name flags...
    n=$1; shift; mkdir -p $V/$n
    /opt/rocm/bin/hipcc $BASE -DXM_BUILD_STAMP='"variant"' "$@" -c -o $V/$n/xm_capi.o xm_capi.hip 2> $V/$n/err.txt &&
    /opt/rocm/bin/hipcc $BASE -shared -o $V/$n/libxmapper_hip.so $V/$n/xm_capi.o ../_lib/xm_index_device.o ../_lib/xm_wave_k*.o && rm -f $V/$n/xm_capi.o
  }
  one basic_plain -mllvm -wwm-regalloc=basic &
  one basic_pattern -mllvm -wwm-regalloc=basic -ftrivial-auto-var-init=pattern &
  one basic_profile -mllvm -wwm-regalloc=basic -DXM_PROFILE=2 &
  one greedy_plain &
  wait
  one greedy_pattern -ftrivial-auto-var-init=pattern &
  one greedy_profile -DXM_PROFILE=2 &
  wait
  ls -la $V/*/libxmapper_hip.so
else
  cd $R
  for d in $V/*/; do n=$(basename $d); echo "== $n"; XM_LIB_PATH=$d/libxmapper_hip.so timeout 300 python3 scripts/gpu_codegen_check_r04.py "X=0,X=1,X=2,XM_FULL_LPW=8" 2>&1 | grep "nodes"; done
fi
