#!/bin/bash
# compiler-flag experiment (run on the GPU box): EXTRA flags appended after -O3
for F in "" "-O2" "-Os" "-fno-unroll-loops" "-mllvm -amdgpu-function-calls=false" "-mllvm -inline-threshold=100"; do
  make -B -C mapper_amd/csrc EXTRA="$F" > /dev/null 2>&1 || { echo "build failed: $F"; continue; }
  for i in 1 2; do timeout 200 python scripts/gpu_prof.py "flags[$F]" 1000000 2>&1 | grep -E "kernel ms" | cut -c1-130; done
done
