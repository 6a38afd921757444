#!/bin/bash
# round 6, final build: differential fuzz of long-read batches on the GPU (thousands of reads per batch: several reads per wave, eight lanes per read) against the oracle with its observer on
R=$GRAFT_REPO_ROOT
cd $R
ulimit -c 0
O=$R/gpurun_out/final; mkdir -p $O
timeout ${2:-1200} python3 scripts/cpu_filter_fuzz.py ${1:-30} ${3:-606} gpu > $O/fuzz_filter_gpu.log 2>&1; echo "fuzz rc=$?"; tail -4 $O/fuzz_filter_gpu.log | cut -c1-400
