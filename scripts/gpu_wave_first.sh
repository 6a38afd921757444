#!/bin/bash
# GPU run of the wave-per-read passes: parity + pass trace on the bench workload (single-end, paired), in-kernel phase timers, then the GPU test suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export XM_TRACE_PASSES=1
timeout 600 python3 scripts/gpu_wave.py se 1000000 30000 > gpurun_out/r02/wave_se.log 2>&1
grep -v "^\[xm\] pass\|rep 0\|rep 1" gpurun_out/r02/wave_se.log | tail -n 12
timeout 600 python3 scripts/gpu_wave.py pe 300000 10000 > gpurun_out/r02/wave_pe.log 2>&1
grep -v "rep 0\|rep 1" gpurun_out/r02/wave_pe.log | tail -n 14
unset XM_TRACE_PASSES
true
if [ "$1" != "notest" ]; then
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.log 2>&1
tail -n 5 gpurun_out/r02/pytest_gpu.log
fi
