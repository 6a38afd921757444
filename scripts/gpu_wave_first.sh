#!/bin/bash
# first GPU run of the wave-per-read passes: parity + pass trace on the bench workload (single-end, paired), then the GPU test suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export XM_TRACE_PASSES=1
timeout 600 python3 scripts/gpu_wave.py se 1000000 30000 > gpurun_out/r02/wave_se.log 2>&1
tail -n 25 gpurun_out/r02/wave_se.log
timeout 600 python3 scripts/gpu_wave.py pe 300000 10000 > gpurun_out/r02/wave_pe.log 2>&1
tail -n 25 gpurun_out/r02/wave_pe.log
unset XM_TRACE_PASSES
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.log 2>&1
tail -n 15 gpurun_out/r02/pytest_gpu.log
