#!/bin/bash
# round 6, final build: the filter's GPU tests with the lane-form knobs, then configs[4]'s reads with milder errors (2 % substitutions, 0.2 % indel events: reads that
# align) at one GPU's share against the 3.1 Gb reference - profiles/r06/bench_config4mild_share.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06big
mkdir -p $O
cd $R
ulimit -c 0
timeout 900 python3 -m pytest tests/test_gpu_bound.py -x -q > $O/pytest_bound.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_bound.log | cut -c1-300
timeout 2400 python3 bench.py --config 4mild --reads ${1:-6250000} --contexts 1 --steps 1 --warmup 0 --seed-probes 0 --stream-batches 0 --single-context-steps 0 --cpu-sample 20000 2> $O/bench_config4mild_share.err | tail -n 1 > $O/bench_config4mild_share.json
cut -c1-700 $O/bench_config4mild_share.json; echo; tail -3 $O/bench_config4mild_share.err
