#!/bin/bash
# configs[3] / configs[4] on ONE GPU against the 3.1 Gb GRCh38-shaped reference at one GPU's share of the configs (6.25 M pairs; 625 000 reads of
# 10 kb = 6.25 M queries), and the milder long reads: bench lines for profiles/r05 (run on the GPU box).  Every line builds the reference (~1.5 min) and its index (~25 s) again.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05big}
mkdir -p $O
cd $R
timeout 1500 python3 bench.py --config 3shape --reads 6250000 --steps 4 --warmup 1 --seed-probes 16000000 --stream-batches 0 --single-context-steps 1 --cpu-sample 200000 2> $O/bench_config3_share.err | tail -n 1 > $O/bench_config3_share.json
timeout 1500 python3 bench.py --config 4mild --reads 200000 --steps 4 --warmup 1 --seed-probes 0 --stream-batches 0 --single-context-steps 1 2> $O/bench_config4mild.err | tail -n 1 > $O/bench_config4mild.json
timeout 2400 python3 bench.py --config 4 --reads 6250000 --contexts 1 --steps 1 --warmup 0 --seed-probes 0 --stream-batches 0 --single-context-steps 0 --cpu-sample 20000 2> $O/bench_config4_share.err | tail -n 1 > $O/bench_config4_share.json
for f in 3_share 4mild 4_share; do cut -c1-700 $O/bench_config$f.json; echo; tail -3 $O/bench_config$f.err; done
