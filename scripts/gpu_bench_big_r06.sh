#!/bin/bash
# round 6: configs[4] on ONE GPU against the 3.1 Gb GRCh38-shaped reference at one GPU's share (625 000 reads of 10 kb = 6.25 M queries), one context, with the rejection
# filter in front of PathAligner (default) - profiles/r06/bench_config4_share.json; "off" as first argument: XM_BOUND_FILTER=0 for comparison.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06big
mkdir -p $O
cd $R
ulimit -c 0
TAG=${1:-on}
if [ "$TAG" = "off" ]; then export XM_BOUND_FILTER=0; fi
[ "$3" = "3rep" ] || timeout 2400 python3 bench.py --config 4 --reads ${2:-6250000} --contexts 1 --steps 1 --warmup 0 --seed-probes 0 --stream-batches 0 --single-context-steps 0 --cpu-sample 20000 2> $O/bench_config4_share_$TAG.err | tail -n 1 > $O/bench_config4_share_$TAG.json
[ "$3" = "3rep" ] || { cut -c1-900 $O/bench_config4_share_$TAG.json; echo; tail -3 $O/bench_config4_share_$TAG.err; }
# configs[3]'s pairs at one GPU's share against the GRCh38 shape with the repeat structure of a genome (third argument "3rep")
if [ "$3" = "3rep" ]; then
timeout 2400 python3 bench.py --config 3rep --reads ${4:-6250000} --steps 4 --warmup 1 --seed-probes 0 --stream-batches 0 --single-context-steps 1 --cpu-sample 200000 2> $O/bench_config3rep.err | tail -n 1 > $O/bench_config3rep.json
cut -c1-1500 $O/bench_config3rep.json; echo; tail -3 $O/bench_config3rep.err
fi
