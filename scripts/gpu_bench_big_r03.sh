#!/bin/bash
# configs[3] / configs[4] on ONE GPU against the 3.1 Gb GRCh38-shaped reference: bench lines for profiles/r03 (run on the GPU box).
# Every line builds the reference (~1.5 min) and its index (~20 s) again: three lines, ~15 minutes.
# (the long-read lines with the default two contexts as well: a batch of them ends with a pass over the few reads that outgrew the gapped pass's scratch, one read
# per wave, as long as its slowest read - 0.7 s and 3.2 s per 100 k queries - which the other context's batch fills)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03}
mkdir -p $O
cd $R
make -j8 -C $R/mapper_amd/csrc > /dev/null || exit 1
timeout 1500 python3 bench.py --config 3shape --steps 4 --warmup 1 --seed-probes 4000000 --stream-batches 2 --single-context-steps 2 2> $O/bench_config3shape.err | tail -n 1 > $O/bench_config3shape.json
timeout 1500 python3 bench.py --config 4mild --reads 200000 --steps 4 --warmup 1 --seed-probes 0 --stream-batches 0 --single-context-steps 1 2> $O/bench_config4mild.err | tail -n 1 > $O/bench_config4mild.json
timeout 1500 python3 bench.py --config 4 --reads 200000 --steps 4 --warmup 1 --seed-probes 0 --stream-batches 0 --single-context-steps 1 --cpu-sample 20000 2> $O/bench_config4.err | tail -n 1 > $O/bench_config4.json
for f in 3shape 4mild 4; do cut -c1-900 $O/bench_config$f.json; echo; tail -3 $O/bench_config$f.err; done
