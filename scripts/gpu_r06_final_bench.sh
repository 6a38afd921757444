#!/bin/bash
# round 6, final build: the bench line again with the PMC summary of this build in profiles/r06 (roofline.traffic), plain and under rocprofv3 --kernel-trace --stats
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r06b; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
timeout 900 python3 bench.py --steps 20 > $O/bench_full.log 2>&1; tail -n 1 $O/bench_full.log > $O/bench_line.json
Q="--cpu-sample 0 --seed-probes 0 --wave-steps 0 --single-context-steps 0 --stream-batches 0 --end-to-end-reads 0"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py $Q --steps 20 > $O/stats.log 2>&1
cd $R
cut -c1-300 $O/bench_line.json; grep -o '"traffic": [0-9.e+]*' $O/bench_line.json | head -2
