#!/bin/bash
# The round's judged measurements (run on the GPU box): bench line, rocprofv3 kernel stats of the same command, HBM traffic PMC passes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01
mkdir -p $O
cd $R
timeout 900 python3 bench.py > $O/bench_full.log 2>&1
tail -n 1 $O/bench_full.log > $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py --cpu-sample 0 --seed-probes 0 > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmcF -o pmc --output-format csv -- python3 $R/bench.py --cpu-sample 0 --seed-probes 0 > $O/pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmcW -o pmc --output-format csv -- python3 $R/bench.py --cpu-sample 0 --seed-probes 0 > $O/pmcW.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmcS -o pmc --output-format csv -- python3 $R/bench.py --cpu-sample 0 --seed-probes 0 > $O/pmcS.log 2>&1
ls -R $O | head -40
cat $O/bench_line.json | cut -c1-600
