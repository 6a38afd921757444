#!/bin/bash
# The round's judged measurements (run on the GPU box): bench line, rocprofv3 kernel stats of the same command, HBM traffic PMC passes.
# usage: scripts/gpu_profile_round.sh [rNN]
R=$GRAFT_REPO_ROOT
RN=${1:-r03}
O=$R/gpurun_out/$RN
mkdir -p $O
cd $R
# the library is built before any profiler starts: nothing under rocprofv3 may spawn make/hipcc (mapper_amd/_capi.py lib() never builds)
make -j8 -C $R/mapper_amd/csrc > /dev/null || exit 1
timeout 900 python3 bench.py > $O/bench_full.log 2>&1
tail -n 1 $O/bench_full.log > $O/bench_line.json
timeout 600 python3 bench.py --config 2 --seed-probes 0 --steps 9 --stream-batches 10 2> $O/bench_config2.err | tail -n 1 > $O/bench_config2.json
timeout 600 python3 bench.py --config 4shape --seed-probes 0 --steps 6 2> $O/bench_config4shape.err | tail -n 1 > $O/bench_config4shape.json
cd /tmp && export TMPDIR=/tmp
Q="--cpu-sample 0 --seed-probes 0 --wave-steps 0 --single-context-steps 0 --stream-batches 0"
# (the profiled command is the headline measurement alone: the default contexts and steps, without the extra measurements of the bench line)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py $Q > $O/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmcF -o pmc --output-format csv -- python3 $R/bench.py $Q > $O/pmcF.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmcW -o pmc --output-format csv -- python3 $R/bench.py $Q > $O/pmcW.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmcS -o pmc --output-format csv -- python3 $R/bench.py $Q > $O/pmcS.log 2>&1
cd $R
python3 scripts/pmc_summary.py $O > $O/pmc_summary.json
ls -R $O | head -40
cut -c1-1500 $O/bench_line.json
