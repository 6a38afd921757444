#!/bin/bash
# The round's judged measurements (run on the GPU box): bench lines, rocprofv3 kernel stats of the headline command, HBM traffic and SQ PMC passes of the
# headline and of --config 2 / 4shape / 1rep, the phase timers of the gapped pass (a library built with -DXM_PROFILE=2 must be in mapper_amd/_lib_prof).
# usage: scripts/gpu_profile_round_r06.sh [rNN] [what: all | bench | prof | pmc | pmc2]
R=$GRAFT_REPO_ROOT
RN=${1:-r06}
WHAT=${2:-all}
O=$R/gpurun_out/$RN
mkdir -p $O
cd $R
Q="--cpu-sample 0 --seed-probes 0 --wave-steps 0 --single-context-steps 0 --stream-batches 0 --end-to-end-reads 0"
# (the HIP runtime reads GPU_MAX_HW_QUEUES when it starts - under rocprofv3 that is before Python runs: exported here for every command)
export GPU_MAX_HW_QUEUES=8
if [ $WHAT = all ] || [ $WHAT = bench ]; then
timeout 900 python3 bench.py --steps 20 > $O/bench_full.log 2>&1
tail -n 1 $O/bench_full.log > $O/bench_line.json
timeout 600 python3 bench.py --config 2 --seed-probes 0 --steps 8 --stream-batches 10 2> $O/bench_config2.err | tail -n 1 > $O/bench_config2.json
timeout 600 python3 bench.py --config 4shape --seed-probes 0 --steps 6 2> $O/bench_config4shape.err | tail -n 1 > $O/bench_config4shape.json
timeout 900 python3 bench.py --config 1rep --seed-probes 0 --steps 12 --wave-steps 0 2> $O/bench_config1rep.err | tail -n 1 > $O/bench_config1rep.json
fi
if [ $WHAT = prof ]; then
export XM_LIB_PATH=$R/mapper_amd/_lib_prof/libxmapper_hip.so
timeout 600 python3 scripts/gpu_phase_prof.py 1 1000000 gapped > $O/phase_gapped.json 2> $O/phase_gapped.err
timeout 600 python3 scripts/gpu_phase_prof.py 1 1000000 all > $O/phase_all.json 2> $O/phase_all.err
timeout 600 python3 scripts/gpu_phase_prof.py rep 1000000 gapped > $O/phase_rep_gapped.json 2> $O/phase_rep_gapped.err
unset XM_LIB_PATH
fi
if [ $WHAT = all ] || [ $WHAT = pmc ]; then
cd /tmp && export TMPDIR=/tmp
# (the profiled command is the headline measurement alone: the default contexts and steps, without the extra measurements of the bench line)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py $Q --steps 20 > $O/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmcF -o pmc --output-format csv -- python3 $R/bench.py $Q --steps 20 > $O/pmcF.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmcW -o pmc --output-format csv -- python3 $R/bench.py $Q --steps 20 > $O/pmcW.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmcS -o pmc --output-format csv -- python3 $R/bench.py $Q --steps 20 > $O/pmcS.log 2>&1
cd $R
python3 scripts/pmc_summary.py $O > $O/pmc_summary.json
fi
if [ $WHAT = pmc2 ]; then
cd /tmp && export TMPDIR=/tmp
for c in 2 4shape 1rep; do
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d $O/pmcF_config$c -o pmc --output-format csv -- python3 $R/bench.py $Q --config $c --steps 6 > $O/pmcF_config$c.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE -d $O/pmcW_config$c -o pmc --output-format csv -- python3 $R/bench.py $Q --config $c --steps 6 > $O/pmcW_config$c.log 2>&1
done
cd $R
for c in 2 4shape 1rep; do python3 scripts/pmc_by_grid.py $O _config$c > $O/pmc_config$c.json; done
fi
ls $O | head -80
cut -c1-1200 $O/bench_line.json
