#!/bin/bash
# HBM traffic and SQ counters of the wave-per-read form (XM_WAVE=1) on the headline workload, per kernel (run on the GPU box)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r02}/wave
mkdir -p $O
# the library is built before any profiler starts: nothing under rocprofv3 may spawn make/hipcc (mapper_amd/_capi.py lib() never builds)
make -j8 -C $R/mapper_amd/csrc > /dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
export XM_WAVE=1
Q="--cpu-sample 0 --seed-probes 0 --wave-steps 0 --contexts 1 --steps 3"
rocprofv3 --pmc FETCH_SIZE -d $O/pmcF -o pmc --output-format csv -- python3 $R/bench.py $Q > $O/pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmcW -o pmc --output-format csv -- python3 $R/bench.py $Q > $O/pmcW.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmcS -o pmc --output-format csv -- python3 $R/bench.py $Q > $O/pmcS.log 2>&1
cd $R
python3 scripts/pmc_by_kernel.py $O/pmcF $O/pmcW $O/pmcS | tee $O/pmc_by_kernel.txt
