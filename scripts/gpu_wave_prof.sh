#!/bin/bash
# wave-per-read passes: SQ instruction counters of the normal build, then in-kernel phase timers of an XM_PROFILE build (single-end bench workload)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $O
W=${1:-se}
N=${2:-1000000}
export XM_WAVE_HEAVY=${XM_WAVE_HEAVY:-1}
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmcS_$W -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gpu_wave.py $W $N 0 > $O/pmcS_$W.log 2>&1)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES -d $O/pmcT_$W -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gpu_wave.py $W $N 0 > $O/pmcT_$W.log 2>&1)
python3 scripts/pmc_by_kernel.py $O/pmcS_$W $O/pmcT_$W | tee $O/pmc_wave_$W.txt
make -C mapper_amd/csrc clean > /dev/null
make -C mapper_amd/csrc EXTRA=-DXM_WAVE_PROFILE=1 > $O/build_prof.log 2>&1 || tail -5 $O/build_prof.log
python3 scripts/gpu_wave_ticks.py $W $N | tee $O/ticks_wave_$W.txt
