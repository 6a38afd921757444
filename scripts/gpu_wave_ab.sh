#!/bin/bash
# A/B of prebuilt library variants (mapper_amd/_lib/libxmapper_hip<suffix>.so) on the light tier of the wave passes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $O
export XM_WAVE_HEAVY=0
for v in "" _w2; do
  echo "== variant '$v'"
  XM_LIB_PATH=$GRAFT_REPO_ROOT/mapper_amd/_lib/libxmapper_hip$v.so python3 scripts/gpu_wave.py se 1000000 0 2>&1 | grep "rep" | sed 's/probes.*//'
done
echo "== profile variant"
XM_LIB_PATH=$GRAFT_REPO_ROOT/mapper_amd/_lib/libxmapper_hip_prof.so python3 scripts/gpu_wave_ticks.py se 1000000 2>&1 | head -3
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/pmcS2 -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gpu_wave.py se 1000000 0 > $O/pmcS2.log 2>&1)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES -d $O/pmcT2 -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gpu_wave.py se 1000000 0 > $O/pmcT2.log 2>&1)
python3 scripts/pmc_by_kernel.py $O/pmcS2 $O/pmcT2 | grep -v align_kernel
