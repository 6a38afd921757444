#!/bin/bash
# PMC passes over the inline-search configuration (run on the GPU box)
R=$GRAFT_REPO_ROOT
# the library is built before any profiler starts: nothing under rocprofv3 may spawn make/hipcc (mapper_amd/_capi.py lib() never builds)
make -j8 -C $R/mapper_amd/csrc > /dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/pmc1 -o pmc --output-format csv -- python3 $R/scripts/gpu_prof.py pmc1 1000000 > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_IFETCH SQ_WAVES SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM -d $R/gpurun_out/pmc2 -o pmc --output-format csv -- python3 $R/scripts/gpu_prof.py pmc2 1000000 > $R/gpurun_out/pmc2.log 2>&1
tail -3 $R/gpurun_out/pmc1.log $R/gpurun_out/pmc2.log
ls -R $R/gpurun_out/pmc1 | head
