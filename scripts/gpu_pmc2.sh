#!/bin/bash
# HBM traffic of the align kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), L2 hit rate in a third
R=$GRAFT_REPO_ROOT
# the library is built before any profiler starts: nothing under rocprofv3 may spawn make/hipcc (mapper_amd/_capi.py lib() never builds)
make -j8 -C $R/mapper_amd/csrc > /dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmcF -o pmc --output-format csv -- python3 $R/scripts/gpu_prof.py pmcF 1000000 > $R/gpurun_out/pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmcW -o pmc --output-format csv -- python3 $R/scripts/gpu_prof.py pmcW 1000000 > $R/gpurun_out/pmcW.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmcH -o pmc --output-format csv -- python3 $R/scripts/gpu_prof.py pmcH 1000000 > $R/gpurun_out/pmcH.log 2>&1
grep -h "kernel ms" $R/gpurun_out/pmcF.log $R/gpurun_out/pmcW.log $R/gpurun_out/pmcH.log
