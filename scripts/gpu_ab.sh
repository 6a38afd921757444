#!/bin/bash
# A/B on one GPU box, alternating runs: A = the library built from the sources in .ab_head/ (prepared with scripts/ab_prepare.sh from a git
# revision), B = the library built from the working tree
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ab; mkdir -p /tmp/ab/A/mapper_amd /tmp/ab/B/mapper_amd
cp -r $R/.ab_head/csrc /tmp/ab/A/mapper_amd/csrc; cp -r $R/.ab_head/include /tmp/ab/A/include
cp -r $R/mapper_amd/csrc /tmp/ab/B/mapper_amd/csrc; cp -r $R/include /tmp/ab/B/include
for v in A B; do (cd /tmp/ab/$v/mapper_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $AB_EXTRA -shared -o /tmp/ab/lib$v.so xm_capi.hip xm_index_device.hip 2>&1 | grep -E "error" ); done
ls -la /tmp/ab/*.so
N=${AB_REPS:-4}
for i in $(seq 1 $N); do for v in A B; do XM_LIB_PATH=/tmp/ab/lib$v.so timeout 300 python $R/scripts/gpu_prof.py $v$i 1000000 2>&1 | grep "kernel ms" | cut -c1-105; done; done
