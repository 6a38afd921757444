#!/bin/bash
# in-kernel wave-time profile of the gapped pass on 1,000 bp reads
R=$GRAFT_REPO_ROOT
# the library is built before any profiler starts: nothing under rocprofv3 may spawn make/hipcc (mapper_amd/_capi.py lib() never builds)
make -j8 -C $R/mapper_amd/csrc > /dev/null || exit 1
cd $R
make -B -C mapper_amd/csrc EXTRA="-DXM_PROFILE=2" > /dev/null 2>&1
cat > /tmp/lp.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import numpy as np
from helpers import se_batch
from mapper_amd import api, synth
ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("r", ref)], mode="mapper", max_query_length=1000)
reads = synth.synthetic_single_end(ref, 100000, read_len=1000, sub_rate=0.01, indel_prob=0.05)[0]
b = se_batch(reads)
for rep in range(2):
    r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
names = ["TOTAL", "PYRAMID", "WALK", "HITS", "STRAIGHT", "ANALYZE", "PATH", "PATH_INIT", "BLOCK", "MATCHER_INDEX", "CONFIDENT", "OUTER", "PA_LOOK", "PA_LOAD", "PA_COMPUTE", "PA_PUT"]
print("kernel ms %.1f" % r.kernel_ms, "PA calls/nodes", list(r.counters[5:7]), "cands", r.counters[4])
print({k: round(x / 1e6, 1) for k, x in zip(names, r.prof)})
PY
XM_PROF_GAPPED_ONLY=1 timeout 600 python /tmp/lp.py 2>&1 | tail -2
