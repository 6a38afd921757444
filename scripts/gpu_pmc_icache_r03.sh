#!/bin/bash
# round 3: instruction-cache counters of the bench's kernels (separate --pmc passes; nothing else traced)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03/icache
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--cpu-sample 0 --seed-probes 0 --wave-steps 0 --single-context-steps 0 --stream-batches 0 --contexts 1 --steps 3"
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $O/a -o pmc --output-format csv -- python3 $R/bench.py $Q > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $O/b -o pmc --output-format csv -- python3 $R/bench.py $Q > $O/b.log 2>&1
timeout 300 rocprofv3 --pmc SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_REQ SQC_DCACHE_MISSES -d $O/c -o pmc --output-format csv -- python3 $R/bench.py $Q > $O/c.log 2>&1
cd $R
python3 - <<'PY'
import csv, os, collections
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03", "icache")
for d in "abc":
    p = os.path.join(root, d, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        print(d, "missing"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        if "xm_align_kernel" in r["Kernel_Name"]:
            acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g, cs in sorted(acc.items()):
        print(d, "grid", g, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(cs.items())})
PY
