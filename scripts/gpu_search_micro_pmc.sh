#!/bin/bash
# Executed instructions per category of ONE PathAligner search (scripts/gpu_search_micro.py, the test entry: one lane of one wave), per library variant.
# usage (on the GPU box): scripts/gpu_search_micro_pmc.sh outdir libdir [libdir ...]
R=$GRAFT_REPO_ROOT
O=$R/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export XM_LIB_PATH=$R/mapper_amd/$v/libxmapper_hip.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d $O/pmcA$v -o p --output-format csv -- python3 $R/scripts/gpu_search_micro.py 0 6 > $O/pmcA$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAVE_CYCLES -d $O/pmcB$v -o p --output-format csv -- python3 $R/scripts/gpu_search_micro.py 0 6 > $O/pmcB$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmcC$v -o p --output-format csv -- python3 $R/scripts/gpu_search_micro.py 0 6 > $O/pmcC$v.log 2>&1
done
cd $R
python3 - $O "$@" <<'PY'
import csv, sys, collections, json, os
O = sys.argv[1]
import re
nodes = None
out = {}
for v in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for p in "ABC":
        f = os.path.join(O, "pmc%s%s" % (p, v), "p_counter_collection.csv")
        if not os.path.exists(f): continue
        for r in csv.DictReader(open(f)):
            if "xm_test_local_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for c, vals in acc.items():
        n = len(vals) // 4
        per = [sorted(vals[i * n:(i + 1) * n])[n // 2] for i in range(4)]
        if nodes is None:
            nodes = [int(x) for x in re.findall(r"nodes put (\d+)", open(os.path.join(O, "pmcA%s.log" % v)).read())]
        res[c] = {"per_search": per, "per_node_put": round((per[2] - per[0]) / (nodes[2] - nodes[0]), 1)}
    res["nodes_put_per_search"] = nodes
    out[v] = res
out["_note"] = "scripts/gpu_search_micro.py: four searches through the test entry (one lane of one wave); per_node_put = (counter of search 3 - counter of search 1) / (nodes 3 - nodes 1): the slot search's executed wave instructions per node put; search 4 starts in HBM mode"
print(json.dumps(out, indent=1))
PY
