#!/usr/bin/env python3
"""HBM traffic of the align kernel's launches from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (scripts/gpu_profile_round_r05.sh), for workloads whose steps are
not "one light launch + one gapped launch of known grids" (pairs, long reads): launches are grouped by grid size.

    python scripts/pmc_by_grid.py gpurun_out/r04 _config2 > profiles/r04/pmc_config2.json
"""
import csv, json, os, sys
from collections import defaultdict
root, suffix = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for d in ("pmcF", "pmcW"):
    path = os.path.join(root, d + suffix, "pmc_counter_collection.csv")
    for r in csv.DictReader(open(path)):
        if "xm_align_kernel" in r["Kernel_Name"]:
            acc[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
groups = []
for g in sorted(acc, reverse=True):
    cs = acc[g]
    n = min(len(v) for v in cs.values())
    f = sum(cs.get("FETCH_SIZE", [0])) / max(1, len(cs.get("FETCH_SIZE", [0])))
    w = sum(cs.get("WRITE_SIZE", [0])) / max(1, len(cs.get("WRITE_SIZE", [0])))
    groups.append({"grid_threads": g, "launches_profiled": n, "fetch_bytes_per_launch": f * 1024.0, "write_bytes_per_launch": w * 1024.0, "hbm_bytes_per_launch": (f + w) * 1024.0})
out = {"kernel": "xm_align_kernel", "launches_by_grid": groups,
       "_note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over `python3 bench.py --config %s --steps 6` without the side measurements (scripts/gpu_profile_round_r05.sh); "
                "KiB as rocprofv3 reports them x 1024 (narrow scattered accesses: no gfx950 wide-load correction applies); a launch is sized by the contexts that exist on the GPU and by the scratch it gets, so one pass can appear under more than one grid" % suffix.replace("_config", "")}
try:
    line = json.load(open(os.path.join(root, "bench%s.json" % suffix)))
    out["build"] = line["build"]
    out["reads_per_step"] = line["config"]["reads_per_gpu"]
except Exception:  # noqa: BLE001
    pass
print(json.dumps(out, indent=1))
