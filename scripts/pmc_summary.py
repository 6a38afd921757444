#!/usr/bin/env python3
"""profiles/<round>/pmc_summary.json from the rocprofv3 --pmc passes of scripts/gpu_profile_round.sh.

    python scripts/pmc_summary.py gpurun_out/r01 > profiles/r01/pmc_summary.json

Every pass directory (pmcF = FETCH_SIZE, pmcW = WRITE_SIZE, pmcS = SQ_*) holds pmc_counter_collection.csv with one row per dispatch
and counter.  xm_align_kernel runs twice per step (light pass = the launch with the larger grid, gapped pass); counters are averaged
per launch over the launches of each pass.  HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) KiB x 1024 (narrow scattered accesses:
the gfx950 wide-load correction of the guide does not apply)."""
import csv
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
suffix = sys.argv[2] if len(sys.argv) > 2 else ""   # pass directories pmcF<suffix>, pmcW<suffix>, pmcS<suffix> (e.g. _config2: the passes over bench.py --config 2)
acc = defaultdict(lambda: defaultdict(list))   # pass -> counter -> values
grids = set()
rows = []
for d in ("pmcF", "pmcW", "pmcS"):
    path = os.path.join(root, d + suffix, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        if "xm_align_kernel" in r["Kernel_Name"]:
            rows.append(r)
            grids.add(int(r["Grid_Size"]))
light = max(grids)
for r in rows:
    name = "light_pass(grid %d)" % light if int(r["Grid_Size"]) == light else "gapped_pass(grid %s)" % r["Grid_Size"]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items()}
launches = {k: len(next(iter(cs.values()))) for k, cs in acc.items()}
hbm = {}
for k, cs in out.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        hbm[k.split("(")[0]] = (cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0
if len(hbm) == 2:
    hbm["mean_over_the_two_launches_of_a_step"] = sum(hbm.values()) / 2.0
out["_note"] = ("rocprofv3 --pmc passes over `python3 bench.py --cpu-sample 0 --seed-probes 0 --wave-steps 0 --single-context-steps 0` (the headline measurement alone; scripts/gpu_profile_round.sh, summarised by "
                "scripts/pmc_summary.py), mean per launch over %s launches of each pass; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them "
                "(narrow scattered accesses: no gfx950 wide-load correction applies); SQ_* in quad-cycles" % sorted(set(launches.values())))
out["hbm_bytes_per_launch"] = hbm
try:  # the build (and the number of contexts) the counters belong to: bench.py quotes roofline.traffic only for the same
    line = json.load(open(os.path.join(root, "bench_line.json" if not suffix else "bench%s.json" % suffix)))
    out["build"] = line["build"]
    out["contexts"] = line["contexts"]["per_gpu"]
except Exception:  # noqa: BLE001
    out["build"] = None
print(json.dumps(out, indent=1))
