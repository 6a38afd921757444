#!/usr/bin/env python3
"""profiles/<round>/pmc_summary.json from the rocprofv3 --pmc passes of scripts/gpu_profile_round.sh.

    python scripts/pmc_summary.py gpurun_out/r01 > profiles/r01/pmc_summary.json

Every pass directory (pmcF = FETCH_SIZE, pmcW = WRITE_SIZE, pmcS = SQ_*) holds pmc_counter_collection.csv with one row per dispatch
and counter.  xm_align_kernel runs twice per step (light pass, gapped pass: told apart by their order on the context's queue); counters are
averaged per launch over the launches of each pass.  HBM bytes per launch = (FETCH_SIZE + WRITE_SIZE) KiB x 1024 (narrow scattered accesses:
the gfx950 wide-load correction of the guide does not apply)."""
import csv
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
suffix = sys.argv[2] if len(sys.argv) > 2 else ""   # pass directories pmcF<suffix>, pmcW<suffix>, pmcS<suffix> (e.g. _config2: the passes over bench.py --config 2)
acc = defaultdict(lambda: defaultdict(list))   # pass -> counter -> values
# A step of this workload is two launches of xm_align_kernel on its context's queue: light pass, then gapped pass.  (Round 5: a launch is sized by the
# number of contexts that are aligning, so the grid no longer tells the passes apart - the first steps of a run see one and two contexts.)  The launches of
# a queue are taken in dispatch order, in twos; a pair whose second grid is larger than its first is not such a step and is left out, as are the first two
# steps of every queue (the sizes of a run that is starting).
for d in ("pmcF", "pmcW", "pmcS"):
    path = os.path.join(root, d + suffix, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    by_queue = defaultdict(dict)   # queue -> dispatch id -> {counter: value, "grid": n}
    for r in csv.DictReader(open(path)):
        if "xm_align_kernel" in r["Kernel_Name"]:
            e = by_queue[r.get("Queue_Id", "0")].setdefault(int(r["Dispatch_Id"]), {"grid": int(r["Grid_Size"])})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
    for q, ds in by_queue.items():
        order = sorted(ds)
        for k in range(4, len(order) - 1, 2):
            a, b = ds[order[k]], ds[order[k + 1]]
            if b["grid"] > a["grid"]:
                continue
            for name, e in (("light_pass", a), ("gapped_pass", b)):
                for c, v in e.items():
                    if c != "grid":
                        acc[name][c].append(v)
                acc[name]["grid_size"].append(e["grid"])
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items()}
launches = {k: len(next(iter(cs.values()))) for k, cs in acc.items()}
hbm = {}
for k, cs in out.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        hbm[k] = (cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0
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
