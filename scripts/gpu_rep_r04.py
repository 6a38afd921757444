"""Experiment (round 4): configs[1]'s reads on the repeat-rich reference (bench.py --config 1rep), one context, pass trace; optional env combos.
usage: gpu_rep_r04.py [nq] [combos]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
combos = [dict(kv.split("=") for kv in item.split("+") if kv) for item in (sys.argv[2] if len(sys.argv) > 2 else "XM_SCHED=0").split(",")]
ref = synth.repeat_rich_reference(5_000_000)
reads = synth.synthetic_single_end(ref, nq, read_len=150, seed=0x5EED0001)[0]
mc = np.ones(nq, np.int32); mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 150
ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
p = api.AlignmentParameters()
db = api.ReferenceDatabase([("rep", ref)], max_query_length=150)
db.upload_arrays(mc, mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq))
base = db.align_resident(p)
os.environ["XM_TRACE_PASSES"] = "1"
seen = set()
for c in combos:
    seen.update(c.keys())
for c in combos:
    for k in seen:
        os.environ.pop(k, None)
    for k, v in c.items():
        os.environ[k] = str(v)
    r = db.align_resident(p)
    same = np.array_equal(r.ints, base.ints) and np.array_equal(r.dbls.view(np.int64), base.dbls.view(np.int64))
    us = list(r.counters[12:16])
    print("%-50s light %.1f ms gapped+reruns %.1f ms launches %d reruns %d same=%s" % (" ".join("%s=%s" % kv for kv in c.items()), us[0] / 1e3, (us[1] + us[2] + us[3]) / 1e3, r.kernel_launches, r.counters[11], same), flush=True)
