"""Experiment (round 4): the gapped pass as the wave scheduler (XM_SCHED, xm_sched_kernel) against the lane-per-read gapped pass, one context, the 1 M-read
bench batch resident; the knobs are read per align call, so one process measures them all on the same box and clock state.
usage: gpu_sched_r04.py [config 1|2] [reps] [combos: comma-separated 'K=V+K=V' items, default a built-in list]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nq = int(os.environ.get("NQ", 1_000_000))
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
if cfg == "2":
    m1, m2 = synth.synthetic_paired_end(ref, nq, read_len=150, seed=0x5EED0002)[:2]
    L = 150
    codes = np.ascontiguousarray(np.concatenate([m1, m2], axis=1).reshape(-1))
    mc = np.full(nq, 2, np.int32)
    mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 2 * L; mo[1::2] = mo[0::2] + L
    ml = np.full(2 * nq, L, np.int32)
    arrays = (mc, mo, ml, codes, np.full(nq, 100.0), np.full(nq, 50.0))
else:
    reads = synth.synthetic_single_end(ref, nq, read_len=150, seed=0x5EED0001)[0]
    mc = np.ones(nq, np.int32); mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 150
    ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
    arrays = (mc, mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq))
p = api.AlignmentParameters()
db = api.ReferenceDatabase([("e", ref)], max_query_length=150)
db.upload_arrays(*arrays)
os.environ["XM_SCHED"] = "0"
base = db.align_resident(p)
if len(sys.argv) > 3:
    combos = [dict(kv.split("=") for kv in item.split("+") if kv) for item in sys.argv[3].split(",")]
else:
    combos = [{"XM_SCHED": 0}, {"XM_SCHED": 1}, {"XM_SCHED": 1, "XM_SCHED_LPW": 16}, {"XM_SCHED": 1, "XM_SCHED_LPW": 64}, {"XM_SCHED": 1, "XM_SCHED_LPW": 8},
              {"XM_SCHED": 1, "XM_SCHED_LPW": 64, "XM_FULL_WAVES": 2}, {"XM_SCHED": 1, "XM_SCHED_LPW": 32, "XM_FULL_WAVES": 8}, {"XM_SCHED": 0}]
seen = set()
for c in combos:
    seen.update(c.keys())
for c in combos:
    for k in seen:
        os.environ.pop(k, None)
    for k, v in c.items():
        os.environ[k] = str(v)
    best = None
    for _ in range(reps):
        r = db.align_resident(p)
        us = list(r.counters[12:16])
        if best is None or sum(us) < sum(best):
            best = us
    same = np.array_equal(r.ints, base.ints) and np.array_equal(r.dbls.view(np.int64), base.dbls.view(np.int64))
    cnt = np.array_equal(np.asarray(r.counters[:11]), np.asarray(base.counters[:11]))
    print("%-60s light %.1f ms  gapped+reruns %.1f ms  launches %d reruns %d same=%s counters=%s" % (
        " ".join("%s=%s" % kv for kv in c.items()) or "(defaults)", best[0] / 1e3, (best[1] + best[2] + best[3]) / 1e3, r.kernel_launches, r.counters[11], same, cnt), flush=True)
