"""Experiment (round 3): one context, the 1 M-read bench batch resident, the launch-shape knobs of the gapped pass in combination
(every knob is read per align call, so one process measures all of them on the same box and clock state).
usage: gpu_sweep_r03.py [config 1|2] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nq = 1_000_000
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
base = db.align_resident(p)
KNOBS = ["XM_HEAVY_HINT", "XM_FULL_SYNC", "XM_FULL_LPW", "XM_FULL_WAVES", "XM_PAIR_LANES", "XM_TAPER_PCT", "XM_LIGHT_WAVES", "XM_LIGHT_LPW", "XM_GAPPED_TMP_PCT", "XM_SCRATCH_GIB"]
combos = [
    {},
    {"XM_HEAVY_HINT": 64},
    {"XM_FULL_SYNC": 1},
    {"XM_FULL_SYNC": 1, "XM_FULL_LPW": 16},
    {"XM_FULL_SYNC": 1, "XM_FULL_LPW": 8},
    {"XM_FULL_SYNC": 1, "XM_HEAVY_HINT": 64},
    {"XM_FULL_SYNC": 1, "XM_HEAVY_HINT": 64, "XM_FULL_LPW": 16},
    {"XM_FULL_SYNC": 1, "XM_HEAVY_HINT": 64, "XM_FULL_LPW": 8},
    {"XM_FULL_SYNC": 1, "XM_HEAVY_HINT": 64, "XM_FULL_LPW": 4},
    {"XM_FULL_LPW": 16},
    {"XM_FULL_LPW": 8},
    {"XM_FULL_LPW": 64, "XM_PAIR_LANES": 0},
    {"XM_PAIR_LANES": 0},
    {"XM_SCRATCH_GIB": 64},
    {"XM_SCRATCH_GIB": 64, "XM_GAPPED_TMP_PCT": 50},
    {},
]
for c in combos:
    for k in KNOBS:
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
    print("%-70s light %.1f ms  gapped+reruns %.1f ms  (chain %.1f search %.1f)  launches %d reruns %d same=%s" % (
        str(c), best[0] / 1e3, best[3] / 1e3, best[1] / 1e3, best[2] / 1e3, r.kernel_launches, r.counters[11], same), flush=True)
