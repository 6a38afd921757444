"""GPU experiment: phase timers (XM_PROFILE build) for single reads = single lanes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mapper_amd import api, synth
ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=150)
reads, _, _ = synth.synthetic_single_end(ref, 1024)
names = ["TOTAL", "PYRAMID", "WALK", "HITS", "STRAIGHT", "ANALYZE", "PATH", "PATH_INIT", "BLOCK", "MATCHER_INDEX", "CONFIDENT", "OUTER"]
for idx in [436, 19, 770, 0, 1, 2, 3]:
    rd = reads[idx:idx + 1]
    nq = 1
    mc = np.ones(nq, np.int32); mo = np.zeros(2, np.int64); ml = np.array([150, 0], np.int32)
    for rep in range(2):
        r = db.align_arrays(mc, mo, ml, np.ascontiguousarray(rd.reshape(-1)), np.zeros(1), np.ones(1), api.AlignmentParameters())
    print("read", idx, "kernel ms %.3f" % r.kernel_ms, "passes us", r.counters[12:16], "ctr", r.counters[:9])
    print("   ", {n: round(t / 1e3, 1) for n, t in zip(names, r.prof[:12]) if t}, "(kticks)", flush=True)
