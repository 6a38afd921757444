"""GPU experiment: phase timers (XM_PROFILE build) on reads that never leave the light pass (no indels, sub_rate given)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mapper_amd import api, synth
label = sys.argv[1]; nq = int(sys.argv[2]); sub = float(sys.argv[3]); indel = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=150)
reads, _, _ = synth.synthetic_single_end(ref, nq, sub_rate=sub, indel_prob=indel)
mc = np.ones(nq, np.int32); mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq) * 150; ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
codes = np.ascontiguousarray(reads.reshape(-1))
for rep in range(2):
    r = db.align_arrays(mc, mo, ml, codes, np.zeros(nq), np.ones(nq), api.AlignmentParameters())
names = ["TOTAL", "PYRAMID", "WALK", "HITS", "STRAIGHT", "ANALYZE", "PATH", "PATH_INIT", "BLOCK", "MATCHER_INDEX", "CONFIDENT", "OUTER", "PA_LOOK", "PA_LOAD", "PA_COMPUTE", "PA_PUT"]
print(label, "nq", nq, "sub", sub, "kernel ms %.2f" % r.kernel_ms, "us light/chain/search/inline", list(r.counters[12:16]), "probes/fetches/hits/cands", list(r.counters[1:5]), "PA calls/nodes", list(r.counters[5:7]), flush=True)
print(label, "Mticks", {n: round(x / 1e6, 1) for n, x in zip(names, r.prof)}, flush=True)
