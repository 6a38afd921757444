"""GPU experiment: phase timers (XM_PROFILE build) on paired-end reads (config 3 shape)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import pe_batch
from mapper_amd import api, synth
label = sys.argv[1]; n = int(sys.argv[2])
ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=150)
m1, m2 = synth.synthetic_paired_end(ref, n)[:2]
b = pe_batch(m1, m2, 100.0, 50.0)
for rep in range(2):
    r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
names = ["TOTAL", "PYRAMID", "WALK", "HITS", "STRAIGHT", "ANALYZE", "PATH", "PATH_INIT", "BLOCK", "MATCHER_INDEX", "CONFIDENT", "OUTER", "PA_LOOK", "PA_LOAD", "PA_COMPUTE", "PA_PUT"]
print(label, "pairs", n, "kernel ms %.2f" % r.kernel_ms, "us light/chain/search/inline", list(r.counters[12:16]), "probes/fetches/hits/cands", list(r.counters[1:5]), "PA calls/nodes", list(r.counters[5:7]), "quick", r.counters[7], flush=True)
print(label, "Mticks", {k: round(x / 1e6, 1) for k, x in zip(names, r.prof)}, flush=True)
