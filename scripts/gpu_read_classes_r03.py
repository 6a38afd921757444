"""GPU experiment (round 3): which reads cost the gapped pass its time - 1 M reads of 150 bp with substitutions only, with indels only, with both (the bench workload)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import se_batch
from mapper_amd import api, synth
n = 1_000_000
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
db = api.ReferenceDatabase([("e", ref)], max_query_length=150)
for sub, ind in ((0.01, 0.05), (0.01, 0.0), (0.0, 0.05), (0.0, 0.5), (0.02, 0.0), (0.005, 0.0)):
    reads = synth.synthetic_single_end(ref, n, read_len=150, seed=0x5EED0001, sub_rate=sub, indel_prob=ind)[0]
    b = se_batch(reads)
    for rep in range(2):
        r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
    c = r.counters
    print("sub", sub, "indel reads", ind, "kernel ms %.1f" % r.kernel_ms, "launches", r.kernel_launches, "by pass (ms) light %.1f gapped+reruns %.1f" % (c[12] / 1e3, c[15] / 1e3),
          "PA calls %d nodes %d cands %d" % (c[5], c[6], c[4]), flush=True)
