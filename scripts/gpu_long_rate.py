"""Throughput on 1,000 bp queries (BASELINE.json configs[4] splits 10 kb reads into 1,000 bp pieces): per-pass times at scale."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import se_batch
from mapper_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("r", ref)], mode="mapper", max_query_length=L)
for sub, ind in ((0.01, 0.05), (0.03, 0.5)):
    reads = synth.synthetic_single_end(ref, n, read_len=L, sub_rate=sub, indel_prob=ind)[0]
    b = se_batch(reads)
    for rep in range(2):
        r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
    aligned = int((r.ints[r.int_off[:-1] + 1] > 0).sum())
    print("len", L, "n", n, "sub", sub, "indel reads", ind, "kernel ms %.1f" % r.kernel_ms, "launches", r.kernel_launches, "reruns", r.counters[11], "aligned", aligned,
          "-> %.3f M reads/s, %.0f M bases/s" % (n / r.kernel_ms / 1e3, n * L / r.kernel_ms / 1e3), flush=True)
