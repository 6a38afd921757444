"""GPU check: 1,000 bp reads (BASELINE.json configs[4] splits 10 kb reads into 1,000 bp queries), long-read-like error rates, against the oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as o
from helpers import se_batch
from mapper_amd import api, synth
ref = synth.synthetic_reference(2_000_000)
for L, n, sub, ind in ((1000, 4000, 0.01, 0.05), (1000, 4000, 0.05, 0.9), (400, 20000, 0.02, 0.3)):
    reads = synth.synthetic_single_end(ref, n, read_len=L, sub_rate=sub, indel_prob=ind)[0]
    b = se_batch(reads)
    db = api.ReferenceDatabase([("r", ref)], mode="mapper", max_query_length=L)
    t = time.time()
    r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
    wall = time.time() - t
    want = o.OracleReference([("r", ref)], mode="mapper").align(b, o.make_params(), threads=os.cpu_count())
    same = np.array_equal(want.ints, r.ints) and np.array_equal(want.dbls.view(np.int64), r.dbls.view(np.int64))
    aligned = int(sum(1 for q in range(n) if r.ints[r.int_off[q] + 1] > 0))
    print("len", L, "n", n, "sub", sub, "indel", ind, "kernel ms %.1f" % r.kernel_ms, "wall %.2f" % wall, "launches", r.kernel_launches, "reruns", r.counters[11], "aligned", aligned, "identical", same, flush=True)
    assert same
    db.close()
print("long ok")
