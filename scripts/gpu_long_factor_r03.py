"""GPU experiment (round 3): chain scale of the gapped pass for long reads (XM_GAPPED_FACTOR: 4 = scale 16, reads that outgrow it rerun at 64; 16 = scale 64 from the start).  argv: n sub indel"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import se_batch
from mapper_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sub = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
ind = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("r", ref)], mode="mapper", max_query_length=1000)
reads = synth.synthetic_single_end(ref, n, read_len=1000, sub_rate=sub, indel_prob=ind)[0]
b = se_batch(reads)
def run():
    return db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
base = run()
for combo in ({}, {"XM_GAPPED_FACTOR": 16}, {"XM_GAPPED_FACTOR": 16, "XM_FULL_LPW": 4}, {"XM_GAPPED_FACTOR": 16, "XM_FULL_LPW": 2, "XM_FULL_WAVES": 16}, {"XM_GAPPED_FACTOR": 8}):
    for k in ("XM_GAPPED_FACTOR", "XM_FULL_LPW", "XM_FULL_WAVES"):
        os.environ.pop(k, None)
    for k, v in combo.items():
        os.environ[k] = str(v)
    r = run()
    same = np.array_equal(r.ints, base.ints) and np.array_equal(r.dbls.view(np.int64), base.dbls.view(np.int64))
    print("%-70s kernel ms %.1f launches %d reruns %d same=%s" % (combo, r.kernel_ms, r.kernel_launches, r.counters[11], same), flush=True)
