"""GPU experiment (round 3): one batch of hard 1 kb queries, run twice; under rocprofv3 --kernel-trace the launches of the second run show what the rerun pass costs.  argv: n sub indel"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import se_batch
from mapper_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sub = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
ind = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("r", ref)], mode="mapper", max_query_length=1000)
reads = synth.synthetic_single_end(ref, n, read_len=1000, sub_rate=sub, indel_prob=ind)[0]
b = se_batch(reads)
for rep in range(2):
    r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
print("n", n, "kernel ms %.1f" % r.kernel_ms, "launches", r.kernel_launches, "reruns", r.counters[11], "pass us", list(r.counters[12:16]))
