import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mapper_amd import api, synth
from test_gpu_bound import long_read_batch
ref = synth.synthetic_reference(2_000_000, seed=0xEC011)
b = long_read_batch(ref, 60, 0.035, 0.02)
db = api.ReferenceDatabase([("r", ref)], max_query_length=1000)
for i in range(2):
    got = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
    print("gpu calls nodes", got.counters[5:7], "extra", got.extra[:4], "reruns", got.counters[11], "launches", got.kernel_launches)
print("expected: calls 41577 nodes 102372226 examined 41577 rejected 11831")
