"""GPU experiment (round 3): the third update of an explored entry from the outcomes of the first two (XM_PA_FORWARD, xm_extend.h) - one library per process
(XM_LIB_PATH), the same batches: 1 M reads of 150 bp, 1 M pairs, 1 kb queries easy and hard; kernel ms by pass and a digest of the result streams."""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import se_batch, pe_batch
from mapper_amd import api, synth
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
def run(db, b, reps=3):
    best = None
    for _ in range(reps):
        r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
        if best is None or r.kernel_ms < best.kernel_ms:
            best = r
    c = best.counters
    return "kernel ms %.1f (light %.1f gapped %.1f) nodes %d digest %s" % (best.kernel_ms, c[12] / 1e3, c[15] / 1e3, c[6], hashlib.sha256(best.ints.tobytes() + best.dbls.tobytes()).hexdigest()[:12])
db = api.ReferenceDatabase([("e", ref)], max_query_length=150)
print("150 bp x 1M :", run(db, se_batch(synth.synthetic_single_end(ref, 1_000_000, read_len=150, seed=0x5EED0001)[0])), flush=True)
m1, m2 = synth.synthetic_paired_end(ref, 500_000, read_len=150, seed=0x5EED0002)[:2]
print("pairs x 500k:", run(db, pe_batch(m1, m2, 100.0, 50.0)), flush=True)
db.close()
db = api.ReferenceDatabase([("e", ref)], max_query_length=1000)
print("1 kb easy 150k:", run(db, se_batch(synth.synthetic_single_end(ref, 150_000, read_len=1000, sub_rate=0.01, indel_prob=0.05)[0]), 2), flush=True)
print("1 kb hard 100k:", run(db, se_batch(synth.synthetic_single_end(ref, 100_000, read_len=1000, sub_rate=0.03, indel_prob=0.5)[0]), 2), flush=True)
