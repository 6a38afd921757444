"""Diagnostic (-DXM_READ_TIMES build in mapper_amd/_lib_rt): how long the gapped pass spends on each read (shader-clock ticks of the lane that ran it)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ["XM_LIB_PATH"] = os.path.join(ROOT, "mapper_amd", "_lib_rt", "libxmapper_hip.so")
os.environ["XM_READ_TIMES_FILE"] = "/tmp/read_times.bin"
from mapper_amd import api, synth
from helpers import se_batch
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
reads = synth.synthetic_single_end(ref, 1_000_000, read_len=150, seed=0x5EED0001)[0]
b = se_batch(reads)
db = api.ReferenceDatabase([("e", ref)], max_query_length=150)
for rep in range(2):
    r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
t = np.fromfile("/tmp/read_times.bin", dtype=np.uint64).astype(np.float64) / 2.4e6  # ms at 2.4 GHz (clock64 = shader clock)
print("kernel ms by pass", list(r.counters[12:16]))
print("reads", len(t), "sum of per-read lane time %.1f s" % (t.sum() / 1e3))
for q in (50, 90, 99, 99.9, 99.99, 99.999):
    print("percentile %g: %.3f ms" % (q, np.percentile(t, q)))
top = np.sort(t)[::-1]
print("top 20 reads (ms):", np.round(top[:20], 2))
for thr in (1, 2, 5, 10, 15, 20):
    print("reads over %d ms: %d (%.1f s of lane time)" % (thr, int((t > thr).sum()), t[t > thr].sum() / 1e3))
