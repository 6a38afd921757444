"""Experiment: throughput of k contexts on one GPU, each aligning its own resident 1 M-read batch at the same time (k host threads, k streams,
k sets of scratch), against one context aligning its batch k times in a row.  The index of the extra contexts is a replica (xm_index_replicate)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth
nq = 1_000_000
gib = sys.argv[1] if len(sys.argv) > 1 else "90"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
os.environ["XM_SCRATCH_GIB"] = gib
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
reads = synth.synthetic_single_end(ref, nq, read_len=150, seed=0x5EED0001)[0]
p = api.AlignmentParameters()
n = len(reads)
mc = np.ones(n, np.int32); mo = np.zeros(2 * n, np.int64); mo[0::2] = np.arange(n, dtype=np.int64) * 150
ml = np.zeros(2 * n, np.int32); ml[0::2] = 150
arrays = (mc, mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(n), np.ones(n))
first = api.ReferenceDatabase([("e", ref)], max_query_length=150)
dbs = [first] + [first.replicate(0) for _ in range(k - 1)]
for db in dbs:
    db.upload_arrays(*arrays)
reps = 4
for db in dbs:
    db.align_resident(p)  # warm-up: scratch allocated
t = time.perf_counter()
for _ in range(reps * k):
    r0 = first.align_resident(p)
seq = (time.perf_counter() - t) / (reps * k)
print("one context, one batch after the other (%s GiB scratch): %.1f ms per 1 M reads (kernel %.1f)" % (gib, seq * 1e3, r0.kernel_ms), flush=True)
out = [None] * k
def work(i):
    for _ in range(reps):
        out[i] = dbs[i].align_resident(p)
t = time.perf_counter()
th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
[x.start() for x in th]; [x.join() for x in th]
par = (time.perf_counter() - t) / (reps * k)
print("%d contexts at the same time: %.1f ms per 1 M reads (kernel ms of each %s) = %.2f x" % (k, par * 1e3, [round(o.kernel_ms, 1) for o in out], seq / par), flush=True)
print("results identical:", all(np.array_equal(o.ints, r0.ints) for o in out))
