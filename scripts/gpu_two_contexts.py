"""Experiment: the 1 M-read batch as two halves aligned at the same time by two contexts on one GPU (two host threads, two streams),
against one context with the whole batch.  (Here: two ReferenceDatabase objects, i.e. two copies of the index.)"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth
nq = 1_000_000
gib = sys.argv[1] if len(sys.argv) > 1 else "96"
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
reads = synth.synthetic_single_end(ref, nq, read_len=150, seed=0x5EED0001)[0]
p = api.AlignmentParameters()
def arrays(r):
    n = len(r)
    mc = np.ones(n, np.int32); mo = np.zeros(2 * n, np.int64); mo[0::2] = np.arange(n, dtype=np.int64) * 150
    ml = np.zeros(2 * n, np.int32); ml[0::2] = 150
    return mc, mo, ml, np.ascontiguousarray(r.reshape(-1)), np.zeros(n), np.ones(n)
one = api.ReferenceDatabase([("e", ref)], max_query_length=150)
one.upload_arrays(*arrays(reads))
for rep in range(4):
    t = time.perf_counter(); r = one.align_resident(p); dt = time.perf_counter() - t
print("one context: %.1f ms per 1 M reads (kernel %.1f)" % (dt * 1e3, r.kernel_ms), flush=True)
whole = r
one.close()
os.environ["XM_SCRATCH_GIB"] = gib
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dbs = [api.ReferenceDatabase([("e", ref)], max_query_length=150) for _ in range(k)]
cut = [nq * i // k for i in range(k + 1)]
for i, db in enumerate(dbs):
    db.upload_arrays(*arrays(reads[cut[i]:cut[i + 1]]))
out = [None] * k
def work(i): out[i] = dbs[i].align_resident(p)
for rep in range(4):
    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t
print("%d contexts (%s GiB scratch each): %.1f ms per 1 M reads (kernel ms of each %s)" % (k, gib, dt * 1e3, [round(o.kernel_ms, 1) for o in out]), flush=True)
same = np.array_equal(np.concatenate([o.ints for o in out]), whole.ints)
print("results identical to the single context:", same)
