"""Streaming throughput with host buffers in (PCIe included): batches through xm_align_batch one after the other against align_stream
(xm_batch_stage of batch k+1 on a second thread and stream while batch k is aligned)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
db = api.ReferenceDatabase([("e", ref)], max_query_length=150)
p = api.AlignmentParameters()
batches = []
for k in range(nb):
    reads = synth.synthetic_single_end(ref, nq, read_len=150, seed=0x5EED0001 + k)[0]
    mc = np.ones(nq, np.int32)
    mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 150
    ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
    batches.append((mc, mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq)))
db.align_arrays(*batches[0], p)  # warm-up
for rep in range(2):
    t = time.perf_counter()
    h2d = 0.0
    for b in batches:
        r = db.align_arrays(*b, p)
        h2d += r.h2d_ms
    t1 = time.perf_counter() - t
    t = time.perf_counter()
    n = sum(1 for _ in db.align_stream(iter(batches), p))
    t2 = time.perf_counter() - t
    print("%d batches x %d reads: one after the other %.1f ms/batch (H2D %.1f ms of it) = %.2f M reads/s; streamed %.1f ms/batch = %.2f M reads/s" %
          (nb, nq, t1 / nb * 1e3, h2d / nb, nb * nq / t1 / 1e6, t2 / n * 1e3, nb * nq / t2 / 1e6), flush=True)
