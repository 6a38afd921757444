"""Experiment (round 3): what a context needs as scratch.  One context with scratch limits from 200 down to 24 GiB (xm_context_set_scratch), then k contexts
that share the index (xm_context_new) and divide the free HBM.  Per-pass kernel times, reruns, results compared with the first run.
usage: gpu_scratch_r03.py [config 1|2]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "1"
nq = 1_000_000
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
if cfg == "2":
    m1, m2 = synth.synthetic_paired_end(ref, nq, read_len=150, seed=0x5EED0002)[:2]
    L = 150
    codes = np.ascontiguousarray(np.concatenate([m1, m2], axis=1).reshape(-1))
    mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 2 * L; mo[1::2] = mo[0::2] + L
    arrays = (np.full(nq, 2, np.int32), mo, np.full(2 * nq, L, np.int32), codes, np.full(nq, 100.0), np.full(nq, 50.0))
else:
    reads = synth.synthetic_single_end(ref, nq, read_len=150, seed=0x5EED0001)[0]
    mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 150
    ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
    arrays = (np.ones(nq, np.int32), mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq))
p = api.AlignmentParameters()
db = api.ReferenceDatabase([("e", ref)], max_query_length=150)
db.upload_arrays(*arrays)
base = db.align_resident(p)
print("free HBM after the index and one run: %.1f GiB" % (api.device_memory(0)[0] / 2**30), flush=True)
for gib in (200, 96, 64, 48, 32, 24):
    db.set_scratch(gib << 30)
    best = None
    for _ in range(3):
        t = time.perf_counter()
        r = db.align_resident(p)
        wall = (time.perf_counter() - t) * 1e3
        us = list(r.counters[12:16])
        if best is None or wall < best[0]:
            best = (wall, us)
    same = np.array_equal(r.ints, base.ints) and np.array_equal(r.dbls.view(np.int64), base.dbls.view(np.int64))
    print("one context, scratch limit %3d GiB: step %.1f ms, light %.1f ms, gapped+reruns %.1f ms, launches %d, reruns %d, same=%s, free HBM now %.1f GiB" % (
        gib, best[0], best[1][0] / 1e3, best[1][3] / 1e3, r.kernel_launches, r.counters[11], same, api.device_memory(0)[0] / 2**30), flush=True)
for k in (2, 3, 4):
    ctx = [db] + [db.new_context() for _ in range(k - 1)]
    n, share = api.divide_scratch(ctx, 0)
    for c in ctx[1:]:
        c.upload_arrays(*arrays)
    for c in ctx:
        c.align_resident(p)
    reps = 4
    out = [None] * k
    def work(i):
        for _ in range(reps):
            out[i] = ctx[i].align_resident(p)
    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    [x.start() for x in th]; [x.join() for x in th]
    per = (time.perf_counter() - t) / (reps * k) * 1e3
    same = all(np.array_equal(o.ints, base.ints) for o in out)
    print("%d contexts sharing the index, %.1f GiB of scratch each (%d used): %.1f ms per batch = %.2f M reads/s, kernel ms of each %s, same=%s" % (
        k, share / 2**30, n, per, nq * (2 if cfg == "2" else 1) / per / 1e3, [round(o.kernel_ms, 1) for o in out], same), flush=True)
    for c in ctx[1:]:
        c.close()
