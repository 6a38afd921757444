"""Diagnostic (round 4): repeat one resident batch under XM_SCHED=0 / 1 with the pass trace on; report differences between runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth
nq = int(os.environ.get("NQ", 1_000_000))
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
reads = synth.synthetic_single_end(ref, nq, read_len=150, seed=0x5EED0001)[0]
mc = np.ones(nq, np.int32); mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 150
ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
arrays = (mc, mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq))
p = api.AlignmentParameters()
db = api.ReferenceDatabase([("e", ref)], max_query_length=150)
db.upload_arrays(*arrays)
os.environ["XM_TRACE_PASSES"] = "1"
prev = None
for mode in sys.argv[1:] or ["0", "0", "1", "1"]:
    os.environ["XM_SCHED"] = mode
    r = db.align_resident(p)
    print("XM_SCHED=%s launches %d reruns %d counters %s" % (mode, r.kernel_launches, r.counters[11], list(r.counters[:11])), flush=True)
    if prev is not None:
        same = np.array_equal(r.int_off, prev.int_off) and np.array_equal(r.ints, prev.ints) and np.array_equal(r.dbls.view(np.int64), prev.dbls.view(np.int64))
        print("  same as previous:", same)
        if not same:
            lens_a = np.diff(r.int_off); lens_b = np.diff(prev.int_off)
            bad = np.nonzero(lens_a != lens_b)[0]
            print("  queries with different stream lengths:", len(bad), bad[:10])
            if len(bad) == 0:
                d = np.nonzero(r.ints != prev.ints)[0]
                qs = np.unique(np.searchsorted(r.int_off, d, side="right") - 1)
                print("  queries with different ints:", len(qs), qs[:10])
                dd = np.nonzero(r.dbls.view(np.int64) != prev.dbls.view(np.int64))[0]
                print("  different doubles:", len(dd))
    prev = r
