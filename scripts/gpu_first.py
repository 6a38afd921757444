"""First GPU run: smoke parity + a timing at growing batch sizes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
g.smoke()
import oracle_lib
from mapper_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
ref = synth.synthetic_reference(N)
t = time.time(); db = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=150); print("index build+upload s", time.time() - t, db.info())
for nreads in [10_000, 100_000, 1_000_000]:
    reads, starts, strand = synth.synthetic_single_end(ref, nreads)
    nq = len(reads)
    mc = np.ones(nq, np.int32); mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq) * 150; ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
    codes = reads.reshape(-1)
    for rep in range(2):
        t = time.time(); r = db.align_arrays(mc, mo, ml, codes, np.zeros(nq), np.ones(nq), api.AlignmentParameters()); dt = time.time() - t
        print("nreads", nreads, "wall s %.3f" % dt, "kernel ms %.2f" % r.kernel_ms, "launches", r.kernel_launches, "Mreads/s(kernel) %.3f" % (nq / r.kernel_ms / 1e3), "h2d %.2f d2h %.2f" % (r.h2d_ms, r.d2h_ms), "counters", r.counters[:12], flush=True)
    if nreads <= 100_000:
        o = oracle_lib.OracleReference([("ecoli_syn", ref)], mode="mapper")
        b = oracle_lib.QueryBatch.from_arrays(mc, mo, ml, codes, np.zeros(nq), np.ones(nq))
        t = time.time(); w = o.align(b, oracle_lib.make_params(), threads=os.cpu_count()); print("oracle s %.2f (incl. index)" % (time.time() - t))
        same = np.array_equal(r.ints, w.ints) and np.array_equal(r.dbls.view(np.int64), w.dbls.view(np.int64)) and np.array_equal(r.int_off, w.int_off)
        print("parity vs oracle:", "IDENTICAL" if same else "DIFFERENT", flush=True)
