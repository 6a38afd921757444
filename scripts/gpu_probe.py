"""GPU experiment: per-pass kernel times for different read populations."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mapper_amd import api, synth
N = 5_000_000
ref = synth.synthetic_reference(N)
db = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=150)
def run(tag, reads):
    nq = len(reads)
    mc = np.ones(nq, np.int32); mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq) * 150; ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
    codes = np.ascontiguousarray(reads.reshape(-1))
    for rep in range(2):
        r = db.align_arrays(mc, mo, ml, codes, np.zeros(nq), np.ones(nq), api.AlignmentParameters())
    print(tag, "nq", nq, "kernel ms %.2f" % r.kernel_ms, "launches", r.kernel_launches, "pass us", r.counters[12:16], "Mreads/s %.3f" % (nq / r.kernel_ms / 1e3), "ctr", r.counters[:12], "prof(Mticks)", [round(x/1e6,1) for x in r.prof[:12]], flush=True)
for n in [1024, 65536, 262144]:
    perfect, _, _ = synth.synthetic_single_end(ref, n, sub_rate=0.0, indel_prob=0.0)
    run("perfect", perfect)
    subs, _, _ = synth.synthetic_single_end(ref, n, sub_rate=0.01, indel_prob=0.0)
    run("subs1pct", subs)
    full, _, _ = synth.synthetic_single_end(ref, n)
    run("default", full)
