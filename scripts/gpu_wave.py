"""GPU experiment: the wave-per-read passes on the bench workload: parity against the oracle on a sample, pass trace, kernel times."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from helpers import se_batch, pe_batch, streams_equal, first_difference
from mapper_amd import api, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "se"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
nCheck = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=150)
if kind == "se":
    reads, _, _ = synth.synthetic_single_end(ref, nq)
    b = se_batch(reads)
else:
    m1, m2 = synth.synthetic_paired_end(ref, nq)[:2]
    b = pe_batch(m1, m2, 100.0, 50.0)
P = api.AlignmentParameters()
for rep in range(3):
    t = time.time()
    r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, P)
    print(kind, "rep", rep, "nq", nq, "wall %.1f ms" % ((time.time() - t) * 1e3), "kernel ms %.2f" % r.kernel_ms, "launches", r.kernel_launches, "us wave light tier / chain tiers / search kernel / lane passes",
          list(r.counters[12:16]), "reruns", r.counters[11], "probes/fetches/hits/cands", list(r.counters[1:5]), "PA calls/nodes", list(r.counters[5:7]), "quick", r.counters[7], flush=True)
if nCheck > 0:
    n = min(nq, nCheck)
    if kind == "se":
        sb = se_batch(reads[:n])
    else:
        sb = pe_batch(m1[:n], m2[:n], 100.0, 50.0)
    O = oracle_lib.OracleReference([("ecoli_syn", ref)], mode="mapper")
    want = O.align(sb, oracle_lib.make_params(), threads=os.cpu_count())
    got = db.align_arrays(sb.mate_count, sb.mate_offset, sb.mate_length, sb.codes, sb.expected_inner, sb.deviation, P)
    same = np.array_equal(got.int_off, want.int_off) and np.array_equal(got.ints, want.ints) and np.array_equal(got.dbls.view(np.int64), np.asarray(want.dbls).view(np.int64))
    print(kind, "parity vs oracle on", n, "queries:", same, flush=True)
    if not same:
        class S: pass
        g = S(); g.ints, g.dbls, g.int_off, g.dbl_off = got.ints, got.dbls, got.int_off, got.dbl_off
        print(first_difference(g, want, n))
