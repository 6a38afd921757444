"""configs[4] at reduced size through the api (no bench plumbing): python gpu_c4_small.py <scale> <reads of 10 kb> [steps] - prints pass times and counters.
XM_LIB_PATH selects another build of the library (A/B)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mapper_amd import api, synth, cli

scale, n_reads = float(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sub, ind = (0.05, 0.05) if os.environ.get("C4_MILD") != "1" else (0.02, 0.002)
contigs, whole, gstarts, gruns = synth.grch38_shaped_reference(scale=scale)
where = synth.genome_wide_starts(gstarts, gruns, n_reads, 10_000 + 2_500 + 8, seed=0x5EED0004 ^ 0xF00D)[0]
strand = (synth.splitmix64(0x5EED0004 ^ 0x57A, n_reads) >> np.uint64(63)).astype(np.uint8)
reads = synth.synthetic_long_reads(whole, where, 10_000, seed=0x5EED0004, sub_rate=sub, indel_rate=ind, strand=strand)
sections = cli.split_sections(10_000, 1000)
k = len(sections); n = n_reads * k
mo = np.zeros(2 * n, np.int64)
mo[0::2] = (np.arange(n_reads, dtype=np.int64)[:, None] * 10_000 + np.array([a for a, _ in sections], dtype=np.int64)[None, :]).reshape(-1)
ml = np.zeros(2 * n, np.int32); ml[0::2] = np.tile(np.array([b - a for a, b in sections], dtype=np.int32), n_reads)
mc = np.ones(n, np.int32); codes = np.ascontiguousarray(reads.reshape(-1)); z = np.zeros(n); o1 = np.ones(n)
db = api.ReferenceDatabase(contigs, mode="mapper", max_query_length=1000, min_interesting_size=13)
db.upload_arrays(mc, mo, ml, codes, z, o1)
p = api.AlignmentParameters()
for s in range(steps + 1):
    t = time.perf_counter()
    r = db.align_resident(p)
    dt = time.perf_counter() - t
    print("step %d: %d queries in %.3f s = %.1f k queries/s; kernel %.1f ms (light %.1f, gapped %.1f), launches %d; calls %d nodes %d; filter: on %d examined %d rejected %d cells %d pieces %d rejected %d; reruns %d" % (
        s, n, dt, n / dt / 1e3, r.kernel_ms, r.counters[12] / 1e3, r.counters[15] / 1e3, r.kernel_launches, r.counters[5], r.counters[6], r.extra[3], r.extra[0], r.extra[1], r.extra[2], r.extra[4], r.extra[5], r.counters[11]), flush=True)
if any(r.prof):
    names = ["TOTAL", "PYRAMID", "WALK", "HITS", "STRAIGHT", "ANALYZE", "PATH", "PATH_INIT", "BLOCK", "MATCHER_INDEX", "CONFIDENT", "OUTER", "BOUND", "PA_LOOK+LOAD", "PA_COMPUTE", "PA_PUT"]
    tot = max(1, r.prof[0])
    print("phase ticks (wave time, %% of TOTAL %d G): " % (tot // 10**9) + ", ".join("%s %.1f" % (n, 100.0 * v / tot) for n, v in zip(names, r.prof) if n != "TOTAL"))
if os.environ.get("C4_CHECK"):
    import oracle_lib as ol
    m = int(os.environ["C4_CHECK"])
    sc = [(nm, np.ascontiguousarray(c)) for nm, c in contigs]
    R = ol.OracleReference(sc, min_interesting_size=13)
    b = ol.QueryBatch.from_arrays(mc[:m], mo[:2 * m], ml[:2 * m], codes, z[:m], o1[:m])
    w = R.align(b, ol.make_params(), threads=os.cpu_count())
    same = np.array_equal(w.ints, r.ints[:r.int_off[m]]) and np.array_equal(w.dbls.view(np.int64), r.dbls[:r.dbl_off[m]].view(np.int64))
    print("first %d queries equal the oracle's streams: %s" % (m, same))
db.close()
