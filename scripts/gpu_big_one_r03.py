"""GPU experiment (round 3): configs[4]-style queries (10 kb reads with per-base error rates, cut at 1000) against the 5 Mb reference, one context, passes traced
(XM_TRACE_PASSES=1) - for A/B runs of libraries (XM_LIB_PATH).  argv: 4|4mild n_queries"""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mapper_amd import api, synth, cli
cfg = sys.argv[1]
nq = int(sys.argv[2])
sub, ind = (0.05, 0.05) if cfg == "4" else (0.02, 0.002)
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
n_reads = nq // 10
starts = (synth.splitmix64(0x5EED0004, n_reads) % np.uint64(len(ref) - 12_600)).astype(np.int64)
strand = (synth.splitmix64(0x5EED0004 ^ 0x57A, n_reads) >> np.uint64(63)).astype(np.uint8)
reads = synth.synthetic_long_reads(ref, starts, 10_000, seed=0x5EED0004, sub_rate=sub, indel_rate=ind, strand=strand)
sections = cli.split_sections(10_000, 1000)
n = n_reads * len(sections)
mo = np.zeros(2 * n, np.int64)
mo[0::2] = (np.arange(n_reads, dtype=np.int64)[:, None] * 10_000 + np.array([a for a, _ in sections], dtype=np.int64)[None, :]).reshape(-1)
ml = np.zeros(2 * n, np.int32); ml[0::2] = np.tile(np.array([b - a for a, b in sections], dtype=np.int32), n_reads)
db = api.ReferenceDatabase([("e", ref)], max_query_length=1000)
for rep in range(2):
    r = db.align_arrays(np.ones(n, np.int32), mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(n), np.ones(n), api.AlignmentParameters())
print(cfg, "queries", n, "kernel ms %.1f" % r.kernel_ms, "launches", r.kernel_launches, "reruns", r.counters[11], "nodes", r.counters[6], "digest", hashlib.sha256(r.ints.tobytes() + r.dbls.tobytes()).hexdigest()[:12])
