"""Round 4 (scripts/gpu_codegen_check_r04.sh): the bench batch through the library XM_LIB_PATH names under a list of knob settings (K=V+K=V,...); prints the pass times and
whether the work counters are the oracle's (31 059 911 search nodes, 1 000 005 alignments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth
nq = 1_000_000
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
reads = synth.synthetic_single_end(ref, nq, read_len=150, seed=0x5EED0001)[0]
mc = np.ones(nq, np.int32); mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 150
ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
p = api.AlignmentParameters()
db = api.ReferenceDatabase([("e", ref)], max_query_length=150)
db.upload_arrays(mc, mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq))
seen = set()
combos = [dict(kv.split("=") for kv in item.split("+") if kv) for item in sys.argv[1].split(",")]
for c in combos: seen.update(c)
for c in combos:
    for k in seen: os.environ.pop(k, None)
    os.environ.update(c)
    r = db.align_resident(p)
    print("%-50s light %.1f ms gapped %.1f ms nodes %d aligned %d %s" % (" ".join("%s=%s" % kv for kv in c.items()) or "default", r.counters[12] / 1e3, sum(r.counters[13:16]) / 1e3, r.counters[6], r.counters[8], "OK" if r.counters[6] == 31059911 and r.counters[8] == 1000005 else "WRONG"), flush=True)
