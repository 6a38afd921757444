"""Bisect (round 3): which of the scratch right-sizing changes breaks results.  262,144 reads, knobs per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from mapper_amd import api, synth
import oracle_lib
nq = 262144
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
reads = synth.synthetic_single_end(ref, nq, read_len=150, seed=0x5EED0001)[0]
mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 150
ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
arrays = (np.ones(nq, np.int32), mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq))
p = api.AlignmentParameters()
o = oracle_lib.OracleReference([("e", ref)])
want = o.align(oracle_lib.QueryBatch.from_arrays(*arrays), oracle_lib.make_params(), threads=os.cpu_count())
KN = ["XM_HANDBACK", "XM_SEARCH_POOL", "XM_GAPPED_TMP_PCT", "XM_REGION_KB", "XM_LIGHT_TMP_KB", "XM_HANDOVER", "XM_PAIR_LANES"]
combos = [
    {},
    {"XM_HANDBACK": 0},
    {"XM_SEARCH_POOL": 0, "XM_GAPPED_TMP_PCT": 100, "XM_REGION_KB": 120, "XM_LIGHT_TMP_KB": 168},   # the old sizes
    {"XM_SEARCH_POOL": 0, "XM_GAPPED_TMP_PCT": 100, "XM_REGION_KB": 120},                            # + small light temporaries
    {"XM_SEARCH_POOL": 0, "XM_GAPPED_TMP_PCT": 100},                                                 # + small regions
    {"XM_SEARCH_POOL": 0, "XM_GAPPED_TMP_PCT": 30},                                                  # + small gapped temporaries, no pool (reruns expected, results must hold)
    {},                                                                                              # everything (pool on)
    {"XM_PAIR_LANES": 0},
    {"XM_HANDOVER": 0},
]
for c in combos[:int(sys.argv[1]) if len(sys.argv) > 1 else len(combos)]:
    for k in KN:
        os.environ.pop(k, None)
    for k, v in c.items():
        os.environ[k] = str(v)
    db = api.ReferenceDatabase([("e", ref)], max_query_length=150)
    for rep in range(2):
        r = db.align_arrays(*arrays, p)
        same = np.array_equal(r.ints, want.ints) and np.array_equal(r.dbls.view(np.int64), want.dbls.view(np.int64))
        print(c, "call", rep, "launches", r.kernel_launches, "reruns", r.counters[11], "us", list(r.counters[12:16]), "equal to the oracle:", same, flush=True)
    db.close()
