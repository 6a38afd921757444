"""Results must not depend on the scratch budget (lanes, region pool size, reads that are seeded again): 200 / 6 / 1 GiB."""
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import numpy as np
from helpers import se_batch, pe_batch
from mapper_amd import api, synth
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
db = api.ReferenceDatabase([("e", ref)], max_query_length=150)
reads = synth.synthetic_single_end(ref, 400000, read_len=150, seed=5)[0]
m1, m2 = synth.synthetic_paired_end(ref, 150000, seed=6)[:2]
outs = {}
for gib in ("200", "6", "1"):
    os.environ["XM_SCRATCH_GIB"] = gib
    res = []
    for b in (se_batch(reads), pe_batch(m1, m2, 100.0, 50.0)):
        r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
        res.append((r.ints.copy(), r.dbls.copy(), r.kernel_ms, r.kernel_launches))
    outs[gib] = res
    print("scratch", gib, "GiB: kernel ms", [round(x[2], 1) for x in res], "launches", [x[3] for x in res], flush=True)
for gib in ("6", "1"):
    for k in range(2):
        assert np.array_equal(outs[gib][k][0], outs["200"][k][0]) and np.array_equal(outs[gib][k][1].view(np.int64), outs["200"][k][1].view(np.int64)), (gib, k)
print("identical across scratch budgets")
