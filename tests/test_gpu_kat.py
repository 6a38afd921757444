"""GPU tier of the reference's component-level known-answer tests (tests/golden/kat_reference.json "local_cases"): the three
PathAligner_Test triples (T/PathAligner_Test.java:10-39) and the four HashBlockAligner_Test triples (T/HashBlockAligner_Test.java:10-48)
through the DEVICE code of the align kernels - the lane-per-read search in the wave's LDS slot and in HBM mode, the wave-cooperative
search with both capacity sets, the lane-private form of xm_wsearch.h, and hashBlockAlign with its searches slot-first, HBM-only and lane-private - via the test-only entry xm_test_local_align.
Asserted: aligned text A, aligned text B and the penalty, exactly as the JUnit tests assert them."""
import ctypes as C
import numpy as np
import pytest

import oracle_lib as o
from helpers import KAT
from mapper_amd import api, _capi

CASES = [(c, m) for c in KAT["local_cases"] for m in ((0, 1, 2, 3, 4) if c["chain"] == 0 else (0, 1, 4))]
MODE_NAMES = {0: "lds-slot", 1: "hbm", 2: "wave-search", 3: "wave-search-inline-capacities", 4: "lane-private"}


def local_align(chain, mode, query, reference, params, max_ins, max_del):
    L = _capi.lib()
    q, r = api.encode(query), api.encode(reference)
    p = api.AlignmentParameters(**{k: v for k, v in params.items()})._c()
    blocks = np.zeros(4 * 64, np.int32)
    nb = C.c_int32(0)
    pen = np.zeros(2)
    nodes = C.c_int64(0)
    rc = L.xm_test_local_align(0, chain, mode, C.byref(p), q.ctypes.data, len(q), r.ctypes.data, len(r), max_ins, max_del, 64, blocks.ctypes.data, C.byref(nb), pen.ctypes.data, C.byref(nodes))
    if rc < 0:
        raise RuntimeError(L.xm_last_error().decode())
    if rc == 1:
        return None
    a, b = [], []
    for k in range(nb.value):
        sa, sb, la, lb = (int(x) for x in blocks[4 * k:4 * k + 4])
        a.append(query[sa:sa + la] if la > 0 else "-" * lb)
        b.append(reference[sb:sb + lb] if lb > 0 else "-" * la)
    return "".join(a), "".join(b), float(pen[0]), int(nodes.value)


@pytest.mark.gpu
@pytest.mark.parametrize("case,mode", CASES, ids=["%s/%s" % (c["name"], MODE_NAMES[m]) for c, m in CASES])
def test_local_aligner_cases_on_the_gpu(case, mode):
    got = local_align(case["chain"], mode, case["query"], case["reference"], case["params"], case["penalty"], case["penalty"])
    assert got is not None
    assert got[0] == case["alignedA"] and got[1] == case["alignedB"]
    if case["exact"]:
        assert got[2] == case["penalty"]
    else:
        assert abs(got[2] - case["penalty"]) <= 0.000001  # tolerance of T/HashBlockAligner_Test.java:76
    # and bit-identical to the oracle's answer for the same call (penalty as the same double)
    want = o.kat_local_align(case["chain"], case["query"], case["reference"], o.make_params(case["params"]), case["penalty"], case["penalty"])
    assert (got[0], got[1]) == (want[0], want[1]) and np.float64(got[2]).view(np.int64) == np.float64(want[2]).view(np.int64)


@pytest.mark.gpu
def test_search_forms_put_the_same_nodes():
    """The five forms of the search are the same best-first search: they put the same number of nodes (PathAligner.java:446-473) on every
    PathAligner case, and on random texts with an indel they agree on blocks, penalty and node count."""
    rng = np.random.default_rng(7)
    cases = [(c["query"], c["reference"], c["params"], c["penalty"]) for c in KAT["local_cases"] if c["chain"] == 0]
    base = dict(KAT["local_cases"][0]["params"])
    for _ in range(12):
        n = int(rng.integers(20, 60))
        ref = "".join("ACGT"[i] for i in rng.integers(0, 4, n + 12))
        cut = int(rng.integers(5, n - 5))
        q = ref[2:cut] + ref[cut + int(rng.integers(1, 4)):n]
        cases.append((q, ref, base, 8.0))
    for q, r, prm, pen in cases:
        res = [local_align(0, m, q, r, prm, pen, pen) for m in (0, 1, 2, 3, 4)]
        assert all((x is None) == (res[0] is None) for x in res)
        if res[0] is not None:
            assert all(x[:3] == res[0][:3] for x in res), (q, r, res)
            assert res[2][3] == res[3][3] and res[0][3] == res[1][3] == res[2][3] == res[4][3], (q, r, [x[3] for x in res])
