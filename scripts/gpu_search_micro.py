"""Micro-benchmark of ONE PathAligner search on an otherwise idle GPU (the test entry xm_test_local_align: one lane of one wave): texts of 150 bases with an
indel in the middle and mismatches, as the gapped pass meets them.  Prints nodes put per search; run under `rocprofv3 --kernel-trace --stats` to get the
kernel's duration per problem (xm_test_local_kernel), or read the host's wall time per call printed here (launch and allocation overhead included).
usage: gpu_search_micro.py [mode 0 lds-slot | 1 hbm | 4 lane-private] [calls]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, _capi
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
L = _capi.lib()
rng = np.random.default_rng(5)
def problem(n, indel, nsub, shift):
    ref = rng.integers(0, 4, n + 2 * shift + 8)
    q = list(ref[shift:shift + n])
    cut = n // 2
    if indel > 0: q = q[:cut] + list(rng.integers(0, 4, indel)) + q[cut:n - indel]
    elif indel < 0: q = q[:cut] + q[cut - indel:] + list(ref[shift + n:shift + n - indel])
    for p in rng.integers(0, len(q), nsub): q[p] = (q[p] + 1) % 4
    return "".join("ACGT"[i] for i in q), "".join("ACGT"[i] for i in ref[:n + 2 * shift])
p = api.AlignmentParameters()._c()
# (texts of up to 128 x 253 bases run in the wave's LDS slot, as nearly all searches of 150 bp reads do: BlockAligner cuts a read into pieces first; longer texts start in HBM mode)
for name, (q, r) in (("60 bases, 1 indel + 1 sub", problem(60, 1, 1, 3)), ("80 bases, 1 indel + 1 sub", problem(80, 1, 1, 3)), ("100 bases, 1 indel + 1 sub", problem(100, -1, 1, 3)), ("150 bases (HBM mode), 1 indel + 3 subs", problem(150, 2, 3, 8))):
    qa, ra = api.encode(q), api.encode(r)
    blocks = np.zeros(4 * 64, np.int32); nb = C.c_int32(0); pen = np.zeros(2); nodes = C.c_int64(0)
    t = []
    for i in range(calls):
        t0 = time.perf_counter()
        rc = L.xm_test_local_align(0, 0, mode, C.byref(p), qa.ctypes.data, len(qa), ra.ctypes.data, len(ra), 10.0, 10.0, 64, blocks.ctypes.data, C.byref(nb), pen.ctypes.data, C.byref(nodes))
        t.append(time.perf_counter() - t0)
    print("%s: rc %d, %d blocks, penalty %.2f, nodes put %d, wall per call %.1f us (min %.1f)" % (name, rc, nb.value, pen[0], nodes.value, 1e6 * np.median(t), 1e6 * min(t)), flush=True)
