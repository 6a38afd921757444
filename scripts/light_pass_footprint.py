"""Where the light pass's bytes go (round 4, verdict item 3): the footprint of one read of the bench workload in a lane's scratch, structure by structure,
from the host simulation of the kernel sources built with -DXM_ARENA_TRACE (every arena allocation is recorded; the arenas start filled with a pattern and the
bytes that no longer hold it after the read's light pass are counted).  Footprint = bytes written at least once: with 262,144 lanes resident the scratch is a
13 GB working set no cache holds, so a byte written is a byte that goes to HBM and, if it is ever read again, comes back from there.
usage: light_pass_footprint.py [reads] [out.json]"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as o
import hostsim_lib as hs
from mapper_amd import synth, _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
out_path = sys.argv[2] if len(sys.argv) > 2 else None
lib = os.path.join(ROOT, "tests", "_build", "libxm_hostsim_trace.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-w", "-DXM_ARENA_TRACE", "-o", lib, hs.SRC])
hs.OUT = lib; hs.build = lambda: lib
ref = synth.synthetic_reference(5_000_000, seed=0xEC011)
reads = synth.synthetic_single_end(ref, n, read_len=150, seed=0x5EED0001)[0]
from helpers import se_batch
b = se_batch(reads)
S = hs.SimReference([("ecoli_syn", ref)])
L = hs.lib()
L.xmsim_light_footprint.restype = C.c_int64
rows = np.zeros(3 * 256, np.int64)
done = C.c_int64(0)
params = o.make_params()
p = _capi.XmParams()
for f, _ in _capi.XmParams._fields_:
    if f != "reserved":
        setattr(p, f, getattr(params, f))
cb, keep = _capi.make_batch(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation)
L.xmsim_light_footprint.argtypes = [C.c_void_p, C.POINTER(_capi.XmParams), C.POINTER(_capi.XmQueryBatch), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
k = L.xmsim_light_footprint(S.h, C.byref(p), C.byref(cb), rows.ctypes.data, 256, C.byref(done))
assert k > 0, (k, done.value)
names_region = ["pyramid blocks (levels >= 1, 16-byte words)", "pyramid level starts", "vote counters", "good-counter list", "history (blocks looked up)", "pending blocks",
                "list: high-priority positions", "list: best positions", "list: all positions", "assembled query matches", "filtered query matches", "nearby scratch list",
                "QueryMatch_Aligner", "accepted alignments (QAl)", "accepted alignments' block pool", "candidate blocks", "best-alignment index"]
table, ri = [], 0
for i in range(k):
    arena, cap, dirty = int(rows[3 * i]), int(rows[3 * i + 1]), int(rows[3 * i + 2])
    if arena == 0:
        name = names_region[ri] if ri < len(names_region) else "region allocation %d" % ri
        ri += 1
    else:
        name = "temporaries allocation %d (matchers alignMatch sets aside and their tables)" % i
    table.append({"structure": name, "arena": "read region" if arena == 0 else "lane temporaries", "capacity_bytes": cap, "bytes_written_per_read": round(dirty / done.value, 1)})
tot = sum(t["bytes_written_per_read"] for t in table)
res = {"workload": "configs[1]: first %d reads of the bench batch (150 bp, seed 0x5EED0001) vs the 5 Mb reference" % done.value, "method": __doc__.split("usage:")[0].strip(),
       "bytes_written_per_read_total": round(tot, 1), "structures": sorted(table, key=lambda t: -t["bytes_written_per_read"])}
print(json.dumps(res, indent=1))
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
