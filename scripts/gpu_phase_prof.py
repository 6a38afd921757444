"""Phase timers of the passes of one xm_align_resident call on a bench batch: needs a library built with -DXM_PROFILE=2 (make -C mapper_amd/csrc EXTRA=-DXM_PROFILE=2 OUT=../_lib_prof;
XM_LIB_PATH points at it).  Ticks of wave time (the lowest active lane of a wave counts), summed over the waves of the launch.
usage: gpu_phase_prof.py [config 1|2|rep|4shape] [nq] [gapped|all]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "1"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
which = sys.argv[3] if len(sys.argv) > 3 else "gapped"
L = 1000 if cfg == "4shape" else 150
ref = synth.repeat_rich_reference(5_000_000) if cfg == "rep" else synth.synthetic_reference(5_000_000, seed=0xEC011)
if cfg == "2":
    m1, m2 = synth.synthetic_paired_end(ref, nq, read_len=150, seed=0x5EED0002)[:2]
    codes = np.ascontiguousarray(np.concatenate([m1, m2], axis=1).reshape(-1))
    mc = np.full(nq, 2, np.int32)
    mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 2 * L; mo[1::2] = mo[0::2] + L
    ml = np.full(2 * nq, L, np.int32)
    arrays = (mc, mo, ml, codes, np.full(nq, 100.0), np.full(nq, 50.0))
else:
    reads = synth.synthetic_single_end(ref, nq, read_len=L, seed=0x5EED0001, **({"sub_rate": 0.02, "indel_prob": 0.3} if cfg == "4shape" else {}))[0]
    mc = np.ones(nq, np.int32); mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * L
    ml = np.zeros(2 * nq, np.int32); ml[0::2] = L
    arrays = (mc, mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq))
p = api.AlignmentParameters()
db = api.ReferenceDatabase([("e", ref)], max_query_length=L)
db.upload_arrays(*arrays)
if which == "gapped":
    os.environ["XM_PROF_GAPPED_ONLY"] = "1"
names = ["TOTAL", "PYRAMID", "WALK", "HITS", "STRAIGHT", "ANALYZE", "PATH", "PATH_INIT", "BLOCK", "MATCHER_INDEX", "CONFIDENT", "OUTER", "PA_LOOK", "PA_LOAD", "PA_COMPUTE", "PA_PUT"]
for rep in range(2):
    r = db.align_resident(p)
out = {"workload": "configs[%s], %d queries, one context, %s, library built with -DXM_PROFILE=2" % (cfg, nq, "gapped pass only (XM_PROF_GAPPED_ONLY=1)" if which == "gapped" else "all passes"),
       "unit": "shader-clock ticks of wave time (the lowest active lane of a wave counts), summed over the waves of the launch",
       "kernel_ms_profile_build": r.kernel_ms, "kernel_us_light_gapped": [int(r.counters[12]), int(r.counters[15])], "pathaligner_calls_nodes": [int(x) for x in r.counters[5:7]],
       "phases": {n: int(x) for n, x in zip(names, r.prof)}}
print(json.dumps(out))
tot = max(1, out["phases"]["TOTAL"])
print({n: "%.1f %%" % (100.0 * v / tot) for n, v in out["phases"].items() if n != "TOTAL"}, file=sys.stderr)
if out["pathaligner_calls_nodes"][1]:
    print("ticks per node put: %.0f" % (out["phases"]["PATH"] / out["pathaligner_calls_nodes"][1]), file=sys.stderr)
