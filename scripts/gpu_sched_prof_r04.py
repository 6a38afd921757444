"""GPU experiment (round 4): the scheduler kernel's phase timers (a library built with -DXM_PROFILE=2: XM_LIB_PATH) on the bench batch.
usage: gpu_sched_prof_r04.py [config 1|2|rep] [nq]   (rep: configs[1]'s reads on synth.repeat_rich_reference)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mapper_amd import api, synth, _capi
cfg = sys.argv[1] if len(sys.argv) > 1 else "1"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ref = synth.repeat_rich_reference(5_000_000) if cfg == "rep" else synth.synthetic_reference(5_000_000, seed=0xEC011)
if cfg == "2":
    m1, m2 = synth.synthetic_paired_end(ref, nq, read_len=150, seed=0x5EED0002)[:2]
    L = 150
    codes = np.ascontiguousarray(np.concatenate([m1, m2], axis=1).reshape(-1))
    mc = np.full(nq, 2, np.int32)
    mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 2 * L; mo[1::2] = mo[0::2] + L
    ml = np.full(2 * nq, L, np.int32)
    arrays = (mc, mo, ml, codes, np.full(nq, 100.0), np.full(nq, 50.0))
else:
    reads = synth.synthetic_single_end(ref, nq, read_len=150, seed=0x5EED0001)[0]
    mc = np.ones(nq, np.int32); mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 150
    ml = np.zeros(2 * nq, np.int32); ml[0::2] = 150
    arrays = (mc, mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq))
p = api.AlignmentParameters()
db = api.ReferenceDatabase([("e", ref)], max_query_length=150)
db.upload_arrays(*arrays)
L = _capi.lib()
L.xm_debug_sched_profile.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
os.environ["XM_PROF_GAPPED_ONLY"] = "1"
names = ["TOTAL", "PYRAMID", "WALK", "HITS", "STRAIGHT", "ANALYZE", "PATH", "PATH_INIT", "BLOCK", "MATCHER_INDEX", "CONFIDENT", "OUTER", "PA_LOOK", "PA_LOAD", "PA_COMPUTE", "PA_PUT"]
snames = ["loop", "chain_fresh", "chain_replay", "search", "ws_state", "iterations", "searches", "search_lane_ticks", "chain_lane_ticks", "ws_poll", "round_max_steps", "round_sum_steps", "ws_list", "ws_lookups", "ws_arith", "ws_puts"]
for rep in range(2):
    out = (C.c_uint64 * 16)()
    L.xm_debug_sched_profile(out, 1)
    r = db.align_resident(p)
    L.xm_debug_sched_profile(out, 1)
print("config", cfg, "nq", nq, "kernel ms %.2f" % r.kernel_ms, "us light/chain/search/inline", list(r.counters[12:16]), "PA calls/nodes", list(r.counters[5:7]), flush=True)
print("gapped-pass Mticks", {n: round(x / 1e6, 1) for n, x in zip(names, r.prof)}, flush=True)
s = {n: int(out[i]) for i, n in enumerate(snames)}
print("scheduler Mticks", {n: round(v / 1e6, 1) for n, v in s.items()}, flush=True)
print("search rounds: explored entries %d, sum over rounds of the longest lane %d -> %.0f ticks of wave time per round-step, %.0f per explored entry" % (s["round_sum_steps"], s["round_max_steps"], s["search"] / max(1, s["round_max_steps"]), s["search"] / max(1, s["round_sum_steps"])), flush=True)
import json
if os.environ.get("XM_PROFILE_JSON"):
    json.dump({"workload": "configs[%s], %d queries, one context, gapped pass only (XM_PROF_GAPPED_ONLY=1), library built with -DXM_PROFILE=2" % (cfg, nq), "XM_SCHED": os.environ.get("XM_SCHED", "0"),
               "unit": "shader-clock ticks of wave time (the lowest active lane of a wave counts), summed over the waves of the launch",
               "kernel_ms_profile_build": r.kernel_ms, "pathaligner_calls_nodes": [int(x) for x in r.counters[5:7]],
               "gapped_pass_phases": {n: int(x) for n, x in zip(names, r.prof)}, "scheduler": s,
               "mean_lanes_searching_at_the_start_of_a_search_phase": (s["search_lane_ticks"] / s["search"]) if s["search"] else None,
               "mean_lanes_stepping_per_lock_step_step": (s["round_sum_steps"] / s["round_max_steps"]) if s["round_max_steps"] else None,
               "mean_lanes_with_chain_work_in_a_chain_phase": (s["chain_lane_ticks"] / max(1, s["chain_fresh"] + s["chain_replay"])) if s["search"] else None},
              open(os.environ["XM_PROFILE_JSON"], "w"), indent=1)
if s["search"]:
    print("mean lanes searching in a search phase: %.2f; mean lanes with chain work in a chain phase: %.2f" % (s["search_lane_ticks"] / s["search"], s["chain_lane_ticks"] / max(1, s["chain_fresh"] + s["chain_replay"])), flush=True)
