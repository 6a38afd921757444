"""Experiment (round 4): 1,000 bp queries (configs[4]'s pieces) through the gapped pass as the wave scheduler (XM_SCHED_LONG=1) against the
lane-per-read gapped pass, one context; 'easy' = 1 % substitutions + an indel in 5 % of the reads, 'hard' = 3 % + an indel in every second read.
usage: gpu_sched_long_r04.py [n easy] [n hard] [combos 'K=V+K=V,...']"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import se_batch
from mapper_amd import api, synth
n_easy = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
n_hard = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
combos = [dict(kv.split("=") for kv in item.split("+") if kv) for item in (sys.argv[3] if len(sys.argv) > 3 else "XM_SCHED_LONG=0,XM_SCHED_LONG=1").split(",")]
L = 1000
ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("r", ref)], mode="mapper", max_query_length=L)
p = api.AlignmentParameters()
seen = set()
for c in combos:
    seen.update(c.keys())
for name, n, sub, ind in (("easy", n_easy, 0.01, 0.05), ("hard", n_hard, 0.03, 0.5)):
    if n <= 0:
        continue
    reads = synth.synthetic_single_end(ref, n, read_len=L, sub_rate=sub, indel_prob=ind)[0]
    b = se_batch(reads)
    db.upload_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation)
    for k in seen:
        os.environ.pop(k, None)
    os.environ["XM_SCHED_LONG"] = "0"
    base = db.align_resident(p)
    for c in combos:
        for k in seen:
            os.environ.pop(k, None)
        for k, v in c.items():
            os.environ[k] = str(v)
        r = db.align_resident(p)
        same = np.array_equal(r.int_off, base.int_off) and np.array_equal(r.ints, base.ints) and np.array_equal(r.dbls.view(np.int64), base.dbls.view(np.int64))
        cnt = np.array_equal(np.asarray(r.counters[:11]), np.asarray(base.counters[:11]))
        us = list(r.counters[12:16])
        print("%-5s n %d  %-70s light %.1f ms gapped+reruns %.1f ms launches %d reruns %d same=%s counters=%s -> %.3f M reads/s" % (
            name, n, " ".join("%s=%s" % kv for kv in c.items()), us[0] / 1e3, (us[1] + us[2] + us[3]) / 1e3, r.kernel_launches, r.counters[11], same, cnt, n / r.kernel_ms / 1e3), flush=True)
