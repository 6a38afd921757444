"""XM_PROFILE build: in-kernel phase timers of the wave-per-read passes (wave time, shader-clock ticks)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import se_batch, pe_batch
from mapper_amd import api, synth
kind = sys.argv[1]; nq = int(sys.argv[2])
ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=150)
if kind == "se":
    b = se_batch(synth.synthetic_single_end(ref, nq)[0])
else:
    m1, m2 = synth.synthetic_paired_end(ref, nq)[:2]; b = pe_batch(m1, m2, 100.0, 50.0)
names = {0: "TOTAL", 2: "WALK", 3: "STEP", 4: "UNGAPPED", 5: "HITS", 6: "CHAIN", 10: "CONFIDENT", 11: "ALIGNMATCH", 12: "MATEINIT", 13: "WRITE", 14: "OPTIMISTIC"}
for heavy in (os.environ.get("TIERS", "3"),):
    os.environ["XM_WAVE_TIERS"] = heavy
    for rep in range(2):
        r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
    print("tiers", heavy, "kernel ms %.2f" % r.kernel_ms, "us by pass", list(r.counters[12:16]), flush=True)
    print("   Mticks (all slots)", {names.get(i, str(i)): round(x / 1e6, 1) for i, x in enumerate(r.prof) if x}, flush=True)
