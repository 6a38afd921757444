"""GPU experiment (XM_PROFILE=2 build in mapper_amd/_lib_prof, XM_LIB_PATH): wave-time phase timers of the gapped pass on 1,000 bp queries,
for the launch shape given in the environment (XM_FULL_LPW / XM_FULL_WAVES).  argv: label [n] [sub_rate] [indel_prob]."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import se_batch
from mapper_amd import api, synth
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150000
sub = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
ind = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("r", ref)], mode="mapper", max_query_length=1000)
reads = synth.synthetic_single_end(ref, n, read_len=1000, sub_rate=sub, indel_prob=ind)[0]
b = se_batch(reads)
for rep in range(2):
    r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
names = ["TOTAL", "PYRAMID", "WALK", "HITS", "STRAIGHT", "ANALYZE", "PATH", "PATH_INIT", "BLOCK", "MATCHER_INDEX", "CONFIDENT", "OUTER", "PA_LOOK", "PA_LOAD", "PA_COMPUTE", "PA_PUT"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "n", n, "sub", sub, "indel", ind, "kernel ms %.1f" % r.kernel_ms, "launches", r.kernel_launches, "reruns", r.counters[11], "PA calls/nodes", list(r.counters[5:7]), "cands", r.counters[4])
print({k: round(x / 1e6, 1) for k, x in zip(names, r.prof)})
