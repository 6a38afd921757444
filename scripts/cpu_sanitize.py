"""CPU check: the kernel sources compiled for the host with AddressSanitizer + UBSan (GPU sanitizers are not available on the pool) and driven
through the reference's KATs, single-end / paired / ambiguous / long-read batches.  Run with scripts/cpu_sanitize.sh."""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np
import hostsim_lib as hs
hs.OUT = os.environ['XM_SANITIZED_LIB']; hs.build = lambda: hs.OUT
import oracle_lib as o
from helpers import se_batch, pe_batch, sprinkle_ambiguity, ambiguous_reference, KAT
from mapper_amd import synth
# KATs
for c in KAT["align_cases"]:
    S = hs.SimReference([("reference-0", c["reference"])], mode="api")
    S.align(o.QueryBatch([(c["mates"], c["expectedInner"], c["deviation"])]), o.make_params(c["params"]))
print("kats ok")
ref = synth.synthetic_reference(300_000, seed=41)
S = hs.SimReference([("r", ref)])
reads = synth.synthetic_single_end(ref, 3000, seed=42, indel_prob=0.3)[0]
S.align(se_batch(reads), o.make_params()); print("se ok")
S.align(se_batch(sprinkle_ambiguity(reads)), o.make_params()); print("se ambiguous ok")
m1, m2 = synth.synthetic_paired_end(ref, 800, seed=43)[:2]
S.align(pe_batch(m1, m2, 100.0, 50.0), o.make_params()); print("pe ok")
aref = ambiguous_reference(100_000, seed=0xA3C, n_runs=30, n_codes=300)
S2 = hs.SimReference([("amb", aref)])
S2.align(se_batch(synth.synthetic_single_end(aref, 1500, seed=51)[0]), o.make_params()); print("ambiguous ref ok")
nref = ref[:120_000].copy(); nref[30_000:36_000] = 15; nref[90_000:90_400] = 15; nref[:50] = 15  # runs of N (one longer than XM_BUILD_SPLICE_MIN when that is set)
S3 = hs.SimReference([("nruns", nref)])
S3.align(se_batch(synth.synthetic_single_end(nref[40_000:88_000], 500, seed=52)[0]), o.make_params()); print("N-run ref ok")
long_reads = synth.synthetic_single_end(ref, 100, read_len=1000, sub_rate=0.05, indel_prob=0.9, seed=77)[0]
S.align(se_batch(long_reads), o.make_params()); print("long ok")
print("all ok")
