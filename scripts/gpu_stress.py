"""GPU stress run: workloads away from the bench's (paired-end, longer reads, noisier reads, repetitive reference); every result is
compared with the oracle on a sample and the kernel time is printed."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as o
from helpers import se_batch, pe_batch, streams_equal, first_difference
from mapper_amd import api, synth


def check(tag, db, R, b, nq, sample):
    t = time.time()
    r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
    wall = time.time() - t
    sb = o.QueryBatch.from_arrays(b.mate_count[:sample], b.mate_offset[:2 * sample], b.mate_length[:2 * sample], b.codes, b.expected_inner[:sample], b.deviation[:sample])
    want = R.align(sb, o.make_params(), threads=os.cpu_count())
    same = np.array_equal(want.ints, r.ints[:r.int_off[sample]]) and np.array_equal(want.dbls.view(np.int64), r.dbls[:r.dbl_off[sample]].view(np.int64))
    aligned = int(sum(1 for q in range(min(nq, 200000)) if r.ints[r.int_off[q] + 1] > 0))
    print(tag, "nq", nq, "kernel ms %.1f" % r.kernel_ms, "wall s %.2f" % wall, "launches", r.kernel_launches, "us", list(r.counters[12:16]), "reruns", r.counters[11],
          "aligned(first 200k)", aligned, "oracle-identical on %d: %s" % (sample, same), flush=True)
    assert same, tag


ref = synth.synthetic_reference(5_000_000)
db = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=250)
R = o.OracleReference([("ecoli_syn", ref)], mode="mapper")
m1, m2 = synth.synthetic_paired_end(ref, 200_000)[:2]
check("paired 2x150", db, R, pe_batch(m1, m2, 100.0, 50.0), 200_000, 20_000)
reads = synth.synthetic_single_end(ref, 300_000, read_len=250)[0]
check("single 250bp", db, R, se_batch(reads), 300_000, 20_000)
reads = synth.synthetic_single_end(ref, 300_000, sub_rate=0.04, indel_prob=0.3)[0]
check("single noisy (4% subs, 30% indel reads)", db, R, se_batch(reads), 300_000, 20_000)
db.close()
# repetitive reference: 2 kb unit repeated with 1% divergence + unique flanks
rng = np.random.default_rng(5)
unit = synth.synthetic_reference(2000, seed=9)
parts = [synth.synthetic_reference(100_000, seed=10)]
for k in range(60):
    u = unit.copy()
    pos = rng.integers(0, len(u), size=20)
    u[pos] = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=20)]
    parts.append(u)
parts.append(synth.synthetic_reference(100_000, seed=11))
rep = np.concatenate(parts)
db = api.ReferenceDatabase([("rep", rep)], mode="mapper", max_query_length=150)
R = o.OracleReference([("rep", rep)], mode="mapper")
reads = synth.synthetic_single_end(rep, 100_000)[0]
check("repetitive reference", db, R, se_batch(reads), 100_000, 10_000)
db.close()
print("stress ok")
