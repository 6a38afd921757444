"""(CPU) differential fuzz of the rejection filter inside the whole chain: batches of long reads with random lengths, error rates and prices through the host simulation of the
kernel sources (tests/hostsim: the device code compiled for the host, the filter on as in the gapped passes of long reads) against the oracle with its observer of the same
bound on.  Per batch: result streams bit for bit; searches examined / rejected equal the observer's; PathAligner calls unchanged; nodes put + the reference's nodes in rejected
searches = the reference's nodes.  The observer raises if a search the bound rejects ever returns an alignment.   usage: cpu_filter_fuzz.py [batches] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as o, hostsim_lib as hs
from helpers import streams_equal, first_difference
from mapper_amd import synth


def batch(rng, ref, n):
    qs = []
    for _ in range(n):
        L = int(rng.choice([330, 500, 800, 1000, 1000, 1500]))
        sub, ind = [(0.01, 0.1), (0.03, 0.3), (0.05, 0.5), (0.08, 0.8), (0.12, 0.95)][int(rng.integers(0, 5))]
        r = synth.synthetic_single_end(ref, 1, read_len=L, seed=int(rng.integers(1, 2**31)), sub_rate=sub, indel_prob=ind)[0][0]
        if rng.random() < 0.15:
            r = r.copy(); r[rng.integers(0, L, int(rng.integers(1, 8)))] = 15
        qs.append(([r], 0.0, 1.0))
    return o.QueryBatch(qs)


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rng = np.random.default_rng(seed)
    t0 = time.time()
    tot = dict(calls=0, nodes=0, null=0, rejects=0, reject_nodes=0)
    for k in range(n_batches):
        ref = synth.synthetic_reference(int(rng.choice([60_000, 200_000, 700_000])), seed=int(rng.integers(1, 2**31)))
        if rng.random() < 0.3:   # a repeat: more candidates per read
            ref = np.concatenate([ref, ref[1000:9000]])
        prm = {}
        if rng.random() < 0.4:
            prm = dict(MutationPenalty=float(rng.choice([1.0, 0.8, 1.5])), InsertionStart_Penalty=float(rng.choice([1.5, 1.0, 2.2])), InsertionExtension_Penalty=float(rng.choice([0.6, 0.35, 0.77])),
                       DeletionStart_Penalty=float(rng.choice([1.5, 1.1])), DeletionExtension_Penalty=float(rng.choice([0.5, 0.3])), MaxErrorRate=float(rng.choice([0.1, 0.07, 0.15])))
        p = o.make_params(prm)
        b = batch(rng, ref, int(rng.integers(6, 14)))
        R, S = o.OracleReference([("r", ref)]), hs.SimReference([("r", ref)])
        with o.observe_bound():
            want = R.align(b, p, threads=os.cpu_count())
        got = S.align(b, p)
        assert streams_equal(want, got), (k, first_difference(want, got, b.nq))
        calls, nodes, null, rejects, reject_nodes, checks = want.counters[6], want.counters[7], want.counters[9], want.counters[11], want.counters[12], want.counters[13]
        assert got.extra[3] == 1 and (got.extra[0], got.extra[1]) == (checks, rejects), (k, got.extra[:4], checks, rejects)
        assert got.counters[5] == calls and got.counters[6] + reject_nodes == nodes, (k, got.counters[5:7], calls, nodes, reject_nodes)
        for key, v in zip(tot, (calls, nodes, null, rejects, reject_nodes)):
            tot[key] += v
        if (k + 1) % 20 == 0:
            print("batch %d: all identical so far; searches %d, returning null %d, rejected by the filter %d; nodes %d, in rejected searches %d; %.0f s" % (
                k + 1, tot["calls"], tot["null"], tot["rejects"], tot["nodes"], tot["reject_nodes"], time.time() - t0), flush=True)
    print("%d batches: result streams and filter counters identical to the oracle's (observer never raised)" % n_batches)


if __name__ == "__main__":
    main()
