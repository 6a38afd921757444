"""(CPU) differential fuzz of the rejection filter inside the whole chain: batches of long reads with random lengths, error rates and prices through the host simulation of the
kernel sources (tests/hostsim: the device code compiled for the host, the filter on as in the gapped passes of long reads) against the oracle with its observer of the same
bound on.  Per batch: result streams bit for bit; pieces and searches examined / rejected equal the observer's; PathAligner calls and nodes = the reference's minus what it
spent in rejected searches and inside rejected pieces (tests/helpers.py filter_counters).  The observer raises if a search the bound rejects ever returns an alignment.   usage: cpu_filter_fuzz.py [batches] [seed] [gpu]
Third argument "gpu": the product on the GPU instead of the host simulation, with batches of thousands of reads (several reads per wave, eight lanes per read: the forms of the
recurrence and of the matcher tables that only exist on the device) - run on the GPU box (scripts/gpu_r06_filter_fuzz.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as o, hostsim_lib as hs
from helpers import streams_equal, first_difference, filter_counters
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
    on_gpu = len(sys.argv) > 3 and sys.argv[3] == "gpu"
    if on_gpu:
        from mapper_amd import api
    rng = np.random.default_rng(seed)
    t0 = time.time()
    tot = dict(calls=0, nodes=0, pieces=0, piece_rejects=0, rejects=0, skipped_nodes=0)
    for k in range(n_batches):
        ref = synth.synthetic_reference(int(rng.choice([60_000, 200_000, 700_000])), seed=int(rng.integers(1, 2**31)))
        if rng.random() < 0.3:   # a repeat: more candidates per read
            ref = np.concatenate([ref, ref[1000:9000]])
        prm = {}
        if rng.random() < 0.4:
            prm = dict(MutationPenalty=float(rng.choice([1.0, 0.8, 1.5])), InsertionStart_Penalty=float(rng.choice([1.5, 1.0, 2.2])), InsertionExtension_Penalty=float(rng.choice([0.6, 0.35, 0.77])),
                       DeletionStart_Penalty=float(rng.choice([1.5, 1.1])), DeletionExtension_Penalty=float(rng.choice([0.5, 0.3])), MaxErrorRate=float(rng.choice([0.1, 0.07, 0.15])))
        p = o.make_params(prm)
        b = batch(rng, ref, int(rng.integers(3000, 9000)) if on_gpu else int(rng.integers(6, 14)))
        R = o.OracleReference([("r", ref)])
        with o.observe_bound():
            want = R.align(b, p, threads=os.cpu_count())
        if on_gpu:
            db = api.ReferenceDatabase([("r", ref)], max_query_length=1500)
            try:
                got = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters(**prm))
            finally:
                db.close()
        else:
            got = hs.SimReference([("r", ref)]).align(b, p)
        assert streams_equal(want, got), (k, first_difference(want, got, b.nq))
        ok, what = filter_counters(got.counters, got.extra, want.counters)
        assert got.extra[3] == 1 and ok, (k, what)
        ref, ob = what["reference"], what["oracle_observer"]
        for key, v in zip(tot, (ref["path_aligner_calls"], ref["nodes"], ob["pieces_examined"], ob["pieces_rejected"], ob["searches_rejected"], ref["nodes_in_rejected_searches"] + ref["nodes_in_rejected_pieces"])):
            tot[key] += v
        if (k + 1) % (5 if on_gpu else 20) == 0:
            print("batch %d: all identical so far; pieces examined %d, rejected %d; searches of the reference %d, rejected by the filter outside rejected pieces %d; nodes of the reference %d, skipped %d; %.0f s" % (
                k + 1, tot["pieces"], tot["piece_rejects"], tot["calls"], tot["rejects"], tot["nodes"], tot["skipped_nodes"], time.time() - t0), flush=True)
    print("%d batches: result streams and filter counters identical to the oracle's (observer never raised)" % n_batches)


if __name__ == "__main__":
    main()
