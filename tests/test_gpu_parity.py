"""GPU tier: libxmapper_hip.so through its C ABI (mapper_amd.api) against the oracle, the committed golden vectors and
size-independent properties at BASELINE.json's full size.  Run with -m gpu on an MI355X."""
import json
import os
import sys
import numpy as np
import pytest

import oracle_lib as o
from helpers import (filter_counters, KAT, streams_equal, first_difference, se_batch, ragged_se_batch, pe_batch, check_align_case, sam_text, sprinkle_ambiguity, ambiguous_reference,
                     heavy_ambiguity, low_complexity_reads)
from mapper_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def gpu_align(db, batch, params=None):
    r = db.align_arrays(batch.mate_count, batch.mate_offset, batch.mate_length, batch.codes, batch.expected_inner, batch.deviation,
                        params or api.AlignmentParameters())
    return o.Streams(r.ints, r.dbls, r.int_off, r.dbl_off, r.counters), r


def to_api_params(d):
    return api.AlignmentParameters(**{k: v for k, v in d.items()})


@pytest.mark.parametrize("case", KAT["align_cases"], ids=lambda c: c["name"])
def test_reference_kats_on_gpu(case):
    """T/AlignerWorker_Test.java through Api.alignOnce's mirror: expectations of the reference's test + bit-identical to the oracle."""
    db = api.newDatabase(case["reference"])
    b = o.QueryBatch([(case["mates"], case["expectedInner"], case["deviation"])])
    got, _ = gpu_align(db, b, to_api_params(case["params"]))
    want = o.OracleReference([("reference-0", case["reference"])], mode="api").align(b, o.make_params(case["params"]))
    assert streams_equal(got, want), first_difference(got, want, 1)
    check_align_case(case, api.decode_streams(got.ints, got.dbls, got.int_off, got.dbl_off, 0), o.encode(case["reference"]))
    db.close()


@pytest.mark.parametrize("case", KAT["sam_cases"], ids=[c["name"] for c in KAT["sam_cases"]])
def test_sam_bodies_on_gpu(case):
    """T/SamWriter_Test.java:18-94 with the alignments computed on the GPU."""
    db = api.ReferenceDatabase([("ref", case["reference"])], dup=(1, 2, 2, 1))
    mates = [m[1] for m in case["mates"]]
    q = api.Query(*mates, expected_inner_distance=case.get("expectedInner", 0.0), spacing_deviation_per_unit_penalty=case.get("deviation", 1.0),
                  names=[m[0] for m in case["mates"]])
    comps = db.align_batch([q], to_api_params(KAT["align_cases"][0]["params"])).query_alignments(0)
    assert sorted(sam_text(q, comps, ["ref"]).splitlines()) == sorted(case["sam"].splitlines())
    db.close()


def test_golden_digests_and_oracle_parity():
    """Committed golden vectors (tests/golden/synthetic_golden.json, made by the oracle) + live oracle comparison."""
    import hashlib
    from make_synthetic_golden import cases, digest
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "synthetic_golden.json")))
    ref, batches = cases()
    db = api.ReferenceDatabase([("ecoli_syn", ref)])
    R = o.OracleReference([("ecoli_syn", ref)])
    for name, b in batches.items():
        got, _ = gpu_align(db, b)
        assert digest(got) == golden["digests"][name]["sha256"], name
        want = R.align(b, o.make_params(), threads=os.cpu_count())
        assert streams_equal(got, want), first_difference(got, want, b.nq)
    del hashlib
    db.close()


def test_full_size_golden_digests():
    """BASELINE.json configs[1] and configs[2] at full size (the batches bench.py aligns): 1,000,000 single-end reads and 1,000,000 pairs against
    the 5 Mb reference.  The whole result streams equal the oracle's (digests committed in tests/golden/synthetic_golden.json, made once by
    tests/golden/make_synthetic_golden.py), and the device's work counters equal the oracle's (the roofline's byte
    counts come from them)."""
    from make_synthetic_golden import full_cases, digest
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "synthetic_golden.json")))["full_digests"]
    ref, batches = full_cases()
    db = api.ReferenceDatabase([("ecoli_syn", ref)], max_query_length=150)
    for name, make in batches.items():
        got, _ = gpu_align(db, make())
        g = golden[name]
        assert (len(got.ints), len(got.dbls)) == (g["num_ints"], g["num_dbls"]), name
        assert digest(got) == g["sha256"], name
        oc = g["oracle_counters"]
        assert [int(x) for x in got.counters[:8]] == [oc[0], oc[1] + oc[2], oc[2], oc[3], oc[5], oc[6], oc[7], oc[8]], name
    db.close()


def test_config2_at_its_stated_size():
    """BASELINE.json configs[2] as stated: 10,000,000 pairs 2 x 150 bp (--spacing 100 50) against the 5 Mb reference, in ONE batch - the whole result streams
    against the digest of the oracle's (tests/golden/synthetic_golden.json, stated_digests, made by `make_synthetic_golden.py stated`), and the work counters."""
    from make_synthetic_golden import stated_size_cases, digest
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "synthetic_golden.json")))["stated_digests"]
    ref, batches = stated_size_cases()
    db = api.ReferenceDatabase([("ecoli_syn", ref)], max_query_length=150)
    for name, make in batches.items():
        got, _ = gpu_align(db, make())
        g = golden[name]
        assert (len(got.ints), len(got.dbls)) == (g["num_ints"], g["num_dbls"]), name
        assert digest(got) == g["sha256"], name
        oc = g["oracle_counters"]
        assert [int(x) for x in got.counters[:8]] == [oc[0], oc[1] + oc[2], oc[2], oc[3], oc[5], oc[6], oc[7], oc[8]], name
    db.close()


PASS_SHAPES = [
    {"XM_LIGHT_LEVEL": "1"},                                                    # light pass keeps the hash-block analysis
    {"XM_TAPER_PCT": "0", "XM_FULL_LPW": "64"},                                 # no end-of-list taper, full waves in the gapped pass
    {"XM_LIGHT_WAVES": "2", "XM_FULL_WAVES": "1", "XM_SCRATCH_GIB": "2"},       # few lanes: every lane aligns many reads in turn
    {"XM_HEAVY_HINT": "64"},                                                    # gapped pass ordered by the cost hint and dealt out (the default for single reads; here for the pairs too)
    {"XM_HEAVY_HINT": "0"},                                                     # gapped pass in list order for the single reads as well
    {"XM_HEAVY_HINT": "16", "XM_FULL_LPW": "8", "XM_FULL_WAVES": "2"},          # nearly every read of the gapped pass "heavy", few lanes
    {"XM_HANDOVER": "0"},                                                       # gapped pass seeds its reads again (no saved regions)
    {"XM_PAIR_LANES": "0"},                                                     # one lane per read in the gapped pass
    {"XM_HANDOVER": "0", "XM_PAIR_LANES": "0", "XM_FULL_LPW": "64"},            # both off, full waves
    {"XM_GAPPED_TMP_PCT": "25", "XM_SCRATCH_GIB": "1"},                         # small temporaries (HBM-mode searches overflow into the rerun passes), tiny region pool
    {"XM_SEARCH_POOL": "0"},                                                    # HBM-mode searches in the lanes' temporaries (no buffer per wave)
    {"XM_SEARCH_POOL": "0", "XM_GAPPED_TMP_PCT": "20"},                         # ... and too small for them: those reads rerun
    {"XM_REGION_KB": "40", "XM_LIGHT_TMP_KB": "24"},                            # regions and light temporaries too small for anything: every read overflows into the reruns
    {"XM_WAVE": "1"},                                                           # the wave-per-read form first (light tier, chain tiers with inline searches), lane-per-read passes for the rest
    {"XM_WAVE": "1", "XM_WAVE_TIERS": "1"},                                     # its light tier only
    {"XM_WAVE": "1", "XM_WAVE_TIERS": "2"},                                     # light + chain tier (no tier with the largest capacities)
    {"XM_WAVE": "1", "XM_WAVE_INLINE_SEARCH": "0"},                             # every PathAligner search through the reads' memos and the search kernel
]


@pytest.mark.parametrize("env", PASS_SHAPES, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_pass_shapes_do_not_change_results(env, monkeypatch):
    """The pass sequence and launch shapes are scheduling only: every variant must return the oracle's streams bit for bit
    (single-end with indels and a paired batch, so that the gapped chain, PathAligner and the join path all run)."""
    ref = synth.synthetic_reference(300_000, seed=77)
    reads, _, _ = synth.synthetic_single_end(ref, 6000, seed=78, indel_prob=0.3)
    b = se_batch(reads)
    m1, m2 = synth.synthetic_paired_end(ref, 1500, seed=79)[:2]
    pb = pe_batch(m1, m2, 100.0, 50.0)
    R = o.OracleReference([("r", ref)])
    want, wantp = R.align(b, o.make_params()), R.align(pb, o.make_params())
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    db = api.ReferenceDatabase([("r", ref)])
    got, _ = gpu_align(db, b)
    assert streams_equal(got, want), first_difference(got, want, len(reads))
    gotp, _ = gpu_align(db, pb)
    assert streams_equal(gotp, wantp), first_difference(gotp, wantp, 1500)
    db.close()


def test_long_reads_on_gpu():
    """1,000 bp queries (BASELINE.json configs[4] splits its 10 kb reads into these) with long-read-like errors and 400 bp reads: the batch starts
    at a larger scratch scale, the searches outgrow the LDS slot and run in HBM mode; results must still be the oracle's."""
    ref = synth.synthetic_reference(400_000, seed=31)
    R = o.OracleReference([("r", ref)])
    for L, n, sub, ind in ((1000, 300, 0.05, 0.9), (400, 1500, 0.02, 0.3)):
        reads = synth.synthetic_single_end(ref, n, read_len=L, sub_rate=sub, indel_prob=ind, seed=32 + L)[0]
        b = se_batch(reads)
        db = api.ReferenceDatabase([("r", ref)], max_query_length=L)
        got, _ = gpu_align(db, b)
        want = R.align(b, o.make_params())
        assert streams_equal(got, want), first_difference(got, want, n)
        db.close()


@pytest.mark.parametrize("env", [{"XM_FULL_WAVES": "1"}, {"XM_FULL_WAVES": "1", "XM_FULL_LPW": "32"}, {"XM_FULL_WAVES": "1", "XM_FULL_LPW": "16", "XM_PAIR_LANES": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_long_reads_sharing_waves_on_gpu(env, monkeypatch):
    """Enough 1,000 bp queries that the gapped pass puts several of them on every wave (its launch shape for long reads: 8 per wave; here 9 000 reads on
    1 024 waves, and the shapes of the short-read pass beside it): the searches of a wave's reads - HBM mode from the start at this chain scale - and
    the two lanes of a read must not disturb each other."""
    ref = synth.synthetic_reference(400_000, seed=41)
    reads = synth.synthetic_single_end(ref, 9000, read_len=1000, sub_rate=0.02, indel_prob=0.3, seed=42)[0]
    b = se_batch(reads)
    want = o.OracleReference([("r", ref)]).align(b, o.make_params(), threads=os.cpu_count())
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    db = api.ReferenceDatabase([("r", ref)], max_query_length=1000)
    got, _ = gpu_align(db, b)
    assert streams_equal(got, want), first_difference(got, want, len(reads))
    db.close()


@pytest.mark.parametrize("env", [{}, {"XM_CONF_SEED": "0"}, {"XM_CONF_SEED": "300"}], ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_many_read_lengths_one_batch_on_gpu(env, monkeypatch):
    """A batch in which every read has its own length (36 ... 700: several hundred distinct (penalty, length) families for the confidence-length table the
    host keeps for the kernels, xm_capi.hip confPrepare) - with the table seeded as in the product, not seeded at all (every key comes in through the reads that
    missed it and run again), and seeded for a few lengths only; then the same lengths again in a second batch (nothing left to seed).  The oracle's streams."""
    ref = synth.synthetic_reference(300_000, seed=51)
    rng = np.random.default_rng(52)
    lens = rng.permutation(np.arange(36, 700))[:420]
    reads = []
    for i, L in enumerate(lens):
        reads.append(synth.synthetic_single_end(ref, 1, read_len=int(L), sub_rate=0.02, indel_prob=0.3, seed=1000 + i)[0][0])
    b = ragged_se_batch(reads)
    want = o.OracleReference([("r", ref)]).align(b, o.make_params(), threads=os.cpu_count())
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    db = api.ReferenceDatabase([("r", ref)], max_query_length=700)
    for _ in range(2):
        got, _ = gpu_align(db, b)
        assert streams_equal(got, want), first_difference(got, want, len(reads))
    db.close()


def test_random_configurations_on_gpu():
    """Differential fuzz (scripts/gpu_fuzz.py): random references (repeats, ambiguity codes), read lengths 36-301, single / paired mixes, ambiguity in
    reads and random alignment parameters; every batch must equal the oracle bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gpu_fuzz
    assert gpu_fuzz.run(rounds=8, seed=77, max_queries=1500) == 0


def test_random_shapes_on_gpu():
    """Second flavour of the fuzz (gpu_fuzz.run_shapes): several contigs with reads across their ends, a length per read inside one batch (36 ... 450; every third
    round 300 ... 1 600: chains at the long-read scales, searches in the form of xm_wsearch.h), mates of unequal length, pairs and single reads mixed."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gpu_fuzz
    assert gpu_fuzz.run_shapes(rounds=9, seed=99, max_queries=1500) == 0


def test_edge_cases_on_gpu():
    rng = np.random.default_rng(3)
    c0 = synth.synthetic_reference(60_000, seed=11)
    c1 = synth.synthetic_reference(9_000, seed=12)
    rep = np.tile(synth.synthetic_reference(500, seed=13), 8)
    contigs = api.sort_reference([("c0", c0), ("c1", c1), ("rep", rep)])
    queries = []
    for L in (1, 2, 7, 12, 20, 33, 75, 151, 400, 1000):
        s = int(rng.integers(0, len(c0) - L))
        queries.append(([c0[s:s + L].copy()], 0.0, 1.0))
    queries.append(([np.concatenate([c1[-40:], np.array([1, 2, 4, 8] * 10, dtype=np.uint8)])], 0.0, 1.0))
    queries.append(([np.concatenate([np.array([8, 4, 2, 1] * 8, dtype=np.uint8), c1[:60]])], 0.0, 1.0))
    queries.append(([rep[100:250].copy()], 0.0, 1.0))
    queries.append(([np.full(150, 1, dtype=np.uint8)], 0.0, 1.0))
    queries.append(([np.array([1, 2, 4, 8], dtype=np.uint8)[rng.integers(0, 4, 150)]], 0.0, 1.0))
    queries.append(([c0[5000:5150].copy(), api.reverse_complement(c0[5300:5450])], 100.0, 50.0))
    queries.append(([c0[5000:5150].copy(), api.reverse_complement(c1[300:450])], 100.0, 50.0))
    queries.append(([c0[5000:5150].copy(), api.reverse_complement(c0[5100:5250])], 100.0, 50.0))
    b = o.QueryBatch(queries)
    db = api.ReferenceDatabase(contigs)
    R = o.OracleReference(contigs)
    for kw in ({}, {"MaxNumMatches": 3}, {"Max_PenaltySpan": 2.0, "MaxErrorRate": 0.15}):
        got, _ = gpu_align(db, b, api.AlignmentParameters(**kw))
        want = R.align(b, o.make_params(**kw))
        assert streams_equal(got, want), first_difference(got, want, len(queries))
    # empty batch
    empty = o.QueryBatch([])
    r = db.align_arrays(empty.mate_count, empty.mate_offset, empty.mate_length, empty.codes, empty.expected_inner, empty.deviation, api.AlignmentParameters())
    assert len(r.ints) == 0 and list(r.int_off) == [0]
    db.close()


def test_reads_with_ambiguous_bases_on_gpu():
    """Reads with IUPAC ambiguity codes (MultiHashBlock / ConditionalHashBlock / SequenceCondition on the read side): single-end and paired,
    bit-identical to the oracle."""
    ref = synth.synthetic_reference(300_000, seed=41)
    reads = sprinkle_ambiguity(synth.synthetic_single_end(ref, 6000, seed=42)[0])
    m1, m2 = synth.synthetic_paired_end(ref, 1500, seed=43)[:2]
    m1, m2 = sprinkle_ambiguity(m1, 4), sprinkle_ambiguity(m2, 5)
    R = o.OracleReference([("r", ref)])
    db = api.ReferenceDatabase([("r", ref)])
    for b, n in ((se_batch(reads), 6000), (pe_batch(m1, m2, 100.0, 50.0), 1500)):
        got, _ = gpu_align(db, b)
        want = R.align(b, o.make_params())
        assert streams_equal(got, want), first_difference(got, want, n)
    one = db.align_batch([api.Query("ACGTACGTACGTANGTACGTACGTACGTACGTACGT")], api.AlignmentParameters())  # (the first implementation refused this read)
    assert len(one.int_off) == 2
    db.close()


def test_reads_of_mostly_ambiguous_bases_on_gpu():
    """Any number of ambiguous bases per mate (the reference bounds the combinations per block, HashBlock_ParentRow.java:10,109,165, not the bases; rounds 1-4 of this
    product failed the batch above 128): N and IUPAC codes at 1 % ... 100 % of a read's positions, N runs at either end up to the whole read, pairs with one or both
    mates affected or one mate all N, reads shorter than minInterestingSize, homopolymers and short-period reads, a batch in which ordinary reads surround them -
    bit-identical to the oracle; an all-N read comes back unaligned, and nothing fails the batch."""
    ref = synth.synthetic_reference(300_000, seed=41)
    R = o.OracleReference([("r", ref)])
    db = api.ReferenceDatabase([("r", ref)])
    plain = synth.synthetic_single_end(ref, 6000, seed=45)[0]
    reads = heavy_ambiguity(synth.synthetic_single_end(ref, 1600, seed=42)[0])
    m1, m2 = synth.synthetic_paired_end(ref, 800, seed=43)[:2]
    m1h, m2h = heavy_ambiguity(m1, 4), heavy_ambiguity(m2, 5)
    m2one = m2.copy(); m2one[::2] = 15
    short = [r[:int(L)] for r, L in zip(heavy_ambiguity(synth.synthetic_single_end(ref, 320, seed=44)[0], 6), np.tile([1, 2, 3, 5, 8, 11, 13, 20], 40))]
    low = low_complexity_reads(240, 150)
    mixed = np.concatenate([plain[:3000], reads, plain[3000:]])
    batches = [("single", se_batch(reads)), ("among ordinary reads", se_batch(mixed)), ("pairs", pe_batch(m1h, m2h)), ("one all-N mate", pe_batch(m1, m2one)),
               ("short", ragged_se_batch(short)), ("low complexity", se_batch(low)), ("all N", se_batch(np.full((64, 150), 15, np.uint8))),
               ("135 N inside", se_batch(np.concatenate([reads[15:16, :10], np.full((1, 135), 15, np.uint8), reads[15:16, 145:]], axis=1)))]
    for name, b in batches:
        got, _ = gpu_align(db, b)
        want = R.align(b, o.make_params(), threads=os.cpu_count())
        assert streams_equal(got, want), name + ": " + str(first_difference(got, want, b.nq))
        if name == "all N":
            assert all(got.ints[got.int_off[q] + 1] == 0 for q in range(b.nq))
    db.close()


def test_ambiguity_fuzz_on_gpu():
    """Third flavour of the fuzz (gpu_fuzz.run_ambiguity): per read an ambiguous fraction from {0, 1 %, 10 %, 50 %, 90 %, 100 %} as N / two-way / three-way codes, N runs
    at the ends, low-complexity and very short reads, single reads and pairs in one batch, references plain / ambiguous / with repeats."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gpu_fuzz
    assert gpu_fuzz.run_ambiguity(rounds=12, seed=515, max_queries=400) == 0


def test_ambiguous_reference_on_gpu():
    """Reference with N runs and IUPAC codes (multi-block index build on the host, ambiguity penalties and SkipHighAmbiguity on the device),
    plain reads and reads with ambiguity codes of their own: bit-identical to the oracle."""
    ref = ambiguous_reference(300_000, seed=0xA3D, n_runs=80, n_codes=900)
    reads = synth.synthetic_single_end(ref, 8000, seed=52)[0]
    reads[4000:] = sprinkle_ambiguity(reads[4000:], 7)
    b = se_batch(reads)
    db = api.ReferenceDatabase([("amb", ref)])
    got, _ = gpu_align(db, b)
    want = o.OracleReference([("amb", ref)]).align(b, o.make_params(), threads=os.cpu_count())
    assert streams_equal(got, want), first_difference(got, want, len(reads))
    db.close()


def _layout(info):  # what describes the index itself (not where or how fast it was built)
    return {k: v for k, v in info.items() if k not in ("built_on_device", "reserved", "bucket_line_bytes", "hash_seconds", "duplication_seconds")}


def _tables_equal(A, B, what):
    ia, ib = (_layout(x.info()) for x in (A, B))
    assert ia == ib, what
    for L in range(0, ia["max_hashed_length"] + 1):
        ta, tb = A.table(L), B.table(L)
        assert ta["capacity"] == tb["capacity"] and ta["maxCount"] == tb["maxCount"], (what, L)
        assert np.array_equal(ta["counts"], tb["counts"]), (what, L)
        assert np.array_equal(ta["positions"], tb["positions"]), (what, L)
    for c in range(ia["num_contigs"]):
        assert np.array_equal(A.dup_keys(c), B.dup_keys(c)), what


@pytest.mark.gpu
@pytest.mark.parametrize("mode,contigs,ambiguous", [("mapper", [300_000], False), ("mapper", [40_000, 9_000, 300], True), ("api", [6000], False)], ids=["300kb", "contigs+ambiguity", "api"])
def test_index_hashed_on_the_gpu_equals_the_oracle_tables(mode, contigs, ambiguous, monkeypatch):
    """The GPU tier's own link from the device-built index to the ORACLE (round-5 verdict: the comparison with the host builder is product against
    product, and the host builder's comparison with the oracle runs in the CPU tier only): every PackedMap of an index hashed on the GPU - capacity, per-key
    limit, per-bucket counts with their overfull marks, packed positions - and the duplication keys equal the oracle's literal restatement of
    HashBlock_Database / DuplicationDetector (M/HashBlock_Database.java:490-616, M/PackedMap.java:99-153, M/DuplicationDetector.java:97-436), for the lengths
    hashed at build time and for those added by lazy growth (M/Readable_HashBlock_Database.java:108-113)."""
    refs = [("c%d" % i, (ambiguous_reference(n, seed=0xA0 + i) if ambiguous and i == 0 else synth.synthetic_reference(n, seed=0xEC011 + i))) for i, n in enumerate(contigs)]
    refs[0] = (refs[0][0], np.concatenate([refs[0][1], refs[0][1][:900]]))  # a repeat: overfull buckets and duplication keys
    R = o.OracleReference(refs, mode=mode)
    R.align(["ACGTACGTACGTAGCATCGACTAGCAGCATCGAC"], o.make_params())  # triggers prepare()
    monkeypatch.setenv("XM_DEVICE_BUILD", "1")
    D = api.ReferenceDatabase(refs, mode=mode)
    try:
        assert D.info()["built_on_device"] == 1
        mn, mx = R.index_info()
        assert (D.info()["min_interesting_size"], D.info()["max_hashed_length"]) == (mn, mx)

        def same(lengths):
            for L in lengths:
                ta, tb = R.table(L), D.table(L)
                assert ta["capacity"] == tb["capacity"] and ta["maxCount"] == tb["maxCount"], L
                assert np.array_equal(ta["counts"], tb["counts"]), L
                assert np.array_equal(ta["positions"], tb["positions"]), L
        same(range(0, mx + 1))
        for c in range(len(refs)):
            assert np.array_equal(R.dup_keys(c), D.dup_keys(c)), c
        R.require_size(mx + 20)
        D.ensure_length(R.index_info()[1])
        same(range(mx + 1, R.index_info()[1] + 1))
    finally:
        D.close()


@pytest.mark.parametrize("mode,contigs,gapmers,group", [("mapper", [300_000], True, None), ("mapper", [50_000, 31_000, 700, 1], True, "20000"),
                                                         ("api", [4000], True, "1"), ("mapper", [120_000], False, None)])
def test_index_hashed_on_the_gpu_equals_host_builder(mode, contigs, gapmers, group, monkeypatch):
    """the tables hashed, sorted and cut into CSR form on the GPU (xm_index_device.hip) are the host builder's
    (which the CPU tier checks against the oracle's literal HashBlock_Database), bucket by bucket, for every hashed length - also when the
    records are made in several groups of tables, and for the tables added later by xm_index_ensure_length."""
    refs = [("c%d" % i, synth.synthetic_reference(n, seed=0xEC011 + i)) for i, n in enumerate(contigs)]
    refs[0] = (refs[0][0], np.concatenate([refs[0][1], refs[0][1][:900]]))  # a repeat: overfull buckets and duplication keys
    if group:
        monkeypatch.setenv("XM_BUILD_GROUP_RECORDS", group)
    monkeypatch.setenv("XM_DEVICE_BUILD", "1")
    D = api.ReferenceDatabase(refs, mode=mode, enable_gapmers=gapmers)
    monkeypatch.setenv("XM_DEVICE_BUILD", "0")
    H = api.ReferenceDatabase(refs, mode=mode, enable_gapmers=gapmers)
    assert D.info()["built_on_device"] == 1 and H.info()["built_on_device"] == 0
    _tables_equal(D, H, "build")
    grow = D.info()["max_hashed_length"] + 23
    monkeypatch.setenv("XM_DEVICE_BUILD", "1")
    D.ensure_length(grow)
    monkeypatch.setenv("XM_DEVICE_BUILD", "0")
    H.ensure_length(grow)
    _tables_equal(D, H, "grown")
    if mode == "mapper" and gapmers and len(contigs) == 1:
        R = o.OracleReference(refs, mode=mode)
        b = se_batch(synth.synthetic_single_end(refs[0][1], 2000, seed=9, indel_prob=0.3)[0])
        got, _ = gpu_align(D, b)
        want = R.align(b, o.make_params())
        assert streams_equal(got, want), first_difference(got, want, 1)
    D.close(); H.close()


def n_run_reference(n, seed, run=10_000, fraction=0.01, codes=200):
    """A reference the shape of SURVEY.md section 8(d) config 4: runs of N of `run` bases over `fraction` of it, plus a few scattered IUPAC codes."""
    ref = synth.synthetic_reference(n, seed=seed).copy()
    rng = np.random.default_rng(seed)
    for _ in range(max(1, int(n * fraction / run))):
        p0 = int(rng.integers(0, n - run))
        ref[p0:p0 + run] = 15
    ref[rng.integers(0, n, size=codes)] = np.array([5, 10, 3, 12, 6, 9, 7, 11, 13, 14, 15], np.uint8)[rng.integers(0, 11, size=codes)]
    return ref


@pytest.mark.parametrize("case,group", [("scattered", None), ("scattered", "30000"), ("n_runs_50Mb", None), ("long_run", None)])
def test_reference_with_ambiguity_codes_hashed_on_the_gpu_equals_host_builder(case, group, monkeypatch):
    """References with ambiguity codes (GRCh38's N-runs; HashBlock_ParentRow.java:109-165, MultiHashBlock.java): the GPU hashes what lies clear of
    the ambiguous bases, the conditional multi blocks come from the host's windows around them (long runs of N with their middle left out) and join
    the records before the sort.  Tables and duplication keys must be those of the host's whole-contig multi builder (which the CPU tier checks
    against the oracle), also with the records made in several groups; and reads align as the oracle says."""
    mis = -1
    if case == "scattered":
        refs = [("a", ambiguous_reference(120_000, seed=31, n_runs=12, n_codes=150)), ("b", ambiguous_reference(9_000, seed=32)), ("clean", synth.synthetic_reference(30_000, seed=33))]
    elif case == "n_runs_50Mb":
        refs = [("chr%d" % i, n_run_reference(n, seed=40 + i)) for i, n in enumerate((30_000_000, 14_000_000, 6_000_000))]
        mis = 13  # (what a 3 Gb reference gets, HashBlock_Database.java:52)
    else:
        r = synth.synthetic_reference(600_000, seed=34).copy()
        r[100_000:300_000] = 15   # longer than the 65,536 positions a window holds of a run
        r[400_000:400_900] = 15
        refs = [("a", r)]
        mis = 13
    if group:
        monkeypatch.setenv("XM_BUILD_GROUP_RECORDS", group)
    monkeypatch.setenv("XM_DEVICE_BUILD", "1")
    D = api.ReferenceDatabase(refs, max_query_length=150, min_interesting_size=mis)
    monkeypatch.setenv("XM_DEVICE_BUILD", "0")
    H = api.ReferenceDatabase(refs, max_query_length=150, min_interesting_size=mis)
    assert D.info()["built_on_device"] == 1 and H.info()["built_on_device"] == 0
    _tables_equal(D, H, "build")
    if case == "scattered":
        R = o.OracleReference(refs)
        b = se_batch(synth.synthetic_single_end(refs[0][1], 3000, seed=35, indel_prob=0.3)[0])
        got, _ = gpu_align(D, b)
        want = R.align(b, o.make_params(), threads=os.cpu_count())
        assert streams_equal(got, want), first_difference(got, want, 1)
    D.close(); H.close()


def test_index_from_cache_aligns_the_same(tmp_path):
    """--cache-dir: an index read back from its file (and grown for longer reads after the load) aligns exactly like the one that was built."""
    ref = synth.synthetic_reference(200_000, seed=77)
    built = api.ReferenceDatabase([("r", ref)], cache_dir=tmp_path)
    cached = api.ReferenceDatabase([("r", ref)], cache_dir=tmp_path)
    assert not built.cache_hit and cached.cache_hit
    for read_len in (150, 260):  # 260 > the hashed lengths of the file: xm_index_ensure_length after the load
        b = se_batch(synth.synthetic_single_end(ref, 3000, read_len=read_len, seed=78, indel_prob=0.3)[0])
        x, _ = gpu_align(built, b)
        y, _ = gpu_align(cached, b)
        assert streams_equal(x, y), first_difference(x, y, 1)
    want = o.OracleReference([("r", ref)]).align(b, o.make_params())
    assert streams_equal(y, want), first_difference(y, want, 1)
    built.close(); cached.close()


def test_streamed_batches_overlap_upload_and_alignment():
    """xm_batch_stage / xm_batch_commit (align_stream): batches of different shapes streamed through the two buffer sets give exactly the
    results of one xm_align_batch call each, in order; stopping early leaves the database usable."""
    ref = synth.synthetic_reference(300_000, seed=31)
    db = api.ReferenceDatabase([("r", ref)])
    batches = []
    for k, (n, read_len) in enumerate([(4000, 150), (100, 150), (3000, 250), (1, 36), (5000, 150)]):
        batches.append(se_batch(synth.synthetic_single_end(ref, n, read_len=read_len, seed=100 + k, indel_prob=0.3)[0]))
    m1, m2 = synth.synthetic_paired_end(ref, 1500, seed=200)[:2]
    batches.append(pe_batch(m1, m2, 100.0, 50.0))
    arrays = [(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation) for b in batches]
    with pytest.raises(RuntimeError, match="no staged batch"):
        db.commit_staged()
    want = [gpu_align(db, b)[0] for b in batches]
    got = list(db.align_stream(iter(arrays), api.AlignmentParameters()))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        x = o.Streams(g.ints, g.dbls, g.int_off, g.dbl_off, g.counters)
        assert streams_equal(x, w), first_difference(x, w, 1)
    gen = db.align_stream(iter(arrays), api.AlignmentParameters())
    first = next(gen)
    gen.close()
    assert np.array_equal(first.ints, want[0].ints)
    again, _ = gpu_align(db, batches[1])
    assert streams_equal(again, want[1])
    db.close()


def test_seed_probe_matches_host_tables():
    """xm_seed_probe (bulk PackedMap.get on the device) against the bucket contents the oracle holds."""
    ref = synth.synthetic_reference(300_000)
    db = api.ReferenceDatabase([("ecoli_syn", ref)])
    R = o.OracleReference([("ecoli_syn", ref)])
    R.align(["ACGTACGTACGTAGCATCGACTAGCAGCATCGAC"], o.make_params())
    rng = np.random.default_rng(5)
    info = db.info()
    for L in (info["min_interesting_size"], 12, 16, 24, info["max_hashed_length"]):
        t = R.table(L)
        keys = rng.integers(-2**31, 2**31 - 1, 4096).astype(np.int32)
        counts, pos, _ = db.seed_probe(np.full(len(keys), L, np.int32), keys, max_per_probe=8)
        starts = np.concatenate([[0], np.cumsum(np.maximum(t["counts"], 0))])
        for i, k in enumerate(keys):
            bucket = int(k) % t["capacity"]
            c = int(t["counts"][bucket])
            want = -1 if (c < 0 or c > t["maxCount"]) else c
            assert counts[i] == want
            if want > 0:
                assert list(pos[i, :min(want, 8)]) == list(t["positions"][starts[bucket]:starts[bucket] + min(want, 8)])
    db.close()


def test_full_size_properties():
    """BASELINE.json configs[1] at full size (1,000,000 x 150 bp vs 5 Mb): properties that need no oracle run."""
    ref = synth.synthetic_reference(5_000_000)
    reads, starts, strand = synth.synthetic_single_end(ref, 1_000_000)
    db = api.ReferenceDatabase([("ecoli_syn", ref)], max_query_length=150)
    b = se_batch(reads)
    a1, r1 = gpu_align(db, b)
    a2, _ = gpu_align(db, b)
    assert streams_equal(a1, a2), "two runs of the same batch differ (non-determinism)"
    # batch-composition invariance: a read's result does not depend on what else is in the batch
    sub = se_batch(reads[250_000:300_000])
    a3, _ = gpu_align(db, sub)
    lo, hi = a1.int_off[250_000], a1.int_off[300_000]
    assert np.array_equal(a3.ints, a1.ints[lo:hi])
    # every read of this workload maps back to its origin (ungapped offset of the first block, strand)
    io = a1.int_off[:-1]
    nal = a1.ints[io + 1]
    assert (nal >= 1).mean() > 0.995
    one = nal >= 1
    contig_rev = a1.ints[io[one] + 5]
    start_a = a1.ints[io[one] + 7]
    start_b = a1.ints[io[one] + 8]
    assert (np.abs((start_b - start_a) - starts[one]) <= 3).mean() > 0.99
    assert (contig_rev == strand[one]).mean() > 0.995
    # the oracle agrees bit for bit on a slice that finishes in seconds
    R = o.OracleReference([("ecoli_syn", ref)])
    n = 60_000
    want = R.align(se_batch(reads[:n]), o.make_params(), threads=os.cpu_count())
    assert np.array_equal(want.ints, a1.ints[:a1.int_off[n]]) and np.array_equal(want.dbls.view(np.int64), a1.dbls[:a1.dbl_off[n]].view(np.int64))
    db.close()


@pytest.mark.parametrize("generator", ["grch38_shaped_reference", "grch38_repeat_rich_reference"], ids=["iid", "repeat-rich"])
def test_grch38_regime_alignments_equal_oracle(generator):
    """The regime of the walk a 3 Gb reference puts the path in, against the oracle: minInterestingSize = 13 (HashBlock_Database.java:52; passed as the
    constructor argument of :34 on both sides, as the reference itself would derive it from 3.1 G bases) changes which tables exist and how
    HashBlockPath.advanceToNextPosition moves (HashBlockPath.java:143-195).  Reference: the GRCh38-shaped one of SURVEY.md section 8(d) at 1/200 of its
    size (24 contigs, 15 Mb, N-runs of 10 kb; synth.grch38_shaped_reference) - what the oracle can hash; 64-bit position arrays forced as a 3 Gb
    index has them.  150 bp reads, pairs (--spacing 100 50) and 1 kb reads sampled genome-wide: streams bit for bit, work counters equal.
    repeat-rich (round 6): the same shape with interspersed repeat families at 85-95 % identity, segmental duplications and tandem repeats
    (synth.grch38_repeat_rich_reference: what `bench.py --config 3rep` measures at full size) - overfull buckets, duplication windows, many candidates per read."""
    contigs, whole, starts, runs = getattr(synth, generator)(scale=0.005)
    assert len(contigs) == 24 and sum(len(r) for r in runs) >= 10
    os.environ["XM_FORCE_POS64"] = "1"
    try:
        db = api.ReferenceDatabase(contigs, max_query_length=1000, min_interesting_size=13)
    finally:
        del os.environ["XM_FORCE_POS64"]
    info = db.info()
    assert info["min_interesting_size"] == 13 and info["position_bytes"] == 8 and info["built_on_device"] == 1
    R = o.OracleReference(contigs, min_interesting_size=13)
    params, oparams = api.AlignmentParameters(), o.make_params()
    batches = {}
    g, _, _ = synth.genome_wide_starts(starts, runs, 6000, 153, seed=0x13A)
    batches["150 bp"] = se_batch(synth.synthetic_single_end(whole, 6000, seed=0x13B, indel_prob=0.3, at=g)[0])
    g, _, _ = synth.genome_wide_starts(starts, runs, 3000, 900, seed=0x13C)
    m1, m2 = synth.synthetic_paired_end(whole, 3000, seed=0x13D, indel_prob=0.2, at=g)[:2]
    batches["pairs"] = pe_batch(m1, m2, 100.0, 50.0)
    g, _, _ = synth.genome_wide_starts(starts, runs, 400, 1003, seed=0x13E)
    batches["1 kb"] = se_batch(synth.synthetic_single_end(whole, 400, read_len=1000, seed=0x13F, indel_prob=0.3, at=g)[0])
    # reads that touch an N-run: the templates start just in front of runs
    near = np.concatenate([starts[c] + runs[c][:2] - 100 for c in range(24) if len(runs[c])])
    batches["at N-runs"] = se_batch(synth.synthetic_single_end(whole, len(near), seed=0x140, at=near)[0])
    for name, b in batches.items():
        got, raw = gpu_align(db, b, params)
        with o.observe_bound():   # (the 1 kb batch runs with the rejection filter in front of PathAligner: the observer says which searches it skips)
            want = R.align(b, oparams, threads=os.cpu_count())
        assert streams_equal(got, want), name + ": " + str(first_difference(got, want, b.nq))
        wc = [int(x) for x in want.counters[:9]]  # (the oracle counts probes and fetches apart; compared as bench.py compares them)
        skipped_calls = int(want.counters[16]) if raw.extra[3] else 0                            # PathAligner calls inside the pieces the filter proved unalignable
        skipped = int(want.counters[12]) + int(want.counters[17]) if raw.extra[3] else 0          # nodes of rejected searches + nodes inside rejected pieces
        assert (raw.extra[3] == 1) == (name == "1 kb"), name
        if raw.extra[3]:
            ok, what = filter_counters(raw.counters, raw.extra, want.counters)
            assert ok, what
        assert [int(x) for x in got.counters[:8]] == [wc[0], wc[1] + wc[2], wc[2], wc[3], wc[5], wc[6] - skipped_calls, wc[7] - skipped, wc[8]], name
    assert R.index_info()[0] == 13
    db.close()


@pytest.mark.gpu
def test_repeat_rich_reference_equals_oracle():
    """configs[1]'s and configs[2]'s read models on synth.repeat_rich_reference (5 Mb: segmental duplications at 90-99.5 % identity, tandem repeats, a 28-mer
    whose buckets overflow - over 30 % of the positions in a segment present at least twice): 100,000 reads and 50,000 pairs, result streams bit for bit and
    work counters equal to the oracle's.  This is the branch a real genome sends reads into and i.i.d. ACGT does not: no early accept in a duplicated window
    (Readable_DuplicationDetector.java:28-47, AlignerWorker.java:494-587), every candidate enumerated, overfull buckets skipped (HashBlock_Database.java:569-577);
    the duplication map itself is built from these buckets (DuplicationDetector.java:129-250)."""
    st = {}
    ref = synth.repeat_rich_reference(5_000_000, stats=st)
    assert st["fraction_in_repeats"] >= 0.3
    R = o.OracleReference([("rep", ref)])
    reads = synth.synthetic_single_end(ref, 100_000, seed=0x5EED0001)[0]
    b = se_batch(reads)
    m1, m2 = synth.synthetic_paired_end(ref, 50_000, seed=0x5EED0002)[:2]
    pb = pe_batch(m1, m2, 100.0, 50.0)
    want, wantp = R.align(b, o.make_params(), threads=os.cpu_count()), R.align(pb, o.make_params(), threads=os.cpu_count())
    db = api.ReferenceDatabase([("rep", ref)])
    for name, batch, w, n in (("reads", b, want, 100_000), ("pairs", pb, wantp, 50_000)):
        got, _ = gpu_align(db, batch)
        assert streams_equal(got, w), first_difference(got, w, n)
        wc = [int(x) for x in w.counters[:9]]
        assert [int(x) for x in got.counters[:8]] == [wc[0], wc[1] + wc[2], wc[2], wc[3], wc[5], wc[6], wc[7], wc[8]], name
    quick = want.counters[8] / 100_000
    assert quick < 0.85, "quick accepts: %.3f of the reads (i.i.d. reference: 0.94)" % quick
    db.close()
