"""The oracle against every known-answer test the reference holds for this path (tests/golden/kat_reference.json,
transcribed from /root/reference/src/test/java; see tests/golden/make_kat.py).  CPU only."""
import ctypes as C
import numpy as np
import pytest

import oracle_lib as o
from helpers import KAT, check_align_case, sam_text
from mapper_amd import api


@pytest.mark.parametrize("case", KAT["align_cases"], ids=[c["name"] for c in KAT["align_cases"]])
def test_alignerworker_cases(case):
    """T/AlignerWorker_Test.java (Api.alignOnce)."""
    ref = o.OracleReference([("reference-0", case["reference"])], mode="api")
    s = ref.align([(case["mates"], case["expectedInner"], case["deviation"])], o.make_params(case["params"]))
    comps = api.decode_streams(s.ints, s.dbls, s.int_off, s.dbl_off, 0)
    check_align_case(case, comps, o.encode(case["reference"]))


@pytest.mark.parametrize("case", KAT["local_cases"], ids=[c["name"] for c in KAT["local_cases"]])
def test_local_aligner_cases(case):
    """T/PathAligner_Test.java, T/HashBlockAligner_Test.java: exact aligned texts and penalties."""
    r = o.kat_local_align(case["chain"], case["query"], case["reference"], o.make_params(case["params"]), case["penalty"], case["penalty"])
    assert r is not None
    assert r[0] == case["alignedA"] and r[1] == case["alignedB"]
    if case["exact"]:
        assert r[2] == case["penalty"]
    else:
        assert abs(r[2] - case["penalty"]) <= 0.000001  # tolerance of T/HashBlockAligner_Test.java:76


@pytest.mark.parametrize("case", KAT["symmetry_cases"], ids=[str(len(c["text"])) for c in KAT["symmetry_cases"]])
def test_hash_symmetry(case):
    """T/HashBlock_Test.java: forward / reverse-complement symmetry of hashes, merge flags and gapmers."""
    assert o.lib().xmo_kat_hash_symmetry(case["text"].encode()) == 0


@pytest.mark.parametrize("case", KAT["counting_cases"], ids=[c["name"] for c in KAT["counting_cases"]])
def test_counting_path(case):
    """T/Counting_HashBlockPath_Test.java."""
    out = np.zeros(256, dtype=np.int32)
    n = o.lib().xmo_kat_counting_path(case["query"].encode(), case["reference"].encode(), 0.1, len(case["query"]), out.ctypes.data, 256)
    if "expectNum" in case:
        assert n == case["expectNum"]
    else:
        assert case["expectContainsOffset"] in list(out[:n])


@pytest.mark.parametrize("case", KAT["paths_cases"], ids=[c["name"] for c in KAT["paths_cases"]])
def test_paths_counter(case):
    """T/HashBlockPaths_Counter_Test.java: mate pairing, inner / outer distance."""
    inner = np.zeros(16, dtype=np.int32)
    outer = np.zeros(16, dtype=np.int32)
    n = o.lib().xmo_kat_paths_counter(case["reference"].encode(), case["seq1"].encode(), case["seq2"].encode(), 10, 20, inner.ctypes.data, outer.ctypes.data, 16)
    assert n == case["num"]
    if "inner" in case:
        assert inner[0] == case["inner"] and outer[0] == case["outer"]


def test_database_order_independent():
    """T/HashBlockDatabase_Test.java:14-27: hashing the reference in reverse job order gives the same tables."""
    rng = np.random.default_rng(7)
    texts = ["".join("ACGT"[i] for i in rng.integers(0, 4, n)) for n in (60000, 51000, 700)]
    arr = (C.c_char_p * len(texts))(*[t.encode() for t in texts])
    assert o.lib().xmo_kat_db_order_independent(len(texts), arr, 40) == 0


@pytest.mark.parametrize("case", KAT["basepairs_cases"], ids=[c["a"] + c["b"] for c in KAT["basepairs_cases"]])
def test_basepairs_penalty(case):
    """T/BasepairsTest.java:9-45: mismatch = SNP, A~N = ambiguity, A~(A|C) = ambiguity / 3 (exact)."""
    got = o.lib().xmo_kat_base_penalty(case["a"].encode(), case["b"].encode(), float(case["mutationPenalty"]), float(case["ambiguityPenalty"]))
    assert got == case["penalty"]


@pytest.mark.parametrize("case", KAT["sam_cases"], ids=[c["name"] for c in KAT["sam_cases"]])
def test_sam_records(case):
    """T/SamWriter_Test.java:18-94: oracle alignments + mapper_amd.sam reproduce the five pinned SAM bodies (as line sets)."""
    ref = o.OracleReference([("ref", case["reference"])], custom_dup=(1, 2, 2, 1))  # DuplicationDetector(db, 1, 2, 2, 1)
    mates = [m[1] for m in case["mates"]]
    q = api.Query(*mates, expected_inner_distance=case.get("expectedInner", 0.0), spacing_deviation_per_unit_penalty=case.get("deviation", 1.0),
                  names=[m[0] for m in case["mates"]])
    s = ref.align([(mates, case.get("expectedInner", 0.0), case.get("deviation", 1.0))], o.make_params(KAT["align_cases"][0]["params"]))
    comps = api.decode_streams(s.ints, s.dbls, s.int_off, s.dbl_off, 0)
    got = sam_text(q, comps, ["ref"])
    assert sorted(got.splitlines()) == sorted(case["sam"].splitlines()), got


def test_examples_plumbing():
    """configs[0]: examples/reference.fasta + examples/queries.fasta; the query names state the expected outcome."""
    ex = KAT["examples"]
    contigs = api.sort_reference([(n, o.encode(t)) for n, t in ex["reference"]])
    ref = o.OracleReference(contigs, mode="mapper")
    s = ref.align([([t], 0.0, 1.0) for _, t in ex["queries"]], o.make_params())
    aligned = {}
    for i, (name, text) in enumerate(ex["queries"]):
        comps = api.decode_streams(s.ints, s.dbls, s.int_off, s.dbl_off, i)
        aligned[name] = comps[0]
    assert len(aligned["query6-too-different"]) == 0
    for name in ("query1-matches", "query2-1SNP", "query3-matches", "query4-insertion", "query5-deletion"):
        assert len(aligned[name]) >= 1, name
    assert aligned["query1-matches"][0].penalty == 0.0 and aligned["query3-matches"][0].penalty == 0.0
    assert aligned["query2-1SNP"][0].penalty == 1.0
    # a 12-base query allows penalty 1.2 < one insertion (2.1): the inserted base ends up unaligned + 1 SNP (penalty 1.1)
    assert aligned["query4-insertion"][0].penalty == 1.1
    assert any(b.lengthA == 0 for al in aligned["query5-deletion"] for b in al.components[0].sections)


def _with_ns(text, k):
    """Every way to replace exactly k bases of text by N, in the order of T/MultiHashBlock_Test.java:173-195 (addAmbiguities)."""
    if k < 1:
        return [text]
    if k > len(text):
        return []
    return ["N" + t for t in _with_ns(text[1:], k - 1)] + [text[0] + t for t in _with_ns(text[1:], k)]


def _multi_pairs():
    out = []
    for c in KAT["multi_cases"]["expanding"]:
        for k in range(c["maxNumAmbiguities"] + 1):
            out += [(c["text"], a, False) for a in _with_ns(c["text"], k)]
    out += [(t, a, True) for t, a in KAT["multi_cases"]["into"]]
    return out


def test_multi_hashblock_expansion():
    """T/MultiHashBlock_Test.java:12-77: an ambiguous text offers, among the possibilities of its blocks, the block of every text it can stand
    for - the oracle's literal MultiHashBlock / HashBlock_ParentRow.expand, and the product's reference-side multi blocks (HostIndex::nextLevelMulti,
    which hashes the windows around ambiguous bases for the index), through the host simulation."""
    import hostsim_lib as hs
    pairs = _multi_pairs()
    assert len(pairs) > 600
    for text, ambiguous, must_be_a_block in pairs:
        r = o.lib().xmo_kat_multi_contains(text.encode(), ambiguous.encode())
        assert r == 0 or (r == 2 and not must_be_a_block), (text, ambiguous, r)
        t, a = o.encode(text), o.encode(ambiguous)
        r2 = hs.lib().xmsim_kat_multi_contains(t.ctypes.data, a.ctypes.data, len(t))
        assert r2 == r, (text, ambiguous, r, r2)


@pytest.mark.parametrize("case", KAT["codec_cases"]["cases"], ids=[c["name"] for c in KAT["codec_cases"]["cases"]])
def test_position_codec_round_trip_at_scale(case):
    """T/SequenceDatabase_Test.java:16-42: encodePosition / decodePosition round trips on 16 x 2^30 and 8192 x 2^21 bases (repeating sequences,
    T/RepeatingSequence.java): the oracle's SequenceDatabase, and the product's codec (HostIndex::encodePosition / decode, and decodePosition of
    xm_seed.h as the kernels run it)."""
    import hostsim_lib as hs
    assert o.lib().xmo_kat_position_codec(case["numSequences"], case["sequenceLength"]) == 0
    assert hs.lib().xmsim_kat_position_codec(case["numSequences"], case["sequenceLength"]) == 0


def test_packed_map_with_large_reference():
    """T/PackedMap_Test.java:13-49: buckets keep their positions when the encoded positions need 35 bits."""
    assert o.lib().xmo_kat_packed_map_large() == 0


def test_fixture_generator_reproduces_the_committed_fixture(tmp_path):
    """tests/golden/make_kat.py is the provenance of kat_reference.json: run into a temporary file it writes the committed fixture byte for byte
    (the three families added by hand in round 2 - MultiHashBlock, position codec, PackedMap - are in the script since round 6)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = tmp_path / "kat.json"
    subprocess.check_call([sys.executable, os.path.join(here, "golden", "make_kat.py"), str(out)], stdout=subprocess.DEVNULL)
    assert out.read_bytes() == open(os.path.join(here, "golden", "kat_reference.json"), "rb").read()
