"""the pile-up of alignments on the reference (device) and the mutations file (host), against what the
reference's own tests pin - tests/golden/mutations_reference.json transcribes src/test/java/MutationsWriter_Test.java:18-134 and
src/test/java/MatchDatabase_Test.java:12-69 (inputs and expected values)."""
import io
import json
import os
import numpy as np
import pytest

from mapper_amd import api, pileup, synth, multi

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "mutations_reference.json")))


def build(query_texts, reference, params=None, query_end_fraction=0.0):
    db = api.ReferenceDatabase([("ref", reference)], mode="api")
    queries = [api.Query(*[api.encode(t) for t in qt]) if isinstance(qt, (list, tuple)) else api.Query(api.encode(qt)) for qt in query_texts]
    r = db.align_batch(queries, api.AlignmentParameters(**(params or FIX["alignment_parameters"])))
    m = pileup.MatchDatabase(db, query_end_fraction)
    m.add_last(queries)
    return db, m, r


@pytest.mark.gpu
@pytest.mark.parametrize("case", FIX["mutation_cases"], ids=[c["name"] for c in FIX["mutation_cases"]])
def test_mutations_writer_cases(case):
    db, m, _ = build([case["query"]], case["reference"], query_end_fraction=case.get("query_end_fraction", 0.0))
    out = io.StringIO()
    m.write_mutations(out, pileup.MutationDetectionParameters(**case.get("filter", {})))
    lines = [l for l in out.getvalue().split("\n") if l and not l.startswith("#") and not l.startswith("CHR")]  # (withoutMetadataLines, MutationsWriter_Test.java:144-154)
    assert lines == case["expected"]
    m.close(); db.close()


@pytest.mark.gpu
def test_match_database_counts():
    c = FIX["match_database_cases"][0]
    db, m, _ = build([c["query"]], c["reference"], dict(FIX["alignment_parameters"], MaxErrorRate=0.5))
    assert np.array_equal(m.depth(0), np.ones(len(c["reference"])))
    m.close(); db.close()
    # overlapping mates: mate 2 is given as sequenced (reverse complement), the pair overlaps on reference positions 3..7
    c = FIX["match_database_cases"][1]
    ref = c["reference"] * 1
    q1, q2 = c["query1"], api.decode(api.reverse_complement(api.encode(c["query2"])))
    db = api.ReferenceDatabase([("ref", ref)], mode="api")
    q = api.Query(api.encode(q1), api.encode(q2), expected_inner_distance=0.0, spacing_deviation_per_unit_penalty=1.0)
    r = db.align_batch([q], api.AlignmentParameters(**dict(FIX["alignment_parameters"], MaxErrorRate=1.0)))
    comps = r.query_alignments(0)
    # the aligner returns the pair as the reference's test builds it (MatchDatabase_Test.java:47-54): one alignment of two sequences at 0 and 3
    assert len(comps) == 1 and len(comps[0]) == 1 and [sa.start_index_b() for sa in comps[0][0].components] == [c["start1"], c["start2"]]
    m = pileup.MatchDatabase(db)
    m.add_last([q])
    assert np.array_equal(m.depth(0), np.ones(len(ref)))
    m.close()
    db.close()


@pytest.mark.gpu
def test_query_ends_and_thresholds_on_a_synthetic_batch():
    """--distinguish-query-ends 0.1 with the reference's default thresholds (Mapper.java:76,534-542) on a deep synthetic pile-up: the middle depth is
    the depth minus what the first and last tenth of every read contribute (recounted on the host), indels near read ends do not count, and the
    thresholds only ever remove lines."""
    ref = synth.synthetic_reference(20_000, seed=91)
    reads = synth.synthetic_single_end(ref, 3000, seed=92, indel_prob=0.3)[0]
    queries = [api.Query(r) for r in reads]
    db = api.ReferenceDatabase([("r", ref)])
    res = db.align_batch(queries, api.AlignmentParameters())
    m0 = pileup.MatchDatabase(db)
    m0.add_last(queries)
    m1 = pileup.MatchDatabase(db, 0.1)
    m1.add_last(queries)
    assert np.array_equal(m0._sum(0)[0], m1._sum(0)[0]) and np.array_equal(m0._middle(0), m0._sum(0)[0])
    mid = np.zeros(len(ref))
    for q in range(len(queries)):
        for comp in res.query_alignments(q):
            for al in comp:
                for sa in al.components:
                    for b in sa.sections:
                        n = len(reads[q])
                        ks = np.arange(b.startA, b.startA + b.lengthA) if b.lengthA == b.lengthB else np.full(b.lengthB, b.startA)
                        inner = ~((ks < 0.1 * n) | (ks >= n - 0.1 * n))
                        np.add.at(mid, b.startB + np.arange(len(ks))[inner], 1.0 / len(comp))
    assert np.allclose(m1._middle(0) / pileup.UNIT, mid, atol=1e-9)
    everything = m1.mutations(pileup.MutationDetectionParameters.emptyFilter())
    # (the defaults - 90 % of a depth of 5 and more - are a variant caller's: sequencing errors of a deep pile-up do not pass them)
    assert m1.mutations(pileup.MutationDetectionParameters.defaultFilter()) == []
    filtered = m1.mutations(pileup.MutationDetectionParameters(5.0, 0.08, 1.0, 0.08, 1.0, 0.08))
    assert 0 < len(filtered) < len(everything) and set((c, p) for c, p, *_ in filtered) <= set((c, p) for c, p, *_ in everything)
    assert len([x for x in everything if "-" in x[2] + x[3]]) < len([x for x in m0.mutations() if "-" in x[2] + x[3]])  # indels near read ends are gone
    m0.close(); m1.close(); db.close()


@pytest.mark.gpu
def test_pileup_on_a_synthetic_batch_and_two_replicas():
    """Depth and substitution counts of a few thousand reads equal a host recount from the decoded alignments; two replicas (batches dealt
    between them) sum to the same pile-up as one."""
    ref = synth.synthetic_reference(150_000, seed=61)
    reads = synth.synthetic_single_end(ref, 4000, seed=62, indel_prob=0.3)[0]
    queries = [api.Query(r) for r in reads]
    params = api.AlignmentParameters()
    db = api.ReferenceDatabase([("r", ref)])
    res = db.align_batch(queries, params)
    m = pileup.MatchDatabase(db)
    n_events = m.add_last(queries)
    depth = np.zeros(len(ref))
    alt = np.zeros((4, len(ref)))
    events = 0
    for q in range(len(queries)):
        comps = res.query_alignments(q)
        for comp in comps:
            for al in comp:
                w = 1.0 / len(comp)
                for sa in al.components:
                    qq = api.reverse_complement(reads[q]) if sa.reference_reversed else reads[q]
                    for b in sa.sections:
                        if b.lengthA == b.lengthB:
                            depth[b.startB:b.startB + b.lengthB] += w
                            for i in range(b.lengthA):
                                if qq[b.startA + i] != ref[b.startB + i]:
                                    alt[{1: 0, 2: 1, 4: 2, 8: 3}[int(qq[b.startA + i])], b.startB + i] += w
                        else:
                            events += 1
                            if b.lengthA == 0:
                                depth[b.startB:b.startB + b.lengthB] += w
    assert n_events == events and events > 500
    got_depth, got_alt = m._sum(0)
    assert np.allclose(got_depth / pileup.UNIT, depth, atol=1e-9) and np.allclose(got_alt / pileup.UNIT, alt, atol=1e-9)
    muts = m.mutations()
    assert len(muts) > 1000 and all(c == 0 and 1 <= p <= len(ref) for c, p, *_ in muts)
    # the same through two replicas
    two = multi.MultiGpuDatabase([("r", ref)], [0, 0])
    m2 = pileup.MatchDatabase(two.replicas)
    half = len(queries) // 2
    for rep, qs in ((0, queries[:half]), (1, queries[half:])):
        two.replicas[rep].align_batch(qs, params)
        m2.add_last(qs, replica=rep)
    d2, a2 = m2._sum(0)
    assert np.array_equal(d2, got_depth) and np.array_equal(a2, got_alt)
    assert m2.mutations() == muts
    m.close(); m2.close(); db.close(); two.close()


@pytest.mark.gpu
def test_cli_out_mutations(tmp_path):
    """`--out-mutations` (Mapper.java:187,758-785) through the command line, one GPU and two contexts, small batches: the same file."""
    from mapper_amd import cli
    ref = synth.synthetic_reference(80_000, seed=71)
    reads = synth.synthetic_single_end(ref, 900, seed=72, indel_prob=0.3)[0]
    with open(tmp_path / "ref.fasta", "w") as f:
        f.write(">chrSyn\n" + api.decode(ref) + "\n")
    with open(tmp_path / "reads.fastq", "w") as f:
        for i, r in enumerate(reads):
            f.write("@r%d\n%s\n+\n%s\n" % (i, api.decode(r), "I" * len(r)))
    outs = []
    for extra in ([], ["--devices", "0,0", "--batch-size", "128"]):
        path = tmp_path / ("mut%d.txt" % len(outs))
        assert cli.run(["--reference", str(tmp_path / "ref.fasta"), "--queries", str(tmp_path / "reads.fastq"), "--distinguish-query-ends", "0", "--out-mutations", str(path),
                        "--snp-threshold", "0", "0", "--indel-threshold", "0", "0"] + extra, out=io.StringIO()) == 0
        outs.append(open(path).read())
    assert outs[0] == outs[1]
    body = [l.split("\t") for l in outs[0].split("\n") if l and not l.startswith("#") and not l.startswith("CHR")]
    assert len(body) > 500 and all(l[0] == "chrSyn" and len(l) == 6 for l in body)
    assert any(set(l[2]) == {"-"} for l in body) and any(set(l[3]) == {"-"} for l in body)  # insertions and deletions are there


@pytest.mark.gpu
def test_cli_out_refs_map_count(tmp_path):
    """`--out-refs-map-count` (Mapper.java:197,747-756): reads from two contigs are counted under the contig they map to ([unpinned] file format)."""
    from mapper_amd import cli
    a, b = synth.synthetic_reference(60_000, seed=81), synth.synthetic_reference(40_000, seed=82)
    ra, rb = synth.synthetic_single_end(a, 300, seed=83)[0], synth.synthetic_single_end(b, 200, seed=84)[0]
    with open(tmp_path / "ref.fasta", "w") as f:
        f.write(">chrA\n" + api.decode(a) + "\n>chrB\n" + api.decode(b) + "\n")
    with open(tmp_path / "reads.fastq", "w") as f:
        for i, r in enumerate(list(ra) + list(rb)):
            f.write("@r%d\n%s\n+\n%s\n" % (i, api.decode(r), "I" * len(r)))
    assert cli.run(["--reference", str(tmp_path / "ref.fasta"), "--queries", str(tmp_path / "reads.fastq"), "--out-refs-map-count", str(tmp_path / "counts.txt")], out=io.StringIO()) == 0
    counts = dict(l.split("\t") for l in open(tmp_path / "counts.txt").read().split("\n") if l)
    assert int(counts["chrA"]) >= 295 and int(counts["chrB"]) >= 195 and sum(int(v) for v in counts.values()) <= 500
