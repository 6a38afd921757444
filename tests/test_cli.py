"""Command-line harness (mapper_amd/cli.py): flag parsing and parameter derivation against Mapper.main (Mapper.java:82-453), the
FASTA/FASTQ readers, and — GPU tier — config 1 of BASELINE.json (the reference's examples/ data, `examples/test.sh:14`)."""
import gzip
import io
import os

import pytest

from mapper_amd import cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "tests", "golden", "examples")


def test_default_parameters_are_mapper_mains():
    o = cli.parse_args(["--reference", "r.fa", "--queries", "q.fa", "--out-sam", "o.sam"])
    p = cli.derive_parameters(o)
    # Mapper.java:27-34 + :409-453: snp 1, indel start 1.5, extension 0.5, max penalty 0.1, span = snp/2, ambiguity = max penalty,
    # additional insertion extension = ambiguity
    assert (p.MutationPenalty, p.DeletionStart_Penalty, p.DeletionExtension_Penalty, p.InsertionStart_Penalty) == (1.0, 1.5, 0.5, 1.5)
    assert p.InsertionExtension_Penalty == 0.5 + 0.1 and p.MaxErrorRate == 0.1 and p.AmbiguityPenalty == 0.1 and p.UnalignedPenalty == 0.1
    assert p.Max_PenaltySpan == 0.5 and p.MaxNumMatches == 2**31 - 1


def test_flags_feed_the_dependent_defaults():
    o = cli.parse_args(["--reference", "r.fa", "--queries", "q.fa", "--no-output", "--snp-penalty", "2", "--max-penalty", "0.2", "--new-indel-penalty", "3",
                        "--extend-indel-penalty", "0.25", "--max-num-matches", "5", "--no-gapmers", "--num-threads", "7", "-v"])
    p = cli.derive_parameters(o)
    assert p.MutationPenalty == 2.0 and p.Max_PenaltySpan == 1.0 and p.AmbiguityPenalty == 0.2 and p.UnalignedPenalty == 0.2
    assert p.InsertionExtension_Penalty == 0.25 + 0.2 and p.InsertionStart_Penalty == 3.0 and p.MaxNumMatches == 5
    assert o["enable_gapmers"] is False


def test_paired_queries_and_spacing():
    o = cli.parse_args(["--reference", "r.fa", "--paired-queries", "a.fq", "b.fq", "--spacing", "250", "40", "--paired-queries", "c.fq", "d.fq", "--out-sam", "-"])
    assert o["paired"] == [("a.fq", "b.fq", 250.0, 40.0), ("c.fq", "d.fq", 100.0, 50.0)] and o["paired_without_spacing"]


def test_split_queries(tmp_path):
    fa = tmp_path / "long.fasta"
    fa.write_text(">r1\n" + "ACGT" * 6 + "A\n>r2\nACGTA\n")  # 25 and 5 bases
    o = cli.parse_args(["--reference", "r.fa", "--split-queries-past-size", "10", "--queries", str(fa), "--no-output"])
    qs = cli.load_queries(o)
    assert [len(q.sequences[0]) for q, _ in qs] == [8, 8, 9, 5]  # SequenceSplitter: (25-1)//10+1 = 3 sections at 25*k//3
    with pytest.raises(cli.UsageError):
        cli.parse_args(["--reference", "r.fa", "--queries", str(fa), "--split-queries-past-size", "10"])


@pytest.mark.parametrize("argv,needle", [
    (["--queries", "q.fa", "--out-sam", "o"], "--reference is required"),
    (["--reference", "r.fa", "--out-sam", "o"], "--queries or --paired-queries is required"),
    (["--reference", "r.fa", "--queries", "q.fa"], "No output specified"),
    (["--reference", "r.fa", "--queries", "q.fa", "--out-sam", "o", "--spacing", "1", "2"], "--spacing is not a top-level argument"),
    (["--reference", "r.fa", "--queries", "q.fa", "--out-sam", "o", "--frobnicate"], "Unrecognized argument: --frobnicate"),
    (["--reference", "r.fa", "--queries", "q.fa", "--out-vcf", "o.vcf"], "handled by the Java host"),
    (["--reference", "r.fa", "--queries", "q.fa", "--out-sam", "o", "--extend-indel-penalty", "0"], "--extend-indel-penalty must be > 0"),
    (["--reference", "r.fa", "--queries", "q.fa", "--out-sam", "o", "--batch-size", "0"], "--batch-size must be >= 1"),
    (["--reference", "r.fa", "--paired-queries", "a", "b", "--out-sam", "o", "--snp-penalty", "2", "--max-penalty", "0.3"], "specify --spacing explicitly"),
])
def test_usage_errors(argv, needle):
    with pytest.raises(cli.UsageError) as e:
        cli.derive_parameters(cli.parse_args(argv))
    assert needle in str(e.value)


def test_readers(tmp_path):
    fa = tmp_path / "a.fasta"
    fa.write_text(">c1 some description\nACGT\nacgtn\n\n>c2\nTTTT\n")
    assert cli.read_sequences(str(fa)) == [("c1", "ACGTACGTN", None), ("c2", "TTTT", None)]
    fq = tmp_path / "a.fastq.gz"
    with gzip.open(fq, "wt") as f:
        f.write("@r1/1 x\nACGT\n+\nIIII\n@r2\nGG\n+r2\n#I\n")
    assert cli.read_sequences(str(fq)) == [("r1/1", "ACGT", "IIII"), ("r2", "GG", "#I")]
    assert cli.sam_header([("c1", "ACGTACGTN"), ("c2", "TTTT")])[:3] == ["@HD\tVN:1.6\tSO:unsorted", "@SQ\tSN:c1\tLN:9", "@SQ\tSN:c2\tLN:4"]


@pytest.mark.gpu
def test_examples_through_the_command_line(tmp_path):
    """BASELINE.json configs[0] / SURVEY.md section 8d config 1: the reference's examples, flags as examples/test.sh:14 (the outputs of this
    path): five of the six queries align as their names say, query6-too-different is written to --out-unaligned."""
    sam_path, un_path = str(tmp_path / "out.sam"), str(tmp_path / "unaligned.fasta")
    log = io.StringIO()
    rc = cli.run(["--reference", os.path.join(EX, "reference.fasta"), "--queries", os.path.join(EX, "queries.fasta"), "--out-sam", sam_path,
                  "--out-unaligned", un_path], out=log)
    assert rc == 0
    lines = open(sam_path).read().splitlines()
    body = [l.split("\t") for l in lines if not l.startswith("@")]
    assert [l for l in lines if l.startswith("@SQ")] == ["@SQ\tSN:contig1\tLN:19", "@SQ\tSN:contig2\tLN:6", "@SQ\tSN:contig3\tLN:40"]
    by_name = {}
    for f in body:
        by_name.setdefault(f[0], []).append(f)
    assert sorted(by_name) == ["query1-matches", "query2-1SNP", "query3-matches", "query4-insertion", "query5-deletion"]
    assert by_name["query1-matches"][0][2:6] == ["contig1", "1", "255", "11M"]
    assert by_name["query2-1SNP"][0][2:6] == ["contig1", "1", "255", "11M"] and by_name["query2-1SNP"][0][-1] == "AS:f:1.0"
    assert {(f[2], f[5]) for f in by_name["query3-matches"]} >= {("contig2", "6M")}
    assert by_name["query5-deletion"][0][2] == "contig3" and "D" in by_name["query5-deletion"][0][5]
    assert open(un_path).read() == ">query6-too-different\nACGCGCTAAACCGAGG\n"
    assert " Alignment rate                : 83% of queries (5/6)" in log.getvalue()
    # --cache-dir (Mapper.java:264): the second run reads the hashed reference back and writes the same SAM
    sam2 = str(tmp_path / "again.sam")
    for _ in range(2):
        assert cli.run(["--reference", os.path.join(EX, "reference.fasta"), "--queries", os.path.join(EX, "queries.fasta"), "--out-sam", sam2,
                        "--cache-dir", str(tmp_path / "cache")], out=io.StringIO()) == 0
        assert open(sam2).read().splitlines() == lines
    assert [f for _, _, fs in os.walk(tmp_path / "cache") for f in fs if f.endswith(".xmidx")] == ["index.xmidx"]
    # --batch-size: the queries in batches of two, streamed (upload of the next batch during the alignment of the current one): same output
    sam3, un3 = str(tmp_path / "batched.sam"), str(tmp_path / "batched_unaligned.fasta")
    assert cli.run(["--reference", os.path.join(EX, "reference.fasta"), "--queries", os.path.join(EX, "queries.fasta"), "--out-sam", sam3,
                    "--out-unaligned", un3, "--batch-size", "2"], out=io.StringIO()) == 0
    assert open(sam3).read().splitlines() == lines and open(un3).read() == open(un_path).read()


@pytest.mark.gpu
def test_streaming_harness_equals_the_per_object_harness(tmp_path):
    """The harness without an object per read (native reader and SAM formatter, batches streamed through several contexts: cli.run_streaming) writes the
    files of the per-object harness (--per-object: api.Query, sam.records - the reference for the formats) byte for byte: single reads from FASTQ with
    unaligned queries kept, pairs with --spacing, long reads cut by --split-queries-past-size; batches smaller than the job, two contexts."""
    import numpy as np
    from mapper_amd import api, synth
    ref = synth.synthetic_reference(200_000, seed=0xEC011)
    dec = np.frombuffer(b"?ACMGRSVTWYHKDBN", dtype=np.uint8)
    (tmp_path / "ref.fa").write_text(">chrA\n%s\n>chrB\n%s\n" % (dec[ref[:120_000]].tobytes().decode(), dec[ref[120_000:]].tobytes().decode()))
    reads = synth.synthetic_single_end(ref[:120_000], 5000, seed=5)[0]
    junk = synth.synthetic_single_end(synth.synthetic_reference(50_000, seed=99), 300, seed=6)[0]   # reads from elsewhere: unaligned
    every = np.concatenate([reads, junk])
    (tmp_path / "se.fq").write_text("".join("@r%d\n%s\n+\n%s\n" % (i, dec[r].tobytes().decode(), "I" * len(r)) for i, r in enumerate(every)))
    m1, m2 = synth.synthetic_paired_end(ref[120_000:], 3000, seed=7)[:2]
    (tmp_path / "p1.fq").write_text("".join("@p%d/1\n%s\n+\n%s\n" % (i, dec[r].tobytes().decode(), "F" * len(r)) for i, r in enumerate(m1)))
    (tmp_path / "p2.fq").write_text("".join("@p%d/2\n%s\n+\n%s\n" % (i, dec[r].tobytes().decode(), "F" * len(r)) for i, r in enumerate(m2)))
    long_reads = synth.synthetic_single_end(ref[:120_000], 40, read_len=2600, seed=8)[0]
    (tmp_path / "long.fa").write_text("".join(">L%d\n%s\n" % (i, dec[r].tobytes().decode()) for i, r in enumerate(long_reads)))
    jobs = {"single": ["--queries", str(tmp_path / "se.fq")],
            "paired": ["--paired-queries", str(tmp_path / "p1.fq"), str(tmp_path / "p2.fq"), "--spacing", "100", "50"],
            "split": ["--split-queries-past-size", "1000", "--queries", str(tmp_path / "long.fa")]}
    for name, args in jobs.items():
        outs = {}
        for mode in ("stream", "object"):
            sam_path, un_path, log = str(tmp_path / (name + mode + ".sam")), str(tmp_path / (name + mode + ".un")), io.StringIO()
            argv = ["--reference", str(tmp_path / "ref.fa")] + args + ["--out-sam", sam_path, "--out-unaligned", un_path, "--batch-size", "1024", "--contexts", "2"]
            assert cli.run(argv + (["--per-object"] if mode == "object" else []), out=log) == 0
            outs[mode] = (open(sam_path).read(), open(un_path).read(), log.getvalue())
        assert outs["stream"][0] == outs["object"][0], name
        assert outs["stream"][1] == outs["object"][1], name
        assert outs["stream"][2] == outs["object"][2], name   # the statistics lines
        assert outs["stream"][0].count("\n") > {"single": 4000, "paired": 4000, "split": 100}[name] and (name != "single" or outs["stream"][1].count("@r") >= 250)
