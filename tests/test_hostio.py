"""libxm_hostio.so (mapper_amd/hostio.py; include/xmapper_hostio.h): the native FASTA / FASTQ reader and SAM formatter of the standalone harness against the
per-object Python path they replace (cli.read_sequences + api.Query, sam.records) - byte for byte - and against the SAM bodies the reference's own
SamWriter_Test pins (T/SamWriter_Test.java:18-94, through the oracle's alignments).  CPU tier: no GPU code is involved."""
import gzip
import os

import numpy as np
import pytest

import oracle_lib as o
from helpers import KAT
from mapper_amd import api, cli, hostio, sam


def _write(path, text, gz=False):
    if gz:
        with gzip.open(path, "wt") as f:
            f.write(text)
    else:
        with open(path, "w", newline="") as f:
            f.write(text)
    return str(path)


def _rand_seq(rng, n, alphabet="ACGT"):
    return "".join(alphabet[i] for i in rng.integers(0, len(alphabet), n))


@pytest.mark.parametrize("gz", [False, True], ids=["plain", "gzip"])
def test_reader_gives_the_arrays_of_the_per_object_path(tmp_path, gz):
    rng = np.random.default_rng(5)
    # FASTQ with lower case, IUPAC codes, an unknown letter, CRLF ends and a header with a description; FASTA with sequences over several lines
    fq = "".join("@r%d some description\n%s\r\n+\n%s\n" % (i, _rand_seq(rng, int(rng.integers(1, 200)), "ACGTNacgtRYKMx"), "I" * 5) for i in range(257))
    fa = ">c1 first\nACGTAC\nGTTA\n\n>c2\n" + _rand_seq(rng, 150) + "\n>c3\nacgu\n"
    for name, text in (("a.fq", fq), ("b.fa", fa)):
        path = _write(tmp_path / (name + (".gz" if gz else "")), text, gz)
        want = cli.read_sequences(path)
        got_names, got_codes, got_quals = [], [], []
        for b in hostio.read_batches(path, batch_size=100, keep_qualities=True):
            raw = b._ptr.contents
            names = bytes((C_ := __import__("ctypes")).string_at(raw.names, int(raw.name_off[2 * len(b)])))
            quals = bytes(C_.string_at(raw.quals, int(raw.qual_off[2 * len(b)])))
            for q in range(len(b)):
                assert b.mate_count[q] == 1 and b.mate_length[2 * q + 1] == 0 and b.expected_inner[q] == 0.0 and b.deviation[q] == 1.0
                got_names.append(names[raw.name_off[2 * q]:raw.name_off[2 * q + 1]].decode())
                got_codes.append(b.codes[b.mate_offset[2 * q]:b.mate_offset[2 * q] + b.mate_length[2 * q]].copy())
                got_quals.append(quals[raw.qual_off[2 * q]:raw.qual_off[2 * q + 1]].decode() if raw.has_qual[2 * q] else None)
        assert got_names == [n for n, _, _ in want]
        assert all(np.array_equal(g, api.encode(t)) for g, (_, t, _) in zip(got_codes, want))
        assert got_quals == [q for _, _, q in want]


def test_reader_pairs_and_sections(tmp_path):
    rng = np.random.default_rng(6)
    r1 = "".join("@p%d/1\n%s\n+\n%s\n" % (i, _rand_seq(rng, 50), "F" * 50) for i in range(33))
    r2 = "".join("@p%d/2\n%s\n+\n%s\n" % (i, _rand_seq(rng, 40), "F" * 40) for i in range(33))
    a, b2 = _write(tmp_path / "r1.fq", r1), _write(tmp_path / "r2.fq", r2)
    n = 0
    for b in hostio.read_batches(a, b2, batch_size=10, expected_inner=100.0, deviation=50.0):
        assert (b.mate_count == 2).all() and (b.expected_inner == 100.0).all() and (b.deviation == 50.0).all()
        assert (b.mate_length[0::2] == 50).all() and (b.mate_length[1::2] == 40).all()
        n += len(b)
    assert n == 33
    _write(tmp_path / "short.fq", "".join("@p%d/2\nACGT\n+\nFFFF\n" % i for i in range(32)))
    with pytest.raises(ValueError, match="different numbers of reads"):
        list(hostio.read_batches(a, str(tmp_path / "short.fq"), batch_size=10))
    # --split-queries-past-size (SequenceSplitter.java:9-38): sections cross batch ends
    fa = _write(tmp_path / "long.fa", ">r1\n" + "ACGT" * 6 + "A\n>r2\nACGTA\n>r3\n" + _rand_seq(rng, 1234) + "\n")
    o_ = cli.parse_args(["--reference", "r.fa", "--split-queries-past-size", "10", "--queries", fa, "--no-output"])
    want = [(q.names[0], q.sequences[0]) for q, _ in cli.load_queries(o_)]
    got = []
    for b in hostio.read_batches(fa, batch_size=7, split=10):
        raw = b._ptr.contents
        names = __import__("ctypes").string_at(raw.names, int(raw.name_off[2 * len(b)]))
        for q in range(len(b)):
            got.append((names[raw.name_off[2 * q]:raw.name_off[2 * q + 1]].decode(), b.codes[b.mate_offset[2 * q]:b.mate_offset[2 * q] + b.mate_length[2 * q]].copy()))
    assert [g[0] for g in got] == [w[0] for w in want] and all(np.array_equal(g[1], w[1]) for g, w in zip(got, want))
    with pytest.raises(ValueError, match="neither a FASTA nor a FASTQ header"):
        list(hostio.read_batches(_write(tmp_path / "bad.fa", "ACGT\n")))


def _random_streams(rng, queries, n_contigs):
    """Result streams (include/xmapper_hip.h layout) with random alignments for `queries` (list of api.Query): unaligned queries, several alignments, paired
    alignments, pairs that fell back to one component per mate, soft clips, indels, both strands."""
    ints, dbls, io, do = [], [], [0], [0]

    def seq_al(L):
        blocks = []
        a = int(rng.integers(0, 4)) if rng.random() < 0.3 else 0
        bpos = int(rng.integers(0, 100000))
        end = L - (int(rng.integers(0, 4)) if rng.random() < 0.3 else 0)
        while a < end:
            kind = rng.random()
            n = int(rng.integers(1, max(2, end - a + 1)))
            if kind < 0.6 or not blocks:
                blocks.append((a, bpos, n, n)); a += n; bpos += n
            elif kind < 0.8:
                n = min(n, 5); blocks.append((a, bpos, n, 0)); a += n
            else:
                n = min(n, 7); blocks.append((a, bpos, 0, n)); bpos += n
        if not blocks:
            blocks.append((0, bpos, max(L, 1), max(L, 1)))
        return [int(rng.integers(0, n_contigs)), int(rng.integers(0, 2)), len(blocks)] + [x for b in blocks for x in b], [float(rng.choice([0.0, 1.0, 2.1, 0.1 * rng.integers(0, 99)])), float(rng.random())]

    for q in queries:
        paired = len(q.sequences) == 2
        mode = rng.random()
        if mode < 0.2:
            ints += [1, 0]
        elif paired and mode < 0.4:   # one component per mate (AlignerWorker.java:602-644)
            ints.append(2)
            for mate in range(2):
                nal = int(rng.integers(0, 3))
                ints.append(nal)
                for _ in range(nal):
                    i_, d_ = seq_al(len(q.sequences[mate]))
                    ints += [0, 1] + i_
                    dbls += [0.0, 1.0, 0.0, float(rng.choice([0.0, 3.5, 12.25, 0.6000000000000001]))] + d_
        else:
            nal = int(rng.integers(1, 4))
            ints += [1, nal]
            for _ in range(nal):
                ints += [int(rng.integers(-5, 300)), 2 if paired else 1]
                dbls += [float(rng.choice([0.0, 0.4, 1.0])) if paired else 0.0, 1.0, 0.0, float(rng.choice([0.0, 1.0, 2.6, 7.300000000000001, 10.0, 1e-4, 123456789.5]))]
                for mate in range(2 if paired else 1):
                    i_, d_ = seq_al(len(q.sequences[mate]))
                    ints += i_
                    dbls += d_
        io.append(len(ints)); do.append(len(dbls))
    return np.array(ints, np.int32), np.array(dbls, np.float64), np.array(io, np.int64), np.array(do, np.int64)


def test_writer_equals_the_per_object_formatter(tmp_path):
    rng = np.random.default_rng(7)
    contig_names = ["chr%d" % i for i in range(5)]
    # the queries as files (so that the native reader makes the batch), single reads with qualities and pairs
    singles = [("s%d" % i, _rand_seq(rng, int(rng.integers(20, 160)), "ACGTN"), "".join(chr(33 + int(x)) for x in rng.integers(0, 40, 1))) for i in range(700)]
    fq = "".join("@%s\n%s\n+\n%s\n" % (n, s, (q * len(s))) for n, s, q in singles)
    fa = "".join(">%s\n%s\n" % (n, s) for n, s, _ in singles[:100])
    r1 = "".join("@p%d/1\n%s\n+\n%s\n" % (i, _rand_seq(rng, 60), "F" * 60) for i in range(300))
    r2 = "".join("@p%d/2\n%s\n+\n%s\n" % (i, _rand_seq(rng, 45), "G" * 45) for i in range(300))
    jobs = [((_write(tmp_path / "s.fq", fq), None), False), ((_write(tmp_path / "s.fa", fa), None), False), ((_write(tmp_path / "r1.fq", r1), _write(tmp_path / "r2.fq", r2)), True)]
    for (p1, p2), paired in jobs:
        recs1 = cli.read_sequences(p1)
        recs2 = cli.read_sequences(p2) if p2 else None
        queries = [api.Query(recs1[i][1], *( [recs2[i][1]] if paired else []), names=[recs1[i][0]] + ([recs2[i][0]] if paired else [])) for i in range(len(recs1))]
        quals = [[recs1[i][2]] + ([recs2[i][2]] if paired else []) for i in range(len(recs1))]
        sam_path, un_path = tmp_path / "out.sam", tmp_path / "un.txt"
        want_sam, want_un = [], []
        aligned = total_len = indels = 0
        total_penalty = 0.0
        with open(sam_path, "w") as fs, open(un_path, "w") as fu:
            fs.write("@HD\tVN:1.6\n")   # (something in front of the records: the writer appends behind what the Python file object has buffered)
            w = hostio.Writer(contig_names, fs, fu, threads=3)
            first = 0
            for b in hostio.read_batches(p1, p2, batch_size=256, keep_qualities=True):
                qs = queries[first:first + len(b)]
                ints, dbls, io, do = _random_streams(rng, qs, len(contig_names))
                res = api.BatchResult(dict(ints=ints, dbls=dbls, int_off=io, dbl_off=do))
                w.write(b, res)
                for k, q in enumerate(qs):
                    comps = api.decode_streams(ints, dbls, io, do, k)
                    if any(len(c) for c in comps):
                        aligned += 1
                        want_sam += sam.records(q, comps, contig_names)
                        for comp in comps:
                            for al in comp:
                                for sa in al.components:
                                    total_len += sum(x.lengthA for x in sa.sections)
                                    indels += sum(1 for x in sa.sections if x.lengthA != x.lengthB)
                                total_penalty += al.penalty
                    else:
                        for name, seq, qual in zip(q.names, q.sequences, quals[first + k]):
                            want_un.append("@%s\n%s\n+\n%s\n" % (name, api.decode(seq), qual) if qual is not None else ">%s\n%s\n" % (name, api.decode(seq)))
                first += len(b)
        assert open(sam_path).read() == "@HD\tVN:1.6\n" + "".join(x + "\n" for x in want_sam)
        assert open(un_path).read() == "".join(want_un)
        st = w.stats
        assert (st.num_queries, st.num_aligned, st.total_aligned_length, st.num_indels) == (len(queries), aligned, total_len, indels)
        assert st.total_penalty == total_penalty   # (a double sum in query order: the same bits)


@pytest.mark.parametrize("case", KAT["sam_cases"], ids=[c["name"] for c in KAT["sam_cases"]])
def test_native_writer_reproduces_the_reference_sam_bodies(case, tmp_path):
    """T/SamWriter_Test.java:18-94: the oracle's alignments through the native reader and writer give the five pinned SAM bodies (as line sets)."""
    ref = o.OracleReference([("ref", case["reference"])], custom_dup=(1, 2, 2, 1))  # DuplicationDetector(db, 1, 2, 2, 1)
    mates = case["mates"]
    files = [_write(tmp_path / ("m%d.fa" % i), ">%s\n%s\n" % (m[0], m[1])) for i, m in enumerate(mates)]
    batches = list(hostio.read_batches(files[0], files[1] if len(files) > 1 else None, expected_inner=case.get("expectedInner", 0.0), deviation=case.get("deviation", 1.0)))
    assert len(batches) == 1
    b = batches[0]
    s = ref.align(o.QueryBatch.from_arrays(*b.arrays()), o.make_params(KAT["align_cases"][0]["params"]))
    out = tmp_path / "o.sam"
    with open(out, "w") as f:
        hostio.Writer(["ref"], f).write(b, s)
    assert sorted(open(out).read().splitlines()) == sorted(case["sam"].splitlines())


def test_java_double_matches_the_python_formatter():
    rng = np.random.default_rng(8)
    vals = [0.0, 1.0, 0.1, 0.5, 10.0, 1e7, 9999999.999, 1e-3, 0.00099999, 1e-5, 1.5e300, 123456.789, 0.30000000000000004, 1 / 3, 1e22, 5e-324, -2.5, 84743373.69372326]
    vals += list(rng.random(2000) * rng.choice([1, 10, 100, 1e-3, 1e4, 1e6, 1e8], 2000)) + [x * 0.1 for x in range(1000)] + [x / 20 for x in range(2000)]
    vals += list(rng.integers(0, 2**63, 2000).astype(np.uint64).view(np.float64))
    for v in vals:
        if v == v:
            got = hostio.java_double(v)
            assert got == sam.java_double(v), v
            assert float(got.replace("E", "e")) == v or abs(v) == float("inf")
    assert [hostio.java_double(v) for v in (0.0, 1.0, 100.0, 1e7, 1e-3, 1e-4, 12345678.0, 0.6000000000000001)] == ["0.0", "1.0", "100.0", "1.0E7", "0.001", "1.0E-4", "1.2345678E7", "0.6000000000000001"]
