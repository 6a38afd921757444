"""BASELINE.json configs[3] and configs[4] on one GPU, on the reference SURVEY.md section 8(d) states for them: synthetic 3.1 Gb in 24 contigs with the
real GRCh38 chromosome lengths, i.i.d. ACGT, 1 % of the positions in N-runs of 10 kb, seed 0x6C38 (mapper_amd/synth.py::grch38_shaped_reference).  No
oracle finishes at this size (the oracle checks the same regime - minInterestingSize 13, several contigs, N-runs - on a 15 Mb reference of the same
shape: tests/test_gpu_parity.py::test_grch38_regime_alignments_equal_oracle), so the checks are properties:
  * configs[3]: 6,250,000 pairs 2 x 150 bp - one GPU's share of the config's 50 M pairs on 8 GPUs - sampled genome-wide (seed 0x5EED0003, --spacing 100 50)
    come back as pairs with both mates at their origins;
  * configs[4]: reads of 10 kb (seed 0x5EED0004) cut by --split-queries-past-size 1000 (the command line's splitter, SequenceSplitter.java:17,35-38)
    into queries of 1 kb (62,500 reads = 625,000 queries here, a tenth of one GPU's share of the config's 5 M reads on 8 GPUs; the whole share is
    profiles/r05/bench_config4_share.json): with the error rates as stated (5 % substitutions + 5 % indel events per base: above --max-penalty, almost nothing
    aligns, and what does is a short chance match) and with milder ones (2 % + 0.2 %), where the sections that align sit where they came from;
  * determinism (the same batch twice gives the same streams) and batch invariance (a query's result does not depend on the batch it travels in);
  * the index takes the paths a 5 Mb reference never takes: 64-bit position arrays, 64-byte bucket lines, tables hashed on the GPU in groups with the
    multi blocks around the N-runs from the host; the bucket-line probe and the CSR probe return the same positions.
The 8-GPU sharding of these configs (index replicated, batches dealt to GPUs) is what tests/test_multi_gpu.py and tests/test_multirank.py cover."""
import numpy as np
import pytest

from helpers import pe_batch, streams_equal
import oracle_lib
from mapper_amd import api, cli, synth


@pytest.fixture(scope="module")
def grch():
    contigs, whole, starts, runs = synth.grch38_shaped_reference()
    db = api.ReferenceDatabase(contigs, max_query_length=1000)
    yield contigs, whole, starts, runs, db
    db.close()


def first_alignment(r):
    """Per query with at least one alignment: (mask, contig, reversed, startB - startA) of the first sequence of the first alignment."""
    io = r.int_off[:-1]
    one = (r.ints[io] >= 1) & (r.ints[np.minimum(io + 1, len(r.ints) - 1)] >= 1)
    idx = io[one]
    return one, r.ints[idx + 4], r.ints[idx + 5], r.ints[idx + 8] - r.ints[idx + 7]


def arrays(b):
    return (b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation)


@pytest.mark.gpu
def test_config3_pairs_on_the_grch38_shaped_reference(grch):
    contigs, whole, starts, runs, db = grch
    info = db.info()
    assert info["num_contigs"] == 24 and info["total_forward_size"] == sum(synth.GRCH38_LENGTHS) == 3_088_269_832
    assert info["position_bytes"] == 8 and info["built_on_device"] == 1 and info["min_interesting_size"] == 13  # HashBlock_Database.java:52
    assert abs(float((whole[:50_000_000] == 15).mean()) - 0.01) < 0.002
    params = api.AlignmentParameters()
    n = 6_250_000  # configs[3]: 50 M pairs sharded over 8 GPUs (Mapper.java:926,957-983 deal queries to workers; here: to GPUs)
    frag = 2 * 150 + 400 + 3 + 153
    g, contig, local = synth.genome_wide_starts(starts, runs, n, frag, seed=0x5EED0003 ^ 0xF00D)
    m1, m2, starts1, inner, strand = synth.synthetic_paired_end(whole, n, seed=0x5EED0003, at=g)
    pb = pe_batch(m1, m2, 100.0, 50.0)
    r1 = db.align_arrays(*arrays(pb), params)
    io = r1.int_off[:-1]
    paired = (r1.ints[io] == 1) & (r1.ints[io + 1] >= 1) & (r1.ints[io + 3] == 2)
    assert paired.mean() > 0.97
    # both mates on the fragment's contig; the first sequence alignment is mate 1's: it starts where the fragment starts (forward fragment) or where
    # its right end lies (reverse fragment), give or take the read's own indel
    idx = io[paired]
    assert (r1.ints[idx + 4] == contig[paired]).mean() > 0.999
    off1 = r1.ints[idx + 8] - r1.ints[idx + 7]
    want = np.where(strand[paired] == 0, local[paired], local[paired] + 150 + inner[paired])
    assert (np.abs(off1 - want) <= 3).mean() > 0.995
    # determinism, on a part of the batch (the whole batch again would only repeat the minute)
    sl = slice(4_400_000, 4_460_000)
    sb = pe_batch(m1[sl], m2[sl], 100.0, 50.0)
    r2 = db.align_arrays(*arrays(sb), params)
    r3 = db.align_arrays(*arrays(sb), params)
    assert streams_equal(r2, r3)
    # batch invariance: the same queries inside the big batch
    lo, hi = sl.start, sl.stop
    assert np.array_equal(r2.ints, r1.ints[r1.int_off[lo]:r1.int_off[hi]]) and np.array_equal(r2.dbls.view(np.int64), r1.dbls[r1.dbl_off[lo]:r1.dbl_off[hi]].view(np.int64))
    c = r1.counters
    print("configs[3] shape on one GPU: %d pairs, kernel %.1f ms, %.1f header probes and %.1f hits per pair, %.2f %% paired" % (n, r1.kernel_ms, c[1] / n, c[3] / n, 100 * paired.mean()))


def split_batch(reads, size):
    """--split-queries-past-size `size` over reads [n, L]: the sections the command line makes (cli.split_sections), as one batch without copying bases."""
    n, L = reads.shape
    sections = cli.split_sections(L, size)
    k = len(sections)
    a = np.array([s for s, _ in sections], dtype=np.int64)
    ln = np.array([e - s for s, e in sections], dtype=np.int32)
    nq = n * k
    mo = np.zeros(2 * nq, np.int64)
    mo[0::2] = (np.arange(n, dtype=np.int64)[:, None] * L + a[None, :]).reshape(-1)
    ml = np.zeros(2 * nq, np.int32)
    ml[0::2] = np.tile(ln, n)
    b = oracle_lib.QueryBatch.from_arrays(np.ones(nq, np.int32), mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq))
    return b, a, k


@pytest.mark.gpu
def test_config4_long_reads_through_the_splitter(grch):
    contigs, whole, starts, runs, db = grch
    params = api.AlignmentParameters()
    L = 10_000
    span = L + L // 4 + 8
    assert cli.split_sections(L, 1000) == [(1000 * k, 1000 * (k + 1)) for k in range(10)] and cli.split_sections(2500, 1000) == [(0, 833), (833, 1666), (1666, 2500)]
    # (a) the config as stated: 5 % substitutions + 5 % indel events per base -> queries of 1,000 bases
    # (one GPU's share of the config is 625,000 reads = 6.25 M queries; these reads are the slowest the path knows - every section walks the whole
    # chain and fails, ~45 us each - so the test tier aligns a tenth of the share; bench.py --config 4 --reads 625000 runs all of it: profiles/r05)
    n = 62_500
    g, contig, local = synth.genome_wide_starts(starts, runs, n, span, seed=0x5EED0004 ^ 0xF00D)
    strand = (synth.splitmix64(0x5EED0004 ^ 0x57A, n) >> np.uint64(63)).astype(np.uint8)
    reads = synth.synthetic_long_reads(whole, g, L, seed=0x5EED0004, sub_rate=0.05, indel_rate=0.05, strand=strand)
    b, a, k = split_batch(reads, 1000)
    r1 = db.align_arrays(*arrays(b), params)
    one, ctg, rev, off = first_alignment(r1)
    # ~15 penalty units per 100 bases against --max-penalty's 10: next to nothing aligns, and what does is a short local match whose other bases
    # count as unaligned (0.1 each, AlignmentParameters.java:73-95), not the section's origin - so no statement about places here, only about numbers
    frac_stated = one.mean()
    assert frac_stated < 0.01
    sl = slice(330_000, 340_000)
    sb = oracle_lib.QueryBatch.from_arrays(b.mate_count[sl], b.mate_offset[2 * sl.start:2 * sl.stop], b.mate_length[2 * sl.start:2 * sl.stop], b.codes, b.expected_inner[sl], b.deviation[sl])
    r2 = db.align_arrays(*arrays(sb), params)
    assert np.array_equal(r2.ints, r1.ints[r1.int_off[sl.start]:r1.int_off[sl.stop]]) and np.array_equal(r2.dbls.view(np.int64), r1.dbls[r1.dbl_off[sl.start]:r1.dbl_off[sl.stop]].view(np.int64))
    # (b) reads a long-read aligner would be given after polishing: 2 % substitutions, 0.2 % indel events per base: the sections come home
    n2 = 4_000
    reads2 = synth.synthetic_long_reads(whole, g[:n2], L, seed=0x5EED0004 + 1, sub_rate=0.02, indel_rate=0.002, strand=strand[:n2])
    b2, a2, _ = split_batch(reads2, 1000)
    r3 = db.align_arrays(*arrays(b2), params)
    one2, ctg2, rev2, off2 = first_alignment(r3)
    assert one2.mean() > 0.9
    q2 = np.nonzero(one2)[0]
    read2, sec2 = q2 // k, q2 % k
    # where a section came from in its contig: forward reads: local + a (+- the indels before it); reverse reads: the read is the reverse
    # complement of the template's first ~L bases, so section j of it covers template [L' - a_j - 1000, L' - a_j) with L' within the indel drift of L
    fwd = strand[read2] == 0
    want = np.where(fwd, local[read2] + a2[sec2], local[read2] + (L - a2[sec2] - 1000))
    assert (ctg2 == contig[read2]).mean() > 0.999 and (rev2 == strand[read2]).mean() > 0.999
    assert (np.abs(off2 - want) <= 120).mean() > 0.99
    r4 = db.align_arrays(*arrays(b2), params)
    assert streams_equal(r3, r4)
    print("configs[4] shape on one GPU: %d queries of 1 kb as stated (%.2f %% aligned), kernel %.1f ms; %d milder ones (%.1f %% aligned), kernel %.1f ms" %
          (len(one), 100 * frac_stated, r1.kernel_ms, len(one2), 100 * one2.mean(), r3.kernel_ms))


@pytest.mark.gpu
def test_bucket_lines_equal_csr_probes_on_the_big_index(grch, monkeypatch):
    db = grch[4]
    info = db.info()
    rng = np.random.default_rng(99)
    n = 2_000_000
    used = rng.integers(info["min_interesting_size"], 151, size=n, dtype=np.int32)
    keys = rng.integers(-2**31, 2**31 - 1, size=n, dtype=np.int64).astype(np.int32)
    monkeypatch.setenv("XM_PROBE_NO_LINES", "1")
    c0, p0, _ = db.seed_probe(used, keys, 8)
    monkeypatch.setenv("XM_PROBE_NO_LINES", "0")
    c1, p1, _ = db.seed_probe(used, keys, 8)
    assert np.array_equal(c0, c1) and np.array_equal(p0, p1)
    assert (c0 > 7).any() and (c0 == -1).any() and ((c0 >= 1) & (c0 <= 7)).mean() > 0.3  # every branch of the line probe was taken
    h0, _, _ = db.seed_probe(used, keys, 0)
    assert np.array_equal(h0, c0)
