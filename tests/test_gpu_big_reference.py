"""GRCh38-scale property test (BASELINE.json configs[3] / configs[4] live on such a reference; no oracle finishes at this size): 2.3 G bases in ten
contigs = more than 2^32 encoded positions, so the index takes the paths a 5 Mb reference never takes - 64-bit position arrays on the device,
64-byte bucket lines, tables hashed on the GPU in groups.  Properties: reads sampled from known places come back to them (contig, strand, offset);
the same batch aligned twice gives the same streams (determinism); a query's result does not depend on the batch it travels in (batch invariance);
pairs (--spacing 100 50) come back with both mates at their origins; the bucket-line probe and the CSR probe return the same positions."""
import os
import numpy as np
import pytest

from helpers import se_batch, pe_batch, streams_equal
from mapper_amd import api, synth

NC, CLEN = 10, 230_000_000
MB = 1_000_000


@pytest.fixture(scope="module")
def big():
    """Ten contigs of 230 M bases; every megabase holds a run of 10,000 N at its middle (1 % of the reference, the shape SURVEY.md section 8(d) gives
    config 4), so the index is built the way GRCh38's is: blocks clear of the N-runs hashed on the GPU, the multi blocks at their ends by the host."""
    contigs = []
    for c in range(NC):
        parts = [synth.synthetic_reference(min(50_000_000, CLEN - o), seed=0xB16 + 1000 * c + o // 50_000_000) for o in range(0, CLEN, 50_000_000)]
        text = np.concatenate(parts)
        runs = text[: (CLEN // MB) * MB].reshape(-1, MB)
        runs[:, 500_000:510_000] = 15
        contigs.append(("chr%02d" % c, text))
    db = api.ReferenceDatabase(contigs, max_query_length=150)
    yield contigs, db
    db.close()


def sample_reads(contig, per, seed):
    """Reads from the first 400 kb of 25 megabases spread over the contig (clear of the N-runs): (reads, starts in the contig, strands)."""
    reads, starts, strands = [], [], []
    for k, mb in enumerate(range(3, CLEN // MB, (CLEN // MB) // 25)[:25]):
        r, st, sd = synth.synthetic_single_end(contig[mb * MB: mb * MB + 400_000], per // 25, seed=seed + k)
        reads.append(r); starts.append(st + mb * MB); strands.append(sd)
    return np.concatenate(reads), np.concatenate(starts), np.concatenate(strands)


def first_alignment(r):
    """Per query: (has exactly >= 1 alignment, contig, reversed, startB - startA) of the first sequence of the first alignment."""
    io = r.int_off[:-1]
    one = r.ints[io + 1] >= 1
    idx = io[one]
    return one, r.ints[idx + 4], r.ints[idx + 5], r.ints[idx + 8] - r.ints[idx + 7]


@pytest.mark.gpu
def test_big_reference_properties(big):
    contigs, db = big
    info = db.info()
    assert info["position_bytes"] == 8 and info["total_forward_size"] == NC * CLEN and info["built_on_device"] == 1  # (N-runs and all)
    params = api.AlignmentParameters()
    per = 20_000
    reads, where = [], []
    for c in (0, NC // 2, NC - 1):
        r, starts, strand = sample_reads(contigs[c][1], per, seed=0x5EED + 100 * c)
        reads.append(r)
        where.append(np.stack([np.full(per, c), starts, strand], axis=1))
    reads, where = np.concatenate(reads), np.concatenate(where)
    b = se_batch(reads)
    arrays = (b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation)
    r1 = db.align_arrays(*arrays, params)
    one, contig, rev, off = first_alignment(r1)
    assert one.mean() > 0.99
    ok = (contig == where[one, 0]) & (rev == where[one, 2]) & (np.abs(off - where[one, 1]) <= 3)
    assert ok.mean() > 0.99
    # determinism
    r2 = db.align_arrays(*arrays, params)
    assert streams_equal(r1, r2)
    # batch invariance: a slice from the middle, aligned on its own
    lo, hi = 25_000, 27_000
    sb = se_batch(reads[lo:hi])
    r3 = db.align_arrays(sb.mate_count, sb.mate_offset, sb.mate_length, sb.codes, sb.expected_inner, sb.deviation, params)
    assert np.array_equal(r3.ints, r1.ints[r1.int_off[lo]:r1.int_off[hi]]) and np.array_equal(r3.dbls.view(np.int64), r1.dbls[r1.dbl_off[lo]:r1.dbl_off[hi]].view(np.int64))
    # pairs from the last contig (the highest encoded positions: beyond 2^32)
    m1, m2, starts1, inner, strand = synth.synthetic_paired_end(contigs[NC - 1][1][200 * MB: 200 * MB + 400_000], 10_000, seed=0x9A1)
    pb = pe_batch(m1, m2, 100.0, 50.0)
    rp = db.align_arrays(pb.mate_count, pb.mate_offset, pb.mate_length, pb.codes, pb.expected_inner, pb.deviation, params)
    io = rp.int_off[:-1]
    paired = (rp.ints[io] == 1) & (rp.ints[io + 1] >= 1) & (rp.ints[io + 3] == 2)
    assert paired.mean() > 0.97
    assert (rp.ints[io[paired] + 4] == NC - 1).mean() > 0.999


@pytest.mark.gpu
def test_bucket_lines_equal_csr_probes_on_the_big_index(big, monkeypatch):
    _, db = big
    info = db.info()
    rng = np.random.default_rng(99)
    n = 2_000_000
    used = rng.integers(info["min_interesting_size"], info["max_hashed_length"] + 1, size=n, dtype=np.int32)
    keys = rng.integers(-2**31, 2**31 - 1, size=n, dtype=np.int64).astype(np.int32)
    monkeypatch.setenv("XM_PROBE_NO_LINES", "1")
    c0, p0, _ = db.seed_probe(used, keys, 8)
    monkeypatch.setenv("XM_PROBE_NO_LINES", "0")
    c1, p1, _ = db.seed_probe(used, keys, 8)
    assert np.array_equal(c0, c1) and np.array_equal(p0, p1)
    assert (c0 > 7).any() and (c0 == -1).any() and ((c0 >= 1) & (c0 <= 7)).mean() > 0.3  # every branch of the line probe was taken
    h0, _, _ = db.seed_probe(used, keys, 0)
    assert np.array_equal(h0, c0)
