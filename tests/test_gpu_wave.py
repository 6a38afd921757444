"""GPU tier: the wave-per-read form of the path (XM_WAVE=1: xm_wave_kernel.hip, one wavefront per read, state in LDS) through the C ABI
against the oracle; the 64-bit position arrays of GRCh38-sized references on a small reference (XM_FORCE_POS64); device counters against
the oracle's."""
import os
import sys
import numpy as np
import pytest

import oracle_lib as o
from helpers import KAT, streams_equal, first_difference, se_batch, pe_batch, check_align_case, sprinkle_ambiguity
from mapper_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_align(db, batch, params=None):
    r = db.align_arrays(batch.mate_count, batch.mate_offset, batch.mate_length, batch.codes, batch.expected_inner, batch.deviation,
                        params or api.AlignmentParameters())
    return o.Streams(r.ints, r.dbls, r.int_off, r.dbl_off, r.counters), r


def counters_match(dev, oracle):
    """Device counters (xm_result.counters) against the oracle's (oracle/xmo_types.h Counters): reads, bucket-header probes (the device
    counts the header read of a fetch as a probe too), bucket fetches, positions fetched, candidates extended, PathAligner calls and nodes,
    quick accepts."""
    d, w = [int(x) for x in dev[:8]], [int(x) for x in oracle[:9]]
    return d == [w[0], w[1] + w[2], w[2], w[3], w[5], w[6], w[7], w[8]], (d, w)


@pytest.mark.parametrize("case", KAT["align_cases"], ids=lambda c: c["name"])
def test_reference_kats_in_wave_form(case, monkeypatch):
    monkeypatch.setenv("XM_WAVE", "1")
    db = api.newDatabase(case["reference"])
    b = o.QueryBatch([(case["mates"], case["expectedInner"], case["deviation"])])
    got, _ = gpu_align(db, b, api.AlignmentParameters(**case["params"]))
    want = o.OracleReference([("reference-0", case["reference"])], mode="api").align(b, o.make_params(case["params"]))
    assert streams_equal(got, want), first_difference(got, want, 1)
    check_align_case(case, api.decode_streams(got.ints, got.dbls, got.int_off, got.dbl_off, 0), o.encode(case["reference"]))
    db.close()


@pytest.mark.parametrize("wave", ["0", "1"])
def test_device_counters_equal_the_oracles(wave, monkeypatch):
    """the probe / hit / candidate counts the roofline is computed from are the oracle's (same batch), in both forms;
    the PathAligner node count pins the search to the reference's exploration order, not only to its result."""
    monkeypatch.setenv("XM_WAVE", wave)
    ref = synth.synthetic_reference(400_000, seed=91)
    reads = synth.synthetic_single_end(ref, 20_000, seed=92)[0]
    m1, m2 = synth.synthetic_paired_end(ref, 4000, seed=93)[:2]
    R = o.OracleReference([("r", ref)])
    db = api.ReferenceDatabase([("r", ref)])
    for b, n in ((se_batch(reads), len(reads)), (pe_batch(m1, m2, 100.0, 50.0), len(m1))):
        want = R.align(b, o.make_params(), threads=1)  # (one thread: the oracle's counters are per worker)
        got, _ = gpu_align(db, b)
        assert streams_equal(got, want), first_difference(got, want, n)
        ok, detail = counters_match(got.counters, want.counters)
        assert ok, detail
    db.close()


@pytest.mark.parametrize("read_len", [36, 100, 150, 250])
def test_wave_form_read_lengths_on_gpu(read_len, monkeypatch):
    monkeypatch.setenv("XM_WAVE", "1")
    ref = synth.synthetic_reference(300_000, seed=77)
    reads = synth.synthetic_single_end(ref, 5000, read_len=read_len, seed=80 + read_len, indel_prob=0.2)[0]
    reads[4000:] = sprinkle_ambiguity(reads[4000:], 3)  # (these go through the lane-per-read passes)
    b = se_batch(reads)
    want = o.OracleReference([("r", ref)]).align(b, o.make_params(), threads=os.cpu_count())
    db = api.ReferenceDatabase([("r", ref)], max_query_length=read_len)
    got, _ = gpu_align(db, b)
    assert streams_equal(got, want), first_difference(got, want, len(reads))
    db.close()


def test_wave_form_random_configurations_on_gpu(monkeypatch):
    """The differential fuzz of scripts/gpu_fuzz.py with the wave-per-read form on."""
    monkeypatch.setenv("XM_WAVE", "1")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gpu_fuzz
    assert gpu_fuzz.run(rounds=8, seed=1077, max_queries=1500) == 0


@pytest.mark.parametrize("wave", ["0", "1"])
def test_64_bit_positions_on_a_small_reference(wave, monkeypatch):
    """BASELINE.json configs[3] / configs[4] (GRCh38: more than 2^32 encoded positions) read the positions from 64-bit arrays; XM_FORCE_POS64=1
    takes that path on a multi-contig reference the oracle finishes in seconds: pairs (--spacing 100 50) and 1,000 bp queries."""
    monkeypatch.setenv("XM_FORCE_POS64", "1")
    monkeypatch.setenv("XM_WAVE", wave)
    contigs = api.sort_reference([("c%d" % i, synth.synthetic_reference(n, seed=300 + i)) for i, n in enumerate((220_000, 150_000, 90_000, 4_000))])
    whole = np.concatenate([c for _, c in contigs])
    R = o.OracleReference(contigs)
    db = api.ReferenceDatabase(contigs, max_query_length=1000)
    assert db.info()["position_bytes"] == 8
    m1, m2 = synth.synthetic_paired_end(contigs[0][1], 2500, seed=301)[:2]
    reads = synth.synthetic_single_end(contigs[1][1], 1500, seed=302, indel_prob=0.3)[0]
    long_reads = synth.synthetic_single_end(contigs[0][1], 150, read_len=1000, sub_rate=0.03, indel_prob=0.6, seed=303)[0]
    del whole
    for b, n in ((pe_batch(m1, m2, 100.0, 50.0), len(m1)), (se_batch(reads), len(reads)), (se_batch(long_reads), len(long_reads))):
        want = R.align(b, o.make_params(), threads=os.cpu_count())
        got, _ = gpu_align(db, b)
        assert streams_equal(got, want), first_difference(got, want, n)
    db.close()


@pytest.mark.gpu
def test_experiment_knobs_are_validated(monkeypatch):
    """XM_* environment knobs outside their range are an error of the call (a scale that is not a power of two would break the hash capacities)."""
    ref = synth.synthetic_reference(40_000, seed=3)
    reads = synth.synthetic_single_end(ref, 64, seed=4)[0]
    b = se_batch(reads)
    db = api.ReferenceDatabase([("c", ref)], mode="mapper")
    try:
        for name, value in (("XM_GAPPED_SCALE", "3"), ("XM_FULL_LPW", "0"), ("XM_ARENA_KB", "1")):
            monkeypatch.setenv(name, value)
            with pytest.raises(RuntimeError, match=name):
                db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
            monkeypatch.delenv(name)
        r = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
        assert len(r) == 64
    finally:
        db.close()
