"""CPU tier: the wave-per-read form of the path (mapper_amd/csrc/xm_wave.h: one wavefront per read, state in LDS) in the host simulation
(a WV_PAR region is a loop over the 64 lanes there) against the oracle, in the pass sequence the product runs: light tier, heavy tier,
lane-per-read passes for what is left."""
import os
import numpy as np
import pytest

import oracle_lib as o
import hostsim_lib as hs
from helpers import KAT, streams_equal, first_difference, se_batch, pe_batch, check_align_case, sprinkle_ambiguity, ambiguous_reference
from mapper_amd import api, synth


@pytest.fixture(autouse=True)
def _wave_mode():
    hs.set_wave_mode(3)
    hs.wave_status_counts()
    yield
    hs.set_wave_mode(-1)


@pytest.mark.parametrize("tiers", [1, 2, 3])
def test_wave_form_single_end(tiers):
    """configs[1] shape: every read is taken by the wave form (nothing left to the lane-per-read passes), results bit-equal to the oracle."""
    hs.set_wave_mode(tiers)
    ref = synth.synthetic_reference(250_000)
    reads = synth.synthetic_single_end(ref, 4000, seed=0x5EED0011)[0]
    b = se_batch(reads)
    want = o.OracleReference([("ecoli_syn", ref)]).align(b, o.make_params(), threads=os.cpu_count())
    got = hs.SimReference([("ecoli_syn", ref)]).align(b, o.make_params())
    assert streams_equal(got, want), first_difference(got, want, len(reads))
    st = hs.wave_status_counts()
    assert st[8] == 0
    if tiers == 3:
        assert st[0] == len(reads)


@pytest.mark.parametrize("read_len", [36, 75, 100, 250, 256])
def test_wave_form_read_lengths(read_len):
    ref = synth.synthetic_reference(200_000, seed=77)
    reads = synth.synthetic_single_end(ref, 1200, read_len=read_len, seed=0x5EED0012 + read_len)[0]
    b = se_batch(reads)
    want = o.OracleReference([("r", ref)]).align(b, o.make_params(), threads=os.cpu_count())
    got = hs.SimReference([("r", ref)]).align(b, o.make_params())
    assert streams_equal(got, want), first_difference(got, want, len(reads))
    assert hs.wave_status_counts()[0] > 0.9 * len(reads)


def test_wave_form_paired_end():
    """configs[2] shape (--spacing 100 50), including pairs whose mates overlap (left to the lane-per-read passes) and pairs that fall back to unpaired alignments."""
    ref = synth.synthetic_reference(300_000)
    m1, m2 = synth.synthetic_paired_end(ref, 2500, seed=0x5EED0013)[:2]
    m2[:40] = synth.synthetic_single_end(ref, 40, seed=5)[0]  # mates from unrelated places: unpaired fallback
    b = pe_batch(m1, m2)
    want = o.OracleReference([("ecoli_syn", ref)]).align(b, o.make_params(), threads=os.cpu_count())
    got = hs.SimReference([("ecoli_syn", ref)]).align(b, o.make_params())
    assert streams_equal(got, want), first_difference(got, want, len(m1))
    st = hs.wave_status_counts()
    assert st[0] > 0.95 * len(m1)


def test_wave_form_leaves_what_it_does_not_take():
    """Reads with ambiguity codes and mates longer than 256 bases go through the lane-per-read passes; a reference with N runs and IUPAC codes is fine."""
    ref = ambiguous_reference(150_000, seed=0xA3D, n_runs=40, n_codes=400)
    reads = synth.synthetic_single_end(ref, 1500, seed=52)[0]
    reads[700:] = sprinkle_ambiguity(reads[700:], 5)
    long_reads = synth.synthetic_single_end(ref, 60, read_len=400, seed=53)[0]
    R = o.OracleReference([("amb", ref)])
    S = hs.SimReference([("amb", ref)])
    for b, n in ((se_batch(reads), len(reads)), (se_batch(long_reads), len(long_reads))):
        want = R.align(b, o.make_params(), threads=os.cpu_count())
        got = S.align(b, o.make_params())
        assert streams_equal(got, want), first_difference(got, want, n)
    st = hs.wave_status_counts()
    assert st[8] >= 860 and st[0] >= 400   # (reads next to an ambiguity code of the reference leave the chain tier too)


@pytest.mark.parametrize("case", KAT["align_cases"], ids=lambda c: c["name"])
def test_wave_form_on_reference_kats(case):
    """The reference's own AlignerWorker_Test cases (non-default penalties, indels, overlaps, read past a contig end) through the wave form."""
    R = o.OracleReference([("reference-0", case["reference"])], mode="api")
    S = hs.SimReference([("reference-0", case["reference"])], mode="api")
    q = [(case["mates"], case["expectedInner"], case["deviation"])]
    p = o.make_params(case["params"])
    sa, sb = R.align(q, p), S.align(q, p)
    assert streams_equal(sa, sb), first_difference(sa, sb, 1)
    check_align_case(case, api.decode_streams(sb.ints, sb.dbls, sb.int_off, sb.dbl_off, 0), o.encode(case["reference"]))


def test_wave_form_random_parameters_and_repeats():
    """Differential fuzz: random penalties / error rates / MaxNumMatches, a reference with repeats, single-end and paired mixes."""
    rng = np.random.default_rng(20261002)
    for trial in range(6):
        unit = synth.synthetic_reference(int(rng.integers(300, 3000)), seed=1000 + trial)
        ref = np.concatenate([synth.synthetic_reference(60_000, seed=2000 + trial), np.tile(unit, 4), synth.synthetic_reference(40_000, seed=3000 + trial)])
        L = int(rng.choice([50, 100, 150, 200]))
        reads = synth.synthetic_single_end(ref, 300, read_len=L, seed=4000 + trial)[0]
        m1, m2 = synth.synthetic_paired_end(ref, 150, read_len=L, seed=5000 + trial)[:2]
        queries = [([r], 0.0, 1.0) for r in reads] + [([m1[i], m2[i]], 100.0, 50.0) for i in range(len(m1))]
        b = o.QueryBatch(queries)
        p = o.make_params(MutationPenalty=float(rng.choice([1.0, 0.7, 1.3])), InsertionStart_Penalty=float(rng.choice([1.5, 2.0, 0.9])), MaxErrorRate=float(rng.choice([0.1, 0.06, 0.15])),
                          Max_PenaltySpan=float(rng.choice([0.5, 0.0, 2.0])), MaxNumMatches=int(rng.choice([2**31 - 1, 3, 1])))
        want = o.OracleReference([("r", ref)]).align(b, p, threads=os.cpu_count())
        got = hs.SimReference([("r", ref)]).align(b, p)
        assert streams_equal(got, want), (trial, first_difference(got, want, len(queries)))
