"""The rejection filter in front of PathAligner's search (mapper_amd/csrc/xm_bound.h; gapped passes of batches of long reads) on the GPU.

What is asserted, and against what:
* the filter alone, on random problems (tests/helpers.py bound_problems), takes and rejects exactly the searches the ORACLE's observer of the same bound
  does (oracle/xmo_extend.h PathAligner::boundObserve) - the observer runs the reference's search beside its bound and raises when a search the bound
  rejects returns an alignment, so "the filter never rejects a search the oracle completes" is checked on every problem;
* batches of long reads (BASELINE.json configs[4]: 10 kb reads cut into 1 kb queries) give the oracle's streams bit for bit with the filter on and off, and the
  counters add up (tests/helpers.py filter_counters): pieces and searches examined / rejected equal the observer's; PathAligner calls and nodes = the reference's minus what it
  spent in rejected searches and inside rejected pieces."""
import ctypes as C
import os
import numpy as np
import pytest

import oracle_lib as o
from helpers import streams_equal, first_difference, bound_problems, filter_counters
from mapper_amd import api, synth, cli, _capi


def device_bound(prm, query, query_rc, start_a, end_a, ref, start_b, end_b, offset, pair):
    L = _capi.lib()
    p = api.AlignmentParameters(**prm)._c()
    q = np.ascontiguousarray(query, dtype=np.uint8)
    r = np.ascontiguousarray(ref, dtype=np.uint8)
    out = (C.c_int64 * 3)()
    if L.xm_test_bound(0, C.byref(p), q.ctypes.data, len(q), 1 if query_rc else 0, start_a, end_a, r.ctypes.data, len(r), start_b, end_b, offset, int(pair), out):
        raise RuntimeError(L.xm_last_error().decode())
    return int(out[0]), int(out[1]), int(out[2])


@pytest.mark.gpu
@pytest.mark.parametrize("pair", [0, 1, 3], ids=["one-lane", "pair-of-lanes", "eight-lanes"])
def test_filter_rejects_exactly_what_the_oracle_observer_rejects(pair):
    taken = rejected = 0
    for prm, q, rc, sa, ea, ref, sb, eb, off in bound_problems(0xB0 + int(pair), 700):
        verdict, found, _ = o.kat_bound(o.make_params(prm), q, rc, sa, ea, ref, sb, eb, off)   # (raises if the bound rejected a search that aligned)
        t, rj, cells = device_bound(prm, q, rc, sa, ea, ref, sb, eb, off, pair)
        assert (t, rj) == (1 if verdict else 0, 1 if verdict == 2 else 0), (prm, rc, sa, ea, sb, eb, off, verdict, found, t, rj)
        assert not (rj and found)
        assert cells <= (ea - sa) * (eb - sb) if t else cells == 0
        taken += t
        rejected += rj
    assert taken > 300 and rejected > 100   # (the problem mix exercises both outcomes)


def long_read_batch(ref, n_reads, sub, indel, seed=0x5EED0004):
    starts = (synth.splitmix64(seed, n_reads) % np.uint64(len(ref) - 12_600)).astype(np.int64)
    strand = (synth.splitmix64(seed ^ 0x57A, n_reads) >> np.uint64(63)).astype(np.uint8)
    reads = synth.synthetic_long_reads(ref, starts, 10_000, seed=seed, sub_rate=sub, indel_rate=indel, strand=strand)
    return o.QueryBatch([([r[a_:b_].copy()], 0.0, 1.0) for r in reads for a_, b_ in cli.split_sections(10_000, 1000)])


@pytest.mark.gpu
@pytest.mark.parametrize("sub,indel", [(0.05, 0.05), (0.02, 0.002), (0.035, 0.02)], ids=["as_stated", "mild", "between"])
def test_long_read_batches_equal_oracle_with_the_filter_and_the_counters_add_up(sub, indel, monkeypatch):
    ref = synth.synthetic_reference(2_000_000, seed=0xEC011)
    b = long_read_batch(ref, 60, sub, indel)
    R = o.OracleReference([("r", ref)])
    with o.observe_bound():
        want = R.align(b, o.make_params(), threads=os.cpu_count())
    calls, nodes = want.counters[6], want.counters[7]
    db = api.ReferenceDatabase([("r", ref)], max_query_length=1000)
    try:
        got = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
        assert streams_equal(want, got), first_difference(want, got, b.nq)
        assert got.extra[3] == 1, "a batch of long reads runs its gapped pass with the filter"
        ok, what = filter_counters(got.counters, got.extra, want.counters)
        assert ok, what
        ref_side = what["reference"]
        if indel == 0.05:   # reads that do not align: most pieces are proved unalignable before their chain runs, most of the reference's search nodes are never put
            assert what["oracle_observer"]["pieces_rejected"] > 0.5 * what["oracle_observer"]["pieces_examined"] > 0, what
            assert ref_side["nodes_in_rejected_searches"] + ref_side["nodes_in_rejected_pieces"] > 0.75 * ref_side["nodes"], what
        # the forms the eight-lane passes were measured against: two lanes per read (every lane computes every cell), and eight lanes that only repeat the pair's work
        for knob in ("XM_GROUP_LANES", "XM_GROUP_SWEEP"):
            monkeypatch.setenv(knob, "0")
            alt = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
            monkeypatch.delenv(knob)
            assert streams_equal(want, alt), (knob, first_difference(want, alt, b.nq))
            ok, what_alt = filter_counters(alt.counters, alt.extra, want.counters)
            assert ok, (knob, what_alt)
        monkeypatch.setenv("XM_BOUND_FILTER", "0")   # the same batch without the filter: same streams, the reference's node count
        off = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
        assert streams_equal(want, off), first_difference(want, off, b.nq)
        assert off.extra[3] == 0 and off.extra[1] == 0 and off.extra[5] == 0 and off.counters[5] == calls and off.counters[6] == nodes
    finally:
        db.close()


@pytest.mark.gpu
def test_short_read_batches_run_without_the_filter():
    """Batches of reads up to 320 bases: their searches use the wave's LDS slot, the filter stays off (it would cost what it saves there: profiles/r06/NOTES.md 1)."""
    ref = synth.synthetic_reference(300_000, seed=0xEC011)
    reads = synth.synthetic_single_end(ref, 3000, seed=0x5EED0001)[0]
    b = o.QueryBatch([([r], 0.0, 1.0) for r in reads])
    want = o.OracleReference([("r", ref)]).align(b, o.make_params(), threads=os.cpu_count())
    db = api.ReferenceDatabase([("r", ref)])
    try:
        got = db.align_arrays(b.mate_count, b.mate_offset, b.mate_length, b.codes, b.expected_inner, b.deviation, api.AlignmentParameters())
    finally:
        db.close()
    assert streams_equal(want, got) and got.extra[3] == 0 and got.counters[6] == want.counters[7]
