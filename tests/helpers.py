"""Shared helpers of the test-suite."""
import json
import os
import numpy as np

import oracle_lib
from mapper_amd import api, sam, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_reference.json")))


def streams_equal(a, b):
    """bit-exact equality of two result streams (ints, doubles compared by bit pattern, offsets)."""
    return (np.array_equal(a.int_off, b.int_off) and np.array_equal(a.dbl_off, b.dbl_off) and np.array_equal(a.ints, b.ints)
            and np.array_equal(np.asarray(a.dbls).view(np.int64), np.asarray(b.dbls).view(np.int64)))


def first_difference(a, b, n):
    for q in range(n):
        x = api.decode_streams(a.ints, a.dbls, a.int_off, a.dbl_off, q)
        y = api.decode_streams(b.ints, b.dbls, b.int_off, b.dbl_off, q)
        fx = [[(al.penalty, [(s.contig, s.reference_reversed, [(k.startA, k.startB, k.lengthA, k.lengthB) for k in s.sections]) for s in al.components]) for al in c] for c in x]
        fy = [[(al.penalty, [(s.contig, s.reference_reversed, [(k.startA, k.startB, k.lengthA, k.lengthB) for k in s.sections]) for s in al.components]) for al in c] for c in y]
        if fx != fy:
            return "query %d: %r != %r" % (q, fx, fy)
    return None


def se_batch(reads):
    nq, L = reads.shape
    mc = np.ones(nq, np.int32)
    mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * L
    ml = np.zeros(2 * nq, np.int32); ml[0::2] = L
    return oracle_lib.QueryBatch.from_arrays(mc, mo, ml, np.ascontiguousarray(reads.reshape(-1)), np.zeros(nq), np.ones(nq))


def ragged_se_batch(reads):
    """single reads of different lengths (a list of code arrays)"""
    nq = len(reads)
    lens = np.array([len(r) for r in reads], np.int64)
    mc = np.ones(nq, np.int32)
    mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.concatenate([[0], np.cumsum(lens)[:-1]])
    ml = np.zeros(2 * nq, np.int32); ml[0::2] = lens
    return oracle_lib.QueryBatch.from_arrays(mc, mo, ml, np.ascontiguousarray(np.concatenate(reads)), np.zeros(nq), np.ones(nq))


def pe_batch(m1, m2, expected=100.0, dev=50.0):
    nq, L = m1.shape
    codes = np.ascontiguousarray(np.concatenate([m1, m2], axis=1).reshape(-1))
    mc = np.full(nq, 2, np.int32)
    mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 2 * L; mo[1::2] = mo[0::2] + L
    ml = np.full(2 * nq, L, np.int32)
    return oracle_lib.QueryBatch.from_arrays(mc, mo, ml, codes, np.full(nq, expected), np.full(nq, dev))


def check_align_case(case, comps, ref_codes):
    """Checks the expectations a reference JUnit case pins on the decoded QueryAlignments (list of components)."""
    e = case["expect"]
    top = comps[0] if len(comps) == 1 else []
    assert len(top) == e["num"], "%s: expected %d alignments, got %d" % (case["name"], e["num"], len(top))
    if "alignedB0" in e:
        s = top[0].components[0]
        q = api.encode(case["mates"][0])
        if s.reference_reversed:
            q = api.reverse_complement(q)
        assert s.aligned_text(q, ref_codes)[1] == e["alignedB0"], case["name"]
    if "startsB" in e:
        got = sorted([s.start_index_b() for s in al.components] for al in top)
        # the order of equal-penalty alignments comes from a HashSet in the reference (QueryMatch_Aligner.java:86-92): compared as a set
        assert got == sorted(e["startsB"]), case["name"]


def sam_text(query, comps, names):
    return "".join(line + "\n" for line in sam.records(query, comps, names))


def sprinkle_ambiguity(reads, seed=3):
    """Copies of `reads` (code arrays [n, L]) with 0-3 IUPAC ambiguity codes each (N, R, Y, M, K, V, B) and, in every 50th read, a run of 2-11 N's."""
    rng = np.random.default_rng(seed)
    out = reads.copy()
    codes = np.array([15, 15, 15, 5, 10, 3, 12, 7, 14], np.uint8)
    for q in range(len(out)):
        k = rng.integers(0, 4)
        pos = rng.integers(0, out.shape[1], size=k)
        out[q, pos] = codes[rng.integers(0, len(codes), size=k)]
        if q % 50 == 0:
            p0 = rng.integers(0, out.shape[1] - 12)
            out[q, p0:p0 + rng.integers(2, 12)] = 15
    return out


def ambiguous_reference(n, seed, n_runs=6, n_codes=40):
    """A synthetic reference (code array) with runs of N (1-40 long) and scattered IUPAC codes."""
    from mapper_amd import synth
    ref = synth.synthetic_reference(n, seed=seed).copy()
    rng = np.random.default_rng(seed)
    for _ in range(n_runs):
        p0 = int(rng.integers(0, n - 50))
        ref[p0:p0 + int(rng.integers(1, 41))] = 15
    codes = np.array([15, 5, 10, 3, 12, 6, 9, 7, 11, 13, 14], np.uint8)
    ref[rng.integers(0, n, size=n_codes)] = codes[rng.integers(0, len(codes), size=n_codes)]
    return ref


IUPAC_2WAY = np.array([3, 5, 9, 6, 10, 12], np.uint8)   # M R W S Y K
IUPAC_3WAY = np.array([7, 11, 13, 14], np.uint8)        # V H D B


def heavy_ambiguity(reads, seed=11):
    """Copies of `reads` (code arrays [n, L]) in which read q carries ambiguity of class q mod 16: a fraction of its positions - 1 %, 10 %, 50 %, 90 %, 100 % -
    replaced by N (classes 0-4), by two-way IUPAC codes (5-9) or by three-way codes (10-12: 10 %, 50 %, 100 %); 13: a run of N at the read's start of a
    third to all of its length; 14: the same at its end; 15: untouched.  What real FASTQ holds (all-N reads, N tails) and what no other test reaches: the
    reference bounds the combinations per block, not the ambiguous bases per read (HashBlock_ParentRow.java:10,109,165)."""
    rng = np.random.default_rng(seed)
    out = reads.copy()
    n, L = out.shape
    fr = [0.01, 0.1, 0.5, 0.9, 1.0]
    for q in range(n):
        c = q % 16
        if c < 5:
            out[q, rng.random(L) < fr[c]] = 15
        elif c < 10:
            m = rng.random(L) < fr[c - 5]
            out[q, m] = IUPAC_2WAY[rng.integers(0, len(IUPAC_2WAY), int(m.sum()))]
        elif c < 13:
            m = rng.random(L) < [0.1, 0.5, 1.0][c - 10]
            out[q, m] = IUPAC_3WAY[rng.integers(0, len(IUPAC_3WAY), int(m.sum()))]
        elif c == 13:
            out[q, :int(rng.integers(L // 3, L + 1))] = 15
        elif c == 14:
            out[q, L - int(rng.integers(L // 3, L + 1)):] = 15
    return out


def low_complexity_reads(n, L, seed=12):
    """homopolymer, dinucleotide and short-period reads (code arrays [n, L]), a few with one ambiguity code in them"""
    rng = np.random.default_rng(seed)
    acgt = np.array([1, 2, 4, 8], np.uint8)
    out = np.zeros((n, L), np.uint8)
    for q in range(n):
        period = [1, 2, 3, 5][q % 4]
        unit = acgt[rng.integers(0, 4, period)]
        out[q] = np.tile(unit, L // period + 1)[:L]
        if q % 5 == 4:
            out[q, int(rng.integers(0, L))] = [15, 5, 14][q % 3]
    return out
