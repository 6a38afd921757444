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


def bound_problems(seed, count):
    """Random problems for the rejection filter in front of PathAligner (mapper_amd/csrc/xm_bound.h) and the oracle's observer of it: tuples
    (params dict, query codes, query_rc, start_a, end_a, reference codes, start_b, end_b, predicted_best_offset).  A query section is a window of the reference
    put through one of several error regimes (none ... unrelated text), so that searches that align, searches that fail narrowly and searches that fail by far
    all occur; windows shorter than the query, windows at the ends of the reference, long windows, wide bands, ambiguity codes and price sets that are not
    multiples of the filter's grid are mixed in."""
    rng = np.random.default_rng(seed)
    acgt = np.array([1, 2, 4, 8], dtype=np.uint8)
    comp = np.zeros(16, dtype=np.uint8)
    for c in range(16):
        comp[c] = ((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3)
    out = []
    for _ in range(count):
        n = int(rng.choice([10, 25, 40, 60, 91, 91, 91, 120, 182, 250, 400]))
        slack_lo, slack_hi = (int(x) for x in rng.choice([0, 0, 3, 10, 25, 60, 140], 2))
        if rng.random() < 0.08:
            slack_lo, slack_hi = int(rng.integers(100, 300)), int(rng.integers(100, 300))   # a window far longer than the query
        R = int(rng.integers(n + slack_lo + slack_hi + 40, n + slack_lo + slack_hi + 400))
        ref = acgt[rng.integers(0, 4, R)]
        where = rng.random()
        if where < 0.08:
            start_b = 0
        elif where < 0.16:
            start_b = R - (n + slack_lo + slack_hi)
        else:
            start_b = int(rng.integers(1, R - (n + slack_lo + slack_hi)))
        end_b = start_b + slack_lo + n + slack_hi
        # the query section: the window's middle, mutated
        regime = rng.choice(["none", "subs", "indel", "mixed", "heavy", "unrelated"], p=[0.08, 0.17, 0.2, 0.25, 0.2, 0.1])
        src = ref[start_b + slack_lo: start_b + slack_lo + n + 40 if start_b + slack_lo + n + 40 <= R else R].copy()
        sub, ind = {"none": (0, 0), "subs": (0.04, 0), "indel": (0.005, 0.01), "mixed": (0.03, 0.02), "heavy": (0.06, 0.05), "unrelated": (0, 0)}[regime]
        seq = []
        i = 0
        while len(seq) < n and i < len(src):
            u = rng.random()
            if u < ind / 2:
                seq.extend(acgt[rng.integers(0, 4, int(rng.integers(1, 4)))])   # insertion
            elif u < ind:
                i += int(rng.integers(1, 4))                                      # deletion
                continue
            b = src[i]
            if rng.random() < sub:
                b = acgt[(int(np.log2(b)) + int(rng.integers(1, 4))) & 3]
            seq.append(b)
            i += 1
        sec = np.array(seq[:n], dtype=np.uint8)
        if regime == "unrelated" or len(sec) < n:
            sec = acgt[rng.integers(0, 4, n)]
        if rng.random() < 0.1:   # ambiguity codes in the query or the window
            k = int(rng.integers(1, 6))
            sec[rng.integers(0, n, k)] = rng.choice([15, 5, 10, 3, 12, 7, 14], k)
            ref[rng.integers(start_b, end_b, k)] = rng.choice([15, 5, 10, 6, 9, 11, 13], k)
        if rng.random() < 0.07:  # a window shorter than the query
            end_b = start_b + max(1, n - int(rng.integers(1, 30)))
        # the section inside a longer query, possibly of its reverse complement
        pre, post = int(rng.integers(0, 50)), int(rng.integers(0, 50))
        view = np.concatenate([acgt[rng.integers(0, 4, pre)], sec, acgt[rng.integers(0, 4, post)]])
        query_rc = bool(rng.random() < 0.5)
        query = comp[view[::-1]] if query_rc else view   # the stored query; the view the aligner sees is `view`
        start_a, end_a = pre, pre + n
        offset = (start_b + slack_lo) - start_a + int(rng.choice([0, 0, 0, 1, -2, 7]))
        prm = {}
        u = rng.random()
        if u < 0.25:
            prm = dict(MutationPenalty=float(rng.choice([1.0, 0.8, 1.37, 2.0])), InsertionStart_Penalty=float(rng.choice([1.5, 0.9, 2.25, 0.41])),
                       InsertionExtension_Penalty=float(rng.choice([0.6, 0.3, 0.77, 1.0])), DeletionStart_Penalty=float(rng.choice([1.5, 1.0, 3.1])),
                       DeletionExtension_Penalty=float(rng.choice([0.5, 0.25, 0.61])), AmbiguityPenalty=float(rng.choice([0.1, 0.0, 0.33])))
        prm["MaxErrorRate"] = float(rng.choice([0.1, 0.1, 0.1, 0.1091, 0.05, 0.02, 0.2, 0.3, 0.0]))
        out.append((prm, query, query_rc, start_a, end_a, ref, start_b, end_b, offset))
    return out


def filter_counters(device_counters, device_extra, oracle_counters):
    """The work counters of a call that ran with the rejection filter (mapper_amd/csrc/xm_bound.h) against the oracle's with its observer on (oracle_lib.observe_bound):
    -> (equal, what).  The product skips the searches the filter proves null (their nodes) and the whole chain of the pieces it proves unalignable (their PathAligner calls and
    nodes); the observer counts both, and what the search-level filter sees outside rejected pieces."""
    oc = [int(x) for x in oracle_counters]
    calls, nodes, rejects, reject_nodes, checks, piece_checks, piece_rejects, skipped_calls, skipped_nodes = oc[6], oc[7], oc[11], oc[12], oc[13], oc[14], oc[15], oc[16], oc[17]
    want = dict(searches_examined=checks, searches_rejected=rejects, pieces_examined=piece_checks, pieces_rejected=piece_rejects, path_aligner_calls=calls - skipped_calls,
                nodes=nodes - reject_nodes - skipped_nodes)
    got = dict(searches_examined=int(device_extra[0]), searches_rejected=int(device_extra[1]), pieces_examined=int(device_extra[4]), pieces_rejected=int(device_extra[5]),
               path_aligner_calls=int(device_counters[5]), nodes=int(device_counters[6]))
    return got == want, dict(device=got, oracle_observer=want, reference=dict(path_aligner_calls=calls, nodes=nodes, searches_returning_null=oc[9], nodes_in_null_searches=oc[10],
                                                                                nodes_in_rejected_searches=reject_nodes, calls_in_rejected_pieces=skipped_calls, nodes_in_rejected_pieces=skipped_nodes))
