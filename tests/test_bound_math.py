"""The identity the eight-lane form of the rejection filter's recurrence rests on (mapper_amd/csrc/xm_bound.h, boundSweep; profiles/r06/NOTES.md 13), checked on the CPU with numpy.

Along a column the deletion state is F(k) = min(F(k-1) + de, H(k-1) + ds + de) with H(k) = min(H0(k), F(k)), H0 = the cell without its deletion state.  The device computes eight
cells at a time from the prefix minimum  F(k) = min_{j<k} (H0(j) - j de) + ds + de + (k - 1) de  - H0 in place of H: a run that starts from a cell which itself came out of a run
pays a second start (ds >= 0), never the minimum - with the minimum of the cells done so far carried from one group of eight to the next, and the cells above the previous column's
interval filled by the run alone.  The test states both forms in plain Python and compares columns of random values, prices and interval shapes.  (The device code itself is
compared with the oracle on the GPU: tests/test_gpu_bound.py, eight-lanes.)
"""
import numpy as np

INF = 0xFFFF
BIG = 1 << 28


def column_sequential(h0, e, thr, dsde, de, k_start, k_end_geom):
    """The reference form: cells k_start .. (k_start + len(h0) - 1) have H0 / E given, cells above them (up to k_end_geom) only the run; -> {k: (h, e)} of the cells written, live lo, hi."""
    out = {}
    f, h_below = INF, INF
    lo, hi = None, None
    k = k_start
    for j in range(len(h0)):
        f = min(f + de, h_below + dsde)
        h = min(h0[j], f, INF)
        out[k] = (h, min(e[j], INF))
        if h <= thr:
            lo = k if lo is None else lo
            hi = k
        h_below = h
        k += 1
    while k <= k_end_geom:
        f = min(f + de, h_below + dsde)
        if f > thr:
            break
        out[k] = (f, INF)
        lo = k if lo is None else lo
        hi = k
        h_below = f
        k += 1
    return out, lo, hi


def column_grouped(h0, e, thr, dsde, de, k_start, k_end_geom, G=8):
    """The device's form: G lanes, lane g takes cell kb + g of every group of G; exclusive prefix minimum across the lanes, carry between groups, the run above the interval in groups as well."""
    out = {}
    lo, hi = None, None
    carry = BIG
    n = len(h0)
    k_main = k_start + n - 1
    for kb in range(k_start, k_main + 1, G):
        ks = [kb + g for g in range(G)]
        ins = [k <= k_main for k in ks]
        v = [(h0[k - k_start] - k * de) if i else BIG for k, i in zip(ks, ins)]
        ex = [min([BIG] + v[:g]) for g in range(G)]                       # exclusive prefix minimum across the lanes
        for g, (k, i) in enumerate(zip(ks, ins)):
            if not i:
                continue
            f = min(carry, ex[g]) + dsde + (k - 1) * de
            h = min(h0[k - k_start], f, INF)
            out[k] = (h, min(e[k - k_start], INF))
            if h <= thr:
                lo = k if lo is None else min(lo, k)
                hi = k if hi is None else max(hi, k)
        carry = min(carry, min(v))
    kb = k_main + 1
    while kb <= k_end_geom:
        live = []
        for g in range(G):
            k = kb + g
            f = carry + dsde + (k - 1) * de
            ok = k <= k_end_geom and f <= thr
            live.append(ok)
            if ok:
                out[k] = (f, INF)
        if not any(live):
            break
        lo = kb if lo is None else min(lo, kb)
        hi = kb + max(g for g in range(G) if live[g])
        if not all(live):
            break
        kb += G
    return out, lo, hi


def test_prefix_minimum_form_equals_the_sequential_recurrence():
    rng = np.random.default_rng(0xB0D)
    for trial in range(3000):
        de = int(rng.integers(1, 60))
        dsde = de + int(rng.integers(0, 120))                              # ds >= 0
        thr = int(rng.integers(0, 4000))
        n = int(rng.integers(1, 70))
        k_start = int(rng.integers(0, 300))
        k_end_geom = k_start + n - 1 + int(rng.integers(0, 40))
        # values around the budget, a few far above it, a few "beyond the budget" as the band stores them
        h0 = [int(x) for x in rng.integers(0, max(2, 2 * thr + 50), n)]
        for j in rng.integers(0, n, int(rng.integers(0, 4))):
            h0[int(j)] = INF
        e = [int(x) for x in rng.integers(0, INF + 200, n)]
        a, alo, ahi = column_sequential(h0, e, thr, dsde, de, k_start, k_end_geom)
        b, blo, bhi = column_grouped(h0, e, thr, dsde, de, k_start, k_end_geom)
        assert (alo, ahi) == (blo, bhi), (trial, alo, ahi, blo, bhi)
        for k, (h, ee) in a.items():
            # values within the budget are what decides; above it both forms only need "above" (the device clamps, it does not reset)
            if h <= thr:
                assert k in b and b[k] == (h, ee), (trial, k, a[k], b.get(k))
            else:
                assert k not in b or b[k][0] > thr, (trial, k, a[k], b.get(k))
        for k, (h, ee) in b.items():
            if h <= thr:
                assert k in a and a[k] == (h, ee), (trial, k, b[k], a.get(k))


# ---------------------------------------------------------------- the matcher's section tables (xm_extend.h, matcherIndexSectionEight; HashBlock_Matcher.java:40-77)
M_NO_MATCHES, M_MULTIPLE = -1, -2


def index_section_sequential(text, start, end, block_length):
    """The reference's loop on a text without ambiguous bases: an entry holds the position of a code that occurs once, M_MULTIPLE for one that occurs more often."""
    digit = {1: 0, 2: 1, 4: 2, 8: 3}
    table = {}
    for i in range(start, end):
        code = 0
        for b in range(block_length):
            code = code * 4 + digit[int(text[i + b])]
        table[code] = (i - start) if table.get(code, M_NO_MATCHES) == M_NO_MATCHES else M_MULTIPLE
    return table


def index_section_eight_lanes(text, start, end, block_length, G=8):
    """The device's order: lane g takes the g-th eighth of the section with a rolling code, a round handles one position per lane - entries read before any of the round's writes,
    codes that meet in the round marked by comparing across the lanes, then the writes."""
    digit = {1: 0, 2: 1, 4: 2, 8: 3}
    mask = (1 << (2 * block_length)) - 1
    table = {}
    count = max(end - start, 0)
    chunk = (count + G - 1) // G
    pos = [start + g * chunk for g in range(G)]
    stop = [min(p + chunk, end) for p in pos]
    code = [0] * G
    for g in range(G):
        if pos[g] < stop[g]:
            for b in range(block_length - 1):
                code[g] = code[g] * 4 + digit[int(text[pos[g] + b])]
    for r in range(chunk):
        enc = []
        for g in range(G):
            if pos[g] < stop[g]:
                code[g] = ((code[g] * 4) & mask) + digit[int(text[pos[g] + block_length - 1])]
                enc.append(code[g])
            else:
                enc.append(-1 - g)
        cur = [table.get(c, M_NO_MATCHES) if c >= 0 else 0 for c in enc]
        new = []
        for g in range(G):
            dup = any(enc[j] == enc[g] for j in range(G) if j != g)
            new.append(M_MULTIPLE if (dup or cur[g] != M_NO_MATCHES) else pos[g] - start)
        for g in range(G):
            if enc[g] >= 0:
                table[enc[g]] = new[g]
                pos[g] += 1
    return table


def test_section_table_does_not_depend_on_the_order_of_the_positions():
    rng = np.random.default_rng(0x5EC7)
    for trial in range(400):
        block_length = int(rng.integers(3, 8))
        n = int(rng.integers(0, 300))
        # few distinct bases now and then: many repeated codes, also inside one round
        alphabet = [1, 2, 4, 8] if rng.random() < 0.7 else [1, 2]
        text = rng.choice(alphabet, n + block_length + 4).astype(np.uint8)
        start = int(rng.integers(0, 3))
        end = start + n
        assert index_section_sequential(text, start, end, block_length) == index_section_eight_lanes(text, start, end, block_length), trial
