"""Deterministic synthetic inputs for the BASELINE.json configs (SURVEY.md §8d / BASELINE.md §4).

Reference: i.i.d. uniform ACGT from SplitMix64.  Reads: uniform start, fair strand, per-base substitution
rate, per-read indel probability (one indel of length 1-3, insertion/deletion fair), no N.
Everything is a pure function of the seed, so the GPU box regenerates exactly what the CPU container saw.
Bases are returned as 4-bit IUPAC codes, one per byte (A=1 C=2 G=4 T=8), the layout the C ABI takes.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n outputs of SplitMix64 started at `seed` (vectorised: state_i = seed + (i+1)*gamma)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


_CODES = np.array([1, 2, 4, 8], dtype=np.uint8)
_COMP = np.zeros(16, dtype=np.uint8)
for _b in range(16):
    _COMP[_b] = ((_b & 1) << 3) | ((_b & 2) << 1) | ((_b & 4) >> 1) | ((_b & 8) >> 3)


def synthetic_reference(length=5_000_000, seed=0xEC011):
    """`ecoli_syn`: one contig of i.i.d. uniform ACGT codes."""
    r = splitmix64(seed, length)
    return _CODES[(r >> np.uint64(62)).astype(np.int64)]


def revcomp_codes(codes):
    return _COMP[codes[::-1]]


def _mutate_reads(ref, starts, strand, read_len, r_sub, r_indel, sub_rate, indel_prob):
    """Builds reads of exactly read_len bases. Returns (codes[n, read_len])."""
    n = len(starts)
    # take read_len + 3 reference bases so that deletions still leave read_len bases
    span = read_len + 3
    idx = starts[:, None] + np.arange(span)[None, :]
    tmpl = ref[idx]  # [n, span]
    # substitutions: threshold on 32 random bits, replacement chosen among the 3 other bases
    sub_bits = (r_sub >> np.uint64(32)).astype(np.uint32).reshape(n, span)
    which = ((r_sub & np.uint64(0xFFFF)) % np.uint64(3)).astype(np.int64).reshape(n, span) + 1
    do_sub = sub_bits < np.uint32(int(sub_rate * 2**32))
    base_idx = np.log2(tmpl).astype(np.int64)
    new_idx = (base_idx + which) & 3
    tmpl = np.where(do_sub, _CODES[new_idx], tmpl)
    # indels: one per chosen read
    ri = r_indel.reshape(n, 4)
    has_indel = (ri[:, 0] >> np.uint64(32)).astype(np.uint32) < np.uint32(int(indel_prob * 2**32))
    indel_len = (ri[:, 1] % np.uint64(3)).astype(np.int64) + 1
    is_ins = (ri[:, 2] & np.uint64(1)).astype(bool)
    pos = (ri[:, 3] % np.uint64(max(read_len - 20, 1))).astype(np.int64) + 10
    out = tmpl[:, :read_len].copy()
    for L in (1, 2, 3):
        # deletions of length L: drop template bases [pos, pos+L)
        sel = np.nonzero(has_indel & ~is_ins & (indel_len == L))[0]
        if len(sel):
            col = np.arange(read_len)[None, :]
            src = np.where(col < pos[sel, None], col, col + L)
            out[sel] = np.take_along_axis(tmpl[sel], src, axis=1)
        # insertions of length L: insert L random bases (derived from ri[:,3]) at pos
        sel = np.nonzero(has_indel & is_ins & (indel_len == L))[0]
        if len(sel):
            col = np.arange(read_len)[None, :]
            p = pos[sel, None]
            src = np.where(col < p, col, np.maximum(col - L, 0))
            o = np.take_along_axis(tmpl[sel], src, axis=1)
            ins_bits = (ri[sel, 3] >> np.uint64(40))
            for k in range(L):
                b = _CODES[((ins_bits >> np.uint64(2 * k)) & np.uint64(3)).astype(np.int64)]
                m = (col == p + k)
                o = np.where(m, b[:, None], o)
            out[sel] = o
    rev = strand.astype(bool)
    if rev.any():
        out[rev] = _COMP[out[rev][:, ::-1]]
    return out


def synthetic_single_end(ref, n_reads, read_len=150, seed=0x5EED0001, sub_rate=0.01, indel_prob=0.05, chunk=200_000):
    """Config 2: n_reads x read_len single-end reads. Returns (codes [n, read_len] uint8, starts, strand)."""
    outs, all_starts, all_strand = [], [], []
    span = read_len + 3
    for c0 in range(0, n_reads, chunk):
        n = min(chunk, n_reads - c0)
        base = np.uint64(seed) + np.uint64(c0) * np.uint64(0x1000003)
        r0 = splitmix64(base, 2 * n)
        starts = (r0[:n] % np.uint64(len(ref) - span)).astype(np.int64)
        strand = (r0[n:] >> np.uint64(63)).astype(np.uint8)
        r_sub = splitmix64(base ^ np.uint64(0xA5A5A5A5), n * span)
        r_ind = splitmix64(base ^ np.uint64(0x5A5A5A5A), n * 4)
        outs.append(_mutate_reads(ref, starts, strand, read_len, r_sub, r_ind, sub_rate, indel_prob))
        all_starts.append(starts)
        all_strand.append(strand)
    return np.concatenate(outs), np.concatenate(all_starts), np.concatenate(all_strand)


def synthetic_paired_end(ref, n_pairs, read_len=150, seed=0x5EED0002, sub_rate=0.01, indel_prob=0.05, chunk=200_000):
    """Config 3: FR pairs, inner distance round(N(100, 30^2)) clipped to [-100, 400]; mate 2 is the reverse
    complement strand (Illumina FR).  Returns (mate1 [n, L], mate2 [n, L], starts1, inner, strand)."""
    m1s, m2s, st, inn, sd = [], [], [], [], []
    span = read_len + 3
    for c0 in range(0, n_pairs, chunk):
        n = min(chunk, n_pairs - c0)
        base = np.uint64(seed) + np.uint64(c0) * np.uint64(0x1000003)
        r0 = splitmix64(base, 4 * n)
        u1 = ((r0[:n] >> np.uint64(11)).astype(np.float64) + 0.5) / 2**53
        u2 = ((r0[n:2 * n] >> np.uint64(11)).astype(np.float64) + 0.5) / 2**53
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * np.pi * u2)
        inner = np.clip(np.rint(100 + 30 * z), -100, 400).astype(np.int64)
        frag = 2 * read_len + inner
        starts1 = (r0[2 * n:3 * n] % np.uint64(len(ref) - 2 * read_len - 400 - span)).astype(np.int64)
        strand = (r0[3 * n:] >> np.uint64(63)).astype(np.uint8)
        starts2 = starts1 + frag - read_len
        z8 = np.zeros(n, dtype=np.uint8)
        a = _mutate_reads(ref, starts1, z8, read_len, splitmix64(base ^ np.uint64(0x11), n * span), splitmix64(base ^ np.uint64(0x12), n * 4), sub_rate, indel_prob)
        b = _mutate_reads(ref, starts2, z8, read_len, splitmix64(base ^ np.uint64(0x21), n * span), splitmix64(base ^ np.uint64(0x22), n * 4), sub_rate, indel_prob)
        b = _COMP[b[:, ::-1]]  # mate 2 is sequenced from the opposite strand
        rev = strand.astype(bool)
        # a fragment from the reverse strand swaps the roles: mate1 = rc(right end), mate2 = left end as-is
        a2 = np.where(rev[:, None], b, a)
        b2 = np.where(rev[:, None], a, b)
        m1s.append(a2); m2s.append(b2); st.append(starts1); inn.append(inner); sd.append(strand)
    return np.concatenate(m1s), np.concatenate(m2s), np.concatenate(st), np.concatenate(inn), np.concatenate(sd)
