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


def _parallel(fn, items, workers=None):
    """fn over items on a few threads (numpy releases the GIL in the array passes these generators are made of)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    workers = workers or max(1, min(16, (os.cpu_count() or 2) // 2))
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(fn, items))


def synthetic_single_end(ref, n_reads, read_len=150, seed=0x5EED0001, sub_rate=0.01, indel_prob=0.05, chunk=200_000, at=None):
    """Config 2: n_reads x read_len single-end reads. Returns (codes [n, read_len] uint8, starts, strand).
    at: template starts to use instead of drawing them (genome_wide_starts: reads sampled over a many-contig reference laid out in one array)."""
    span = read_len + 3

    def one(c0):  # (a chunk is a pure function of the seed and its offset: the chunks are made on a few threads)
        n = min(chunk, n_reads - c0)
        base = np.uint64(seed) + np.uint64(c0) * np.uint64(0x1000003)
        r0 = splitmix64(base, 2 * n)
        starts = (r0[:n] % np.uint64(len(ref) - span)).astype(np.int64) if at is None else np.asarray(at[c0:c0 + n], dtype=np.int64)
        strand = (r0[n:] >> np.uint64(63)).astype(np.uint8)
        r_sub = splitmix64(base ^ np.uint64(0xA5A5A5A5), n * span)
        r_ind = splitmix64(base ^ np.uint64(0x5A5A5A5A), n * 4)
        return _mutate_reads(ref, starts, strand, read_len, r_sub, r_ind, sub_rate, indel_prob), starts, strand
    parts = _parallel(one, range(0, n_reads, chunk)) if n_reads > chunk else [one(0)]
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])


def synthetic_paired_end(ref, n_pairs, read_len=150, seed=0x5EED0002, sub_rate=0.01, indel_prob=0.05, chunk=200_000, at=None):
    """Config 3: FR pairs, inner distance round(N(100, 30^2)) clipped to [-100, 400]; mate 2 is the reverse
    complement strand (Illumina FR).  Returns (mate1 [n, L], mate2 [n, L], starts1, inner, strand).
    at: fragment starts to use instead of drawing them (a fragment spans at most 2 * read_len + 400 + 3 bases)."""
    span = read_len + 3

    def one(c0):
        n = min(chunk, n_pairs - c0)
        base = np.uint64(seed) + np.uint64(c0) * np.uint64(0x1000003)
        r0 = splitmix64(base, 4 * n)
        u1 = ((r0[:n] >> np.uint64(11)).astype(np.float64) + 0.5) / 2**53
        u2 = ((r0[n:2 * n] >> np.uint64(11)).astype(np.float64) + 0.5) / 2**53
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * np.pi * u2)
        inner = np.clip(np.rint(100 + 30 * z), -100, 400).astype(np.int64)
        frag = 2 * read_len + inner
        starts1 = (r0[2 * n:3 * n] % np.uint64(len(ref) - 2 * read_len - 400 - span)).astype(np.int64) if at is None else np.asarray(at[c0:c0 + n], dtype=np.int64)
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
        return a2, b2, starts1, inner, strand
    parts = _parallel(one, range(0, n_pairs, chunk)) if n_pairs > chunk else [one(0)]
    return tuple(np.concatenate([p[k] for p in parts]) for k in range(5))


# ---------------------------------------------------------------- configs[3] / configs[4]: the GRCh38-shaped reference of SURVEY.md section 8(d)
# chr1..chr22, chrX, chrY of GRCh38 (primary assembly): 3,088,269,832 bases in 24 contigs
GRCH38_LENGTHS = (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309,
                  114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415)
GRCH38_NAMES = tuple("chr%d" % i for i in range(1, 23)) + ("chrX", "chrY")


def grch38_shaped_reference(scale=1.0, seed=0x6C38, n_fraction=0.01, n_run=10_000, chunk=50_000_000):
    """24 contigs with the real chromosome lengths (times `scale`, for tests that want the shape at a size an
    oracle can hash), i.i.d. uniform ACGT, and `n_fraction` of the positions in runs of `n_run` N (one run at a seeded place in each of a contig's
    equal segments).  Returns (contigs, whole, starts, runs): contigs = [(name, view into whole)], starts[c] = offset of contig c in `whole`,
    runs[c] = sorted starts of the N-runs of contig c.  `whole` is one array so that reads can be sampled genome-wide with one gather."""
    lengths = [max(int(n * scale), 4 * n_run if n_fraction > 0 else 1000) for n in GRCH38_LENGTHS]
    total = int(sum(lengths))
    whole = np.empty(total, dtype=np.uint8)
    starts = np.zeros(len(lengths) + 1, dtype=np.int64)
    starts[1:] = np.cumsum(lengths)
    def fill(o):
        n = min(chunk, total - o)
        whole[o:o + n] = synthetic_reference(n, seed=seed + 0x9E37 * (o // chunk))
    _parallel(fill, range(0, total, chunk))
    contigs, runs = [], []
    for c, n in enumerate(lengths):
        view = whole[starts[c]:starts[c + 1]]
        k = int(round(n * n_fraction / n_run)) if n_fraction > 0 else 0
        rs = np.zeros(k, dtype=np.int64)
        if k > 0:
            seg = n // k
            r = splitmix64(seed ^ (0xA11CE + 7919 * c), k)
            rs = np.arange(k, dtype=np.int64) * seg + (r % np.uint64(max(seg - n_run, 1))).astype(np.int64)
            for s in rs:
                view[s:s + n_run] = 15
        contigs.append((GRCH38_NAMES[c], view))
        runs.append(rs)
    return contigs, whole, starts, runs


def genome_wide_starts(starts, runs, n, span, seed, n_run=10_000):
    """n template starts (global offsets into `whole`) drawn uniformly over the genome; a template of `span` bases that would cross a contig end or
    touch an N-run is moved to the first place behind the obstacle where it fits (reads come from sequenced DNA: no N inside them).
    Returns (global starts, contig of each, start within the contig)."""
    total = int(starts[-1])
    g = (splitmix64(seed, n) % np.uint64(total)).astype(np.int64)
    contig = np.searchsorted(starts, g, side="right") - 1
    local = g - starts[contig]
    for c in np.unique(contig):
        sel = np.nonzero(contig == c)[0]
        clen = int(starts[c + 1] - starts[c])
        x = np.minimum(local[sel], clen - span)
        rs = runs[c]
        for _ in range(3):  # (moving behind one run can land on nothing else: runs are a segment apart; the loop is belt and braces)
            if len(rs) == 0:
                break
            j = np.searchsorted(rs, x + span - 1, side="right") - 1   # last run starting at or before the template's end
            hit = (j >= 0) & (rs[np.maximum(j, 0)] + n_run > x)
            x = np.where(hit, rs[np.maximum(j, 0)] + n_run, x)
            over = x > clen - span
            x = np.where(over, np.maximum(rs[np.maximum(j, 0)] - span, 0), x)
        local[sel] = x
    return starts[contig] + local, contig, local


def synthetic_long_reads(whole, gstarts, read_len=10_000, seed=0x5EED0004, sub_rate=0.05, indel_rate=0.05, strand=None):
    """configs[4]'s long reads: per base a substitution with probability sub_rate and an indel EVENT (insertion or deletion, fair, of 1-3 bases)
    with probability indel_rate.  (SURVEY.md section 8(d) says "5 % substitution + 5 % indel (long-read-like)"; read per base, as here, such reads carry
    ~15 penalty units per 100 bases against the default --max-penalty of 10 per 100, so the reference aligns almost none of them: the tests use these
    rates for the config as stated and milder ones to see reads come home.)  Template = whole[g : g + read_len * 1.25]; returns codes [n, read_len]."""
    n = len(gstarts)
    span = read_len + read_len // 4 + 8
    out = np.empty((n, read_len), dtype=np.uint8)
    thr_sub = np.uint32(min(int(sub_rate * 2**32), 2**32 - 1))
    thr_indel = np.uint32(min(int(indel_rate * 2**32), 2**32 - 1))
    gstarts = np.asarray(gstarts, dtype=np.int64)
    batch = 128

    def some(b0):  # `batch` reads at once, their templates laid end to end (every pass below is one array operation over all of them)
        m = min(batch, n - b0)
        t = whole[(gstarts[b0:b0 + m, None] + np.arange(span)[None, :]).reshape(-1)]
        T = len(t)
        r = splitmix64(np.uint64(seed) + np.uint64(b0) * np.uint64(0x1000003), 2 * T)
        ra, rb = r[:T], r[T:]
        do_sub = ((ra >> np.uint64(32)).astype(np.uint32) < thr_sub) & (t != 15)
        which = ((ra & np.uint64(0xFFFF)) % np.uint64(3)).astype(np.int64) + 1
        idx = np.log2(np.where(t == 15, 1, t)).astype(np.int64)
        t = np.where(do_sub, _CODES[(idx + which) & 3], t)
        ev = (rb >> np.uint64(32)).astype(np.uint32) < thr_indel
        is_ins = ((rb >> np.uint64(8)) & np.uint64(1)).astype(bool)
        ln = ((rb >> np.uint64(9)) % np.uint64(3)).astype(np.int64) + 1
        ins = np.where(ev & is_ins, ln, 0)          # bases inserted in front of template base j
        dele = np.zeros(T + 4, dtype=np.int64)      # a deletion event at j drops bases j .. j + ln - 1 (the tail of a template is never reached)
        for L in (1, 2, 3):
            at = np.nonzero(ev & ~is_ins & (ln == L))[0]
            for k in range(L):
                dele[at + k] = 1
        count = ins + (1 - dele[:T])
        ends = np.cumsum(count)
        src = np.repeat(np.arange(T), count)
        within = np.arange(len(src)) - (ends - count)[src]
        inserted = within < ins[src]
        rnd = _CODES[((rb[src] >> (np.uint64(12) + np.uint64(2) * within.astype(np.uint64))) & np.uint64(3)).astype(np.int64)]
        seq = np.where(inserted, rnd, t[src])
        first = np.concatenate([[0], ends[span - 1::span][:-1]])  # where each read's bases start in `seq`
        if ((ends[span - 1::span] - first) < read_len).any():
            raise ValueError("template too short for the requested indel rate")
        out[b0:b0 + m] = seq[first[:, None] + np.arange(read_len)[None, :]]
    _parallel(some, range(0, n, batch))
    if strand is not None:
        rev = strand.astype(bool)
        out[rev] = _COMP[out[rev][:, ::-1]]
    return out


# ---------------------------------------------------------------- a repeat-rich reference (round 4)
def repeat_rich_reference(length=5_000_000, seed=0x4E9EA7, n_segdups=100, n_tandem=220, n_hot=240, stats=None):
    """A reference with the structure a real genome has and i.i.d. ACGT lacks: the branch of the path that a duplicated window sends a read into
    (no early accept - AlignerWorker.java:494-587 via Readable_DuplicationDetector.java:28-47 -, every candidate enumerated, overfull buckets skipped -
    HashBlock_Database.java:569-577).  On an i.i.d. background:
      * segmental duplications: n_segdups source segments of 1-20 kb, each copied to 1-3 other places at 90-99.5 % identity (substitutions), a third of the
        copies reverse-complemented;
      * tandem repeats: n_tandem loci of a 2-60 bp unit repeated 5-50 times, 2 % of the unit copies carrying a substitution;
      * one 28-mer planted at n_hot places (its buckets overflow max(L^2, 5) entries and are marked overfull).
    Pure function of the arguments.  stats (a dict, optional) receives the bases written by each kind and the fraction of positions that lie in a
    segment present at least twice."""
    ref = synthetic_reference(length, seed=seed).copy()
    covered = np.zeros(length, dtype=bool)
    r = splitmix64(seed ^ 0xD0B1E, 16 * (n_segdups + n_tandem + n_hot) + 64)
    k = 0

    def nxt():
        nonlocal k
        v = int(r[k]); k += 1
        return v
    dup_bases = 0
    for _ in range(n_segdups):
        seg_len = 1000 + nxt() % 19001
        src = nxt() % (length - seg_len)
        copies = 1 + nxt() % 3
        covered[src:src + seg_len] = True
        for _c in range(copies):
            dst = nxt() % (length - seg_len)
            ident_ppm = 900_000 + nxt() % 95_001          # 90 % .. 99.5 %
            rc = (nxt() % 3) == 0
            seg = ref[src:src + seg_len].copy()
            rs = splitmix64(np.uint64(nxt() & 0x7FFFFFFFFFFFFFFF), seg_len)
            do_sub = (rs >> np.uint64(44)).astype(np.int64) < (1_000_000 - ident_ppm) * (1 << 20) // 1_000_000
            which = ((rs & np.uint64(0xFFFF)) % np.uint64(3)).astype(np.int64) + 1
            seg = np.where(do_sub, _CODES[(np.log2(seg).astype(np.int64) + which) & 3], seg)
            if rc:
                seg = _COMP[seg[::-1]]
            ref[dst:dst + seg_len] = seg
            covered[dst:dst + seg_len] = True
            dup_bases += seg_len
    tandem_bases = 0
    for _ in range(n_tandem):
        unit = 2 + nxt() % 59
        reps = 5 + nxt() % 46
        at = nxt() % (length - unit * reps)
        u = _CODES[(splitmix64(np.uint64(nxt() & 0x7FFFFFFFFFFFFFFF), unit) >> np.uint64(62)).astype(np.int64)]
        t = np.tile(u, reps)
        rs = splitmix64(np.uint64(nxt() & 0x7FFFFFFFFFFFFFFF), len(t))
        do_sub = (rs >> np.uint64(44)).astype(np.int64) < (1 << 20) // 50
        which = ((rs & np.uint64(0xFFFF)) % np.uint64(3)).astype(np.int64) + 1
        t = np.where(do_sub, _CODES[(np.log2(t).astype(np.int64) + which) & 3], t)
        ref[at:at + len(t)] = t
        covered[at:at + len(t)] = True
        tandem_bases += len(t)
    hot = _CODES[(splitmix64(np.uint64(seed ^ 0x407), 28) >> np.uint64(62)).astype(np.int64)]
    for _ in range(n_hot):
        at = nxt() % (length - 28)
        ref[at:at + 28] = hot
    if stats is not None:
        stats.update(segdup_copy_bases=int(dup_bases), tandem_bases=int(tandem_bases), hot_kmer_copies=int(n_hot), fraction_in_repeats=float(covered.mean()))
    return ref


# ---------------------------------------------------------------- the GRCh38 shape with the repeat structure of a genome (round 6)
def grch38_repeat_rich_reference(scale=1.0, seed=0x6C38, n_families=6, copies=120_000, n_segdups=2_000, n_tandem=50_000, n_hot=5_000, stats=None):
    """The GRCh38-shaped reference of SURVEY.md section 8(d) (grch38_shaped_reference: 24 contigs with the real length ratios, 1 % of the positions in N-runs of
    10 kb) with what a real genome has and i.i.d. ACGT lacks, at genome scale (counts are for scale 1.0 and follow `scale`):
      * interspersed repeats: n_families consensus sequences of 300 bp, `copies` copies in all (family sizes 1 : 2 : 3 ...), each at 85-95 % identity to its
        consensus (substitutions), either strand, anywhere in the genome - the Alu-like families whose k-mers fill buckets past max(L^2, 5) entries
        (HashBlock_Database.java:569-577) and whose copies are what DuplicationDetector.java:129-250 has to tell from unique sequence;
      * segmental duplications: n_segdups segments of 1-50 kb copied to 1-2 other places (any contig) at 95-99.5 % identity, a third reverse-complemented;
      * tandem repeats: n_tandem loci of a 2-60 bp unit repeated 5-50 times, 2 % of the unit copies with a substitution;
      * one 28-mer at n_hot places.
    Returns (contigs, whole, starts, runs) like grch38_shaped_reference; the N-runs are put back last (a repeat never overwrites one).  Pure function of the
    arguments.  stats (a dict, optional): bases written by each kind, fraction of the positions in a repeat."""
    contigs, whole, starts, runs = grch38_shaped_reference(scale=scale, seed=seed)
    total = len(whole)
    covered = np.zeros(total, dtype=bool)
    rng = np.random.default_rng(seed ^ 0x3AE9)

    def mutate(seg, rate_ppm, r):
        do_sub = (r >> np.uint64(44)).astype(np.int64) < np.asarray(rate_ppm, dtype=np.int64) * (1 << 20) // 1_000_000
        which = ((r & np.uint64(0xFFFF)) % np.uint64(3)).astype(np.int64) + 1
        return np.where(do_sub & (seg != 15), _CODES[(np.log2(np.where(seg == 15, 1, seg)).astype(np.int64) + which) & 3], seg)

    # interspersed repeats, all copies of a family at once
    n_copies = max(int(copies * scale), 20 * n_families)
    fam_len = 300
    weights = np.arange(1, n_families + 1, dtype=np.float64)
    per_family = np.maximum((n_copies * weights / weights.sum()).astype(np.int64), 2)
    family_bases = 0
    for f in range(n_families):
        cons = _CODES[(splitmix64(np.uint64(seed ^ (0xFA111 + f)), fam_len) >> np.uint64(62)).astype(np.int64)]
        k = int(per_family[f])
        at = rng.integers(0, total - fam_len, k)
        ident_ppm = rng.integers(850_000, 950_001, k)
        rs = splitmix64(np.uint64(seed ^ (0xC0B1E5 + f)), k * fam_len).reshape(k, fam_len)
        cp = mutate(np.broadcast_to(cons, (k, fam_len)), (1_000_000 - ident_ppm)[:, None], rs)
        rc = rng.random(k) < 0.5
        cp[rc] = _COMP[cp[rc][:, ::-1]]
        idx = at[:, None] + np.arange(fam_len)[None, :]
        whole[idx.reshape(-1)] = cp.reshape(-1)   # (later copies overwrite earlier ones where they overlap)
        covered[idx.reshape(-1)] = True
        family_bases += k * fam_len
    # segmental duplications
    dup_bases = 0
    for _ in range(max(int(n_segdups * scale), 10)):
        seg_len = int(rng.integers(1000, 50_001))
        if seg_len * 4 > total:
            seg_len = total // 8
        src = int(rng.integers(0, total - seg_len))
        covered[src:src + seg_len] = True
        for _c in range(int(rng.integers(1, 3))):
            dst = int(rng.integers(0, total - seg_len))
            seg = mutate(whole[src:src + seg_len].copy(), 1_000_000 - int(rng.integers(950_000, 995_001)), splitmix64(np.uint64(rng.integers(0, 2**62)), seg_len))
            if rng.random() < 1 / 3:
                seg = _COMP[seg[::-1]]
            whole[dst:dst + seg_len] = seg
            covered[dst:dst + seg_len] = True
            dup_bases += seg_len
    # tandem repeats
    tandem_bases = 0
    for _ in range(max(int(n_tandem * scale), 50)):
        unit, reps = int(rng.integers(2, 61)), int(rng.integers(5, 51))
        at = int(rng.integers(0, total - unit * reps))
        t = np.tile(_CODES[rng.integers(0, 4, unit)], reps)
        t = mutate(t, 20_000, splitmix64(np.uint64(rng.integers(0, 2**62)), len(t)))
        whole[at:at + len(t)] = t
        covered[at:at + len(t)] = True
        tandem_bases += len(t)
    hot = _CODES[(splitmix64(np.uint64(seed ^ 0x407), 28) >> np.uint64(62)).astype(np.int64)]
    k_hot = max(int(n_hot * scale), 60)
    at = rng.integers(0, total - 28, k_hot)
    whole[(at[:, None] + np.arange(28)[None, :]).reshape(-1)] = np.tile(hot, k_hot)
    # contig ends and N-runs as they were: a repeat that ran over a contig end is cut there (the arrays are views of `whole`), the N-runs are stamped again
    for c in range(len(contigs)):
        for s in runs[c]:
            whole[starts[c] + s:starts[c] + s + 10_000] = 15
    if stats is not None:
        stats.update(interspersed_copies=int(per_family.sum()), interspersed_bases=int(family_bases), segdup_copy_bases=int(dup_bases), tandem_bases=int(tandem_bases),
                     hot_kmer_copies=int(k_hot), fraction_in_repeats=float(covered.mean()))
    return contigs, whole, starts, runs
