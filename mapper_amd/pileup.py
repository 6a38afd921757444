"""Pile-up of alignments on the reference and the mutations file.

Mapper.run hands every batch of QueryAlignments to a MatchDatabase (Mapper.java:700-708) and, when the run is over, writes
`--out-mutations` / `--out-vcf` from MatchDatabase.groupByPosition() (Mapper.java:758-785).  MatchDatabase, Alignments, AlignmentPosition,
MutationsWriter and MutationDetectionParameters live in the un-vendored QuickVariants module; what this file reproduces is what the
reference's own tests pin: src/test/java/MatchDatabase_Test.java:12-69 (every aligned reference base counts 1; the two mates of a pair count
1 together where they overlap) and src/test/java/MutationsWriter_Test.java:18-134 (the line format `contig, 1-based position, reference
allele, query allele, allele depth, total depth`; a deletion is reported at its first base with the deleted bases against dashes, an insertion
at the base before it with dashes against the inserted bases; consecutive substitutions are separate lines; the total-depth filter).
--distinguish-query-ends (MatchDatabase(queryEndFraction), Mapper.java:76,351-353,700): "when detecting indels, only consider the middle of each query"
(Mapper.java:532) - the indel thresholds look at the *middle* depth, the depth from query bases that are not within the fraction of the query's
length of either query end (Mapper.java:538-542 "total (middle) depth"), and an indel near a query end does not support itself; pinned by
MutationsWriter_Test.java:114-134 (fraction 0.5 = every base is near an end: the insertion is not reported once a middle depth of 1 is asked for).
[unpinned]: the header lines (the reference's test strips lines that start with '#' or 'CHR'), the weight of a query with several
alignments (taken as 1/n each), how ambiguous bases count (depth only), the order of lines at one position, where exactly "near the end" stops
(here: query index k with k < f * length or k >= length - f * length), the supporting-depth fractions and the continuation thresholds
(Mapper.java:208-230: taken as alt / total; an indel is cut where its continuation fails).

The accumulation itself runs on the GPU (xm_pileup_kernel: one lane per query walks its result stream in HBM and adds to per-position
integer counters, so the result does not depend on the order of the atomic adds); this module holds the host side."""
import ctypes as C

import numpy as np

from . import _capi
from .api import decode, reverse_complement

UNIT = 1441440  # XM_PILEUP_UNIT (include/xmapper_hip.h)


class MutationDetectionParameters:
    """MutationDetectionParameters [QuickVariants]: the six thresholds Mapper.main fills in (Mapper.java:208-230).  emptyFilter() = everything is
    reported (what the reference's tests start from); defaultFilter() = the defaults --out-mutations documents (Mapper.java:534-542)."""

    def __init__(self, minSNPTotalDepth=0.0, minSNPDepthFraction=0.0, minIndelTotalStartDepth=0.0, minIndelStartDepthFraction=0.0,
                 minIndelContinuationTotalDepth=0.0, minIndelContinuationDepthFraction=0.0):
        # Mapper.main narrows every threshold to float - (float)Double.parseDouble(...), Mapper.java:203-230 - so they are kept as float32 values here and
        # the products with the depths are taken in float32 too (_below): 0.7 of a depth of 10 is 7.0 in float and 7.000000000000001 in double, and an
        # allele seen 7 times must not be dropped by the latter.  [inferred: the fields of MutationDetectionParameters live in QuickVariants]
        f32 = lambda x: float(np.float32(x))  # noqa: E731
        self.minSNPTotalDepth, self.minSNPDepthFraction = f32(minSNPTotalDepth), f32(minSNPDepthFraction)
        self.minIndelTotalStartDepth, self.minIndelStartDepthFraction = f32(minIndelTotalStartDepth), f32(minIndelStartDepthFraction)
        self.minIndelContinuationTotalDepth, self.minIndelContinuationDepthFraction = f32(minIndelContinuationTotalDepth), f32(minIndelContinuationDepthFraction)

    @staticmethod
    def emptyFilter():
        return MutationDetectionParameters()

    @staticmethod
    def defaultFilter():
        return MutationDetectionParameters(5.0, 0.9, 1.0, 0.8, 1.0, 0.7)


def _below(support, fraction, total):
    """support < fraction * total in float arithmetic (the thresholds are floats in the reference: Mapper.java:203-230)."""
    return bool(np.float32(support) < np.float32(fraction) * np.float32(total))


def _number(x):
    """Depths are sums of 1/n: whole numbers print without a fraction (MutationsWriter_Test: `1`), others like a Java float."""
    return str(int(x)) if float(x).is_integer() else repr(float(np.float32(x)))


class MatchDatabase:
    """MatchDatabase(queryEndFraction) + groupByPosition() for one ReferenceDatabase (or the replicas of a MultiGpuDatabase: one device pile-up
    each, summed on the host in replica order)."""

    def __init__(self, databases, query_end_fraction=0.0):
        self.dbs = list(databases) if isinstance(databases, (list, tuple)) else [databases]
        self.query_end_fraction = float(query_end_fraction)
        self._L = _capi.lib()
        self._h = []
        for db in self.dbs:
            h = C.c_void_p()
            if self._L.xm_pileup_new(db._h, C.byref(h)):
                raise RuntimeError(self._L.xm_last_error().decode())
            self._h.append(h)
            if self._L.xm_pileup_set_query_ends(h, self.query_end_fraction):
                raise RuntimeError(self._L.xm_last_error().decode())
        self.contigs = self.dbs[0].contigs
        self._reads = {}   # query ordinal (per replica) -> mates, kept only for queries with an insertion
        self._batches = [[] for _ in self.dbs]

    def add_last(self, queries=None, replica=0):
        """addAlignments(List<QueryAlignments>) for the batch replica `replica` aligned last (its streams are still in HBM).  `queries`: the
        batch's Query objects or mate arrays, needed for the text of insertions (None: insertions are reported by length only)."""
        n = C.c_int64(0)
        if self._L.xm_pileup_add_last(self._h[replica], C.byref(n)):
            raise RuntimeError(self._L.xm_last_error().decode())
        self._batches[replica].append(queries)
        return n.value

    def close(self):
        for h in self._h:
            self._L.xm_pileup_free(h)
        self._h = []

    def _sum(self, contig):
        n = len(self.contigs[contig][1])
        depth = np.zeros(n, np.uint64)
        alt = np.zeros((4, n), np.uint64)
        for h in self._h:  # fixed replica order (integer sums: any order gives the same, the order is kept for the record)
            d = np.zeros(n, np.uint64)
            a = np.zeros((4, n), np.uint64)
            if self._L.xm_pileup_read(h, contig, 0, n, d.ctypes.data, a.ctypes.data):
                raise RuntimeError(self._L.xm_last_error().decode())
            depth += d
            alt += a
        return depth, alt

    def depth(self, contig):
        """AlignmentPosition.getCount() for every position of the contig."""
        return self._sum(contig)[0].astype(np.float64) / UNIT

    def _middle(self, contig):
        """The depth from query bases away from the query ends (all of it without a query-end fraction), summed over the replicas."""
        n = len(self.contigs[contig][1])
        mid = np.zeros(n, np.uint64)
        for h in self._h:
            d = np.zeros(n, np.uint64)
            if self._L.xm_pileup_read_middle(h, contig, 0, n, d.ctypes.data):
                raise RuntimeError(self._L.xm_last_error().decode())
            mid += d
        return mid

    def _events(self):
        """(replica, contig, startB, type, length, query ordinal, mate | reversed << 1, startA, weight) of every insertion / deletion block."""
        out = []
        buf = np.zeros((1 << 16, 8), np.int64)
        for r, h in enumerate(self._h):
            first = 0
            while True:
                m = self._L.xm_pileup_events(h, first, len(buf), buf.ctypes.data)
                if m <= 0:
                    break
                out += [(r,) + tuple(int(x) for x in row) for row in buf[:m]]
                first += m
        return out

    def _mate(self, replica, ordinal, mate):
        at = 0
        for qs in self._batches[replica]:
            if qs is None:
                return None
            if ordinal < at + len(qs):
                q = qs[ordinal - at]
                seqs = q.sequences if hasattr(q, "sequences") else q
                return np.asarray(seqs[mate], dtype=np.uint8)
            at += len(qs)
        return None

    def mutations(self, parameters=None):
        """-> list of (contig index, 1-based position, reference allele, query allele, allele depth, total depth), in contig and position order."""
        f = parameters or MutationDetectionParameters.emptyFilter()
        rows = []
        indels = {}
        for r, contig, pos, kind, length, ordinal, flags, start_a, weight in self._events():
            if flags & 4:   # near a query end: "when detecting indels, only consider the middle of each query" (Mapper.java:532)
                continue
            ref = self.contigs[contig][1]
            if kind == 1:
                mate = self._mate(r, ordinal, flags & 1)
                if mate is None:
                    text = "N" * length
                else:
                    oriented = reverse_complement(mate) if (flags >> 1) & 1 else mate
                    text = decode(oriented[start_a:start_a + length])
                key = (contig, pos, 1, "-" * length, text)   # reported at the base before the insertion: 1-based position = startB
            else:
                key = (contig, pos + 1, 2, decode(ref[pos:pos + length]), "-" * length)
            indels[key] = indels.get(key, 0) + weight
        for c in range(len(self.contigs)):
            depth, alt = self._sum(c)
            mid = self._middle(c) if indels else depth
            ref = self.contigs[c][1]
            for b, letter in enumerate("ACGT"):
                for pos in np.nonzero(alt[b])[0]:
                    total = depth[pos] / UNIT
                    support = alt[b][pos] / UNIT
                    if total < f.minSNPTotalDepth or _below(support, f.minSNPDepthFraction, total):
                        continue
                    rows.append((c, int(pos) + 1, 0, decode(ref[pos:pos + 1]), letter, support, total))
            for (cc, pos1, kind, ra, qa), w in indels.items():
                if cc != c:
                    continue
                support = w / UNIT
                at = min(max(pos1 - 1, 0), len(ref) - 1)
                total = mid[at] / UNIT     # the total (middle) depth where the indel starts (Mapper.java:538-539)
                if total < f.minIndelTotalStartDepth or _below(support, f.minIndelStartDepthFraction, total):
                    continue
                # continuation (Mapper.java:541-542) [inferred]: every further base of the indel is judged where it lies (a deletion's bases on the
                # reference, an insertion's at its one position); the indel is cut in front of the first base that fails
                n = max(len(ra), len(qa))
                keep = 1
                while keep < n:
                    here = min(at + keep, len(ref) - 1) if kind == 2 else at
                    t = mid[here] / UNIT
                    if t < f.minIndelContinuationTotalDepth or _below(support, f.minIndelContinuationDepthFraction, t):
                        break
                    keep += 1
                rows.append((c, pos1, kind, ra[:keep], qa[:keep], support, total))
        rows.sort(key=lambda t: (t[0], t[1], t[2], t[3], t[4]))
        return [(c, pos1, ra, qa, w, total) for c, pos1, _, ra, qa, w, total in rows]

    def write_mutations(self, out, parameters=None):
        """MutationsWriter.write: one line per mutation.  The two header lines are [unpinned] (the reference's test drops '#' and 'CHR' lines)."""
        out.write("# mutations of the aligned queries against the reference\n")
        out.write("CHR\tPOS\tREF\tALT\tALT DEPTH\tTOTAL DEPTH\n")
        for c, pos1, ra, qa, w, total in self.mutations(parameters):
            out.write("%s\t%d\t%s\t%s\t%s\t%s\n" % (self.contigs[c][0], pos1, ra, qa, _number(w), _number(total)))
