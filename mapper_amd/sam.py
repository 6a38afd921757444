"""SAM records of the result types, in the format the reference's SamWriter produces.

The writer itself lives in the un-vendored QuickVariants module; its record format is pinned only by the five cases of
src/test/java/SamWriter_Test.java:18-94 (flags 0 / 99 / 147 / 73, MAPQ 255, column 9 = read length, RNEXT = contig name,
mate 2 printed as aligned, `cs:f:` only for paired queries, `AS:f:` penalty).  Everything those cases do not show
(reverse-strand single-end flag 16, soft clips for unaligned ends, the value behind cs:f:) follows the SAM specification
and is marked [unpinned].  This formatter is host-side plumbing for the standalone harness; in the drop-in deployment the
Java SamWriter keeps writing the records from the QueryAlignments the C ABI returns.
"""
from .api import decode, reverse_complement


def java_double(x):
    """Double.toString: the shortest decimal that reads back as the same double (what repr() gives), in plain notation with at least one fractional digit for
    1e-3 <= |x| < 1e7 and as d.dddE<n> otherwise."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Infinity" if x > 0 else "-Infinity"
    if x == 0:
        return "0.0"
    a = abs(float(x))
    r = repr(a)
    mant, _, e = r.partition("e")
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    exp10 = (int(e) if e else 0) + len(ip) - 1 - (len(ip + fp) - len((ip + fp).lstrip("0")) if ip.strip("0") == "" else 0)
    digits = digits.rstrip("0") or "0"
    sign = "-" if x < 0 else ""
    if 1e-3 <= a < 1e7:
        if exp10 >= 0:
            whole = digits[:exp10 + 1].ljust(exp10 + 1, "0")
            frac = digits[exp10 + 1:] or "0"
            return sign + whole + "." + frac
        return sign + "0." + "0" * (-exp10 - 1) + digits
    return "%s%s.%sE%d" % (sign, digits[0], digits[1:] or "0", exp10)


def cigar(seq_al, query_len):
    parts = []
    first, last = seq_al.sections[0], seq_al.sections[-1]
    if first.startA > 0:
        parts.append("%dS" % first.startA)  # [unpinned]
    for b in seq_al.sections:
        if b.lengthA == b.lengthB:
            op, n = "M", b.lengthA
        elif b.lengthB == 0:
            op, n = "I", b.lengthA
        else:
            op, n = "D", b.lengthB
        if parts and parts[-1].endswith(op):
            parts[-1] = "%d%s" % (int(parts[-1][:-1]) + n, op)
        else:
            parts.append("%d%s" % (n, op))
    tail = query_len - (last.startA + last.lengthA)
    if tail > 0:
        parts.append("%dS" % tail)  # [unpinned]
    return "".join(parts)


def records(query, query_alignments, contig_names):
    """SAM lines (no header) for one Query and its QueryAlignments (list of components, each a list of QueryAlignment)."""
    paired = len(query.sequences) > 1
    lines = []
    if len(query_alignments) == 1:
        for al in query_alignments[0]:
            for k, sa in enumerate(al.components):
                mate = k
                name = query.names[mate]
                seq = query.sequences[mate]
                shown = reverse_complement(seq) if sa.reference_reversed else seq
                if paired:
                    other = al.components[1 - k]
                    flag = 1 | 2 | (0x10 if sa.reference_reversed else 0) | (0x20 if other.reference_reversed else 0) | (0x40 if k == 0 else 0x80)
                    rnext, pnext = contig_names[other.contig], other.start_index_b() + 1
                else:
                    flag = 0x10 if sa.reference_reversed else 0  # 16 is [unpinned]
                    rnext, pnext = "*", 0
                fields = [name, str(flag), contig_names[sa.contig], str(sa.start_index_b() + 1), "255", cigar(sa, len(seq)), rnext, str(pnext), str(len(seq)),
                          decode(shown), "*"]
                if paired:
                    fields.append("cs:f:" + java_double(al.spacing_penalty))
                fields.append("AS:f:" + java_double(al.penalty))
                lines.append("\t".join(fields))
    else:  # a pair that fell back to unpaired alignments (AlignerWorker.java:602-644): one component per mate
        for mate, comp in enumerate(query_alignments):
            for al in comp:
                sa = al.components[0]
                seq = query.sequences[mate]
                shown = reverse_complement(seq) if sa.reference_reversed else seq
                flag = 1 | 8 | (0x10 if sa.reference_reversed else 0) | (0x40 if mate == 0 else 0x80)
                fields = [query.names[mate], str(flag), contig_names[sa.contig], str(sa.start_index_b() + 1), "255", cigar(sa, len(seq)), "*", "0", str(len(seq)),
                          decode(shown), "*", "cs:f:" + java_double(0.0), "AS:f:" + java_double(al.penalty)]
                lines.append("\t".join(fields))
    return lines
