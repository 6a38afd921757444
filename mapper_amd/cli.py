"""Command-line harness in the shape of `java -jar x-mapper.jar` (Mapper.main, Mapper.java:37-453; Mapper.run :639-812) for the part
of it that this repository accelerates: reference + queries in, alignments out as SAM.

    python -m mapper_amd --reference ref.fasta --queries reads.fastq --out-sam out.sam
    python -m mapper_amd --reference ref.fasta --paired-queries r1.fq r2.fq --spacing 100 50 --out-sam out.sam --out-unaligned un.fasta

SURVEY.md section 8(f) rank 1.  What is pinned by the reference: the flag names and their defaults (Mapper.java:82-453), the
derivation of AlignmentParameters from them (:409-453), reference sorting (:1151-1172), the SAM record format
(SamWriter_Test.java) and the statistics lines of Mapper.run (:786-796).  The SAM header, `--out-unaligned` and the other writers
live in the un-vendored QuickVariants module: the header written here is the minimal SAM-spec one and is marked [unpinned]; the
VCF / ancestry outputs stay with the Java host (`--cache-dir` is
honoured: the index goes to one file under it, api.index_cache_path) and are refused here with a message saying so.
"""
import gzip
import sys

import numpy as np

from . import api, sam

JAVA_HOST_ONLY = {"--out-vcf": 1, "--out-ancestor": 1, "--infer-ancestors": 0, "--verify-consistent-db": 0,
                  "--vcf-exclude-non-mutations": 0, "--vcf-omit-support-reads": 0}
IGNORED = {"--verbose": 0, "-v": 0, "-vv": 0, "--verbose-alignment": 0, "--verbose-reference": 0, "--verbosity-auto": 0, "--num-threads": 1,
           "--no-infer-ancestors": 0, "--allow-duplicate-contig-names": 0, "--version": 0}


class UsageError(Exception):
    pass


last_timing = None  # run_streaming: queries and seconds of the last job's streaming phase (bench.py's end_to_end leg)


def _open(path):
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "r")


def read_sequences(path):
    """FASTA or FASTQ (optionally .gz) -> list of (name, text, quality or None).  The name is the header up to the first blank."""
    out = []
    with _open(path) as f:
        lines = f.read().splitlines()
    i, n = 0, len(lines)
    while i < n:
        line = lines[i].rstrip("\r")
        if not line:
            i += 1
            continue
        if line[0] == ">":
            name = line[1:].split()[0] if len(line) > 1 else ""
            i += 1
            parts = []
            while i < n and not lines[i].startswith(">"):
                parts.append(lines[i].strip())
                i += 1
            out.append((name, "".join(parts).upper(), None))
        elif line[0] == "@":
            name = line[1:].split()[0] if len(line) > 1 else ""
            seq = lines[i + 1].strip().upper()
            qual = lines[i + 3].strip() if i + 3 < n else ""
            i += 4
            out.append((name, seq, qual))
        else:
            raise ValueError("%s: line %d is neither a FASTA nor a FASTQ header" % (path, i + 1))
    return out


def parse_args(argv):
    """Mapper.main's flag loop (Mapper.java:82-385) for the flags of this path."""
    o = dict(references=[], queries=[], paired=[], out_sam=None, out_unaligned=None, no_output=False, enable_gapmers=True,
             mutationPenalty=-1.0, indelStart_penalty=1.5, indelExtension_penalty=0.5, additional_insertionExtension_penalty=-1.0,
             maxErrorRate=-1.0, ambiguityPenalty=-1.0, maxNumMatches=2**31 - 1, max_penaltySpan=-1.0, device=0,
             paired_without_spacing=False, help=False, split=0)
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "--help":
            o["help"] = True
        elif a == "--reference":
            o["references"].append(argv[i + 1]); i += 1
        elif a == "--queries":
            o["queries"].append((argv[i + 1], o["split"])); i += 1
        elif a == "--split-queries-past-size":  # Mapper.java:142-148: applies to the --queries that follow
            if o["queries"] or o["paired"]:
                raise UsageError("Sorry, --split-queries-past-size currently is only supported before --queries")
            o["split"] = int(argv[i + 1]); i += 1
        elif a == "--paired-queries":
            if o["split"] > 0:
                raise UsageError("Sorry, --paired-queries is not currently supported with --split-queries-past-size")
            left, right = argv[i + 1], argv[i + 2]
            i += 2
            expected, deviation = 100.0, 50.0  # Mapper.java:41-42
            if i + 1 < len(argv) and argv[i + 1] == "--spacing":
                expected, deviation = float(argv[i + 2]), float(argv[i + 3])
                i += 3
            else:
                o["paired_without_spacing"] = True
            o["paired"].append((left, right, expected, deviation))
        elif a == "--out-sam":
            o["out_sam"] = argv[i + 1]; i += 1
        elif a == "--out-unaligned":
            o["out_unaligned"] = argv[i + 1]; i += 1
        elif a == "--no-output":
            o["no_output"] = True
        elif a == "--no-gapmers":
            o["enable_gapmers"] = False
        elif a == "--new-indel-penalty":
            o["indelStart_penalty"] = float(argv[i + 1]); i += 1
        elif a == "--extend-indel-penalty":
            o["indelExtension_penalty"] = float(argv[i + 1]); i += 1
        elif a == "--additional-extend-insertion-penalty":
            o["additional_insertionExtension_penalty"] = float(argv[i + 1]); i += 1
        elif a == "--snp-penalty":
            o["mutationPenalty"] = float(argv[i + 1]); i += 1
            if o["mutationPenalty"] <= 0:
                raise UsageError("--snp-penalty must be > 0")
        elif a == "--max-penalty":
            o["maxErrorRate"] = float(argv[i + 1]); i += 1
            if o["maxErrorRate"] < 0:
                raise UsageError("--max-penalty must be >= 0")
        elif a == "--max-penalty-span":
            o["max_penaltySpan"] = float(argv[i + 1]); i += 1
            if o["max_penaltySpan"] < 0:
                raise UsageError("--max-penalty-span must be >= 0")
        elif a == "--ambiguity-penalty":
            o["ambiguityPenalty"] = float(argv[i + 1]); i += 1
            if o["ambiguityPenalty"] < 0:
                raise UsageError("--ambiguity-penalty must be >= 0")
        elif a == "--max-num-matches":
            o["maxNumMatches"] = int(argv[i + 1]); i += 1
        elif a == "--out-refs-map-count":  # Mapper.java:197: how many queries mapped to each combination of references
            o["out_refs_map_count"] = argv[i + 1]; i += 1
        elif a == "--out-mutations":  # Mapper.java:203-237: the mutations file (mapper_amd/pileup.py; accumulated on the GPU) and its nested thresholds
            o["out_mutations"] = argv[i + 1]; i += 1
            f = o.setdefault("mutation_filter", {})
            while i + 1 < len(argv):
                sub = argv[i + 1]
                if sub == "--snp-threshold":
                    f["minSNPTotalDepth"], f["minSNPDepthFraction"] = float(argv[i + 2]), float(argv[i + 3])
                elif sub == "--indel-start-threshold":
                    f["minIndelTotalStartDepth"], f["minIndelStartDepthFraction"] = float(argv[i + 2]), float(argv[i + 3])
                elif sub == "--indel-continue-threshold":
                    f["minIndelContinuationTotalDepth"], f["minIndelContinuationDepthFraction"] = float(argv[i + 2]), float(argv[i + 3])
                elif sub == "--indel-threshold":
                    f["minIndelTotalStartDepth"] = f["minIndelContinuationTotalDepth"] = float(argv[i + 2])
                    f["minIndelStartDepthFraction"] = f["minIndelContinuationDepthFraction"] = float(argv[i + 3])
                else:
                    break  # maybe this argument is a top-level argument
                i += 3
        elif a == "--distinguish-query-ends":  # Mapper.java:351-355
            o["query_end_fraction"] = float(argv[i + 1]); i += 1
            if not 0 <= o["query_end_fraction"] < 1:
                raise UsageError("--distinguish-query-ends must be >= 0 and < 1")
        elif a == "--cache-dir":  # Mapper.java:264: keep the hashed reference between runs
            o["cache_dir"] = argv[i + 1]; i += 1
        elif a == "--batch-size":  # (not a Mapper flag) queries per GPU batch; batches are streamed (upload of the next one during the alignment of the current one)
            o["batch_size"] = int(argv[i + 1]); i += 1
            if o["batch_size"] < 1:
                raise UsageError("--batch-size must be >= 1")
        elif a == "--gpus":  # (not a Mapper flag) align on GPUs 0..N-1: index replicated, batches dealt round-robin (mapper_amd/multi.py)
            o["gpus"] = int(argv[i + 1]); i += 1
            if o["gpus"] < 1:
                raise UsageError("--gpus must be >= 1")
        elif a == "--contexts":  # (not a Mapper flag) contexts per GPU: batches are aligned by several contexts of a GPU at the same time (fills idle wave slots)
            o["contexts"] = int(argv[i + 1]); i += 1
            if o["contexts"] < 1:
                raise UsageError("--contexts must be >= 1")
        elif a == "--devices":  # (not a Mapper flag) explicit GPU ordinals, comma-separated; an ordinal may repeat (two contexts on one GPU)
            o["devices"] = [int(x) for x in argv[i + 1].split(",")]; i += 1
        elif a == "--per-object":  # (not a Mapper flag) the harness's first implementation: one Python object per read and per alignment (the reference for the formats; tests)
            o["per_object"] = True
        elif a == "--device":  # (not a Mapper flag) which GPU
            o["device"] = int(argv[i + 1]); i += 1
        elif a == "--spacing":
            raise UsageError("--spacing is not a top-level argument: try --paired-queries <queries> <queries2> --spacing <expected> <distancePerPenalty>")
        elif a in JAVA_HOST_ONLY:
            raise UsageError("%s is handled by the Java host (VCF / mutation / ancestry subsystems are outside the accelerated path, SURVEY.md section 8)" % a)
        elif a in IGNORED:
            i += IGNORED[a]
        else:
            raise UsageError("Unrecognized argument: " + a)
        i += 1
    return o


def derive_parameters(o):
    """Mapper.java:386-453: validation and the defaults that depend on other flags."""
    if len(o["references"]) < 1:
        raise UsageError("--reference is required")
    if len(o["queries"]) + len(o["paired"]) < 1:
        raise UsageError("--queries or --paired-queries is required")
    if o["out_sam"] is None and o["out_unaligned"] is None and not o.get("out_mutations") and not o.get("out_refs_map_count") and not o["no_output"]:
        raise UsageError("No output specified. Try --out-sam <output path>, or if you really don't want to generate an output file, --no-output")
    if o["maxErrorRate"] >= 0 and o["mutationPenalty"] >= 0 and o["paired_without_spacing"]:
        raise UsageError("Customized alignment penalties (--snp-penalty) and penalty threshold (--max-penalty) without customizing spacing penalty "
                         "between paired-end queries: please specify --spacing explicitly")
    maxErrorRate = o["maxErrorRate"] if o["maxErrorRate"] >= 0 else 0.1
    mutationPenalty = o["mutationPenalty"] if o["mutationPenalty"] > 0 else 1.0
    if o["indelExtension_penalty"] <= 0:
        raise UsageError("--extend-indel-penalty must be > 0")
    if o["indelStart_penalty"] <= 0:
        raise UsageError("--new-indel-penalty must be > 0")
    if o["maxNumMatches"] < 1:
        raise UsageError("--max-num-matches must be >= 1")
    span = o["max_penaltySpan"] if o["max_penaltySpan"] >= 0 else mutationPenalty / 2
    ambiguity = o["ambiguityPenalty"] if o["ambiguityPenalty"] >= 0 else maxErrorRate
    additional = o["additional_insertionExtension_penalty"] if o["additional_insertionExtension_penalty"] >= 0 else ambiguity
    return api.AlignmentParameters(MutationPenalty=mutationPenalty, DeletionStart_Penalty=o["indelStart_penalty"], DeletionExtension_Penalty=o["indelExtension_penalty"],
                                   InsertionStart_Penalty=o["indelStart_penalty"], InsertionExtension_Penalty=o["indelExtension_penalty"] + additional,
                                   MaxErrorRate=maxErrorRate, AmbiguityPenalty=ambiguity, UnalignedPenalty=ambiguity, MaxNumMatches=o["maxNumMatches"],
                                   Max_PenaltySpan=span)


def split_sections(length, max_length):
    """SequenceSplitter.java:9-38: a sequence longer than max_length becomes (length - 1) / max_length + 1 sections, section k covering
    [length * k / n, length * (k + 1) / n) (integer division in 64 bits, :35-38).  -> list of (start, end)."""
    num = (length - 1) // max_length + 1
    return [(length * k // num, length * (k + 1) // num) for k in range(num)]


def load_queries(o):
    """-> list of (api.Query, qualities or None) in input order: --queries files first, then --paired-queries files, as Mapper.main adds them."""
    out = []
    for path, split in o["queries"]:
        for name, text, qual in read_sequences(path):
            if split > 0:  # SequenceSplitter.java:9-40: equal sections of at most `split` bases (the sections carry no quality; their names are [unpinned])
                for a, b in split_sections(len(text), split):
                    out.append((api.Query(text[a:b], name=name), [None]))
            else:
                out.append((api.Query(text, name=name), [qual]))
    for left, right, expected, deviation in o["paired"]:
        ls, rs = read_sequences(left), read_sequences(right)
        if len(ls) != len(rs):
            raise UsageError("paired query files have different numbers of reads: %s %d, %s %d" % (left, len(ls), right, len(rs)))
        for (n1, t1, q1), (n2, t2, q2) in zip(ls, rs):
            out.append((api.Query(t1, t2, expected_inner_distance=expected, spacing_deviation_per_unit_penalty=deviation, names=[n1, n2]), [q1, q2]))
    return out


def java_float(x):
    """Float.toString of (float)x for the statistics lines: shortest decimal that round-trips the 32-bit value."""
    t = str(np.float32(x))
    return t if ("." in t or "e" in t or "n" in t) else t + ".0"


def sam_header(contigs):
    """[unpinned: SamWriter is un-vendored] minimal SAM-spec header: @HD, one @SQ per contig in the order of the reference files, @PG."""
    lines = ["@HD\tVN:1.6\tSO:unsorted"]
    lines += ["@SQ\tSN:%s\tLN:%d" % (name, len(text)) for name, text in contigs]
    lines.append("@PG\tID:xmapper-mi355x\tPN:xmapper-mi355x")
    return lines


def run_streaming(o, params, contigs, out):
    """The harness without an object per read (round 6): the query files are parsed by libxm_hostio.so straight into batch arrays (mapper_amd/hostio.py), the
    batches stream through the GPU contexts (align_stream: the copy of batch k + 1 overlaps the alignment of batch k; several contexts align side by side),
    and a writer thread turns each batch's result streams into SAM text in native code while the next batches are being aligned.  Memory is O(batches in
    flight), not O(job).  Same records, same statistics lines as the per-object path below, which stays for the outputs that need objects
    (--out-mutations, --out-refs-map-count)."""
    import queue
    import threading
    from . import hostio
    ordered = api.sort_reference(contigs)  # Mapper.sortAndComplementReference: alignment results refer to this order
    names = [n for n, _ in ordered]
    batch_size = o.get("batch_size") or 1_000_000
    keep_qual = bool(o["out_unaligned"])

    def batches():
        for path, split in o["queries"]:
            yield from hostio.read_batches(path, None, batch_size, split=split, keep_qualities=keep_qual)
        for left, right, expected, deviation in o["paired"]:
            yield from hostio.read_batches(left, right, batch_size, keep_qualities=keep_qual, expected_inner=expected, deviation=deviation)

    # the first batch decides the launch shapes (longest mate) and the number of contexts
    it = batches()
    first = next(it, None)
    if first is None:
        raise UsageError("no queries found")
    max_len = int(first.mate_length.max())
    devices = o.get("devices") or (list(range(o["gpus"])) if o.get("gpus", 1) > 1 else None)
    contexts = o.get("contexts")
    if contexts is None:  # (as the per-object path chooses them: three contexts for single reads of up to 320 bases, else two; one for a job of one batch)
        full = len(first) >= batch_size
        contexts = 1 if not full or devices is not None else (3 if int(first.mate_count.max()) == 1 and max_len <= 320 else 2)
    if contexts > 1:
        devices = [d for d in (devices or [o["device"]]) for _ in range(contexts)]
    # (a job whose later reads are longer than the first batch's grows the index on demand: xm_index_ensure_length)
    if devices and len(devices) > 1:
        from . import multi
        db = multi.MultiGpuDatabase(ordered, devices, mode="mapper", enable_gapmers=o["enable_gapmers"], max_query_length=max_len, cache_dir=o.get("cache_dir"))
    else:
        db = api.ReferenceDatabase(ordered, mode="mapper", enable_gapmers=o["enable_gapmers"], device=devices[0] if devices else o["device"],
                                   max_query_length=max_len, cache_dir=o.get("cache_dir"))
    sam_out = un_out = None
    if o["out_sam"]:
        sam_out = sys.stdout if o["out_sam"] == "-" else open(o["out_sam"], "w")
        sam_out.write("\n".join(sam_header(contigs)) + "\n")
    if o["out_unaligned"]:
        un_out = open(o["out_unaligned"], "w")
    writer = hostio.Writer(names, sam_out, un_out)
    in_flight = queue.Queue()     # batches in the order they were dealt to the GPUs
    to_write = queue.Queue(maxsize=2 * max(1, len(devices or [0])))
    failure = []

    def feed():
        for b in [first]:
            in_flight.put(b)
            yield b.arrays()
        for b in it:
            in_flight.put(b)
            yield b.arrays()

    def write_loop():
        try:
            while True:
                item = to_write.get()
                if item is None:
                    return
                b, r = item
                writer.write(b, r)
                b.close()
        except BaseException as e:  # noqa: BLE001  (handed to the main thread)
            failure.append(e)

    wt = threading.Thread(target=write_loop, daemon=True)
    wt.start()
    import time
    t_stream = time.perf_counter()
    try:
        for r in db.align_stream(feed(), params):
            if failure:
                break
            to_write.put((in_flight.get(), r))
    finally:
        to_write.put(None)
        wt.join()
        db.close()
    if failure:
        raise failure[0]
    if sam_out is not None and sam_out is not sys.stdout:
        sam_out.close()
    if un_out is not None:
        un_out.close()
    st = writer.stats
    n = int(st.num_queries)
    global last_timing
    last_timing = {"queries": n, "stream_seconds": time.perf_counter() - t_stream, "contexts": len(devices) if devices else 1}  # first read of the query files .. last byte of the outputs (bench.py's end_to_end leg)
    out.write("\nStatistics: \n")
    out.write(" Alignment rate                : %d%% of queries (%d/%d)\n" % (st.num_aligned * 100 // n if n else 0, st.num_aligned, n))
    if st.total_aligned_length:
        out.write(" Average penalty               : %s per base (%d/%d) in aligned queries\n" % (java_float(st.total_penalty / st.total_aligned_length), int(st.total_penalty), st.total_aligned_length))
        out.write(" Num indels                    : %s per base (%d/%d) in aligned queries\n" % (java_float(st.num_indels / st.total_aligned_length), st.num_indels, st.total_aligned_length))
    return 0


def run(argv, out=sys.stdout):
    o = parse_args(argv)
    if o["help"] or not argv:
        out.write(__doc__ + "\n")
        return 0
    params = derive_parameters(o)
    contigs = []
    for path in o["references"]:
        contigs += [(name, text) for name, text, _ in read_sequences(path)]
    out.write("%d reference files:\n" % len(o["references"]))
    for path in o["references"]:
        out.write("Reference path = %s\n" % path)
    if not o.get("out_mutations") and not o.get("out_refs_map_count") and not o.get("per_object"):
        return run_streaming(o, params, contigs, out)
    queries = load_queries(o)
    ordered = api.sort_reference(contigs)  # Mapper.sortAndComplementReference: alignment results refer to this order
    names = [n for n, _ in ordered]
    devices = o.get("devices") or (list(range(o["gpus"])) if o.get("gpus", 1) > 1 else None)
    batch_size = o.get("batch_size") or 1_000_000
    contexts = o.get("contexts")
    if contexts is None:
        # a job of several batches: two contexts per GPU align their batches at the same time (+15-18 % reads per second on MI355X; on reads
        # from i.i.d. references more contexts add nothing, profiles/r03/NOTES.md; on a repeat-rich reference, whose passes end with a long tail of
        # few heavy reads, `--contexts 4` gives 1.45x over two, profiles/r04/NOTES.md 13).  Contexts share the index (xm_context_new), so a genome-sized one is no obstacle; they
        # divide the HBM that is free once it is resident, and a GPU with room for fewer contexts uses fewer (api.divide_scratch)
        n_batches = (len(queries) + batch_size - 1) // batch_size
        contexts = 2 if devices is None and n_batches >= 2 else 1
        # single reads of up to 320 bases in three batches or more: three contexts, each sized for a third of the GPU's wave slots, overlap best (+7 % over
        # two, profiles/r04/NOTES.md 15; pairs and long reads are within 2 % from two to four contexts)
        if contexts == 2 and n_batches >= 3 and all(len(q.sequences) == 1 for q, _ in queries) and max(len(s) for q, _ in queries for s in q.sequences) <= 320:
            contexts = 3
    if contexts > 1:
        devices = [d for d in (devices or [o["device"]]) for _ in range(contexts)]
    max_query_length = max([len(s) for q, _ in queries for s in q.sequences] + [1])
    if devices and len(devices) > 1:
        from . import multi
        # (--out-mutations: every context accumulates its own pile-up on its GPU - depth 8 B, four alternative counts 32 B and, with a query-end fraction, the
        # middle depth 8 B per reference base: 149 GB for a 3.1 Gb reference - so the contexts of a GPU are counted with it: api.divide_scratch)
        pile_up = (48 if o.get("query_end_fraction", 0.1) > 0 else 40) * sum(len(t) for _, t in ordered) + (64 << 20) if o.get("out_mutations") else 0
        db = multi.MultiGpuDatabase(ordered, devices, mode="mapper", enable_gapmers=o["enable_gapmers"], max_query_length=max_query_length, cache_dir=o.get("cache_dir"),
                                    per_context_extra=pile_up)
    else:
        db = api.ReferenceDatabase(ordered, mode="mapper", enable_gapmers=o["enable_gapmers"], device=devices[0] if devices else o["device"],
                                   max_query_length=max_query_length, cache_dir=o.get("cache_dir"))
    sam_out = None
    if o["out_sam"]:
        sam_out = sys.stdout if o["out_sam"] == "-" else open(o["out_sam"], "w")
        sam_out.write("\n".join(sam_header(contigs)) + "\n")
    un_out = open(o["out_unaligned"], "w") if o["out_unaligned"] else None
    num_aligned = total_len = num_indels = 0
    total_penalty = 0.0
    refs_map = {} if o.get("out_refs_map_count") else None
    match_db = None
    on_aligned = None
    if o.get("out_mutations"):  # Mapper.java:700-708: the MatchDatabase listens to every batch; here it accumulates on the GPU while the batch is resident
        from . import pileup
        match_db = pileup.MatchDatabase(db.replicas if hasattr(db, "replicas") else db, o.get("query_end_fraction", 0.1))  # (default 0.1: Mapper.java:76)
        on_aligned = lambda replica, first_query, qs: match_db.add_last(qs, replica=replica)  # noqa: E731
    results = db.align_batches([q for q, _ in queries], params, batch_size, on_aligned=on_aligned)
    first, result, nxt = 0, None, 0
    for qi, (q, quals) in enumerate(queries):
        if qi >= nxt:  # the next batch's results (queries are written in input order)
            first, result = next(results)
            nxt = first + len(result)
        comps = result.query_alignments(qi - first)
        aligned = any(len(c) > 0 for c in comps)
        if aligned:
            num_aligned += 1
            if refs_map is not None:  # ReferenceAlignmentCounter [QuickVariants, inferred]: the set of contigs a query's alignments lie on
                key = tuple(sorted({names[sa.contig] for comp in comps for al in comp for sa in al.components}))
                refs_map[key] = refs_map.get(key, 0) + 1
            for comp in comps:  # AlignmentCounter [QuickVariants, inferred]: every reported alignment's aligned length, penalty and indels
                for al in comp:
                    for k, sa in enumerate(al.components):
                        total_len += sum(b.lengthA for b in sa.sections)
                        num_indels += sum(1 for b in sa.sections if b.lengthA != b.lengthB)
                    total_penalty += al.penalty
            if sam_out:
                for line in sam.records(q, comps, names):
                    sam_out.write(line + "\n")
        elif un_out:  # [unpinned format] the query as it came in: FASTQ when it had qualities, else FASTA
            for name, seq, qual in zip(q.names, q.sequences, quals):
                text = api.decode(seq)
                un_out.write("@%s\n%s\n+\n%s\n" % (name, text, qual) if qual is not None else ">%s\n%s\n" % (name, text))
    if sam_out and sam_out is not sys.stdout:
        sam_out.close()
    if un_out:
        un_out.close()
    if refs_map is not None:  # Mapper.java:747-756 (referenceAlignmentCounter.sumAlignments); [unpinned format]: one line per combination, most frequent first
        with open(o["out_refs_map_count"], "w") as f:
            for key, count in sorted(refs_map.items(), key=lambda kv: (-kv[1], kv[0])):
                f.write("%s\t%d\n" % (",".join(key), count))
    if match_db is not None:  # Mapper.java:758-785
        with open(o["out_mutations"], "w") as f:
            filt = pileup.MutationDetectionParameters.defaultFilter()  # Mapper.java:56; --snp-threshold etc. override it
            for k, v in o.get("mutation_filter", {}).items():
                setattr(filt, k, v)
            match_db.write_mutations(f, filt)
        match_db.close()
    n = len(queries)
    out.write("\nStatistics: \n")
    out.write(" Alignment rate                : %d%% of queries (%d/%d)\n" % (num_aligned * 100 // n if n else 0, num_aligned, n))
    if total_len:
        out.write(" Average penalty               : %s per base (%d/%d) in aligned queries\n" % (java_float(total_penalty / total_len), int(total_penalty), total_len))
        out.write(" Num indels                    : %s per base (%d/%d) in aligned queries\n" % (java_float(num_indels / total_len), num_indels, total_len))
    db.close()
    return 0


def main(argv=None):
    from . import _capi
    _capi.want_hardware_queues(8)  # (an entry point: before the HIP runtime starts; three contexts per GPU need queues of their own, _capi.want_hardware_queues)
    try:
        return run(sys.argv[1:] if argv is None else argv)
    except UsageError as e:
        sys.stderr.write("Error: %s\n" % e)
        return 1


if __name__ == "__main__":
    sys.exit(main())
