"""Host-side mirror of the reference's programmatic API for the seed-and-extend path.

Mirrors src/main/java/mapper/Api.java:18-107 (newDatabase / align / alignOnce), AlignmentParameters.java:8-35 and the
Query / QueryAlignment / SequenceAlignment / AlignedBlock result types the Java host consumes, on top of the C ABI of
include/xmapper_hip.h.  Everything is aligned on the GPU by libxmapper_hip.so; there is no CPU path.
"""
import ctypes as C
import os
import numpy as np

from . import _capi

_CODE = np.full(256, 15, dtype=np.uint8)
for _ch, _v in {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "R": 5, "Y": 10, "S": 6, "W": 9, "K": 12, "M": 3, "B": 14, "D": 13, "H": 11, "V": 7, "N": 15}.items():
    _CODE[ord(_ch)] = _v
    _CODE[ord(_ch.lower())] = _v
_DECODE = "?ACMGRSVTWYHKDBN"
_COMP = np.array([((b & 1) << 3) | ((b & 2) << 1) | ((b & 4) >> 1) | ((b & 8) >> 3) for b in range(16)], dtype=np.uint8)


def encode(text):
    """IUPAC text -> 4-bit Basepairs codes, one per byte."""
    return _CODE[np.frombuffer(text.encode("ascii"), dtype=np.uint8)].copy()


def decode(codes):
    return "".join(_DECODE[int(c) & 15] for c in codes)


def reverse_complement(codes):
    return _COMP[np.asarray(codes, dtype=np.uint8)[::-1]]


class AlignmentParameters:
    """AlignmentParameters.java:8-35; defaults are Mapper.main's (Mapper.java:66-73, 409-453)."""

    def __init__(self, MutationPenalty=1.0, InsertionStart_Penalty=1.5, InsertionExtension_Penalty=0.5 + 0.1, DeletionStart_Penalty=1.5,
                 DeletionExtension_Penalty=0.5, MaxErrorRate=0.1, UnalignedPenalty=0.1, AmbiguityPenalty=0.1, MaxNumMatches=2**31 - 1,
                 Max_PenaltySpan=0.5):
        self.MutationPenalty = MutationPenalty
        self.InsertionStart_Penalty = InsertionStart_Penalty
        self.InsertionExtension_Penalty = InsertionExtension_Penalty
        self.DeletionStart_Penalty = DeletionStart_Penalty
        self.DeletionExtension_Penalty = DeletionExtension_Penalty
        self.MaxErrorRate = MaxErrorRate
        self.UnalignedPenalty = UnalignedPenalty
        self.AmbiguityPenalty = AmbiguityPenalty
        self.MaxNumMatches = MaxNumMatches
        self.Max_PenaltySpan = Max_PenaltySpan

    def _c(self):
        p = _capi.XmParams()
        for f, _ in _capi.XmParams._fields_:
            if f != "reserved":
                setattr(p, f, getattr(self, f))
        return p


class Query:
    """Query(seq) / Query(seq1, seq2, expectedInnerDistance, spacingDeviationPerUnitPenalty) [QuickVariants]."""

    def __init__(self, *sequences, expected_inner_distance=0.0, spacing_deviation_per_unit_penalty=1.0, name=None, names=None):
        self.sequences = [encode(s) if isinstance(s, str) else np.ascontiguousarray(s, dtype=np.uint8) for s in sequences]
        if not 1 <= len(self.sequences) <= 2:
            raise ValueError("a Query has 1 or 2 sequences")
        self.expected_inner_distance = float(expected_inner_distance)
        self.spacing_deviation_per_unit_penalty = float(spacing_deviation_per_unit_penalty)
        self.names = names or ([name] * len(self.sequences) if name else ["query"] * len(self.sequences))


class AlignedBlock:
    __slots__ = ("startA", "startB", "lengthA", "lengthB")

    def __init__(self, sa, sb, la, lb):
        self.startA, self.startB, self.lengthA, self.lengthB = sa, sb, la, lb


class SequenceAlignment:
    """SequenceAlignment [QuickVariants]: ordered AlignedBlocks + referenceReversed + penalties."""

    def __init__(self, contig, reference_reversed, blocks, total_penalty, aligned_penalty):
        self.contig, self.reference_reversed, self.sections = contig, bool(reference_reversed), blocks
        self.penalty, self.aligned_penalty = total_penalty, aligned_penalty

    def start_index_b(self):
        return self.sections[0].startB

    def end_index_b(self):
        return self.sections[-1].startB + self.sections[-1].lengthB

    def aligned_text(self, query_codes, ref_codes):
        """(getAlignedTextA, getAlignedTextB); query_codes must be the strand that was aligned."""
        a, b = [], []
        for s in self.sections:
            a.append(decode(query_codes[s.startA:s.startA + s.lengthA]) if s.lengthA > 0 else "-" * s.lengthB)
            b.append(decode(ref_codes[s.startB:s.startB + s.lengthB]) if s.lengthB > 0 else "-" * s.lengthA)
        return "".join(a), "".join(b)


class QueryAlignment:
    """QueryAlignment(components, spacingPenalty, overlapMultiplier, duplicationBonus, totalPenalty, innerDistance)
    (QueryMatch_Aligner.java:267)."""

    def __init__(self, components, spacing_penalty, overlap_multiplier, duplication_bonus, penalty, inner_distance):
        self.components = components
        self.spacing_penalty, self.overlap_multiplier, self.duplication_bonus = spacing_penalty, overlap_multiplier, duplication_bonus
        self.penalty, self.inner_distance = penalty, inner_distance


def decode_streams(ints, dbls, int_off, dbl_off, q):
    """-> QueryAlignments of query q as list (components) of lists of QueryAlignment."""
    ii = ints[int_off[q]:int_off[q + 1]]
    dd = dbls[dbl_off[q]:dbl_off[q + 1]]
    i = d = 0
    comps = []
    ncomp = int(ii[i]); i += 1
    for _ in range(ncomp):
        nal = int(ii[i]); i += 1
        als = []
        for _ in range(nal):
            inner, nseq = int(ii[i]), int(ii[i + 1]); i += 2
            sp, om, db, tp = (float(x) for x in dd[d:d + 4]); d += 4
            seqs = []
            for _ in range(nseq):
                contig, rev, nb = int(ii[i]), int(ii[i + 1]), int(ii[i + 2]); i += 3
                blocks = [AlignedBlock(int(ii[i + 4 * k]), int(ii[i + 4 * k + 1]), int(ii[i + 4 * k + 2]), int(ii[i + 4 * k + 3])) for k in range(nb)]
                i += 4 * nb
                seqs.append(SequenceAlignment(contig, rev, blocks, float(dd[d]), float(dd[d + 1])))
                d += 2
            als.append(QueryAlignment(seqs, sp, om, db, tp, inner))
        comps.append(als)
    return comps


class BatchResult:
    def __init__(self, d):
        self.__dict__.update(d)

    def __len__(self):
        return len(self.int_off) - 1

    def query_alignments(self, q):
        return decode_streams(self.ints, self.dbls, self.int_off, self.dbl_off, q)


def index_cache_path(cache_dir, contigs, opts):
    """<cache_dir>/cache/<digest>/index.xmidx, the digest over the property text DirCache would hash (DirCache.java:19-60): the
    sequence database's keys (here: names, lengths and a digest of the bases) + enableGapmers, minInterestingSize, maxNumShortMatches,
    formatVersion, type (HashBlock_Database.java:106-114) + the duplication settings this library stores in the same file."""
    import hashlib
    seq = hashlib.sha256()
    for name, codes in contigs:
        seq.update(name.encode() + b"\0" + str(len(codes)).encode() + b"\0")
        seq.update(np.ascontiguousarray(codes, dtype=np.uint8).tobytes())
    props = {"sequences": seq.hexdigest(), "enableGapmers": str(bool(opts.enable_gapmers)).lower(), "minInterestingSize": str(opts.min_interesting_size),
             "maxNumShortMatches": "5", "formatVersion": "xmidx-1", "type": "HashBlock_Database",
             "duplications": "%d,%d,%d,%d" % (opts.dup_window, opts.dup_min_copies, opts.dup_min_length, opts.dup_max_length)}
    text = "{\n" + "".join('%s:"%s",\n' % (k, props[k]) for k in sorted(props)) + "}"
    d = os.path.join(os.fspath(cache_dir), "cache", hashlib.sha256(text.encode()).hexdigest()[:32])
    os.makedirs(d, exist_ok=True)
    meta = os.path.join(d, "metadata")
    if not os.path.exists(meta):
        with open(meta + ".tmp.%d" % os.getpid(), "w") as f:
            f.write(text)
        os.replace(meta + ".tmp.%d" % os.getpid(), meta)
    return os.path.join(d, "index.xmidx")


class ReferenceDatabase:
    """ReferenceDatabase.java: HashBlock_Database + DuplicationDetector of a reference, resident in HBM."""

    def __init__(self, contigs, mode="mapper", enable_gapmers=True, max_query_length=0, device=-1, host_only=False, dup=None, cache_dir=None, min_interesting_size=-1):
        """contigs: list of (name, IUPAC text or code array), already in Mapper.sortAndComplementReference order
        (use sort_reference()).  mode 'mapper' = Mapper.run assembly (duplication window 1000), 'api' = Api.newDatabase (window 1).
        cache_dir: --cache-dir (Mapper.java:264, DirCache.java:19-60): the index is read from / written to a file under
        <cache_dir>/cache/ that is named after the reference's cache keys (HashBlock_Database.java:106-114) and this build's settings."""
        self._L = _capi.lib()
        self.contigs = [(n, encode(s) if isinstance(s, str) else np.ascontiguousarray(s, dtype=np.uint8)) for n, s in contigs]
        ref, self._keep = _capi.make_ref(self.contigs)
        o = _capi.XmBuildOpts()
        o.enable_gapmers = 1 if enable_gapmers else 0
        o.min_interesting_size = int(min_interesting_size)  # (-1: from the reference's size, HashBlock_Database.java:52; tests pin the value a 3 Gb reference gets)
        o.max_hashed_length = int(max_query_length)
        o.dup_window = 1 if mode == "api" else 1000
        o.dup_min_copies = 2
        o.dup_min_length = o.dup_max_length = -1
        if dup:
            o.dup_min_length, o.dup_max_length, o.dup_min_copies, o.dup_window = dup
        o.device = device
        o.host_only = 1 if host_only else 0
        h = C.c_void_p()
        self.cache_file = None
        self.cache_hit = False
        if cache_dir is not None:
            self.cache_file = index_cache_path(cache_dir, self.contigs, o)
            if os.path.exists(self.cache_file) and self._L.xm_index_load(self.cache_file.encode(), C.byref(ref), C.byref(o), C.byref(h)) == 0:
                self._h = h
                self.cache_hit = True
                return
            # (a file that does not load - other settings behind the same name, truncated, older format - is rebuilt and replaced)
        if self._L.xm_index_build(C.byref(ref), C.byref(o), C.byref(h)):
            raise RuntimeError(self._L.xm_last_error().decode())
        self._h = h
        if self.cache_file is not None:
            self.save(self.cache_file)

    def save(self, path):
        """xm_index_save: the reference, every table hashed so far and the duplication map into one file."""
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        if self._L.xm_index_save(self._h, os.fspath(path).encode()):
            raise RuntimeError(self._L.xm_last_error().decode())

    @classmethod
    def load(cls, path, device=-1, host_only=False, max_query_length=0):
        """xm_index_load without a build request to check against: whatever reference and settings the file holds."""
        self = cls.__new__(cls)
        self._L = _capi.lib()
        self.contigs, self._keep, self.cache_file, self.cache_hit = None, None, os.fspath(path), True
        o = _capi.XmBuildOpts()
        o.enable_gapmers = 1
        o.max_hashed_length = int(max_query_length)
        o.device = device
        o.host_only = 1 if host_only else 0
        h = C.c_void_p()
        if self._L.xm_index_load(os.fspath(path).encode(), None, C.byref(o), C.byref(h)):
            raise RuntimeError(self._L.xm_last_error().decode())
        self._h = h
        return self

    def replicate(self, device):
        """xm_index_replicate: a further context of this index on GPU `device`.  The host tables are shared; on another GPU the tables are copied
        HBM to HBM (nothing built or uploaded twice), on this index's own GPU the context reads the very same tables (= new_context)."""
        other = ReferenceDatabase.__new__(ReferenceDatabase)
        other._L = self._L
        other.contigs, other._keep, other.cache_file, other.cache_hit = self.contigs, self._keep, self.cache_file, self.cache_hit
        h = C.c_void_p()
        if self._L.xm_index_replicate(self._h, int(device), C.byref(h)):
            raise RuntimeError(self._L.xm_last_error().decode())
        other._h = h
        return other

    def new_context(self):
        """xm_context_new: a further context on the same GPU - own stream, batch buffers, scratch and results over the same tables, as the reference's
        AlignerWorker threads share one HashBlock_Database through per-thread views (HashBlock_Database.java:129-133)."""
        other = ReferenceDatabase.__new__(ReferenceDatabase)
        other._L = self._L
        other.contigs, other._keep, other.cache_file, other.cache_hit = self.contigs, self._keep, self.cache_file, self.cache_hit
        h = C.c_void_p()
        if self._L.xm_context_new(self._h, C.byref(h)):
            raise RuntimeError(self._L.xm_last_error().decode())
        other._h = h
        return other

    def set_scratch(self, nbytes):
        """xm_context_set_scratch: upper limit of the HBM this context allocates as scratch for its passes (0: the default)."""
        if self._L.xm_context_set_scratch(self._h, int(nbytes)):
            raise RuntimeError(self._L.xm_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.xm_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def info(self):
        i = _capi.XmIndexInfo()
        if self._L.xm_index_get_info(self._h, C.byref(i)):
            raise RuntimeError(self._L.xm_last_error().decode())
        return {f: getattr(i, f) for f, _ in _capi.XmIndexInfo._fields_}

    def ensure_length(self, n):
        if self._L.xm_index_ensure_length(self._h, int(n)):
            raise RuntimeError(self._L.xm_last_error().decode())

    def table_shape(self, used_length):
        """(capacity, per-key limit) of the table of one gapmer length."""
        cap, mx = C.c_int32(), C.c_int32()
        if self._L.xm_index_table_shape(self._h, int(used_length), C.byref(cap), C.byref(mx)):
            raise RuntimeError(self._L.xm_last_error().decode())
        return cap.value, mx.value

    def table(self, used_length):
        cap, mx, n, o = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        if self._L.xm_index_table_info(self._h, used_length, C.byref(cap), C.byref(mx), C.byref(n), C.byref(o)):
            return None
        counts = np.zeros(cap.value, dtype=np.int32)
        pos = np.zeros(max(n.value, 1), dtype=np.int64)
        self._L.xm_index_table_dump(self._h, used_length, counts.ctypes.data, pos.ctypes.data)
        return dict(capacity=cap.value, maxCount=mx.value, counts=counts, positions=pos[:n.value])

    def bucket_stats(self):
        """{buckets, occupied, overfull} over all hashed tables (overfull: more than max(L^2, 5) entries, HashBlock_Database.java:569-577)."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        if self._L.xm_index_bucket_stats(self._h, C.byref(a), C.byref(b), C.byref(c)):
            raise RuntimeError(self._L.xm_last_error().decode())
        return {"buckets": a.value, "occupied": b.value, "overfull": c.value, "overfull_share_of_occupied": round(c.value / max(1, b.value + c.value), 6)}

    def dup_keys(self, contig):
        n = self._L.xm_index_dup_keys(self._h, contig, None, 0)
        out = np.zeros(max(n, 1), dtype=np.int32)
        self._L.xm_index_dup_keys(self._h, contig, out.ctypes.data, n)
        return out[:n]

    def align_arrays(self, mate_count, mate_offset, mate_length, codes, expected_inner, deviation, parameters):
        """One AlignerWorker.process() batch (AlignerWorker.java:177-231) from flat arrays -> BatchResult."""
        b, keep = _capi.make_batch(mate_count, mate_offset, mate_length, codes, expected_inner, deviation)
        p = parameters._c() if isinstance(parameters, AlignmentParameters) else parameters
        res = C.POINTER(_capi.XmResult)()
        if self._L.xm_align_batch(self._h, C.byref(p), C.byref(b), C.byref(res)):
            raise RuntimeError("Failed to align: " + self._L.xm_last_error().decode())
        d = _capi.view_result(self._L, res)
        return BatchResult(d)

    def upload_arrays(self, mate_count, mate_offset, mate_length, codes, expected_inner, deviation):
        """xm_batch_upload: validate + copy one batch to HBM; it stays resident for align_resident()."""
        b, keep = _capi.make_batch(mate_count, mate_offset, mate_length, codes, expected_inner, deviation)
        if self._L.xm_batch_upload(self._h, C.byref(b)):
            raise RuntimeError(self._L.xm_last_error().decode())

    def stage_arrays(self, mate_count, mate_offset, mate_length, codes, expected_inner, deviation):
        """xm_batch_stage: copy the NEXT batch to HBM on its own stream (may run while align_resident() works on the resident batch)."""
        b, keep = _capi.make_batch(mate_count, mate_offset, mate_length, codes, expected_inner, deviation)
        if self._L.xm_batch_stage(self._h, C.byref(b)):
            raise RuntimeError(self._L.xm_last_error().decode())

    def commit_staged(self):
        """xm_batch_commit: the staged batch becomes the resident one."""
        if self._L.xm_batch_commit(self._h):
            raise RuntimeError(self._L.xm_last_error().decode())

    def align_stream(self, batches, parameters, on_aligned=None):
        """Aligns a sequence of batches (each a tuple of upload_arrays' six arrays) and yields their BatchResults in order.  The copy of
        batch k+1 to HBM runs on a second host thread and a second stream while batch k is being aligned (the
        reference's workers fetch their next batch the same way, AlignerWorker.java:92-175)."""
        import queue
        import threading
        ready = queue.Queue()
        aligned = threading.Semaphore(0)
        stop = threading.Event()

        def uploader():
            try:
                first = True
                for arrays in batches:
                    if stop.is_set():
                        break
                    self.stage_arrays(*arrays)
                    if not first:
                        aligned.acquire()  # the previous batch has been aligned: its buffers may be swapped away
                        if stop.is_set():
                            break
                    self.commit_staged()
                    first = False
                    ready.put(True)
                ready.put(None)
            except BaseException as e:  # noqa: BLE001  (handed to the consumer)
                ready.put(e)

        t = threading.Thread(target=uploader, daemon=True)
        t.start()
        count = 0
        try:
            while True:
                token = ready.get()
                if token is None:
                    break
                if isinstance(token, BaseException):
                    raise token
                r = self.align_resident(parameters)
                if on_aligned is not None:  # (while the batch and its result streams are still resident: pileup.MatchDatabase.add_last)
                    on_aligned(count)
                count += 1
                aligned.release()
                yield r
        finally:
            stop.set()
            aligned.release()
            t.join(timeout=60)

    def align_resident(self, parameters):
        p = parameters._c() if isinstance(parameters, AlignmentParameters) else parameters
        res = C.POINTER(_capi.XmResult)()
        if self._L.xm_align_resident(self._h, C.byref(p), C.byref(res)):
            raise RuntimeError("Failed to align: " + self._L.xm_last_error().decode())
        d = _capi.view_result(self._L, res)
        return BatchResult(d)

    @staticmethod
    def batch_arrays(queries):
        """The six arrays of upload_arrays / align_arrays for a list of Query objects."""
        nq = len(queries)
        mc = np.zeros(nq, np.int32); mo = np.zeros(2 * nq, np.int64); ml = np.zeros(2 * nq, np.int32)
        ei = np.zeros(nq); dv = np.ones(nq)
        chunks, off = [], 0
        for i, q in enumerate(queries):
            mc[i] = len(q.sequences)
            ei[i], dv[i] = q.expected_inner_distance, q.spacing_deviation_per_unit_penalty
            for m, s in enumerate(q.sequences):
                mo[2 * i + m], ml[2 * i + m] = off, len(s)
                chunks.append(s)
                off += len(s)
        codes = np.concatenate(chunks) if chunks else np.zeros(1, np.uint8)
        return mc, mo, ml, codes, ei, dv

    def align_batches(self, queries, parameters, batch_size, on_aligned=None):
        """Aligns `queries` in batches of `batch_size` (AlignerWorker takes its queries batch by batch too, AlignerWorker.java:92-231) and
        yields (first query index, BatchResult) per batch; the next batch is packed and copied to HBM while the current one is aligned
        (align_stream).  on_aligned(replica, first query index, queries of the batch) is called while the batch is still resident."""
        starts = list(range(0, len(queries), max(1, int(batch_size))))
        arrays = (self.batch_arrays(queries[s:s + batch_size]) for s in starts)
        hook = (lambda k: on_aligned(0, starts[k], queries[starts[k]:starts[k] + batch_size])) if on_aligned else None
        for s, r in zip(starts, self.align_stream(arrays, parameters, on_aligned=hook)):
            yield s, r

    def align_batch(self, queries, parameters):
        nq = len(queries)
        mc = np.zeros(nq, np.int32); mo = np.zeros(2 * nq, np.int64); ml = np.zeros(2 * nq, np.int32)
        ei = np.zeros(nq); dv = np.ones(nq)
        chunks, off = [], 0
        for i, q in enumerate(queries):
            mc[i] = len(q.sequences)
            ei[i], dv[i] = q.expected_inner_distance, q.spacing_deviation_per_unit_penalty
            for m, s in enumerate(q.sequences):
                mo[2 * i + m], ml[2 * i + m] = off, len(s)
                chunks.append(s)
                off += len(s)
        codes = np.concatenate(chunks) if chunks else np.zeros(1, np.uint8)
        return self.align_arrays(mc, mo, ml, codes, ei, dv, parameters)

    def seed_probe(self, used_length, keys, max_per_probe=8, unpack=True):
        """Bulk PackedMap.get (PackedMap.java:160-172) on the device -> (counts, positions[n, max_per_probe] with -1 behind a probe's positions, kernel_ms).
        max_per_probe: 0 ... 15 (a bucket line holds seven; xm_seed_probe_packed refuses more than 15).  The C entry packs the positions of every 64 consecutive
        probes one behind the other: unpacked here.  unpack=False: (counts, None, kernel_ms) - the kernel fetches the positions all the same (that is what a
        measurement times), but they are not copied back to the host (n * max_per_probe * 8 bytes: 3.6 GB for 64 M probes with seven positions)."""
        used = np.ascontiguousarray(used_length, dtype=np.int32)
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        n = len(used)
        counts = np.zeros(n, np.int32)
        packed = np.zeros(max(max_per_probe, 1) * max(n, 1), np.int64) if unpack else None
        ms = C.c_double()
        if self._L.xm_seed_probe_packed(self._h, n, used.ctypes.data, keys.ctypes.data, max_per_probe, counts.ctypes.data, packed.ctypes.data if unpack else None, C.byref(ms)):
            raise RuntimeError(self._L.xm_last_error().decode())
        if not unpack:
            return counts, packed, ms.value
        pos = np.full((n, max(max_per_probe, 1)), -1, np.int64)
        if max_per_probe > 0 and n > 0:
            m = np.minimum(np.maximum(counts, 0), max_per_probe).astype(np.int64)
            chunk = np.arange(n, dtype=np.int64) // 64
            run = np.cumsum(m) - m                                    # exclusive running sum over all probes ...
            start = run - run[chunk * 64] + chunk * 64 * max_per_probe  # ... made relative to the probe's chunk
            for j in range(max_per_probe):
                sel = m > j
                pos[sel, j] = packed[start[sel] + j]
        return counts, pos, ms.value


def measure_random_gather(table_bytes=4 << 30, accesses=1 << 26, device=0):
    """Random 64-byte-sector reads per second the GPU sustains (xm_measure_random_gather) -> (sectors/s, kernel ms)."""
    L = _capi.lib()
    ms = C.c_double()
    if L.xm_measure_random_gather(device, table_bytes, accesses, C.byref(ms)):
        raise RuntimeError(L.xm_last_error().decode())
    return accesses / (ms.value * 1e-3), ms.value


def device_memory(device=0):
    """xm_device_memory: (free, total) bytes of HBM on GPU `device` right now."""
    L = _capi.lib()
    f, t = C.c_int64(), C.c_int64()
    if L.xm_device_memory(int(device), C.byref(f), C.byref(t)):
        raise RuntimeError(L.xm_last_error().decode())
    return f.value, t.value


def divide_scratch(contexts, device, reserve=24 << 30, most=200 << 30, per_context_extra=0):
    """Several contexts on one GPU: what is free now (the index is resident) minus a reserve for batches and result arenas and minus what every
    context will allocate beside its scratch (per_context_extra: a pile-up of 40-48 bytes per reference base with --out-mutations), in equal
    parts; a context that would get less than 8 GiB is not worth having -> (how many of `contexts` to use, bytes each)."""
    for c in contexts:
        c.set_scratch(1 << 20)  # (a context that already holds scratch gives it back first: what is free is then what there is to divide)
    free, _ = device_memory(device)
    reserve = min(int(reserve), int(free) // 4)  # (a small or busy GPU: the reserve is a share of what there is, not a fixed claim)
    n = len(contexts)
    while n > 1 and (free - reserve - n * per_context_extra) // n < (8 << 30):
        n -= 1
    share = max(256 << 20, min(most, (free - reserve - n * per_context_extra) // max(n, 1)))
    for c in contexts[:n]:
        c.set_scratch(share)
    for c in contexts[n:]:
        c.set_scratch(0)
    return n, share


def sort_reference(contigs):
    """Mapper.sortAndComplementReference (Mapper.java:1151-1172): length-descending, stable within equal lengths."""
    return sorted(contigs, key=lambda c: -len(c[1]))


# ---- Api.java:18-107
def newDatabase(references, **kw):
    """Api.newDatabase(List<String>) (Api.java:25-31): contigs are named reference-<i>."""
    if isinstance(references, str):
        references = [references]
    if isinstance(references, dict):
        items = list(references.items())
    else:
        items = [("reference-%d" % i, r) for i, r in enumerate(references)]
    kw.setdefault("mode", "api")
    return ReferenceDatabase(items, **kw)


def align(query, reference_database, parameters):
    """Api.align (Api.java:79-92) -> List<QueryAlignment> (QueryAlignments.getTopLevelAlignments())."""
    if isinstance(query, str):
        query = Query(query)
    comps = reference_database.align_batch([query], parameters).query_alignments(0)
    return comps[0] if len(comps) == 1 else []


def alignOnce(query, reference_text, parameters):
    """Api.alignOnce (Api.java:96-107)."""
    db = newDatabase(reference_text)
    try:
        return align(query, db, parameters)
    finally:
        db.close()
