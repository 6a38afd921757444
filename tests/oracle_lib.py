"""ctypes binding of the CPU oracle (oracle/_build/libxmo.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (mapper_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libxmo.so")

_CODE = np.full(256, 15, dtype=np.uint8)
for _ch, _v in {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "R": 5, "Y": 10, "S": 6, "W": 9, "K": 12, "M": 3,
                "B": 14, "D": 13, "H": 11, "V": 7, "N": 15}.items():
    _CODE[ord(_ch)] = _v
    _CODE[ord(_ch.lower())] = _v
_DECODE = "?ACMGRSVTWYHKDBN"


def encode(text):
    """IUPAC text -> one 4-bit code per byte (A=1 C=2 G=4 T=8, unions = OR)."""
    return _CODE[np.frombuffer(text.encode("ascii"), dtype=np.uint8)].copy()


def decode(codes):
    return "".join(_DECODE[int(c) & 15] for c in codes)


def revcomp_text(text):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "R": "Y", "Y": "R", "S": "S", "W": "W", "K": "M", "M": "K",
            "B": "V", "V": "B", "D": "H", "H": "D"}
    return "".join(comp[c] for c in reversed(text.upper()))


class Params(C.Structure):
    """Mirrors xm_params (include/xmapper_hip.h) = AlignmentParameters (M/AlignmentParameters.java:8-35)."""
    _fields_ = [("MutationPenalty", C.c_double), ("InsertionStart_Penalty", C.c_double), ("InsertionExtension_Penalty", C.c_double),
                ("DeletionStart_Penalty", C.c_double), ("DeletionExtension_Penalty", C.c_double), ("MaxErrorRate", C.c_double),
                ("UnalignedPenalty", C.c_double), ("AmbiguityPenalty", C.c_double), ("Max_PenaltySpan", C.c_double),
                ("MaxNumMatches", C.c_int32), ("reserved", C.c_int32)]


def make_params(d=None, **kw):
    """Defaults of Mapper.main (M/Mapper.java:66-73,409-453) unless overridden."""
    base = dict(MutationPenalty=1.0, InsertionStart_Penalty=1.5, InsertionExtension_Penalty=0.5 + 0.1, DeletionStart_Penalty=1.5,
                DeletionExtension_Penalty=0.5, MaxErrorRate=0.1, UnalignedPenalty=0.1, AmbiguityPenalty=0.1, Max_PenaltySpan=0.5,
                MaxNumMatches=2**31 - 1)
    if d:
        base.update(d)
    base.update(kw)
    p = Params()
    for k, v in base.items():
        setattr(p, k, v)
    return p


def build_oracle():
    if not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
            for f in os.listdir(ORACLE_DIR) if f.endswith((".h", ".cpp"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return LIB_PATH


class _Result(C.Structure):
    _fields_ = [("nq", C.c_int64), ("nInts", C.c_int64), ("nDbls", C.c_int64), ("ints", C.POINTER(C.c_int32)),
                ("dbls", C.POINTER(C.c_double)), ("intOff", C.POINTER(C.c_int64)), ("dblOff", C.POINTER(C.c_int64)),
                ("counters", C.c_int64 * 24)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        L.xmo_last_error.restype = C.c_char_p
        L.xmo_ref_new.restype = C.c_void_p
        L.xmo_ref_free.argtypes = [C.c_void_p]
        L.xmo_ref_add_contig.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.xmo_ref_add_contig_codes.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
        L.xmo_ref_finish.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.xmo_ref_finish_min_interesting.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.xmo_ref_finish_custom_dup.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.xmo_ref_require_size.argtypes = [C.c_void_p, C.c_int]
        L.xmo_index_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.xmo_index_table_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.xmo_index_table_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.xmo_dup_keys.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        L.xmo_dup_keys.restype = C.c_int64
        L.xmo_dup_granularity.argtypes = [C.c_void_p]
        L.xmo_dup_granularity.restype = C.c_double
        L.xmo_align_batch.restype = C.POINTER(_Result)
        L.xmo_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.xmo_result_free.argtypes = [C.POINTER(_Result)]
        L.xmo_kat_bound.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.xmo_observe_bound.argtypes = [C.c_int]
        L.xmo_observe_bound.restype = None
        L.xmo_kat_multi_contains.argtypes = [C.c_char_p, C.c_char_p]
        L.xmo_kat_position_codec.argtypes = [C.c_int, C.c_int]
        L.xmo_kat_packed_map_large.argtypes = []
        L.xmo_kat_local_align.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_void_p, C.c_double, C.c_double, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_double)]
        L.xmo_kat_hash_symmetry.argtypes = [C.c_char_p]
        L.xmo_kat_base_penalty.argtypes = [C.c_char, C.c_char, C.c_double, C.c_double]
        L.xmo_kat_base_penalty.restype = C.c_double
        L.xmo_kat_counting_path.argtypes = [C.c_char_p, C.c_char_p, C.c_double, C.c_int, C.c_void_p, C.c_int]
        L.xmo_kat_paths_counter.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.xmo_kat_db_order_independent.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int]
        L.xmo_pyramid_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        L.xmo_pyramid_dump.restype = C.c_int64
        L.xmo_pyramid_dump_multi.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        L.xmo_pyramid_dump_multi.restype = C.c_int64
        _lib = L
    return _lib


class observe_bound:
    """with observe_bound(): the oracle also evaluates, beside every PathAligner search, the bound of the product's rejection filter (xmo_extend.h
    PathAligner::boundObserve) - it raises when a search the filter rejects returns an alignment, and Streams.counters[9:14] then hold: searches that returned
    null, their nodes, searches the filter rejects, the reference's nodes in those, searches the filter takes.  Off by default (the CPU baseline does not pay for it)."""

    def __enter__(self):
        lib().xmo_observe_bound(1)
        return self

    def __exit__(self, *a):
        lib().xmo_observe_bound(0)


class QueryBatch:
    """Flat SoA batch of queries (1 or 2 mates each) shared by the oracle and the product C ABI."""

    def __init__(self, queries):
        """queries: list of (mates, expectedInner, deviation); mates = list of 1-2 code arrays or IUPAC strings."""
        nq = len(queries)
        self.nq = nq
        self.mate_count = np.zeros(nq, dtype=np.int32)
        self.mate_offset = np.zeros(nq * 2, dtype=np.int64)
        self.mate_length = np.zeros(nq * 2, dtype=np.int32)
        self.expected_inner = np.zeros(nq, dtype=np.float64)
        self.deviation = np.ones(nq, dtype=np.float64)
        chunks = []
        off = 0
        for i, q in enumerate(queries):
            if isinstance(q, (str, np.ndarray)):
                q = ([q], 0.0, 1.0)
            mates, exp, dev = q
            self.mate_count[i] = len(mates)
            self.expected_inner[i] = exp
            self.deviation[i] = dev
            for m, mate in enumerate(mates):
                codes = encode(mate) if isinstance(mate, str) else np.asarray(mate, dtype=np.uint8)
                self.mate_offset[i * 2 + m] = off
                self.mate_length[i * 2 + m] = len(codes)
                chunks.append(codes)
                off += len(codes)
        self.codes = np.concatenate(chunks) if chunks else np.zeros(1, dtype=np.uint8)

    @staticmethod
    def from_arrays(mate_count, mate_offset, mate_length, codes, expected_inner, deviation):
        b = QueryBatch([])
        b.nq = len(mate_count)
        b.mate_count, b.mate_offset, b.mate_length = mate_count, mate_offset, mate_length
        b.codes, b.expected_inner, b.deviation = codes, expected_inner, deviation
        return b


class Streams:
    """Decoded result streams (layout documented in include/xmapper_hip.h)."""

    def __init__(self, ints, dbls, int_off, dbl_off, counters=None):
        self.ints, self.dbls, self.int_off, self.dbl_off, self.counters = ints, dbls, int_off, dbl_off, counters

    def query(self, q):
        """-> list (components) of list (alignments) of dicts."""
        ints = self.ints[self.int_off[q]:self.int_off[q + 1]]
        dbls = self.dbls[self.dbl_off[q]:self.dbl_off[q + 1]]
        i = d = 0
        ncomp = int(ints[i]); i += 1
        comps = []
        for _ in range(ncomp):
            nal = int(ints[i]); i += 1
            als = []
            for _ in range(nal):
                al = dict(innerDistance=int(ints[i]), spacingPenalty=float(dbls[d]), overlapMultiplier=float(dbls[d + 1]),
                          duplicationBonus=float(dbls[d + 2]), totalPenalty=float(dbls[d + 3]), sequences=[])
                nseq = int(ints[i + 1]); i += 2; d += 4
                for _ in range(nseq):
                    contig, rev, nb = int(ints[i]), int(ints[i + 1]), int(ints[i + 2]); i += 3
                    blocks = [tuple(int(x) for x in ints[i + 4 * k:i + 4 * k + 4]) for k in range(nb)]
                    i += 4 * nb
                    al["sequences"].append(dict(contig=contig, referenceReversed=rev, blocks=blocks, totalPenalty=float(dbls[d]), alignedPenalty=float(dbls[d + 1])))
                    d += 2
                als.append(al)
            comps.append(als)
        return comps


class OracleReference:
    def __init__(self, contigs, mode="mapper", enable_gapmers=True, custom_dup=None, min_interesting_size=-1):
        """contigs: list of (name, text-or-codes) in the order Mapper.sortAndComplementReference would produce."""
        self.L = lib()
        self.h = C.c_void_p(self.L.xmo_ref_new())
        self.contigs = []
        for name, seq in contigs:
            codes = encode(seq) if isinstance(seq, str) else np.ascontiguousarray(seq, dtype=np.uint8)
            self.contigs.append((name, codes))
            self.L.xmo_ref_add_contig_codes(self.h, name.encode(), codes.ctypes.data, len(codes))
        if custom_dup:
            rc = self.L.xmo_ref_finish_custom_dup(self.h, *custom_dup)
        elif min_interesting_size > 0:  # HashBlock_Database's constructor argument (HashBlock_Database.java:34), as a 3 Gb reference would derive it (:52)
            rc = self.L.xmo_ref_finish_min_interesting(self.h, 1 if mode == "api" else 0, 1 if enable_gapmers else 0, int(min_interesting_size))
        else:
            rc = self.L.xmo_ref_finish(self.h, 1 if mode == "api" else 0, 1 if enable_gapmers else 0)
        if rc:
            raise RuntimeError(self.L.xmo_last_error().decode())

    def __del__(self):
        try:
            self.L.xmo_ref_free(self.h)
        except Exception:
            pass

    def align(self, batch, params, threads=1):
        if not isinstance(batch, QueryBatch):
            batch = QueryBatch(batch)
        res = self.L.xmo_align_batch(self.h, C.byref(params), threads, batch.nq, batch.mate_count.ctypes.data, batch.mate_offset.ctypes.data,
                                     batch.mate_length.ctypes.data, batch.codes.ctypes.data, batch.expected_inner.ctypes.data, batch.deviation.ctypes.data)
        if not res:
            raise RuntimeError(self.L.xmo_last_error().decode())
        r = res.contents
        ints = np.ctypeslib.as_array(r.ints, shape=(max(r.nInts, 1),))[:r.nInts].copy()
        dbls = np.ctypeslib.as_array(r.dbls, shape=(max(r.nDbls, 1),))[:r.nDbls].copy()
        io = np.ctypeslib.as_array(r.intOff, shape=(r.nq + 1,)).copy()
        do = np.ctypeslib.as_array(r.dblOff, shape=(r.nq + 1,)).copy()
        counters = list(r.counters)
        self.L.xmo_result_free(res)
        return Streams(ints, dbls, io, do, counters)

    def index_info(self):
        a, b = C.c_int(), C.c_int()
        self.L.xmo_index_info(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def require_size(self, size):
        if self.L.xmo_ref_require_size(self.h, size):
            raise RuntimeError(self.L.xmo_last_error().decode())

    def table(self, L_):
        cap, mx, n, o = C.c_int(), C.c_int(), C.c_int64(), C.c_int64()
        if self.L.xmo_index_table_info(self.h, L_, C.byref(cap), C.byref(mx), C.byref(n), C.byref(o)):
            return None
        counts = np.zeros(cap.value, dtype=np.int32)
        pos = np.zeros(max(n.value, 1), dtype=np.int64)
        self.L.xmo_index_table_dump(self.h, L_, counts.ctypes.data, pos.ctypes.data)
        return dict(capacity=cap.value, maxCount=mx.value, counts=counts, positions=pos[:n.value])

    def dup_keys(self, contig):
        n = self.L.xmo_dup_keys(self.h, contig, None, 0)
        out = np.zeros(max(n, 1), dtype=np.int32)
        self.L.xmo_dup_keys(self.h, contig, out.ctypes.data, n)
        return out[:n]


def aligned_texts(seq_al, query_codes, ref_codes):
    """getAlignedTextA/B of one sequence alignment (for KATs that pin aligned reference text)."""
    a, b = [], []
    for (sa, sb, la, lb) in seq_al["blocks"]:
        a.append(decode(query_codes[sa:sa + la]) if la > 0 else "-" * lb)
        b.append(decode(ref_codes[sb:sb + lb]) if lb > 0 else "-" * la)
    return "".join(a), "".join(b)


def kat_local_align(chain, query, ref, params, max_ins, max_del):
    L = lib()
    a = C.create_string_buffer(4096)
    b = C.create_string_buffer(4096)
    pen = C.c_double()
    rc = L.xmo_kat_local_align(chain, query.encode(), ref.encode(), C.byref(params), max_ins, max_del, a, b, 4096, C.byref(pen))
    if rc:
        return None
    return a.value.decode(), b.value.decode(), pen.value


def pyramid_dump(codes):
    L = lib()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    cap = 64 * (len(codes) + 4) + 64
    while True:
        out = np.zeros((cap, 14), dtype=np.int32)
        n = L.xmo_pyramid_dump(codes.ctypes.data, len(codes), out.ctypes.data, cap)
        if n <= cap:
            return out[:n]
        cap = n


def pyramid_dump_multi(codes):
    """every block of the read-side pyramid, the possibilities of multi blocks included (12 ints per possibility: xmo_capi.cpp)"""
    L = lib()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    cap = 4096
    while True:
        out = np.zeros((cap, 12), dtype=np.int32)
        n = L.xmo_pyramid_dump_multi(codes.ctypes.data, len(codes), out.ctypes.data, cap)
        if n <= cap:
            return out[:n]
        cap = n


def kat_bound(params, query, query_rc, start_a, end_a, reference, start_b, end_b, predicted_best_offset=0):
    """PathAligner.align on one problem with the observer of the product's rejection filter beside it -> (verdict: 0 not taken / 1 taken / 2 rejected,
    found, nodes the search put).  Raises if the observer's bound rejected a search that returned an alignment."""
    q = np.ascontiguousarray(query, dtype=np.uint8)
    r = np.ascontiguousarray(reference, dtype=np.uint8)
    out = (C.c_int64 * 4)()
    if lib().xmo_kat_bound(C.byref(params), q.ctypes.data, len(q), 1 if query_rc else 0, start_a, end_a, r.ctypes.data, len(r), start_b, end_b, predicted_best_offset, out):
        raise RuntimeError(lib().xmo_last_error().decode())
    return int(out[0]), int(out[1]), int(out[2])
