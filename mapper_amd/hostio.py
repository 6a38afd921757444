"""ctypes binding of libxm_hostio.so (include/xmapper_hostio.h): FASTA / FASTQ files -> batch arrays, result streams -> SAM text, in native code.

Host-side plumbing of the standalone harness, not part of the alignment path and not part of the drop-in boundary: the
Java host keeps its own readers and writers (Mapper.java:699-732).  The per-object path of mapper_amd/cli.py (api.Query, sam.records) remains the
reference for the formats; tests/test_hostio.py holds the two equal byte for byte."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_lib", "libxm_hostio.so")


class XmioBatch(C.Structure):
    _fields_ = [("num_queries", C.c_int64), ("mate_count", C.POINTER(C.c_int32)), ("mate_offset", C.POINTER(C.c_int64)), ("mate_length", C.POINTER(C.c_int32)),
                ("codes", C.POINTER(C.c_uint8)), ("codes_length", C.c_int64), ("names", C.c_void_p), ("name_off", C.POINTER(C.c_int64)),
                ("quals", C.c_void_p), ("qual_off", C.POINTER(C.c_int64)), ("has_qual", C.POINTER(C.c_uint8)), ("store", C.c_void_p)]


class XmioStats(C.Structure):
    _fields_ = [("num_queries", C.c_int64), ("num_aligned", C.c_int64), ("total_aligned_length", C.c_int64), ("num_indels", C.c_int64), ("total_penalty", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it first (make -C mapper_amd/hostio, or __graft_entry__.build())" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.xmio_last_error.restype = C.c_char_p
        L.xmio_open.restype = C.c_void_p
        L.xmio_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32]
        L.xmio_next.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.POINTER(XmioBatch))]
        L.xmio_batch_free.argtypes = [C.POINTER(XmioBatch)]
        L.xmio_close.argtypes = [C.c_void_p]
        L.xmio_write_batch.argtypes = [C.POINTER(XmioBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_int32,
                                       C.POINTER(XmioStats)]
        L.xmio_java_double.argtypes = [C.c_double, C.c_char_p, C.c_int32]
        _lib = L
    return _lib


class Batch:
    """One batch of queries as read from the files: numpy views of the native arrays (valid while this object lives)."""

    def __init__(self, ptr, expected_inner, deviation):
        self._ptr = ptr
        b = ptr.contents
        n = int(b.num_queries)
        self.num_queries = n
        self.mate_count = np.ctypeslib.as_array(b.mate_count, shape=(n,))
        self.mate_offset = np.ctypeslib.as_array(b.mate_offset, shape=(2 * n,))
        self.mate_length = np.ctypeslib.as_array(b.mate_length, shape=(2 * n,))
        self.codes = np.ctypeslib.as_array(b.codes, shape=(max(int(b.codes_length), 1),))
        paired = self.mate_count > 1
        self.expected_inner = np.where(paired, float(expected_inner), 0.0)   # single-end Query: expected inner distance 0, deviation 1 (api.Query)
        self.deviation = np.where(paired, float(deviation), 1.0)

    def arrays(self):
        return self.mate_count, self.mate_offset, self.mate_length, self.codes, self.expected_inner, self.deviation

    def __len__(self):
        return self.num_queries

    def close(self):
        if self._ptr is not None:
            lib().xmio_batch_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_batches(path1, path2=None, batch_size=1_000_000, split=0, keep_qualities=False, expected_inner=0.0, deviation=1.0):
    """Generator of Batch objects over a FASTA / FASTQ file (or a pair of them, read in step)."""
    L = lib()
    r = L.xmio_open(os.fsencode(path1), os.fsencode(path2) if path2 else None, int(split), 1 if keep_qualities else 0)
    if not r:
        raise OSError(L.xmio_last_error().decode())
    r = C.c_void_p(r)
    try:
        while True:
            ptr = C.POINTER(XmioBatch)()
            if L.xmio_next(r, int(batch_size), C.byref(ptr)):
                raise ValueError(L.xmio_last_error().decode())
            if not ptr:
                return
            yield Batch(ptr, expected_inner, deviation)
    finally:
        L.xmio_close(r)


class Writer:
    """SAM records (and the unaligned-query file) of the batches of a job, written in query order through file descriptors; keeps Mapper.run's statistics."""

    def __init__(self, contig_names, sam_file=None, unaligned_file=None, threads=None):
        self._names = (C.c_char_p * len(contig_names))(*[n.encode() for n in contig_names])
        self._n = len(contig_names)
        self._sam, self._un = sam_file, unaligned_file
        self.stats = XmioStats()
        self.threads = int(threads or min(32, os.cpu_count() or 1))

    def write(self, batch, result):
        """result: anything with ints / dbls / int_off / dbl_off (api.BatchResult)."""
        for f in (self._sam, self._un):
            if f is not None:
                f.flush()
        ints = np.ascontiguousarray(result.ints, dtype=np.int32)
        dbls = np.ascontiguousarray(result.dbls, dtype=np.float64)
        io = np.ascontiguousarray(result.int_off, dtype=np.int64)
        do = np.ascontiguousarray(result.dbl_off, dtype=np.int64)
        if len(io) != batch.num_queries + 1:
            raise ValueError("result streams of %d queries for a batch of %d" % (len(io) - 1, batch.num_queries))
        if lib().xmio_write_batch(batch._ptr, ints.ctypes.data, dbls.ctypes.data, io.ctypes.data, do.ctypes.data, self._n, self._names,
                                  self._sam.fileno() if self._sam is not None else -1, self._un.fileno() if self._un is not None else -1, self.threads, C.byref(self.stats)):
            raise RuntimeError(lib().xmio_last_error().decode())


def java_double(x):
    buf = C.create_string_buffer(64)
    n = lib().xmio_java_double(float(x), buf, 64)
    return buf.raw[:n].decode()
