"""ctypes binding of libxmapper_hip.so (include/xmapper_hip.h).  No CPU fallback: if the HIP library cannot be loaded
the import of this module fails loudly."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "_lib", "libxmapper_hip.so")


class XmParams(C.Structure):
    """xm_params = AlignmentParameters (M/AlignmentParameters.java:8-35)."""
    _fields_ = [("MutationPenalty", C.c_double), ("InsertionStart_Penalty", C.c_double), ("InsertionExtension_Penalty", C.c_double),
                ("DeletionStart_Penalty", C.c_double), ("DeletionExtension_Penalty", C.c_double), ("MaxErrorRate", C.c_double),
                ("UnalignedPenalty", C.c_double), ("AmbiguityPenalty", C.c_double), ("Max_PenaltySpan", C.c_double),
                ("MaxNumMatches", C.c_int32), ("reserved", C.c_int32)]


class XmRef(C.Structure):
    _fields_ = [("num_contigs", C.c_int32), ("names", C.POINTER(C.c_char_p)), ("codes", C.POINTER(C.c_void_p)), ("lengths", C.POINTER(C.c_int64))]


class XmBuildOpts(C.Structure):
    _fields_ = [("enable_gapmers", C.c_int32), ("min_interesting_size", C.c_int32), ("max_hashed_length", C.c_int32), ("dup_window", C.c_int32),
                ("dup_min_copies", C.c_int32), ("dup_min_length", C.c_int32), ("dup_max_length", C.c_int32), ("device", C.c_int32),
                ("host_only", C.c_int32), ("reserved", C.c_int32)]


class XmQueryBatch(C.Structure):
    _fields_ = [("num_queries", C.c_int64), ("mate_count", C.c_void_p), ("mate_offset", C.c_void_p), ("mate_length", C.c_void_p),
                ("codes", C.c_void_p), ("codes_length", C.c_int64), ("expected_inner", C.c_void_p), ("deviation", C.c_void_p)]


class XmResult(C.Structure):
    _fields_ = [("num_queries", C.c_int64), ("num_ints", C.c_int64), ("num_dbls", C.c_int64), ("ints", C.POINTER(C.c_int32)),
                ("dbls", C.POINTER(C.c_double)), ("int_off", C.POINTER(C.c_int64)), ("dbl_off", C.POINTER(C.c_int64)),
                ("counters", C.c_int64 * 16), ("kernel_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
                ("kernel_launches", C.c_int32), ("reserved", C.c_int32), ("prof", C.c_int64 * 16), ("extra", C.c_int64 * 8)]


ABI_VERSION = 2  # include/xmapper_hip.h, xm_abi_version(): xm_result.extra[] appended; xm_seed_probe_packed (the packed output layout under its own name)


class XmIndexInfo(C.Structure):
    _fields_ = [("num_contigs", C.c_int32), ("min_interesting_size", C.c_int32), ("max_hashed_length", C.c_int32), ("enable_gapmers", C.c_int32),
                ("dup_window", C.c_int32), ("position_bytes", C.c_int32), ("total_forward_size", C.c_int64), ("index_bytes", C.c_int64),
                ("num_positions", C.c_int64), ("dup_granularity", C.c_double), ("built_on_device", C.c_int32), ("bucket_line_bytes", C.c_int32),
                ("hash_seconds", C.c_double), ("duplication_seconds", C.c_double)]


EXPORTS = ["xm_last_error", "xm_build_stamp", "xm_abi_version", "xm_pinned_host_bytes", "xm_device_count", "xm_index_build", "xm_index_replicate", "xm_context_new", "xm_context_set_scratch", "xm_device_memory", "xm_index_save", "xm_index_load", "xm_index_ensure_length", "xm_index_free", "xm_index_get_info",
           "xm_index_table_info", "xm_index_table_shape", "xm_index_table_dump", "xm_index_bucket_stats", "xm_index_dup_keys", "xm_align_batch", "xm_result_free", "xm_batch_upload", "xm_batch_stage", "xm_batch_commit", "xm_align_resident", "xm_seed_probe_packed", "xm_measure_random_gather", "xm_test_local_align", "xm_test_bound_counters", "xm_test_bound", "xm_pileup_new", "xm_pileup_set_query_ends", "xm_pileup_read_middle", "xm_pileup_add_last", "xm_pileup_read", "xm_pileup_events", "xm_pileup_free"]


def build_library(force=False):
    """Compile libxmapper_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "xmapper_hip.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-j8", "-C", CSRC], stdout=subprocess.DEVNULL)
    return LIB_PATH


def source_stamp():
    """The digest the Makefile compiles into the library (xm_build_stamp): SHA-256 over csrc/*.h, *.hip in name order, then include/xmapper_hip.h."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(f for f in os.listdir(CSRC) if f.endswith(".h") or f.endswith(".hip")):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "xmapper_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def build_stamp():
    return lib().xm_build_stamp().decode()


def check_stamp():
    """Raises when the loaded library was not built from the sources in this tree."""
    have, want = build_stamp(), source_stamp()
    if have != want:
        raise RuntimeError("libxmapper_hip.so is stale: built from sources %s, the tree holds %s (run make -j8 -C mapper_amd/csrc)" % (have, want))
    return have


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("XM_LIB_PATH") or LIB_PATH  # (XM_LIB_PATH: A/B experiments with another build of the same library)
        if not os.path.exists(path):
            # no lazy build here: a process that runs under a profiler's preloaded library must not start make/hipcc children
            # (and a half-built library must never be picked up silently); __graft_entry__.build() or `make -C mapper_amd/csrc` builds it
            raise ImportError("%s is missing: build it first (python -c 'import __graft_entry__ as g; g.build()' or make -j8 -C mapper_amd/csrc); "
                              "mapper_amd has no CPU fallback" % path)
        global _hw_queues_at_load
        _hw_queues_at_load = os.environ.get("GPU_MAX_HW_QUEUES")  # what the HIP runtime will see when the first GPU-touching call initialises it
        L = C.CDLL(path)  # (XM_LIB_PATH: A/B experiments with another build of the same library)
        if not hasattr(L, "xm_abi_version") or L.xm_abi_version() != ABI_VERSION:
            raise ImportError("%s implements another version of include/xmapper_hip.h than this binding (%d): rebuild it (make -j8 -C mapper_amd/csrc)" % (path, ABI_VERSION))
        L.xm_last_error.restype = C.c_char_p
        L.xm_build_stamp.restype = C.c_char_p
        L.xm_pinned_host_bytes.restype = C.c_int64
        L.xm_pinned_host_bytes.argtypes = [C.POINTER(C.c_int64)]
        L.xm_index_build.argtypes = [C.POINTER(XmRef), C.POINTER(XmBuildOpts), C.POINTER(C.c_void_p)]
        L.xm_index_save.argtypes = [C.c_void_p, C.c_char_p]
        L.xm_index_replicate.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        L.xm_context_new.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.xm_context_set_scratch.argtypes = [C.c_void_p, C.c_int64]
        L.xm_device_memory.argtypes = [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.xm_index_load.argtypes = [C.c_char_p, C.POINTER(XmRef), C.POINTER(XmBuildOpts), C.POINTER(C.c_void_p)]
        L.xm_index_ensure_length.argtypes = [C.c_void_p, C.c_int32]
        L.xm_index_free.argtypes = [C.c_void_p]
        L.xm_index_get_info.argtypes = [C.c_void_p, C.POINTER(XmIndexInfo)]
        L.xm_index_table_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.xm_index_table_shape.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.xm_index_bucket_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.xm_index_table_dump.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.xm_index_dup_keys.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.xm_index_dup_keys.restype = C.c_int64
        L.xm_align_batch.argtypes = [C.c_void_p, C.POINTER(XmParams), C.POINTER(XmQueryBatch), C.POINTER(C.POINTER(XmResult))]
        L.xm_result_free.argtypes = [C.POINTER(XmResult)]
        L.xm_batch_upload.argtypes = [C.c_void_p, C.POINTER(XmQueryBatch)]
        L.xm_batch_stage.argtypes = [C.c_void_p, C.POINTER(XmQueryBatch)]
        L.xm_batch_commit.argtypes = [C.c_void_p]
        L.xm_align_resident.argtypes = [C.c_void_p, C.POINTER(XmParams), C.POINTER(C.POINTER(XmResult))]
        L.xm_seed_probe_packed.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.xm_measure_random_gather.argtypes = [C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_double)]
        L.xm_test_local_align.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(XmParams), C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32,
                                          C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int64)]
        L.xm_test_bound_counters.argtypes = [C.POINTER(C.c_int64)]
        L.xm_test_bound_counters.restype = None
        L.xm_test_bound.argtypes = [C.c_int32, C.POINTER(XmParams), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.POINTER(C.c_int64)]
        L.xm_pileup_new.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.xm_pileup_set_query_ends.argtypes = [C.c_void_p, C.c_double]
        L.xm_pileup_read_middle.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]
        L.xm_pileup_add_last.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.xm_pileup_read.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        L.xm_pileup_events.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.xm_pileup_events.restype = C.c_int64
        L.xm_pileup_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def make_ref(contigs):
    """contigs: list of (name, uint8 code array).  Returns (XmRef, keepalive)."""
    n = len(contigs)
    names = (C.c_char_p * n)(*[nm.encode() for nm, _ in contigs])
    arrays = [np.ascontiguousarray(c, dtype=np.uint8) for _, c in contigs]
    codes = (C.c_void_p * n)(*[a.ctypes.data for a in arrays])
    lengths = (C.c_int64 * n)(*[len(a) for a in arrays])
    ref = XmRef(n, names, codes, lengths)
    return ref, (names, arrays, codes, lengths)


def make_batch(mate_count, mate_offset, mate_length, codes, expected_inner, deviation):
    arrs = (np.ascontiguousarray(mate_count, dtype=np.int32), np.ascontiguousarray(mate_offset, dtype=np.int64),
            np.ascontiguousarray(mate_length, dtype=np.int32), np.ascontiguousarray(codes, dtype=np.uint8),
            np.ascontiguousarray(expected_inner, dtype=np.float64), np.ascontiguousarray(deviation, dtype=np.float64))
    b = XmQueryBatch(len(arrs[0]), arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data, len(arrs[3]),
                     arrs[4].ctypes.data, arrs[5].ctypes.data)
    return b, arrs


def copy_result(r):
    """XmResult -> dict of numpy arrays (copies; the C result can be freed afterwards)."""
    ints = np.ctypeslib.as_array(r.ints, shape=(max(r.num_ints, 1),))[:r.num_ints].copy()
    dbls = np.ctypeslib.as_array(r.dbls, shape=(max(r.num_dbls, 1),))[:r.num_dbls].copy()
    io = np.ctypeslib.as_array(r.int_off, shape=(r.num_queries + 1,)).copy()
    do = np.ctypeslib.as_array(r.dbl_off, shape=(r.num_queries + 1,)).copy()
    return dict(ints=ints, dbls=dbls, int_off=io, dbl_off=do, counters=list(r.counters), kernel_ms=r.kernel_ms, h2d_ms=r.h2d_ms,
                d2h_ms=r.d2h_ms, kernel_launches=r.kernel_launches, prof=list(r.prof), extra=list(r.extra))


class _ResultOwner:
    """Frees the C result when the last numpy view of its streams is gone."""

    def __init__(self, L, res):
        self._L, self._res = L, res

    def __del__(self):
        try:
            self._L.xm_result_free(self._res)
        except Exception:
            pass


def view_result(L, res):
    """POINTER(XmResult) -> dict of numpy arrays that alias the (pinned) C streams: no copy of the ~100 MB a 1M-read batch
    returns.  Every array keeps the owner alive through its ctypes base object; the owner frees the result."""
    r = res.contents
    owner = _ResultOwner(L, res)

    def view(ptr, ctype, n):
        buf = (ctype * max(int(n), 1)).from_address(C.addressof(ptr.contents))
        buf._xm_owner = owner
        return np.frombuffer(buf, dtype=np.dtype(ctype))[:int(n)]

    return dict(ints=view(r.ints, C.c_int32, r.num_ints), dbls=view(r.dbls, C.c_double, r.num_dbls), int_off=view(r.int_off, C.c_int64, r.num_queries + 1),
                dbl_off=view(r.dbl_off, C.c_int64, r.num_queries + 1), counters=list(r.counters), kernel_ms=r.kernel_ms, h2d_ms=r.h2d_ms,
                d2h_ms=r.d2h_ms, kernel_launches=r.kernel_launches, prof=list(r.prof), extra=list(r.extra))


_hw_queues_at_load = None  # GPU_MAX_HW_QUEUES as it stood when the library was loaded (lib()): the value the HIP runtime reads when it initialises


def want_hardware_queues(n=8):
    """The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4); every context has a compute and a copy stream, so with
    more than two contexts of a GPU launches of different contexts would share a queue and run one after the other (repeat-rich reads, four contexts: 1.5 -> 2.2 M
    reads/s with 8 queues, profiles/r04/NOTES.md 13).  The variable is read when the runtime initialises, so it is the ENTRY POINTS that set it (python -m mapper_amd,
    bench.py) - importing the package does not touch the process environment.  A program that embeds the library and wants three contexts or more per GPU calls
    this (or sets the variable) before anything in the process touches the GPU; returns False when the runtime is already up with another setting."""
    import warnings
    if _lib is None:
        cur = os.environ.get("GPU_MAX_HW_QUEUES")
        if cur is None:
            os.environ["GPU_MAX_HW_QUEUES"] = str(n)
            return True
        if int(cur) >= n:
            return True
        warnings.warn("GPU_MAX_HW_QUEUES is set to %s: with fewer than %d hardware queues, three or more contexts of a GPU share queues and their launches run one after the other" % (cur, n))
        return False
    # the library is loaded: what counts is what the environment held at that moment (a value set afterwards is never read by the runtime; under rocprofv3 the
    # profiler's preloaded library initialises the runtime before Python starts, so the variable must be exported in the shell in front of rocprofv3)
    seen = _hw_queues_at_load
    if seen is not None and int(seen) >= n:
        return True
    warnings.warn("GPU_MAX_HW_QUEUES was %s when the library was loaded and the HIP runtime may already be initialised: more than two contexts per GPU will share hardware queues "
                  "(set the variable before the first import that touches the GPU)" % (seen or "unset"))
    return False


def pinned_host_bytes():
    """(bytes of page-locked host memory the library's result-buffer pool holds now, the most it ever held) in this process."""
    hw = C.c_int64()
    now = lib().xm_pinned_host_bytes(C.byref(hw))
    return int(now), int(hw.value)
