"""Host simulation of the device kernel (tests/hostsim/xm_hostsim.cpp) — TEST HARNESS ONLY, never used by the product."""
import ctypes as C
import os
import subprocess
import numpy as np

from mapper_amd import _capi
import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "hostsim", "xm_hostsim.cpp")
OUT = os.path.join(ROOT, "tests", "_build", "libxm_hostsim.so")


def build():
    """XMSIM_POISON=1 (read when the library is first loaded): the variant built with -DXM_ARENA_POISON - every arena allocation starts as garbage, so a structure
    that is read before it is written (on the GPU: a result that depends on which read used the lane before) shows up as a difference from the oracle."""
    poison = os.environ.get("XMSIM_POISON") == "1"
    out = OUT.replace(".so", "_poison.so") if poison else OUT
    deps = [SRC] + [os.path.join(ROOT, "mapper_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "mapper_amd", "csrc")) if f.endswith(".h")]
    import fcntl
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out + ".lock", "w") as lock:  # (pytest-xdist workers build side by side)
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"] +
                                  (["-DXM_ARENA_POISON"] if poison else []) + ["-o", out + ".tmp", SRC])
            os.replace(out + ".tmp", out)
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.xmsim_last_error.restype = C.c_char_p
        L.xmsim_index_build.restype = C.c_void_p
        L.xmsim_index_build.argtypes = [C.POINTER(_capi.XmRef), C.POINTER(_capi.XmBuildOpts)]
        L.xmsim_index_free.argtypes = [C.c_void_p]
        L.xmsim_align_batch.argtypes = [C.c_void_p, C.POINTER(_capi.XmParams), C.POINTER(_capi.XmQueryBatch), C.POINTER(C.POINTER(_capi.XmResult))]
        L.xmsim_result_free.argtypes = [C.POINTER(_capi.XmResult)]
        L.xmsim_table_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        L.xmsim_table_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.xmsim_index_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.xmsim_ensure_length.argtypes = [C.c_void_p, C.c_int]
        L.xmsim_dup_keys.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        L.xmsim_dup_keys.restype = C.c_int64
        L.xmsim_pyramid_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        L.xmsim_pyramid_dump.restype = C.c_int64
        L.xmsim_pyramid_dump_multi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        L.xmsim_pyramid_dump_multi.restype = C.c_int64
        L.xmsim_kat_multi_contains.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.xmsim_kat_position_codec.argtypes = [C.c_int, C.c_int]
        L.xmsim_test_bound.argtypes = [C.POINTER(_capi.XmParams), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.xmsim_set_wave_mode.argtypes = [C.c_int]
        L.xmsim_wave_status_counts.argtypes = [C.c_void_p, C.c_int]
        _lib = L
    return _lib


def build_opts(mode="mapper", enable_gapmers=True, custom_dup=None, max_hashed_length=0, host_only=0):
    o = _capi.XmBuildOpts()
    o.enable_gapmers = 1 if enable_gapmers else 0
    o.min_interesting_size = -1
    o.max_hashed_length = max_hashed_length
    o.dup_window = 1 if mode == "api" else 1000
    o.dup_min_copies = 2
    o.dup_min_length = o.dup_max_length = -1
    if custom_dup:
        o.dup_min_length, o.dup_max_length, o.dup_min_copies, o.dup_window = custom_dup
    o.device = -1
    o.host_only = host_only
    return o


class SimReference:
    def __init__(self, contigs, mode="mapper", enable_gapmers=True, custom_dup=None):
        self.L = lib()
        cs = [(n, oracle_lib.encode(s) if isinstance(s, str) else np.ascontiguousarray(s, dtype=np.uint8)) for n, s in contigs]
        ref, self._keep = _capi.make_ref(cs)
        o = build_opts(mode, enable_gapmers, custom_dup)
        self.h = self.L.xmsim_index_build(C.byref(ref), C.byref(o))
        if not self.h:
            raise RuntimeError(self.L.xmsim_last_error().decode())
        self.h = C.c_void_p(self.h)

    def __del__(self):
        try:
            self.L.xmsim_index_free(self.h)
        except Exception:
            pass

    def align(self, batch, params):
        if not isinstance(batch, oracle_lib.QueryBatch):
            batch = oracle_lib.QueryBatch(batch)
        b, keep = _capi.make_batch(batch.mate_count, batch.mate_offset, batch.mate_length, batch.codes, batch.expected_inner, batch.deviation)
        p = _capi.XmParams()
        for f, _ in _capi.XmParams._fields_:
            if f != "reserved":
                setattr(p, f, getattr(params, f))
        res = C.POINTER(_capi.XmResult)()
        if self.L.xmsim_align_batch(self.h, C.byref(p), C.byref(b), C.byref(res)):
            raise RuntimeError(self.L.xmsim_last_error().decode())
        d = _capi.copy_result(res.contents)
        self.L.xmsim_result_free(res)
        st = oracle_lib.Streams(d["ints"], d["dbls"], d["int_off"], d["dbl_off"], d["counters"])
        st.extra = d["extra"]  # the rejection filter in front of PathAligner: searches taken, rejected, cells, active (include/xmapper_hip.h)
        return st

    def index_info(self):
        a, b = C.c_int32(), C.c_int32()
        self.L.xmsim_index_info(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def ensure_length(self, n):
        if self.L.xmsim_ensure_length(self.h, n):
            raise RuntimeError(self.L.xmsim_last_error().decode())

    def table(self, L_):
        cap, mx, n = C.c_int32(), C.c_int32(), C.c_int64()
        if self.L.xmsim_table_info(self.h, L_, C.byref(cap), C.byref(mx), C.byref(n)):
            return None
        counts = np.zeros(cap.value, dtype=np.int32)
        pos = np.zeros(max(n.value, 1), dtype=np.int64)
        self.L.xmsim_table_dump(self.h, L_, counts.ctypes.data, pos.ctypes.data)
        return dict(capacity=cap.value, maxCount=mx.value, counts=counts, positions=pos[:n.value])

    def dup_keys(self, contig):
        n = self.L.xmsim_dup_keys(self.h, contig, None, 0)
        out = np.zeros(max(n, 1), dtype=np.int32)
        self.L.xmsim_dup_keys(self.h, contig, out.ctypes.data, n)
        return out[:n]


def pyramid_dump(codes):
    L = lib()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    cap = 64 * (len(codes) + 4) + 64
    out = np.zeros((cap, 14), dtype=np.int32)
    n = L.xmsim_pyramid_dump(codes.ctypes.data, len(codes), out.ctypes.data, cap)
    assert 0 <= n <= cap
    return out[:n]


def pyramid_dump_multi(codes, scale=1):
    """the product's read-side pyramid with its multi blocks, pools sized for `scale` as compInit sizes them; None = a capacity was exceeded at that scale"""
    L = lib()
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    cap = 4096
    while True:
        out = np.zeros((cap, 12), dtype=np.int32)
        n = L.xmsim_pyramid_dump_multi(codes.ctypes.data, len(codes), scale, out.ctypes.data, cap)
        if n < 0:
            return None
        if n <= cap:
            return out[:n]
        cap = n


def set_wave_mode(mode):
    """0: lane-per-read pass sequence only; 1: the wave-per-read form's light tier first; 2: + its chain tier; 3: + its search tier (what the product runs)."""
    lib().xmsim_set_wave_mode(mode)


def wave_status_counts(reset=True):
    """How the wave form left the reads since the last reset: index = status (0 finished there, 8 left to the lane-per-read passes, 9 handed to the heavy tier)."""
    out = np.zeros(16, dtype=np.int64)
    lib().xmsim_wave_status_counts(out.ctypes.data, 1 if reset else 0)
    return out


def test_bound(params, query, query_rc, start_a, end_a, reference, start_b, end_b, predicted_best_offset=0):
    """the rejection filter of xm_bound.h alone (host-compiled) -> (taken, rejected, cells)"""
    q = np.ascontiguousarray(query, dtype=np.uint8)
    r = np.ascontiguousarray(reference, dtype=np.uint8)
    p = _capi.XmParams()
    for f, _ in _capi.XmParams._fields_:
        if f != "reserved":
            setattr(p, f, getattr(params, f))
    out = (C.c_int64 * 3)()
    lib().xmsim_test_bound(C.byref(p), q.ctypes.data, len(q), 1 if query_rc else 0, start_a, end_a, r.ctypes.data, len(r), start_b, end_b, predicted_best_offset, out)
    return int(out[0]), int(out[1]), int(out[2])
