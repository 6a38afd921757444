"""The drop-in boundary used by a foreign binder: bindings/c/binding_test.c calls xm_index_build / xm_align_batch / xm_result_free through the
JNI shim's marshalling functions (bindings/java/xmapper_jni.c, part 1) from plain C - no Python, no ctypes in the process that aligns
(AlignerWorker.java:177-231, 256-261 is what that call replaces).  CPU tier: the program links and the error contract holds; GPU tier: its result
streams equal the oracle's bit for bit."""
import os
import struct
import subprocess
import numpy as np
import pytest

import oracle_lib as o
from helpers import se_batch, pe_batch
from mapper_amd import synth

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
EXE = os.path.join(ROOT, "bindings", "_build", "binding_test")


def build():
    import fcntl
    os.makedirs(os.path.join(ROOT, "bindings", "_build"), exist_ok=True)
    with open(os.path.join(ROOT, "bindings", "_build", ".lock"), "w") as lock:  # (pytest-xdist workers build side by side)
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bindings")], stdout=subprocess.DEVNULL)
    return EXE


def test_plain_c_program_links_every_entry_point():
    exe = build()
    out = subprocess.run([exe, "symbols"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("symbols 33 stamp "), out.stdout + out.stderr


def test_error_contract_from_c():
    """A failing call returns non-zero, produces no partial result, and xm_last_error() describes it (include/xmapper_hip.h)."""
    exe = build()
    out = subprocess.run([exe, "errors"], capture_output=True, text=True)
    assert out.returncode == 0 and "errors ok" in out.stdout, out.stdout + out.stderr


def test_jni_shim_is_complete_source():
    """bindings/java/xmapper_jni.c carries the whole marshalling (no elisions): every native method NativeAligner.java declares has its
    JNI entry point, and both parts compile as C99 (the JNI part is syntax-checked only where a jni.h exists)."""
    src = open(os.path.join(ROOT, "bindings", "java", "xmapper_jni.c")).read()
    java = open(os.path.join(ROOT, "bindings", "java", "mapper", "NativeAligner.java")).read()
    for name in ("buildIndex", "newContext", "freeIndex", "alignBatch"):
        assert "native" in java and name in java
        assert "Java_mapper_NativeAligner_" + name in src
    assert "/* pin each" not in src and "..." not in src.split("part 2")[1]
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "bindings", "java", "xmapper_jni.c")])


def write_case(path, contigs, batch, params):
    with open(path, "wb") as f:
        f.write(struct.pack("<q", len(contigs)))
        for _, codes in contigs:
            f.write(struct.pack("<q", len(codes)))
            f.write(np.ascontiguousarray(codes, np.uint8).tobytes())
        f.write(struct.pack("<qq", batch.nq, len(batch.codes)))
        f.write(struct.pack("<9d", params.MutationPenalty, params.InsertionStart_Penalty, params.InsertionExtension_Penalty, params.DeletionStart_Penalty,
                            params.DeletionExtension_Penalty, params.MaxErrorRate, params.UnalignedPenalty, params.AmbiguityPenalty, params.Max_PenaltySpan))
        f.write(struct.pack("<q", params.MaxNumMatches))
        for a, t in ((batch.mate_count, np.int32), (batch.mate_offset, np.int64), (batch.mate_length, np.int32), (batch.codes, np.uint8), (batch.expected_inner, np.float64),
                     (batch.deviation, np.float64)):
            f.write(np.ascontiguousarray(a, t).tobytes())


@pytest.mark.gpu
def test_plain_c_alignment_equals_oracle(tmp_path):
    exe = build()
    ref = synth.synthetic_reference(120_000, seed=0xC0DE)
    contigs = [("c0", ref[:80_000]), ("c1", ref[80_000:])]
    reads = synth.synthetic_single_end(ref[:80_000], 3000, seed=11)[0]
    m1, m2 = synth.synthetic_paired_end(ref[:80_000], 1000, seed=12)[:2]
    se, pe = se_batch(reads), pe_batch(m1, m2)
    queries = [([r], 0.0, 1.0) for r in reads] + [([m1[i], m2[i]], 100.0, 50.0) for i in range(len(m1))]
    batch = o.QueryBatch(queries)
    params = o.make_params()
    write_case(tmp_path / "case.bin", contigs, batch, params)
    out = subprocess.run([exe, "align", str(tmp_path / "case.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = open(tmp_path / "out.bin", "rb").read()
    nq, ni, nd = struct.unpack_from("<qqq", raw, 0)
    at = 24
    ints = np.frombuffer(raw, np.int32, ni, at); at += 4 * ni
    dbls = np.frombuffer(raw, np.float64, nd, at); at += 8 * nd
    int_off = np.frombuffer(raw, np.int64, nq + 1, at); at += 8 * (nq + 1)
    dbl_off = np.frombuffer(raw, np.int64, nq + 1, at)
    want = o.OracleReference(contigs, mode="mapper").align(batch, params)
    assert nq == batch.nq
    assert np.array_equal(int_off, want.int_off) and np.array_equal(dbl_off, want.dbl_off)
    assert np.array_equal(ints, want.ints) and np.array_equal(dbls.view(np.int64), want.dbls.view(np.int64))
    assert se.nq + pe.nq == nq
