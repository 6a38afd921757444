"""GPU tier: the lane-per-read kernel's results must not depend on how its code happens to be generated (profiles/r04/NOTES.md 14).  ROCm 7.2's default allocator for
the VGPRs that hold spilled SGPRs miscompiles these kernels' out-of-line device functions under partial execution masks: a build that differs from the product only by
the in-kernel timers (-DXM_PROFILE=2) lost the results of ~0.3 % of the reads of the bench batch whenever several reads shared a wavefront.  The Makefile's
-mllvm -wwm-regalloc=basic is what prevents it; this test compiles that instrumented variant with the Makefile's flags and aligns the bench batch with it."""
import os
import re
import shutil
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mapper_amd", "csrc")
LIB = os.path.join(ROOT, "mapper_amd", "_lib")
HIPCC = "/opt/rocm/bin/hipcc"


def test_instrumented_build_gives_the_product_s_results():
    objs = [os.path.join(LIB, f) for f in sorted(os.listdir(LIB)) if f.endswith(".o") and f != "xm_capi.o"] if os.path.isdir(LIB) else []
    if not os.path.exists(HIPCC) or len(objs) < 3:
        pytest.skip("needs hipcc and the product's object files beside the library (make -C mapper_amd/csrc)")
    flags = re.search(r"^FLAGS := (.*)$", open(os.path.join(CSRC, "Makefile")).read(), re.M).group(1)
    assert "-wwm-regalloc=basic" in flags, "the Makefile lost the allocator flag (profiles/r04/NOTES.md 14)"
    flags = flags.replace("$(ARCH)", "gfx950").replace("$(EXTRA)", "").split()
    out = os.path.join(ROOT, "tests", "_build", "codegen")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    obj, lib = os.path.join(out, "xm_capi.o"), os.path.join(out, "libxmapper_hip.so")
    subprocess.run([HIPCC] + flags + ["-Wno-unused-command-line-argument", "-DXM_BUILD_STAMP=\"variant\"", "-DXM_PROFILE=2", "-c", "-o", obj, "xm_capi.hip"], cwd=CSRC, check=True,
                   capture_output=True, timeout=1500)
    subprocess.run([HIPCC] + flags + ["-Wno-unused-command-line-argument", "-shared", "-o", lib, obj] + objs, cwd=CSRC, check=True, capture_output=True, timeout=600)
    env = dict(os.environ, XM_LIB_PATH=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_codegen_check_r04.py"), "X=0,XM_FULL_LPW=8"], env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if "nodes" in l]
    assert r.returncode == 0 and len(lines) == 2, r.stdout[-2000:] + r.stderr[-2000:]
    assert all(l.rstrip().endswith("OK") for l in lines), "\n".join(lines)
    shutil.rmtree(out, ignore_errors=True)
