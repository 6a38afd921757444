"""bench.py's multi-rank flow without a launcher (`python bench.py --gpus N` starts its own ranks before touching a GPU): two ranks over gloo that
share GPU 0 (--force-device 0; the one-GPU boxes of the test tier have no second GPU), the barrier / max-over-ranks timing, rank 0's one JSON line."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--force-device", "0", "--reads", "60000", "--steps", "2", "--warmup", "1",
           "--contexts", "1", "--cpu-sample", "0", "--seed-probes", "0", "--wave-steps", "0", "--single-context-steps", "0", "--stream-batches", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints, the other rank does not
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2 and line["unit"] == "Mreads/s"
    assert line["value"] > 0 and abs(line["value"] - 2 * 60000 * 2 / (line["ms_per_step"] * 2 * 1e-3) / 1e6) < 0.02 * line["value"]   # whole-job reads over the slowest rank's time
    assert line["cpu_baseline"] is None and line["seed_probe"] is None and line["wave_form"] is None   # side measurements are N=1 only
    assert line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_ranks_share_one_reference_and_one_index(tmp_path):
    """configs[3]'s shape on N ranks of one node (here two gloo ranks that share GPU 0, the GRCh38-shaped reference at 1/200 scale): rank 0 generates the
    reference and builds the index once, the other rank maps the reference file and loads the index (xm_index_load) - no second hashing, no second copy
    of the generator's work; both ranks then align their own pairs."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    share = tmp_path / "share"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--force-device", "0", "--config", "3shape", "--big-scale", "0.005", "--share-dir", str(share),
           "--reads", "20000", "--steps", "2", "--warmup", "1", "--contexts", "1", "--cpu-sample", "0", "--seed-probes", "0", "--wave-steps", "0", "--single-context-steps", "0", "--stream-batches", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["aligned_reads"] > 0.95 * 20000
    assert (share / "whole.npy").exists() and (share / "index.xmidx").exists() and (share / "layout.npz").exists()
