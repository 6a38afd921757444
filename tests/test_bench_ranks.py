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
