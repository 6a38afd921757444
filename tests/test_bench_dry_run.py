"""bench.py's multi-rank flow without GPUs (round-5 verdict, "multi-GPU readiness without hardware"): `python bench.py --gpus 8 --engine hostsim` starts eight
ranks (fresh child processes, gloo rendezvous on 127.0.0.1), every rank aligns its own shard through the host simulation of the kernel sources, the timed
region is bracketed by barriers, the time is the maximum over the ranks, and rank 0 prints ONE line.  Checked: n_gpus, weak scaling, the ranks' batches differ
pairwise (seed shift 7919 x rank), the value is the whole job's reads over the slowest rank's time.  The line says it is a dry run; it measures nothing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks", [8])
def test_bench_eight_ranks_dry_run_prints_one_rank0_line(ranks):
    reads = 48
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--engine", "hostsim", "--backend", "gloo", "--reads", str(reads), "--ref-len", "60000",
           "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--seed-probes", "0", "--stream-batches", "0", "--single-context-steps", "0", "--wave-steps", "0", "--contexts", "2",
           "--end-to-end-reads", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert "DRY RUN" in d["data"]
    assert len(d["rank_batch_digests"]) == ranks and len(set(d["rank_batch_digests"])) == ranks   # every rank aligned other reads
    # value = all ranks' reads over the (max-over-ranks) time of the timed steps
    assert abs(d["value"] - ranks * reads * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3) / 1e6) <= 1e-3 * max(d["value"], 1e-9) + 1e-4
    assert d["config"]["reads_per_gpu"] == reads and d["config"]["parallelism"] == "reads sharded x%d" % ranks
