#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: M reads/s aligned (150 bp) on MI355X.

A "step" is one pass of the hot path (xm_align_resident: every kernel pass + the copy of the result streams back to the
host and their ordering) over one batch of synthetic input that is already resident in HBM: BASELINE.json configs[1],
1,000,000 synthetic 150 bp single-end reads vs the 5 Mb synthetic E. coli-sized reference, per GPU.
With --gpus N each rank owns one GPU, holds a replica of the index and aligns its own 1 M reads (no collective on the data
path; scaling = weak).  Rank 0 prints ONE JSON line.

The line carries `roofline` (algorithmic bytes of SURVEY.md §8(d) per second of align-kernel time, against the 8 TB/s
HBM peak) and `cpu_baseline` (the CPU oracle, a port of the Java path, timed on this box's host cores on a bounded
sample).  The oracle is only the checker/baseline here; nothing under oracle/ is on the measured GPU path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step (configs[1]: 1,000,000)")
    ap.add_argument("--ref-len", type=int, default=5_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--seed-probes", type=int, default=16_000_000, help="bucket-header probes of the seed-lookup micro-benchmark (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to exercise the multi-rank path on a one-GPU box)")
    ap.add_argument("--force-device", type=int, default=-1, help="testing: every rank uses this GPU instead of its LOCAL_RANK")
    ap.add_argument("--cpu-sample", type=int, default=200_000, help="reads of the same workload timed on the host cores (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if args.force_device >= 0:
        local_rank = args.force_device
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    assert args.gpus == world, "--gpus must equal the number of launched ranks"

    from mapper_amd import api, synth

    ref = synth.synthetic_reference(args.ref_len, seed=0xEC011)
    reads, _, _ = synth.synthetic_single_end(ref, args.reads, read_len=args.read_len, seed=0x5EED0001 + 7919 * rank)
    nq = len(reads)
    mc = np.ones(nq, np.int32)
    mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * args.read_len
    ml = np.zeros(2 * nq, np.int32); ml[0::2] = args.read_len
    codes = np.ascontiguousarray(reads.reshape(-1))
    params = api.AlignmentParameters()  # Mapper.main defaults

    t0 = time.time()
    db = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=args.read_len, device=local_rank)
    index_build_s = time.time() - t0
    db.upload_arrays(mc, mo, ml, codes, np.zeros(nq), np.ones(nq))  # inputs resident in HBM before the timed region

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        r = db.align_resident(params)
    barrier()
    t_start = time.perf_counter()
    kernel_ms = 0.0
    launches = 0
    d2h_ms = 0.0
    for _ in range(args.steps):
        r = db.align_resident(params)
        kernel_ms += r.kernel_ms
        launches += r.kernel_launches
        d2h_ms += r.d2h_ms
    barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        info = db.info()
        c = r.counters
        # algorithmic bytes of one step: B_in + 8*P + sum_hits(B_pos + 20) + window bytes + B_out
        pos_bytes = info["position_bytes"]
        alg_bytes = c[10] + 8 * c[1] + c[3] * (pos_bytes + 20) + c[9] + 4 * len(r.ints) + 8 * len(r.dbls)
        avg_launch_ms = kernel_ms / max(launches, 1)
        achieved = (alg_bytes * args.steps / max(launches, 1)) / (avg_launch_ms * 1e-3) / 1e9  # GB/s: bytes per launch / avg launch duration
        aligned = int(sum(1 for q in range(nq) if r.ints[r.int_off[q] + 1] > 0)) if nq <= 2_000_000 else -1

        cpu = None
        if args.cpu_sample > 0:
            import oracle_lib
            n = min(args.cpu_sample, nq)
            o = oracle_lib.OracleReference([("ecoli_syn", ref)], mode="mapper")
            cores = os.cpu_count() or 1
            warm = oracle_lib.QueryBatch.from_arrays(mc[:64], mo[:128], ml[:128], codes, np.zeros(64), np.ones(64))
            o.align(warm, oracle_lib.make_params(), threads=1)   # builds the index (not timed, like the reference's prepare())
            o.require_size(args.read_len)
            b = oracle_lib.QueryBatch.from_arrays(mc[:n], mo[:2 * n], ml[:2 * n], codes, np.zeros(n), np.ones(n))
            t1 = time.perf_counter()
            w = o.align(b, oracle_lib.make_params(), threads=cores)
            cpu_s = time.perf_counter() - t1
            same = bool(np.array_equal(w.ints, r.ints[:r.int_off[n]]) and np.array_equal(w.dbls.view(np.int64), r.dbls[:r.dbl_off[n]].view(np.int64)))
            cpu = {"value": round(n / cpu_s / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port",
                   "sample": "first %d reads of the same batch, oracle (C++ port of the Java path) with one worker thread per host core, index build excluded; "
                             "GPU result bit-identical on the sample: %s" % (n, same)}

        # seed-lookup micro-kernel: bulk PackedMap.get on the device, bytes = 8 per bucket-header probe + B_pos per
        # fetched position, next to the measured random-64 B-sector ceiling of this GPU
        seed = None
        if args.seed_probes > 0:
            rng = np.random.default_rng(12345)
            lo, hi = info["min_interesting_size"], info["max_hashed_length"]
            used = rng.integers(lo, hi + 1, size=args.seed_probes, dtype=np.int32)
            keys = rng.integers(-2**31, 2**31 - 1, size=args.seed_probes, dtype=np.int64).astype(np.int32)
            db.seed_probe(used[:4096], keys[:4096], 0)
            counts, _, ms_hdr = db.seed_probe(used, keys, 0)
            sectors_per_s, gather_ms = api.measure_random_gather(4 << 30, 1 << 26, local_rank)
            n2 = args.seed_probes // 4
            c2, _, ms_pos = db.seed_probe(used[:n2], keys[:n2], 4)
            fetched = int(np.minimum(np.maximum(c2, 0), 4).sum())
            hdr_gbs = 8.0 * args.seed_probes / (ms_hdr * 1e-3) / 1e9
            seed = {"kernel": "xm_seed_probe_kernel", "probes": args.seed_probes, "kernel_ms": round(ms_hdr, 4),
                    "probes_per_s": round(args.seed_probes / (ms_hdr * 1e-3), 1), "achieved": round(hdr_gbs, 2), "unit": "GB/s", "peak": 8000.0,
                    "frac": round(hdr_gbs / 8000.0, 5),
                    "random_64B_gather_ceiling_sectors_per_s": round(sectors_per_s, 1),
                    "frac_of_gather_ceiling": round(args.seed_probes / (ms_hdr * 1e-3) / sectors_per_s, 4),
                    "with_positions": {"probes": n2, "positions_fetched": fetched, "kernel_ms": round(ms_pos, 4),
                                       "achieved_GBps": round((8.0 * n2 + pos_bytes * fetched) / (ms_pos * 1e-3) / 1e9, 2)},
                    "note": "algorithmic bytes: 8 B header per probe (+ %d B per fetched position); every probe touches one random 64 B sector, so the "
                            "sector ceiling (%.1f G sectors/s = %.0f GB/s of sector traffic) is the bound that applies" % (pos_bytes, sectors_per_s / 1e9, sectors_per_s * 64 / 1e9)}

        # HBM bytes per launch from the PMC passes committed with this round's profiles (same command, scripts/gpu_profile_round.sh);
        # only quoted for the workload they were collected on
        traffic = None
        try:
            if (args.reads, args.ref_len, args.read_len) == (1_000_000, 5_000_000, 150):
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_summary.json")))
                traffic = int(pm["hbm_bytes_per_launch"]["mean_over_the_two_launches_of_a_step"])
        except Exception:
            traffic = None

        value = world * nq * args.steps / elapsed / 1e6
        line = {
            "metric": "M reads/s aligned (150 bp)", "value": round(value, 4), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %d synthetic %d bp single-end reads per GPU vs %d bp synthetic E. coli-sized reference (index replicated, reads sharded, no collective)" % (nq, args.read_len, args.ref_len),
                       "reads_per_gpu": nq, "read_len": args.read_len, "reference_len": args.ref_len, "parallelism": "reads sharded x%d" % world,
                       "aligned_reads": aligned, "index_build_s": round(index_build_s, 3), "index_bytes": info["index_bytes"],
                       "index_build": {"hashed_on": "gpu" if info["built_on_device"] else "host", "hash_s": round(info["hash_seconds"], 3),
                                       "duplication_map_s": round(info["duplication_seconds"], 3)}},
            "roofline": {"bound": "hbm", "kernel": "xm_align_kernel", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 6), "traffic": traffic,
                         "algorithmic_bytes_per_step": int(alg_bytes), "bytes_per_read": round(alg_bytes / nq, 1),
                         "kernel_ms_per_step": round(kernel_ms / args.steps, 3), "launches_per_step": launches / args.steps,
                         "result_d2h_ms_per_step": round(d2h_ms / args.steps, 3),
                         "traffic_rate": None if traffic is None else round(traffic / (avg_launch_ms * 1e-3) / 1e9, 1),
                         "note": "achieved/peak/frac: algorithmic bytes against the 8 TB/s stream peak.  traffic (PMC) is per-lane scratch in HBM, touched in "
                                 "scattered 32-64 B pieces: traffic_rate (GB/s, traffic / average launch duration) is to be read against this GPU's measured "
                                 "random-64-B-sector ceiling (seed_probe.random_64B_gather_ceiling_sectors_per_s x 64 B), not against the stream peak"},
            "cpu_baseline": cpu,
            "seed_probe": seed,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    db.close()


if __name__ == "__main__":
    main()
