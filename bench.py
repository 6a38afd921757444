#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: M reads/s aligned (150 bp) on MI355X.

A "step" is one pass of the hot path (xm_align_resident: every kernel pass + the copy of the result streams back to the
host and their ordering) over one batch of synthetic input that is already resident in HBM: BASELINE.json configs[1],
1,000,000 synthetic 150 bp single-end reads vs the 5 Mb synthetic E. coli-sized reference, per GPU.
With --gpus N each rank owns one GPU, holds a replica of the index and aligns its own 1 M reads (no collective on the data
path; scaling = weak).  Rank 0 prints ONE JSON line.

The steps of a rank are dealt to --contexts contexts of its GPU (default 3: the index replicated on the GPU with xm_index_replicate,
each context with its own resident copy of the batch, host thread, stream and share of the scratch) that align at the same time - how
the product aligns a stream of batches (`python -m mapper_amd --contexts 3`, mapper_amd/multi.py): the wave slots one context's gapped
pass leaves idle (its tail, the host gaps between its passes, its result copy) are filled by the others' passes, +12 % reads/s.  At
N=1 the line also carries `single_context`: the same kernel with one launch on the GPU at a time (--contexts 1 makes that the headline).

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
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step (configs[1]: 1,000,000)")
    ap.add_argument("--ref-len", type=int, default=5_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--seed-probes", type=int, default=16_000_000, help="bucket-header probes of the seed-lookup micro-benchmark (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to exercise the multi-rank path on a one-GPU box)")
    ap.add_argument("--force-device", type=int, default=-1, help="testing: every rank uses this GPU instead of its LOCAL_RANK")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="reads of the same workload timed on the host cores (0 = skip); default: the whole batch (about a second on the GPU box's 256 cores)")
    ap.add_argument("--config", default="1", choices=["1", "2", "4shape"], help="1: BASELINE.json configs[1] (single-end 150 bp, the headline); 2: configs[2] shape (2x150 bp pairs, --spacing 100 50); "
                    "4shape: the 1,000 bp queries --split-queries-past-size 1000 makes of configs[4]'s reads, against the same 5 Mb reference")
    ap.add_argument("--contexts", type=int, default=3, help="contexts per GPU: the steps are dealt to this many contexts of the GPU that align their resident batches at the same time (1: one launch at a time)")
    ap.add_argument("--single-context-steps", type=int, default=3, help="steps of the one-launch-at-a-time measurement beside the headline at N=1 (0 = skip)")
    ap.add_argument("--wave-steps", type=int, default=2, help="steps of the opt-in wave-per-read form (XM_WAVE=1) measured beside the headline (0 = skip)")
    args = ap.parse_args()

    rc = 0
    line = None
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

    from mapper_amd import api, synth, _capi
    build = _capi.check_stamp()  # refuses to measure a library that was not built from the sources in the tree

    ref = synth.synthetic_reference(args.ref_len, seed=0xEC011)
    if args.config == "2":
        m1, m2 = synth.synthetic_paired_end(ref, args.reads, read_len=args.read_len, seed=0x5EED0002 + 7919 * rank)[:2]
        nq, L = m1.shape
        codes = np.ascontiguousarray(np.concatenate([m1, m2], axis=1).reshape(-1))
        mc = np.full(nq, 2, np.int32)
        mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * 2 * L; mo[1::2] = mo[0::2] + L
        ml = np.full(2 * nq, L, np.int32)
        exp_in, dev_in = np.full(nq, 100.0), np.full(nq, 50.0)  # --spacing 100 50
        reads_per_query = 2
    else:
        if args.config == "4shape":
            args.read_len = 1000
            reads = synth.synthetic_single_end(ref, args.reads, read_len=1000, seed=0x5EED0004 + 7919 * rank)[0]
        else:
            reads = synth.synthetic_single_end(ref, args.reads, read_len=args.read_len, seed=0x5EED0001 + 7919 * rank)[0]
        nq, L = reads.shape
        codes = np.ascontiguousarray(reads.reshape(-1))
        mc = np.ones(nq, np.int32)
        mo = np.zeros(2 * nq, np.int64); mo[0::2] = np.arange(nq, dtype=np.int64) * L
        ml = np.zeros(2 * nq, np.int32); ml[0::2] = L
        exp_in, dev_in = np.zeros(nq), np.ones(nq)
        reads_per_query = 1
    params = api.AlignmentParameters()  # Mapper.main defaults

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(contexts, steps):
        """`steps` passes of the hot path, each over one resident batch, dealt to the contexts (one host thread and stream each) as they become free.
        -> (seconds, summed kernel ms, launches, d2h ms, microseconds by pass, last result of context 0)"""
        import threading
        lock = threading.Lock()
        state = {"next": 0, "kernel_ms": 0.0, "launches": 0, "d2h_ms": 0.0, "pass_us": np.zeros(4), "last": None, "error": None}

        def work(i):
            try:
                while True:
                    with lock:
                        if state["next"] >= steps:
                            return
                        state["next"] += 1
                    rr = contexts[i].align_resident(params)
                    with lock:
                        state["kernel_ms"] += rr.kernel_ms
                        state["launches"] += rr.kernel_launches
                        state["d2h_ms"] += rr.d2h_ms
                        state["pass_us"] += np.asarray(rr.counters[12:16], dtype=np.float64)
                        if i == 0 or state["last"] is None:
                            state["last"] = rr
            except BaseException as e:  # noqa: BLE001
                state["error"] = e
        t_start = time.perf_counter()
        if len(contexts) == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(i,)) for i in range(len(contexts))]
            [x.start() for x in th]
            [x.join() for x in th]
        if state["error"] is not None:
            raise state["error"]
        return time.perf_counter() - t_start, state["kernel_ms"], state["launches"], state["d2h_ms"], state["pass_us"], state["last"]

    # One context at a time first (at N=1, when the headline uses several): the kernel's own numbers, one launch on the GPU at a time
    single = None
    n_ctx = max(1, args.contexts)
    if n_ctx > 1 and world == 1 and args.single_context_steps > 0:
        one = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=args.read_len, device=local_rank)
        one.upload_arrays(mc, mo, ml, codes, exp_in, dev_in)
        run_steps([one], max(1, args.warmup))
        torch.cuda.synchronize()
        k1 = args.single_context_steps
        sec, kms, nl, _, pus, _ = run_steps([one], k1)
        single = {"value": round(nq * reads_per_query * k1 / sec / 1e6, 4), "unit": "Mreads/s", "steps": k1, "ms_per_step": round(sec / k1 * 1e3, 3),
                  "kernel_ms_per_step": round(kms / k1, 3), "launches_per_step": nl / k1,
                  "kernel_ms_by_pass": {"light_pass": round(pus[0] / k1 / 1e3, 3), "gapped_and_rerun_passes": round(pus[3] / k1 / 1e3, 3)},
                  "note": "one context with the whole scratch budget, one launch on the GPU at a time (the round-1 way of running the same kernel)"}
        one.close()
    t0 = time.time()
    db = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=args.read_len, device=local_rank)
    index_build_s = time.time() - t0
    ctx = [db] + [db.new_context() for _ in range(n_ctx - 1)]  # contexts share the index (host tables and tables in HBM), xm_context_new
    if n_ctx > 1:
        n_use, scratch_each = api.divide_scratch(ctx, local_rank)  # what is free now, in equal parts
        ctx = ctx[:n_use]
        n_ctx = len(ctx)
    for c_ in ctx:
        c_.upload_arrays(mc, mo, ml, codes, exp_in, dev_in)  # inputs resident in HBM before the timed region: every context has its batch

    if args.warmup > 0:
        run_steps(ctx, args.warmup * n_ctx)  # (every context allocates its scratch and runs every pass once)
    barrier()
    elapsed, kernel_ms, launches, d2h_ms, pass_us, r = run_steps(ctx, args.steps)
    barrier()
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

        # host buffers in (xm_align_batch: H2D copy + the same passes), every context at the same time: the PCIe-inclusive rate, never the headline value
        import threading
        rps = [None] * len(ctx)

        def with_host_buffers(i):
            for _ in range(2):
                rps[i] = ctx[i].align_arrays(mc, mo, ml, codes, exp_in, dev_in, params)
        t1 = time.perf_counter()
        th = [threading.Thread(target=with_host_buffers, args=(i,)) for i in range(len(ctx))]
        [x.start() for x in th]
        [x.join() for x in th]
        pcie_s = (time.perf_counter() - t1) / (2 * len(ctx))
        pcie = {"value": round(nq * reads_per_query / pcie_s / 1e6, 4), "unit": "Mreads/s", "ms_per_step": round(pcie_s * 1e3, 3), "h2d_ms": round(rps[0].h2d_ms, 3),
                "note": "xm_align_batch with host buffers in (pageable numpy arrays) in every context, no overlap of a context's copy with its own alignment"}

        # the opt-in wave-per-read form (XM_WAVE=1: one wavefront per read, state in LDS, xm_wave_kernel.hip) on the same resident batch
        wave = None
        if args.wave_steps > 0 and args.read_len <= 256:
            os.environ["XM_WAVE"] = "1"
            try:
                rw = db.align_resident(params)
                t1 = time.perf_counter()
                us = np.zeros(4)
                for _ in range(args.wave_steps):
                    rw = db.align_resident(params)
                    us += np.asarray(rw.counters[12:16], dtype=np.float64)
                w_s = (time.perf_counter() - t1) / args.wave_steps
                same_w = bool(np.array_equal(rw.ints, r.ints) and np.array_equal(rw.dbls.view(np.int64), r.dbls.view(np.int64)) and np.array_equal(rw.int_off, r.int_off))
                wave = {"value": round(nq * reads_per_query / w_s / 1e6, 4), "unit": "Mreads/s", "ms_per_step": round(w_s * 1e3, 3), "steps": args.wave_steps,
                        "kernel_ms_by_pass": {"light_tier": round(us[0] / args.wave_steps / 1e3, 3), "chain_tiers_with_inline_searches": round(us[1] / args.wave_steps / 1e3, 3),
                                              "search_kernel": round(us[2] / args.wave_steps / 1e3, 3), "lane_per_read_passes_for_the_rest": round(us[3] / args.wave_steps / 1e3, 3)},
                        "launches_per_step": rw.kernel_launches, "bit_identical_to_default_path": same_w,
                        "note": "opt-in (XM_WAVE=1); not the headline: slower than the lane-per-read passes this round (profiles/r02/NOTES.md)"}
            finally:
                os.environ["XM_WAVE"] = "0"

        # the committed golden digest of this exact batch (tests/golden/synthetic_golden.json, made once by the oracle): checked even with --cpu-sample 0
        golden = None
        try:
            key = {"1": "configs1_single_end_1000000", "2": "configs2_paired_end_1000000"}.get(args.config)
            if key and rank == 0 and (args.reads, args.ref_len, args.read_len) == (1_000_000, 5_000_000, 150):
                import hashlib
                want_g = json.load(open(os.path.join(ROOT, "tests", "golden", "synthetic_golden.json")))["full_digests"][key]
                hsh = hashlib.sha256()
                for a in (r.int_off, r.dbl_off, r.ints, np.asarray(r.dbls).view(np.int64)):
                    hsh.update(np.ascontiguousarray(a).tobytes())
                golden = {"sha256": hsh.hexdigest(), "matches_committed": hsh.hexdigest() == want_g["sha256"]}
        except (OSError, KeyError):
            golden = None

        cpu = None
        same = None
        counters = {"device": [int(x) for x in r.counters[:11]]}
        if args.cpu_sample > 0:
            import oracle_lib
            n = min(args.cpu_sample, nq)
            o = oracle_lib.OracleReference([("ecoli_syn", ref)], mode="mapper")
            cores = os.cpu_count() or 1
            warm = oracle_lib.QueryBatch.from_arrays(mc[:64], mo[:128], ml[:128], codes, exp_in[:64], dev_in[:64])
            o.align(warm, oracle_lib.make_params(), threads=1)   # builds the index (not timed, like the reference's prepare())
            o.require_size(args.read_len)
            b = oracle_lib.QueryBatch.from_arrays(mc[:n], mo[:2 * n], ml[:2 * n], codes, exp_in[:n], dev_in[:n])
            t1 = time.perf_counter()
            w = o.align(b, oracle_lib.make_params(), threads=cores)
            cpu_s = time.perf_counter() - t1
            same = bool(np.array_equal(w.ints, r.ints[:r.int_off[n]]) and np.array_equal(w.dbls.view(np.int64), r.dbls[:r.dbl_off[n]].view(np.int64)))
            cpu = {"value": round(n * reads_per_query / cpu_s / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port", "seconds": round(cpu_s, 3),
                   "sample": "%s %d queries of the same batch, oracle (C++ port of the Java path) with one worker thread per host core taking jobs of >= 50,000 bases "
                             "(Mapper.java:926), index build excluded" % ("all" if n == nq else "first", n)}
            if n == nq:  # the counts the algorithmic bytes are computed from, device against oracle on the same batch
                wc = [int(x) for x in w.counters[:9]]
                counters["oracle"] = wc
                counters["equal"] = counters["device"][:8] == [wc[0], wc[1] + wc[2], wc[2], wc[3], wc[5], wc[6], wc[7], wc[8]]
            java = java_reference(ref, codes, nq, args)
            if java is not None:
                cpu["java_reference"] = java

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
        # only quoted for the workload they were collected on, with their source
        traffic = None
        traffic_source = None
        try:
            if args.config == "1" and (args.reads, args.ref_len, args.read_len) == (1_000_000, 5_000_000, 150):
                pm = json.load(open(os.path.join(ROOT, "profiles", "r02", "pmc_summary.json")))
                traffic = int(pm["hbm_bytes_per_launch"]["mean_over_the_two_launches_of_a_step"])
                traffic_source = "profiles/r02/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, build %s)" % pm.get("build", "?")
        except Exception:
            traffic = None

        value = world * nq * reads_per_query * args.steps / elapsed / 1e6
        line = {
            "metric": "M reads/s aligned (150 bp)", "value": round(value, 4), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": {"1": "configs[1]: %d synthetic %d bp single-end reads per GPU vs %d bp synthetic E. coli-sized reference (index replicated, reads sharded, no collective)",
                                    "2": "configs[2] shape: %d synthetic 2 x %d bp pairs (--spacing 100 50) per GPU vs %d bp synthetic E. coli-sized reference",
                                    "4shape": "configs[4] shape: %d synthetic %d bp queries (what --split-queries-past-size 1000 makes of 10 kb reads) per GPU vs %d bp synthetic reference"}[args.config] % (nq, args.read_len, args.ref_len),
                       "reads_per_gpu": nq * reads_per_query, "read_len": args.read_len, "reference_len": args.ref_len, "parallelism": "reads sharded x%d" % world,
                       "aligned_reads": aligned, "index_build_s": round(index_build_s, 3), "index_bytes": info["index_bytes"],
                       "index_build": {"hashed_on": "gpu" if info["built_on_device"] else "host", "hash_s": round(info["hash_seconds"], 3),
                                       "duplication_map_s": round(info["duplication_seconds"], 3)}},
            "roofline": {"bound": "hbm", "kernel": "xm_align_kernel", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 6), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_step": int(alg_bytes), "bytes_per_read": round(alg_bytes / nq, 1),
                         "concurrent_launches": n_ctx, "frac_of_all_concurrent_launches": round(alg_bytes * args.steps / elapsed / 1e9 / 8000.0, 6),
                         "kernel_ms_per_step": round(kernel_ms / args.steps, 3), "launches_per_step": launches / args.steps,
                         "result_d2h_ms_per_step": round(d2h_ms / args.steps, 3),
                         "kernel_ms_by_pass": {"light_pass": round(pass_us[0] / args.steps / 1e3, 3), "gapped_and_rerun_passes": round(pass_us[3] / args.steps / 1e3, 3)},
                         "traffic_rate": None if traffic is None else round(traffic / (avg_launch_ms * 1e-3) / 1e9, 1),
                         "note": "achieved/peak/frac: algorithmic bytes of one launch over that launch's duration, against the 8 TB/s stream peak; with several contexts "
                                 "the launches of the contexts share the GPU, so a launch lasts longer than it would alone (single_context has the kernel's numbers with one "
                                 "launch at a time) and frac_of_all_concurrent_launches is the algorithmic rate of the GPU as a whole.  traffic (PMC) is per-lane scratch in HBM, touched in "
                                 "scattered 32-64 B pieces: traffic_rate (GB/s, traffic / average launch duration) is to be read against this GPU's measured "
                                 "random-64-B-sector ceiling (seed_probe.random_64B_gather_ceiling_sectors_per_s x 64 B), not against the stream peak"},
            "contexts": {"per_gpu": n_ctx, "scratch_gib_each": round(scratch_each / 2**30, 1) if n_ctx > 1 else None,
                         "note": "a step is one whole pass of the hot path over one resident batch; the steps are dealt to %d contexts of the GPU (index replicated with "
                                 "xm_index_replicate, a resident copy of the batch, a host thread, a stream and a share of the scratch each) that align at the same time: the wave "
                                 "slots one context's gapped pass leaves idle are filled by the others' passes (profiles/r02/NOTES.md 12, 14)" % n_ctx if n_ctx > 1 else "one context"},
            "single_context": single,
            "cpu_baseline": cpu,
            "build": build,
            "bit_identical": same,
            "golden": golden,
            "counters": counters,
            "pcie_inclusive": pcie,
            "wave_form": wave,
            "seed_probe": seed,
        }
        if same is False or (golden is not None and not golden["matches_committed"]) or (wave is not None and not wave["bit_identical_to_default_path"]) or counters.get("equal") is False:
            rc = 1  # a parity failure is not a measurement
    for c_ in ctx:
        c_.close()
    if rank == 0 and line is not None:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def java_reference(ref, codes, nq, args):
    """BASELINE.md section 3.4 / SURVEY.md section 8(d): when a JDK and the reference's jar are on the box (env XMAPPER_JAR), the Java path itself is timed
    on the same inputs (`java -jar x-mapper.jar --reference R --queries Q --num-threads N --no-output`); otherwise None ("Java reference: unavailable")."""
    import shutil
    import subprocess
    import tempfile
    jar = os.environ.get("XMAPPER_JAR")
    if not jar or not os.path.exists(jar) or shutil.which("java") is None or args.config != "1":
        return None
    try:
        from mapper_amd import api as _api
        d = tempfile.mkdtemp(prefix="xm_java_")
        with open(os.path.join(d, "ref.fasta"), "w") as f:
            f.write(">ecoli_syn\n" + _api.decode(ref) + "\n")
        n = min(nq, 200_000)
        with open(os.path.join(d, "reads.fastq"), "w") as f:
            for q in range(n):
                s_ = _api.decode(codes[q * args.read_len: (q + 1) * args.read_len])
                f.write("@r%d\n%s\n+\n%s\n" % (q, s_, "I" * len(s_)))
        cores = os.cpu_count() or 1
        t1 = time.perf_counter()
        subprocess.run(["java", "-jar", jar, "--reference", os.path.join(d, "ref.fasta"), "--queries", os.path.join(d, "reads.fastq"), "--num-threads", str(cores), "--no-output"],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        sec = time.perf_counter() - t1
        shutil.rmtree(d, ignore_errors=True)
        return {"value": round(n / sec / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "reference", "sample": "%d reads, wall time of the whole java run (index build included)" % n}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:200]}


if __name__ == "__main__":
    sys.exit(main())
