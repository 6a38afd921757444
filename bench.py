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
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="reads of the same workload timed on the host cores (0 = skip); default: the whole batch (about a second on the GPU box's 256 cores)")
    ap.add_argument("--config", default="1", choices=["1", "2", "4shape"], help="1: BASELINE.json configs[1] (single-end 150 bp, the headline); 2: configs[2] shape (2x150 bp pairs, --spacing 100 50); "
                    "4shape: the 1,000 bp queries --split-queries-past-size 1000 makes of configs[4]'s reads, against the same 5 Mb reference")
    ap.add_argument("--contexts", type=int, default=3, help="contexts of the extra pipelined measurement at N=1 (several contexts of the GPU aligning their batches at the same time; 1 = skip)")
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

    t0 = time.time()
    db = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=args.read_len, device=local_rank)
    index_build_s = time.time() - t0
    db.upload_arrays(mc, mo, ml, codes, exp_in, dev_in)  # inputs resident in HBM before the timed region

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
    pass_us = np.zeros(4)
    for _ in range(args.steps):
        r = db.align_resident(params)
        kernel_ms += r.kernel_ms
        launches += r.kernel_launches
        d2h_ms += r.d2h_ms
        pass_us += np.asarray(r.counters[12:16], dtype=np.float64)
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

        # host buffers in (xm_align_batch: H2D copy + the same passes): the PCIe-inclusive rate, never the headline value
        t1 = time.perf_counter()
        for _ in range(2):
            rp = db.align_arrays(mc, mo, ml, codes, exp_in, dev_in, params)
        pcie_s = (time.perf_counter() - t1) / 2
        pcie = {"value": round(nq * reads_per_query / pcie_s / 1e6, 4), "unit": "Mreads/s", "ms_per_step": round(pcie_s * 1e3, 3), "h2d_ms": round(rp.h2d_ms, 3),
                "note": "xm_align_batch with host buffers in (pageable numpy arrays), one batch after the other, no overlap of copy and alignment"}

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
                         "kernel_ms_per_step": round(kernel_ms / args.steps, 3), "launches_per_step": launches / args.steps,
                         "result_d2h_ms_per_step": round(d2h_ms / args.steps, 3),
                         "kernel_ms_by_pass": {"light_pass": round(pass_us[0] / args.steps / 1e3, 3), "gapped_and_rerun_passes": round(pass_us[3] / args.steps / 1e3, 3)},
                         "traffic_rate": None if traffic is None else round(traffic / (avg_launch_ms * 1e-3) / 1e9, 1),
                         "note": "achieved/peak/frac: algorithmic bytes against the 8 TB/s stream peak.  traffic (PMC) is per-lane scratch in HBM, touched in "
                                 "scattered 32-64 B pieces: traffic_rate (GB/s, traffic / average launch duration) is to be read against this GPU's measured "
                                 "random-64-B-sector ceiling (seed_probe.random_64B_gather_ceiling_sectors_per_s x 64 B), not against the stream peak"},
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
    db.close()
    if rank == 0 and world == 1 and args.contexts > 1 and line is not None:
        # Several contexts on the GPU, each with its own resident copy of the batch and its own scratch, aligning at the same time (mapper_amd/multi.py,
        # `--contexts` of the command line): the idle wave slots of one context's gapped pass are filled by the others' passes.  Reported beside the
        # headline, which stays the single-context number (one launch at a time: the roofline's per-launch accounting stays clean).
        try:
            import threading
            os.environ["XM_SCRATCH_GIB"] = str(max(8, 240 // args.contexts))
            first = api.ReferenceDatabase([("ecoli_syn", ref)], mode="mapper", max_query_length=args.read_len, device=local_rank)
            ctx = [first] + [first.replicate(local_rank) for _ in range(args.contexts - 1)]
            for c_ in ctx:
                c_.upload_arrays(mc, mo, ml, codes, exp_in, dev_in)
                c_.align_resident(params)
            reps = 3
            outs = [None] * len(ctx)

            def work(i):
                for _ in range(reps):
                    outs[i] = ctx[i].align_resident(params)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            th = [threading.Thread(target=work, args=(i,)) for i in range(len(ctx))]
            [x.start() for x in th]
            [x.join() for x in th]
            torch.cuda.synchronize()
            per_step = (time.perf_counter() - t1) / (reps * len(ctx))
            same_p = all(np.array_equal(o_.ints, r.ints) and np.array_equal(o_.dbls.view(np.int64), r.dbls.view(np.int64)) for o_ in outs)
            line["pipelined_contexts"] = {"contexts": args.contexts, "scratch_gib_each": int(os.environ["XM_SCRATCH_GIB"]), "steps": reps * len(ctx),
                                          "value": round(nq * reads_per_query / per_step / 1e6, 4), "unit": "Mreads/s", "ms_per_step": round(per_step * 1e3, 3),
                                          "kernel_ms_of_one_step_in_each_context": [round(o_.kernel_ms, 1) for o_ in outs], "bit_identical_to_headline": bool(same_p),
                                          "note": "not the headline: %d contexts (index replicated on the same GPU, a copy of the batch and a third of the scratch each) align their "
                                                  "batches at the same time; a step is still one whole pass over one batch" % args.contexts}
            for c_ in ctx:
                c_.close()
            if not same_p:
                rc = 1
        except Exception as e:  # noqa: BLE001  (an extra measurement must not lose the headline line)
            line["pipelined_contexts"] = {"error": str(e)[:300]}
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
