#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: M reads/s aligned (150 bp) on MI355X.

A "step" is one pass of the hot path (xm_align_resident: every kernel pass + the copy of the result streams back to the
host and their ordering) over one batch of synthetic input that is already resident in HBM: BASELINE.json configs[1],
1,000,000 synthetic 150 bp single-end reads vs the 5 Mb synthetic E. coli-sized reference, per GPU.
With --gpus N each rank owns one GPU, holds a replica of the index and aligns its own 1 M reads (no collective on the data
path; scaling = weak).  Rank 0 prints ONE JSON line.

The steps of a rank are dealt to --contexts contexts of its GPU (default 3 for this config; xm_context_new: the contexts share the index - host tables
and tables in HBM - and each has its own resident copy of the batch, host thread, stream and share of the scratch) that align at the
same time - how the product aligns a stream of batches (`python -m mapper_amd --contexts N`, mapper_amd/multi.py): the wave slots one
context's gapped pass leaves idle (its tail, the host gaps between its passes, its result copy) are filled by the other's passes,
+15-18 % reads/s (configs[1]: three and four contexts measure the same or less; a workload whose passes end in a long tail - --config 1rep -
gains up to five contexts, given hardware queues for them: GPU_MAX_HW_QUEUES, profiles/r04/NOTES.md 13).  At N=1 the line also carries
`single_context`: the same kernel with one launch on the GPU at a time (--contexts 1 makes that the headline).

The line carries `roofline` (algorithmic bytes of SURVEY.md §8(d) per second of align-kernel time, against the 8 TB/s
HBM peak) and `cpu_baseline` (the CPU oracle, a port of the Java path, timed on this box's host cores on a bounded
sample).  The oracle is only the checker/baseline here; nothing under oracle/ is on the measured GPU path.
"""
import argparse
import json
import os
import sys
import time

# (before anything initialises the HIP runtime: contexts beyond two need hardware queues of their own, mapper_amd/_capi.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
    ap.add_argument("--seed-probes", type=int, default=64_000_000, help="bucket-header probes of the seed-lookup micro-benchmark (0 = skip); the default is the size of the gather-ceiling measurement it is held against (2^26 accesses); a quarter of them are run again with up to four positions each on the workload's index, all of them with up to seven on the HBM-resident index")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to exercise the multi-rank path on a one-GPU box)")
    ap.add_argument("--force-device", type=int, default=-1, help="testing: every rank uses this GPU instead of its LOCAL_RANK")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="reads of the same workload timed on the host cores (0 = skip); default: the whole batch (about a second on the GPU box's 256 cores)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="the CPU baseline's timed run is repeated until the oracle has worked this long (best and median run reported)")
    ap.add_argument("--end-to-end-reads", type=int, default=2_000_000, help="reads of the end_to_end leg (--config 1 / 2, N = 1): `python -m mapper_amd` from a FASTQ file on disk to a SAM file on disk (0 = skip)")
    ap.add_argument("--config", default="1", choices=["1", "1rep", "2", "4shape", "3shape", "3rep", "4", "4mild"],
                    help="1: BASELINE.json configs[1] (single-end 150 bp, the headline); 2: configs[2] shape (2x150 bp pairs, --spacing 100 50); "
                    "4shape: the 1,000 bp queries --split-queries-past-size 1000 makes of configs[4]'s reads, against the same 5 Mb reference; "
                    "3shape / 4 / 4mild: configs[3] / configs[4] on ONE GPU against the 3.1 Gb GRCh38-shaped reference of SURVEY.md section 8(d) "
                    "(pairs; 10 kb reads split at 1000 with the error rates as stated; the same with 2 %% substitutions + 0.2 %% indel events); "
                    "3rep: configs[3]'s pairs against the same shape with the repeat structure of a genome (synth.grch38_repeat_rich_reference: 120 k copies of 300 bp families at 85-95 %% "
                    "identity, 2 000 segmental duplications, 50 000 tandem repeats)")
    ap.add_argument("--stream-batches", type=int, default=4, help="batches of the PCIe-inclusive streamed measurement (api.align_stream: the upload of batch k+1 overlaps the alignment of batch k; 0 = skip)")
    ap.add_argument("--seed-index-mb", type=int, default=500, help="size (M bases) of the second, HBM-resident index the seed-probe leg builds so that its probes miss every cache (0 = probes on the workload's own index only)")
    ap.add_argument("--big-scale", type=float, default=1.0, help="testing: the GRCh38-shaped reference of --config 3shape / 4 / 4mild at this fraction of its size (1.0 = the 3.1 Gb of SURVEY.md section 8(d))")
    ap.add_argument("--share-dir", default=None, help="--gpus N with --config 3shape / 4 / 4mild: where rank 0 leaves the synthetic reference (memory-mapped by the other ranks) and the index "
                    "it built (xm_index_save; the other ranks xm_index_load it): one generation and one hashing per node instead of N (default: a directory under the system's temporary directory named after MASTER_PORT)")
    ap.add_argument("--contexts", type=int, default=None, help="contexts per GPU (default: --config 1 three, --config 1rep four, the others two): the steps are dealt to this many contexts of the GPU (xm_context_new: they share the index) that align their resident batches at the same time (1: one launch at a time)")
    ap.add_argument("--scratch-gib", type=float, default=0.0, help="upper limit of a context's scratch in GiB (0: what is free, divided between the contexts, up to 200 GiB): the headline against the scratch budget (profiles/r06/scratch_budget.log)")
    ap.add_argument("--single-context-steps", type=int, default=3, help="steps of the one-launch-at-a-time measurement beside the headline at N=1 (0 = skip)")
    ap.add_argument("--engine", default="gpu", choices=["gpu", "hostsim"], help="testing only: hostsim = a DRY RUN of this script's flow (ranks, barriers, max-over-ranks timing, the rank-0 line) "
                    "with the kernel sources compiled for the host (tests/hostsim, tests/sim_engine.py) instead of the GPU library; the line it prints is marked as such and is not a measurement")
    ap.add_argument("--wave-steps", type=int, default=2, help="steps of the opt-in wave-per-read form (XM_WAVE=1) measured beside the headline (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # launched bare: this process starts the N ranks itself (fresh children, one per GPU) before it has touched a GPU, and only waits for them
        return spawn_ranks(args.gpus)

    rc = 0
    line = None
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if args.force_device >= 0:
        local_rank = args.force_device
    dry = args.engine == "hostsim"
    if dry and args.backend == "nccl":
        args.backend = "gloo"

    class _NoGpu:  # (dry run: nothing to select or to wait for)
        @staticmethod
        def set_device(i): pass
        @staticmethod
        def synchronize(): pass
    gpu = _NoGpu if dry else torch.cuda
    if world > 1:
        import torch.distributed as dist
        gpu.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    else:
        dist = None
        gpu.set_device(local_rank)
    assert args.gpus == world, "--gpus must equal the number of launched ranks"

    from mapper_amd import api, synth, _capi
    build = _capi.check_stamp()  # refuses to measure a library that was not built from the sources in the tree
    if dry:  # TEST HARNESS ONLY: the api's database replaced by the host simulation of the kernel sources (the product has no CPU path)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import sim_engine
        api = sim_engine.Api(api)

    big = args.config in ("3shape", "3rep", "4", "4mild")   # the 3.1 Gb GRCh38-shaped reference of SURVEY.md section 8(d), on one GPU
    build_kw = {}
    share = None
    if big and world > 1:
        # N ranks on one node: ONE generation of the 3 GB reference and ONE hashing of it (rank 0); the other ranks map the reference and load the index
        # (HashBlock_Database.java:477-487 / DirCache.java:19-60: the reference's workers share one database; Mapper.java:1026-1040)
        import tempfile
        share = args.share_dir or os.path.join(tempfile.gettempdir(), "xm_bench_share_%s" % os.environ.get("MASTER_PORT", "0"))
        os.makedirs(share, exist_ok=True)
    if big:
        if share is None or rank == 0:
            rep_stats = {}
            if args.config == "3rep":
                contigs, whole, gstarts, gruns = synth.grch38_repeat_rich_reference(scale=args.big_scale, stats=rep_stats)
            else:
                contigs, whole, gstarts, gruns = synth.grch38_shaped_reference(scale=args.big_scale)
            if share is not None:
                np.save(os.path.join(share, "whole.tmp.npy"), whole)
                os.replace(os.path.join(share, "whole.tmp.npy"), os.path.join(share, "whole.npy"))
                np.savez(os.path.join(share, "layout.npz"), starts=gstarts, n_runs=np.array([len(r_) for r_ in gruns]), runs=np.concatenate(gruns) if len(gruns) else np.zeros(0, np.int64))
        if share is not None:
            dist.barrier()
            if rank != 0:
                whole = np.load(os.path.join(share, "whole.npy"), mmap_mode="r")
                lay = np.load(os.path.join(share, "layout.npz"))
                gstarts = lay["starts"]
                cuts = np.concatenate([[0], np.cumsum(lay["n_runs"])])
                gruns = [lay["runs"][cuts[c_]:cuts[c_ + 1]] for c_ in range(len(lay["n_runs"]))]
                contigs = [(synth.GRCH38_NAMES[c_], whole[gstarts[c_]:gstarts[c_ + 1]]) for c_ in range(len(gstarts) - 1)]
        ref = whole
        args.ref_len = int(len(whole))
    elif args.config == "1rep":
        # configs[1]'s reads against a reference with the structure i.i.d. ACGT lacks (segmental duplications at 90-99.5 % identity, tandem repeats, a 28-mer
        # whose buckets overflow): the branch a real genome sends reads into - no early accept in a duplicated window, every candidate enumerated
        rep_stats = {}
        ref = synth.repeat_rich_reference(args.ref_len, stats=rep_stats)
        contigs = [("ecoli_rep", ref)]
    else:
        ref = synth.synthetic_reference(args.ref_len, seed=0xEC011)
        contigs = [("ecoli_syn", ref)]

    def make_queries(cfg, n_reads, source, seed_shift, where=None):
        """-> (mc, mo, ml, codes, exp_in, dev_in, reads_per_query, read_len) of one batch of config `cfg` sampled from `source` (one array)."""
        if cfg in ("2", "3shape", "3rep"):
            seed = (0x5EED0002 if cfg == "2" else 0x5EED0003) + seed_shift  # (3rep: configs[3]'s pairs)
            m1, m2 = synth.synthetic_paired_end(source, n_reads, read_len=150, seed=seed, at=where)[:2]
            n, L = m1.shape
            codes_ = np.ascontiguousarray(np.concatenate([m1, m2], axis=1).reshape(-1))
            mo_ = np.zeros(2 * n, np.int64); mo_[0::2] = np.arange(n, dtype=np.int64) * 2 * L; mo_[1::2] = mo_[0::2] + L
            return np.full(n, 2, np.int32), mo_, np.full(2 * n, L, np.int32), codes_, np.full(n, 100.0), np.full(n, 50.0), 2, L  # --spacing 100 50
        if cfg in ("4", "4mild"):
            # 10 kb reads cut by --split-queries-past-size 1000 (cli.split_sections = SequenceSplitter.java:17,35-38): a query = a section
            from mapper_amd import cli
            sub, ind = (0.05, 0.05) if cfg == "4" else (0.02, 0.002)
            strand = (synth.splitmix64((0x5EED0004 + seed_shift) ^ 0x57A, n_reads) >> np.uint64(63)).astype(np.uint8)
            reads_ = synth.synthetic_long_reads(source, where, 10_000, seed=0x5EED0004 + seed_shift, sub_rate=sub, indel_rate=ind, strand=strand)
            sections = cli.split_sections(10_000, 1000)
            k = len(sections)
            n = n_reads * k
            mo_ = np.zeros(2 * n, np.int64)
            mo_[0::2] = (np.arange(n_reads, dtype=np.int64)[:, None] * 10_000 + np.array([a_ for a_, _ in sections], dtype=np.int64)[None, :]).reshape(-1)
            ml_ = np.zeros(2 * n, np.int32); ml_[0::2] = np.tile(np.array([b_ - a_ for a_, b_ in sections], dtype=np.int32), n_reads)
            return np.ones(n, np.int32), mo_, ml_, np.ascontiguousarray(reads_.reshape(-1)), np.zeros(n), np.ones(n), 1, 1000
        L = 1000 if cfg == "4shape" else args.read_len
        reads_ = synth.synthetic_single_end(source, n_reads, read_len=L, seed=(0x5EED0004 if cfg == "4shape" else 0x5EED0001) + seed_shift, at=where)[0]
        n = len(reads_)
        mo_ = np.zeros(2 * n, np.int64); mo_[0::2] = np.arange(n, dtype=np.int64) * L
        ml_ = np.zeros(2 * n, np.int32); ml_[0::2] = L
        return np.ones(n, np.int32), mo_, ml_, np.ascontiguousarray(reads_.reshape(-1)), np.zeros(n), np.ones(n), 1, L

    def sample_starts(cfg, n_reads, starts_, runs_, seed_shift):
        span = {"3shape": 2 * 150 + 400 + 3 + 153, "3rep": 2 * 150 + 400 + 3 + 153, "4": 10_000 + 2_500 + 8, "4mild": 10_000 + 2_500 + 8}[cfg]
        return synth.genome_wide_starts(starts_, runs_, n_reads, span, seed=(0x5EED0003 if cfg in ("3shape", "3rep") else 0x5EED0004) ^ 0xF00D ^ seed_shift)[0]

    if big:
        n_src = args.reads if args.config in ("3shape", "3rep") else max(1, args.reads // 10)   # --reads counts queries (sections) for the long reads
        mc, mo, ml, codes, exp_in, dev_in, reads_per_query, args.read_len = make_queries(args.config, n_src, whole, 7919 * rank, sample_starts(args.config, n_src, gstarts, gruns, 7919 * rank))
    else:
        mc, mo, ml, codes, exp_in, dev_in, reads_per_query, args.read_len = make_queries(args.config, args.reads, ref, 7919 * rank)
    nq = len(mc)
    params = api.AlignmentParameters()  # Mapper.main defaults

    def barrier():
        if dist is not None:
            dist.barrier()
        gpu.synchronize()

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

    t0 = time.time()
    if share is not None:
        index_file = os.path.join(share, "index.xmidx")
        if rank == 0:
            db = api.ReferenceDatabase(contigs, mode="mapper", max_query_length=args.read_len, device=local_rank)
            db.save(index_file + ".tmp")
            os.replace(index_file + ".tmp", index_file)
        dist.barrier()
        if rank != 0:
            db = api.ReferenceDatabase.load(index_file, device=local_rank, max_query_length=args.read_len)
    else:
        db = api.ReferenceDatabase(contigs, mode="mapper", max_query_length=args.read_len, device=local_rank)
    index_build_s = time.time() - t0
    # One context at a time first (at N=1, when the headline uses several): the kernel's own numbers, one launch on the GPU at a time
    single = None
    if args.contexts is None:
        # a pass of the repeat-rich workload ends with a long tail of few heavy reads (a quarter of its wave slots busy on average): more contexts fill it
        # configs[1]: three contexts, each sized for a third of the GPU's wave slots, overlap best (14.1-14.3 M reads/s against 13.1-13.2 with two); pairs and long
        # reads are within 2 % from two to four (profiles/r04/NOTES.md 13, 15)
        args.contexts = 4 if args.config == "1rep" else (3 if args.config == "1" else 2)
    n_ctx = max(1, args.contexts)
    scratch_each = None
    if n_ctx > 1 and world == 1 and args.single_context_steps > 0:
        db.upload_arrays(mc, mo, ml, codes, exp_in, dev_in)
        run_steps([db], max(1, args.warmup))
        gpu.synchronize()
        k1 = args.single_context_steps
        sec, kms, nl, _, pus, _ = run_steps([db], k1)
        single = {"value": round(nq * reads_per_query * k1 / sec / 1e6, 4), "unit": "Mreads/s", "steps": k1, "ms_per_step": round(sec / k1 * 1e3, 3),
                  "kernel_ms_per_step": round(kms / k1, 3), "launches_per_step": nl / k1,
                  "kernel_ms_by_pass": {"light_pass": round(pus[0] / k1 / 1e3, 3), "gapped_and_rerun_passes": round(pus[3] / k1 / 1e3, 3)},
                  "note": "one context with the whole scratch budget, one launch on the GPU at a time"}
    ctx = [db] + [db.new_context() for _ in range(n_ctx - 1)]  # contexts share the index (host tables and tables in HBM), xm_context_new
    if n_ctx > 1:
        n_use, scratch_each = api.divide_scratch(ctx, local_rank, **({"most": int(args.scratch_gib * 2**30)} if args.scratch_gib > 0 else {}))  # what is free now (the index is resident), in equal parts
        for c_ in ctx[n_use:]:
            c_.close()
        ctx = ctx[:n_use]
        n_ctx = len(ctx)
    if n_ctx == 1 and args.scratch_gib > 0:
        db.set_scratch(int(args.scratch_gib * 2**30))
        scratch_each = int(args.scratch_gib * 2**30)
    for c_ in ctx:
        c_.upload_arrays(mc, mo, ml, codes, exp_in, dev_in)  # inputs resident in HBM before the timed region: every context has its batch

    if args.warmup > 0:
        run_steps(ctx, args.warmup * n_ctx)  # (every context allocates its scratch and runs every pass once)
    barrier()
    elapsed, kernel_ms, launches, d2h_ms, pass_us, r = run_steps(ctx, args.steps)
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dry or args.backend != "nccl" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # which reads every rank aligned (a digest of its batch): the ranks' shards must differ (seed shift 7919 x rank)
        import hashlib
        digests = [None] * world
        dist.all_gather_object(digests, hashlib.sha256(np.ascontiguousarray(codes).tobytes()).hexdigest()[:16])
        pinned = [None] * world
        dist.all_gather_object(pinned, _capi.pinned_host_bytes()[1])   # page-locked host memory of every rank's result buffers (high-water mark): eight ranks share one host

    if world == 1:
        digests = None
        pinned = [_capi.pinned_host_bytes()[1]]
    if rank == 0:
        info = db.info()
        overfull = db.bucket_stats() if args.config in ("1rep", "3rep", "3shape") else None
        c = r.counters
        # algorithmic bytes of one step: B_in + 8*P + sum_hits(B_pos + 20) + window bytes + B_out
        pos_bytes = info["position_bytes"]
        alg_bytes = c[10] + 8 * c[1] + c[3] * (pos_bytes + 20) + c[9] + 4 * len(r.ints) + 8 * len(r.dbls)
        avg_launch_ms = kernel_ms / max(launches, 1)
        achieved = (alg_bytes * args.steps / max(launches, 1)) / (avg_launch_ms * 1e-3) / 1e9  # GB/s: bytes per launch / avg launch duration
        aligned = int(sum(1 for q in range(nq) if r.ints[r.int_off[q] + 1] > 0)) if nq <= 2_000_000 else -1

        extras = world == 1   # (N > 1: the line carries the headline only; the other ranks must not wait at the barrier for rank 0's side measurements)
        # host buffers in (xm_align_batch: H2D copy + the same passes), every context at the same time: the PCIe-inclusive rate, never the headline value
        import threading
        rps = [None] * len(ctx)

        def with_host_buffers(i):
            for _ in range(2):
                rps[i] = ctx[i].align_arrays(mc, mo, ml, codes, exp_in, dev_in, params)
        pcie = None
        if extras:
            t1 = time.perf_counter()
            th = [threading.Thread(target=with_host_buffers, args=(i,)) for i in range(len(ctx))]
            [x.start() for x in th]
            [x.join() for x in th]
            pcie_s = (time.perf_counter() - t1) / (2 * len(ctx))
            pcie = {"value": round(nq * reads_per_query / pcie_s / 1e6, 4), "unit": "Mreads/s", "ms_per_step": round(pcie_s * 1e3, 3), "h2d_ms": round(rps[0].h2d_ms, 3),
                    "note": "xm_align_batch with host buffers in (pageable numpy arrays) in every context, no overlap of a context's copy with its own alignment"}
        # the product's way of taking host buffers (api.align_stream, what `python -m mapper_amd` runs): batches staged on a second stream and host thread
        # while the previous one is aligned, one stream of batches per context; PCIe-inclusive, never the headline value
        streamed = None
        if extras and args.stream_batches > 0:
            counts = [0] * len(ctx)

            def stream(i):
                for _ in ctx[i].align_stream(iter([(mc, mo, ml, codes, exp_in, dev_in)] * args.stream_batches), params):
                    counts[i] += 1
            t1 = time.perf_counter()
            th = [threading.Thread(target=stream, args=(i,)) for i in range(len(ctx))]
            [x.start() for x in th]
            [x.join() for x in th]
            st_s = (time.perf_counter() - t1) / max(1, sum(counts))
            streamed = {"value": round(nq * reads_per_query / st_s / 1e6, 4), "unit": "Mreads/s", "ms_per_batch": round(st_s * 1e3, 3), "batches": sum(counts), "contexts": len(ctx),
                        "note": "api.align_stream: host buffers in, the H2D copy of batch k+1 overlapped with the alignment of batch k, %d batches per context" % args.stream_batches}
            for c_ in ctx:
                c_.upload_arrays(mc, mo, ml, codes, exp_in, dev_in)  # (the legs below align the resident batch again)

        # the opt-in wave-per-read form (XM_WAVE=1: one wavefront per read, state in LDS, xm_wave_kernel.hip) on the same resident batch
        wave = None
        if extras and args.wave_steps > 0 and args.read_len <= 256 and not big:
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
                        "note": "opt-in (XM_WAVE=1); not the headline: slower than the lane-per-read passes (profiles/r02/NOTES.md)"}
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

        def bound_filter_check(dev, orc, batch_):
            # The rejection filter in front of PathAligner (xm_bound.h; batches of long reads) skips searches it proves null, so the device puts fewer search nodes
            # than the reference.  The oracle, run once more (untimed) with its observer of the same bound on, says which searches the filter takes and rejects
            # and how many nodes the reference spent in them (and raises if a search the filter rejects returned an alignment): the counts must add up.
            if not dev.extra[3]:
                return None
            import oracle_lib as ol_
            with ol_.observe_bound():
                w_ = orc.align(batch_, ol_.make_params(), threads=os.cpu_count() or 1)
            oc = [int(x) for x in w_.counters]
            want_ = {"searches_examined": oc[13], "searches_rejected": oc[11], "pieces_examined": oc[14], "pieces_rejected": oc[15], "path_aligner_calls": oc[6] - oc[16], "nodes": oc[7] - oc[12] - oc[17]}
            got_ = {"searches_examined": int(dev.extra[0]), "searches_rejected": int(dev.extra[1]), "pieces_examined": int(dev.extra[4]), "pieces_rejected": int(dev.extra[5]),
                    "path_aligner_calls": int(dev.counters[5]), "nodes": int(dev.counters[6])}
            f_ = {"device": dict(got_, cells=int(dev.extra[2])), "oracle_observer": want_,
                  "reference": {"path_aligner_calls": oc[6], "nodes": oc[7], "searches_returning_null": oc[9], "nodes_in_null_searches": oc[10], "nodes_in_rejected_searches": oc[12],
                                "calls_in_rejected_pieces": oc[16], "nodes_in_rejected_pieces": oc[17]},
                  "equal": got_ == want_}
            return f_
        if extras and args.cpu_sample > 0 and big:
            # No oracle hashes 3.1 G bases in bounded time (15 Mb take it a minute), so the CPU path is timed on the same workload against the
            # SAME-SHAPED reference at 1/200 of its size (24 contigs, N-runs) with the minInterestingSize a 3 Gb reference gets (13,
            # HashBlock_Database.java:52) - the regime of the walk is the big reference's, the tables are smaller (cache-friendlier: flatters the CPU).
            import oracle_lib
            sc, sw, ss, sr = (synth.grch38_repeat_rich_reference if args.config == "3rep" else synth.grch38_shaped_reference)(scale=0.005)
            n_small = min(args.cpu_sample, 100_000 if args.config in ("3shape", "3rep") else 20_000)
            n_src = n_small if args.config in ("3shape", "3rep") else max(1, n_small // 10)
            qs = make_queries(args.config, n_src, sw, 0, sample_starts(args.config, n_src, ss, sr, 0))
            o = oracle_lib.OracleReference(sc, mode="mapper", min_interesting_size=13)
            cores = os.cpu_count() or 1
            o.align(oracle_lib.QueryBatch.from_arrays(qs[0][:16], qs[1][:32], qs[2][:32], qs[3], qs[4][:16], qs[5][:16]), oracle_lib.make_params(), threads=1)  # builds the index (not timed)
            o.require_size(args.read_len)
            b = oracle_lib.QueryBatch.from_arrays(*qs[:6])
            t1 = time.perf_counter()
            w = o.align(b, oracle_lib.make_params(), threads=cores)
            cpu_s = time.perf_counter() - t1
            small = api.ReferenceDatabase(sc, mode="mapper", max_query_length=args.read_len, device=local_rank, min_interesting_size=13)
            rs = small.align_arrays(*qs[:6], params)
            small.close()
            same = bool(np.array_equal(w.ints, rs.ints) and np.array_equal(w.dbls.view(np.int64), rs.dbls.view(np.int64)))
            counters["bound_filter"] = bound_filter_check(rs, o, b)
            if counters["bound_filter"] is not None:
                counters["equal"] = counters["bound_filter"]["equal"]
            cpu = {"value": round(len(qs[0]) * qs[6] / cpu_s / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port", "seconds": round(cpu_s, 3),
                   "sample": "%d queries of the same read model against the GRCh38-shaped reference at 1/200 scale (15 Mb, 24 contigs, N-runs, minInterestingSize 13), oracle (C++ port of the "
                             "Java path) on every host core, index build excluded; the GPU's streams on that sample are compared with the oracle's (bit_identical)" % len(qs[0])}
        elif extras and args.cpu_sample > 0:
            import oracle_lib
            n = min(args.cpu_sample, nq)
            o = oracle_lib.OracleReference([("ecoli_syn", ref)], mode="mapper")
            cores = os.cpu_count() or 1
            warm = oracle_lib.QueryBatch.from_arrays(mc[:64], mo[:128], ml[:128], codes, exp_in[:64], dev_in[:64])
            o.align(warm, oracle_lib.make_params(), threads=1)   # builds the index (not timed, like the reference's prepare())
            o.require_size(args.read_len)
            b = oracle_lib.QueryBatch.from_arrays(mc[:n], mo[:2 * n], ml[:2 * n], codes, exp_in[:n], dev_in[:n])
            # (round-5 verdict: one run of ~1.3 s on 256 threads is thread start-up and page faults to a visible degree and wandered 0.75-1.08 M reads/s from
            # round to round: the run is repeated until the oracle has worked for --cpu-seconds; `value` is the best run, the median is beside it)
            cpu_runs = []
            w = None
            while not cpu_runs or (sum(cpu_runs) < args.cpu_seconds and len(cpu_runs) < 64):
                t1 = time.perf_counter()
                w = o.align(b, oracle_lib.make_params(), threads=cores)
                cpu_runs.append(time.perf_counter() - t1)
            cpu_s = min(cpu_runs)
            same = bool(np.array_equal(w.ints, r.ints[:r.int_off[n]]) and np.array_equal(w.dbls.view(np.int64), r.dbls[:r.dbl_off[n]].view(np.int64)))
            cpu = {"value": round(n * reads_per_query / cpu_s / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port", "seconds": round(cpu_s, 3),
                   "runs": len(cpu_runs), "total_seconds": round(sum(cpu_runs), 3), "median_value": round(n * reads_per_query / float(np.median(cpu_runs)) / 1e6, 4),
                   "sample": "%s %d queries of the same batch, oracle (C++ port of the Java path) with one worker thread per host core taking jobs of >= 50,000 bases "
                             "(Mapper.java:926), index build excluded; the run repeated until %.0f s of oracle time: value = best run, median_value = median run"
                             % ("all" if n == nq else "first", n, args.cpu_seconds)}
            if n == nq:  # the counts the algorithmic bytes are computed from, device against oracle on the same batch
                wc = [int(x) for x in w.counters[:9]]
                counters["oracle"] = wc
                bf = bound_filter_check(r, o, b)
                counters["bound_filter"] = bf
                skipped = bf["reference"]["nodes_in_rejected_searches"] + bf["reference"]["nodes_in_rejected_pieces"] if bf else 0  # (nodes the filter made unnecessary)
                skipped_calls = bf["reference"]["calls_in_rejected_pieces"] if bf else 0
                counters["equal"] = counters["device"][:8] == [wc[0], wc[1] + wc[2], wc[2], wc[3], wc[5], wc[6] - skipped_calls, wc[7] - skipped, wc[8]] and (bf is None or bf["equal"])
            java = java_reference(ref, codes, nq, args)
            if java is not None:
                cpu["java_reference"] = java

        # seed-lookup micro-kernel: bulk PackedMap.get on the device, bytes = 8 per bucket-header probe + B_pos per
        # fetched position, next to the measured random-64 B-sector ceiling of this GPU
        seed = None
        if extras and args.seed_probes > 0:
            rng = np.random.default_rng(12345)
            lo, hi = info["min_interesting_size"], info["max_hashed_length"]
            used = rng.integers(lo, hi + 1, size=args.seed_probes, dtype=np.int32)
            keys = rng.integers(-2**31, 2**31 - 1, size=args.seed_probes, dtype=np.int64).astype(np.int32)
            db.seed_probe(used[:4096], keys[:4096], 0)
            counts, _, ms_hdr = db.seed_probe(used, keys, 0)
            sectors_per_s, gather_ms = api.measure_random_gather(4 << 30, 1 << 26, local_rank)
            n2 = args.seed_probes // 4
            c2, _, ms_pos = db.seed_probe(used[:n2], keys[:n2], 4, unpack=False)
            fetched = int(np.minimum(np.maximum(c2, 0), 4).sum())
            hdr_gbs = 8.0 * args.seed_probes / (ms_hdr * 1e-3) / 1e9
            seed = {"kernel": "xm_seed_probe_lines_kernel<%s>" % ("64-byte lines" if info["bucket_line_bytes"] == 64 else "32-byte lines") if info.get("bucket_line_bytes") else "xm_seed_probe_kernel", "probes": args.seed_probes, "kernel_ms": round(ms_hdr, 4),
                    "probes_per_s": round(args.seed_probes / (ms_hdr * 1e-3), 1), "achieved": round(hdr_gbs, 2), "unit": "GB/s", "peak": 8000.0,
                    "frac": round(hdr_gbs / 8000.0, 5),
                    "random_64B_gather_ceiling_sectors_per_s": round(sectors_per_s, 1),
                    "frac_of_gather_ceiling": round(args.seed_probes / (ms_hdr * 1e-3) / sectors_per_s, 4),
                    "with_positions": {"probes": n2, "positions_fetched": fetched, "kernel_ms": round(ms_pos, 4),
                                       "achieved_GBps": round((8.0 * n2 + pos_bytes * fetched) / (ms_pos * 1e-3) / 1e9, 2)},
                    "note": "algorithmic bytes: 8 B header per probe (+ %d B per fetched position); every probe touches one random 64 B sector, so the "
                            "sector ceiling (%.1f G sectors/s = %.0f GB/s of sector traffic) is the bound that applies; THIS index (%d MB) is cache-resident: "
                            "hbm_resident_index below is the HBM number" % (pos_bytes, sectors_per_s / 1e9, sectors_per_s * 64 / 1e9, info["index_bytes"] >> 20)}
            if args.seed_index_mb > 0 or big:
                for c_ in ctx:
                    c_.set_scratch(1 << 20)  # (the contexts give their scratch back: the second index and its build need the room)
                seed["hbm_resident_index"] = seed_probe_hbm(api, synth, db if big else None, args.seed_index_mb, local_rank, sectors_per_s, args.seed_probes, rng)

        # HBM bytes per launch from the PMC passes committed with this round's profiles (same command, scripts/gpu_profile_round.sh);
        # only quoted for the workload they were collected on, with their source
        traffic = None
        traffic_source = None
        try:
            if args.config == "1" and (args.reads, args.ref_len, args.read_len) == (1_000_000, 5_000_000, 150):
                rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, "pmc_summary.json")))
                pm = json.load(open(os.path.join(ROOT, "profiles", rounds[-1], "pmc_summary.json")))
                if pm.get("build") == build and int(pm.get("contexts", n_ctx)) == n_ctx:
                    traffic = int(pm["hbm_bytes_per_launch"]["mean_over_the_two_launches_of_a_step"])
                    traffic_source = "profiles/%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, build %s)" % (rounds[-1], pm["build"])
                else:  # counters of another build (or another number of contexts) say nothing about this one
                    traffic_source = "none: profiles/%s/pmc_summary.json was collected on build %s, this is %s" % (rounds[-1], pm.get("build", "?"), build)
            elif args.config in ("2", "4shape", "1rep") and (args.reads, args.ref_len) == (1_000_000, 5_000_000):
                name = "pmc_config%s.json" % args.config
                rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, name)))
                pm = json.load(open(os.path.join(ROOT, "profiles", rounds[-1], name)))
                if pm.get("build") == build:
                    gs = [g for g in pm["launches_by_grid"] if g["launches_profiled"] > 0]
                    traffic = int(sum(g["hbm_bytes_per_launch"] * g["launches_profiled"] for g in gs) / sum(g["launches_profiled"] for g in gs))  # mean over the launches of a step
                    traffic_source = "profiles/%s/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload by launch grid, build %s; mean over the launches)" % (rounds[-1], name, pm["build"])
                else:
                    traffic_source = "none: profiles/%s/%s was collected on build %s, this is %s" % (rounds[-1], name, pm.get("build", "?"), build)
        except Exception:
            traffic = None

        value = world * nq * reads_per_query * args.steps / elapsed / 1e6
        line = {
            "metric": "M reads/s aligned (150 bp)", "value": round(value, 4), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not dry else "synthetic; DRY RUN on the host simulation of the kernel sources (--engine hostsim): the multi-rank flow is exercised, nothing is measured",
            "rank_batch_digests": digests if world > 1 else None,
            "pinned_host_bytes_high_water_per_rank": pinned,
            "config": {"workload": {"1": "configs[1]: %d synthetic %d bp single-end reads per GPU vs %d bp synthetic E. coli-sized reference (index replicated, reads sharded, no collective)",
                                    "1rep": "configs[1]'s reads on a repeat-rich reference: %d synthetic %d bp single-end reads per GPU vs %d bp synthetic reference with segmental duplications (90-99.5 %% identity), tandem repeats and an overfull-bucket 28-mer (synth.repeat_rich_reference)",
                                    "2": "configs[2] shape: %d synthetic 2 x %d bp pairs (--spacing 100 50) per GPU vs %d bp synthetic E. coli-sized reference",
                                    "4shape": "configs[4] shape: %d synthetic %d bp queries (what --split-queries-past-size 1000 makes of 10 kb reads) per GPU vs %d bp synthetic reference",
                                    "3shape": "configs[3] on one GPU: %d synthetic 2 x %d bp pairs (--spacing 100 50, seed 0x5EED0003) sampled genome-wide vs the %d bp GRCh38-shaped synthetic reference (24 contigs, real chromosome lengths, 1 %% N-runs of 10 kb, seed 0x6C38)",
                                    "3rep": "configs[3]'s read model against a genome that looks like one, on one GPU: %d synthetic 2 x %d bp pairs (--spacing 100 50, seed 0x5EED0003) sampled genome-wide vs the %d bp GRCh38-shaped reference with interspersed repeat families (85-95 %% identity), segmental duplications, tandem repeats and N-runs (synth.grch38_repeat_rich_reference)",
                                    "4": "configs[4] on one GPU: %d queries of %d bp = 10 kb reads (5 %% substitutions + 5 %% indel events per base, seed 0x5EED0004) cut by --split-queries-past-size 1000, vs the %d bp GRCh38-shaped synthetic reference",
                                    "4mild": "configs[4] shape on one GPU with milder reads: %d queries of %d bp = 10 kb reads (2 %% substitutions + 0.2 %% indel events per base) cut by --split-queries-past-size 1000, vs the %d bp GRCh38-shaped synthetic reference"}[args.config] % (nq, args.read_len, args.ref_len),
                       "reads_per_gpu": nq * reads_per_query, "read_len": args.read_len, "reference_len": args.ref_len, "parallelism": "reads sharded x%d" % world,
                       "aligned_reads": aligned, "quick_accept_fraction": round(int(c[7]) / max(1, nq), 4), "candidates_extended_per_query": round(int(c[4]) / max(1, nq), 3),
                       "header_probes_per_query": round(int(c[1]) / max(1, nq), 2),
                       "reference_repeats": rep_stats if args.config in ("1rep", "3rep") else None,
                       "overfull_buckets": overfull,
                       "index_build_s": round(index_build_s, 3), "index_bytes": info["index_bytes"],
                       "index_build": {"hashed_on": "gpu" if info["built_on_device"] else "host", "hash_s": round(info["hash_seconds"], 3),
                                       "duplication_map_s": round(info["duplication_seconds"], 3)}},
            "roofline": {"bound": "hbm", "kernel": "xm_align_kernel", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 6), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_step": int(alg_bytes), "bytes_per_read": round(alg_bytes / nq, 1),
                         "concurrent_launches": n_ctx, "frac_of_all_concurrent_launches": round(alg_bytes * args.steps / elapsed / 1e9 / 8000.0, 6),
                         "frac_one_launch_at_a_time": None if not single else round(alg_bytes / (single["kernel_ms_per_step"] * 1e-3) / 1e9 / 8000.0, 6),
                         "kernel_ms_per_step": round(kernel_ms / args.steps, 3), "launches_per_step": launches / args.steps,
                         "result_d2h_ms_per_step": round(d2h_ms / args.steps, 3),
                         "kernel_ms_by_pass": {"light_pass": round(pass_us[0] / args.steps / 1e3, 3), "gapped_and_rerun_passes": round(pass_us[3] / args.steps / 1e3, 3)},
                         "traffic_rate": None if traffic is None else round(traffic / (avg_launch_ms * 1e-3) / 1e9, 1),
                         "note": "achieved/peak/frac: algorithmic bytes of one launch over that launch's duration, against the 8 TB/s stream peak; with several contexts "
                                 "the launches of the contexts share the GPU, so a launch lasts longer than it would alone (single_context has the kernel's numbers with one "
                                 "launch at a time; frac_one_launch_at_a_time is this fraction from that leg) and frac_of_all_concurrent_launches is the algorithmic rate of the GPU as a whole.  traffic (PMC) is per-lane scratch in HBM, touched in "
                                 "scattered 32-64 B pieces: traffic_rate (GB/s, traffic / average launch duration) is to be read against this GPU's measured "
                                 "random-64-B-sector ceiling (seed_probe.random_64B_gather_ceiling_sectors_per_s x 64 B), not against the stream peak"},
            "contexts": {"per_gpu": n_ctx, "scratch_gib_each": round(scratch_each / 2**30, 1) if scratch_each else None,
                         "note": "a step is one whole pass of the hot path over one resident batch; the steps are dealt to %d contexts of the GPU (xm_context_new: one index - host "
                                 "tables and tables in HBM - shared by all; a resident copy of the batch, a host thread, a stream and a share of the scratch each) that align at the same time: "
                                 "the wave slots one context's gapped pass leaves idle are filled by the others' passes (profiles/r02/NOTES.md 12, 14; profiles/r03/NOTES.md)" % n_ctx if n_ctx > 1 else "one context"},
            "single_context": single,
            "cpu_baseline": cpu,
            "build": build,
            "bit_identical": same,
            "golden": golden,
            "counters": counters,
            "pcie_inclusive": pcie,
            "streamed": streamed,
            "wave_form": wave,
            "seed_probe": seed,
        }
        if same is False or (golden is not None and not golden["matches_committed"]) or (wave is not None and not wave["bit_identical_to_default_path"]) or counters.get("equal") is False:
            rc = 1  # a parity failure is not a measurement
    for c_ in ctx:
        c_.close()
    if rank == 0 and line is not None and world == 1 and args.end_to_end_reads > 0 and args.config in ("1", "2") and args.ref_len <= 50_000_000:
        # (after the contexts of the timed region are closed: the command line builds its own index and contexts)
        line["end_to_end"] = end_to_end(args, ref, synth)
    if rank == 0 and line is not None:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def end_to_end(args, ref, synth):
    """`python -m mapper_amd --reference R --queries Q --out-sam S` (cli.run, in this process) from a FASTQ file on disk to a SAM file on disk: the whole harness of
    SURVEY.md section 8(f) rank 1 - native reader, batches streamed through the GPU contexts, native SAM formatter (mapper_amd/hostio.py).  `value` counts from the
    first byte read of the query file to the last byte written of the SAM file (the reference is parsed and hashed before that: wall_seconds has everything).
    PCIe- and disk-inclusive: never the headline value."""
    import io
    import shutil
    import tempfile
    from mapper_amd import cli
    n = args.end_to_end_reads
    d = tempfile.mkdtemp(prefix="xm_e2e_")
    try:
        dec = np.frombuffer(b"?ACMGRSVTWYHKDBN", dtype=np.uint8)
        with open(os.path.join(d, "ref.fa"), "wb") as f:
            f.write(b">ecoli_syn\n" + dec[ref].tobytes() + b"\n")

        def fastq(path, reads_, tag):  # fixed-width records, written as one byte matrix
            m, L = reads_.shape
            head = np.frombuffer(("@r%09d" + tag + "\n") .encode() % 0, dtype=np.uint8)
            rows = np.empty((m, len(head) + L + 3 + L + 1), dtype=np.uint8)
            rows[:, :len(head)] = head
            idx = np.arange(m)
            for k in range(9):  # the nine digits of the read number
                rows[:, 2 + k] = 48 + (idx // 10 ** (8 - k)) % 10
            rows[:, len(head):len(head) + L] = dec[reads_]
            rows[:, len(head) + L:len(head) + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
            rows[:, len(head) + L + 3:len(head) + 2 * L + 3] = 73  # 'I'
            rows[:, -1] = 10
            rows.tofile(path)
        if args.config == "2":
            m1, m2 = synth.synthetic_paired_end(ref, n // 2, read_len=150, seed=0x5EED0E2E)[:2]
            fastq(os.path.join(d, "r1.fq"), m1, "/1"); fastq(os.path.join(d, "r2.fq"), m2, "/2")
            qargs = ["--paired-queries", os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq"), "--spacing", "100", "50"]
            n_queries, n_reads = n // 2, 2 * (n // 2)
        else:
            fastq(os.path.join(d, "reads.fq"), synth.synthetic_single_end(ref, n, read_len=args.read_len, seed=0x5EED0E2E)[0], "")
            qargs = ["--queries", os.path.join(d, "reads.fq")]
            n_queries = n_reads = n
        in_bytes = sum(os.path.getsize(os.path.join(d, f_)) for f_ in os.listdir(d) if f_.endswith(".fq"))
        log = io.StringIO()
        t0 = time.perf_counter()
        rc = cli.run(["--reference", os.path.join(d, "ref.fa")] + qargs + ["--out-sam", os.path.join(d, "out.sam")], out=log)
        wall = time.perf_counter() - t0
        t = cli.last_timing or {}
        sam_bytes = os.path.getsize(os.path.join(d, "out.sam"))
        rate_line = [l_ for l_ in log.getvalue().splitlines() if "Alignment rate" in l_]
        return {"value": round(n_reads / t["stream_seconds"] / 1e6, 4) if rc == 0 and t.get("queries") == n_queries else None, "unit": "Mreads/s", "reads": n_reads,
                "stream_seconds": round(t.get("stream_seconds", 0.0), 3), "wall_seconds": round(wall, 3), "contexts": t.get("contexts"), "fastq_bytes": in_bytes, "sam_bytes": sam_bytes,
                "alignment_rate_line": rate_line[0].strip() if rate_line else None,
                "note": "python -m mapper_amd (cli.run) from FASTQ on disk to SAM on disk, batches of 1 M queries; value = reads / (first byte of the query files read .. last byte of the SAM "
                        "file written); wall_seconds adds parsing and hashing the reference and starting the contexts"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def seed_probe_hbm(api, synth, db, index_mb, device, sectors_per_s, n_probes, rng):
    """The seed-lookup micro-kernel on an index that no cache holds (the 5 Mb workload's 47 MB index is L2 / Infinity-Cache resident): a second
    index of `index_mb` M synthetic bases (hashed on the GPU; or the workload's own 3.1 Gb one), random (length, key) probes.  Three ways of reading the
    rate: (a) algorithmic bytes (8 B header + B_pos per returned position) against the 8 TB/s stream peak, (b) the 64-byte sectors the probes
    touch against the same peak, (c) probes against this GPU's measured random-64-B-sector gather rate.  `sorted`: the same probes issued in
    (table, bucket) order - what batching a read batch's probes by address could gain at best."""
    own = db is None
    t0 = time.time()
    if own:
        db = api.ReferenceDatabase([("probe_ref", np.concatenate([synth.synthetic_reference(min(50_000_000, index_mb * 1_000_000 - o), seed=0x9B0 + o // 50_000_000)
                                                                    for o in range(0, index_mb * 1_000_000, 50_000_000)]))], max_query_length=150, device=device)
    build_s = time.time() - t0
    info = db.info()
    pb = info["position_bytes"]
    lo, hi = info["min_interesting_size"], min(info["max_hashed_length"], 150)
    used = rng.integers(lo, hi + 1, size=n_probes, dtype=np.int32)
    keys = rng.integers(-2**31, 2**31 - 1, size=n_probes, dtype=np.int64).astype(np.int32)
    out = {"index_bytes": info["index_bytes"], "reference_bases": info["total_forward_size"], "position_bytes": pb, "build_s": round(build_s, 2) if own else None, "probes": n_probes}

    def measure(u, k, label):
        db.seed_probe(u[:4096], k[:4096], 0)
        _, _, ms_hdr = db.seed_probe(u, k, 0)
        n2 = len(u)  # (the same probes with their positions: up to seven each, what a bucket line holds)
        c2, _, ms_pos = db.seed_probe(u[:n2], k[:n2], 7, unpack=False)
        fetched = int(np.minimum(np.maximum(c2, 0), 7).sum())
        hdr_rate, pos_rate = len(u) / (ms_hdr * 1e-3), n2 / (ms_pos * 1e-3)
        out[label] = {"header_only": {"probes_per_s": round(hdr_rate, 1), "algorithmic_frac_of_peak": round(8.0 * hdr_rate / 8e12, 5), "sector_traffic_frac_of_peak": round(64.0 * hdr_rate / 8e12, 4),
                                      "frac_of_gather_ceiling": round(hdr_rate / sectors_per_s, 4)},
                      "with_positions": {"probes_per_s": round(pos_rate, 1), "positions_per_probe": round(fetched / n2, 3),
                                         "algorithmic_frac_of_peak": round((8.0 * n2 + pb * fetched) / (ms_pos * 1e-3) / 8e12, 5),
                                         "sector_traffic_frac_of_peak": round(64.0 * pos_rate / 8e12, 4), "frac_of_gather_ceiling": round(pos_rate / sectors_per_s, 4)}}
    measure(used, keys, "random_order")
    # the same probes in (table, bucket) order: bucket = key mod capacity, non-negative (PackedMap.getPackedKey, PackedMap.java:210-215)
    caps = np.ones(hi + 2, dtype=np.int64)
    for L in range(lo, hi + 1):
        caps[L] = max(1, db.table_shape(L)[0])
    bucket = np.mod(keys.astype(np.int64), caps[used])
    order = np.lexsort((bucket, used))
    measure(used[order], keys[order], "sorted_by_table_and_bucket")
    out["note"] = ("a probe moves one 64-byte sector (a bucket line) to use 8 B of header and ~%d B of positions: the algorithmic fraction is bounded by used bytes / 64 x the sector rate; "
                   "random sectors reach ~40 %% of the stream peak on this GPU (random_64B_gather_ceiling), so north_star's 40 %% of peak in algorithmic bytes is out of reach for a "
                   "hash probe whatever the kernel does; sorting the probes by address is the upper bound of what probe batching could add" % int(pb * out["random_order"]["with_positions"]["positions_per_probe"]))
    if own:
        db.close()
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: N child processes of this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would start them.  The parent never initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


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
