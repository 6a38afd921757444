"""In-process multi-GPU driver for the alignment path: the index is built once and replicated to every GPU
(xm_index_replicate: HBM-to-HBM copies), batch k of the query stream goes to GPU k mod N, every GPU has its own host thread and two streams
(the upload of its next batch overlaps the alignment of its current one, ReferenceDatabase.align_stream), and the results come back in batch
order.  No collective: reads are independent (the reference shares one HashBlock_Database between its AlignerWorker threads and hands them
batches of queries the same way, Mapper.java:912-1134, AlignerWorker.java:92-175).

bench.py's --gpus N keeps the one-process-per-GPU form of the benchmark contract (torch.distributed over RCCL); this module is the product's
form of the same sharding, used by `python -m mapper_amd.cli --gpus N`.

The functions below the class (shard_range, gather_streams, reduce_histograms) are the one-process-per-GPU form of the same sharding:
rank r aligns the contiguous shard shard_range(nq, r, world), the per-rank results are concatenated in rank order on the host, per-GPU
histograms are summed on the host in fixed rank order as BASELINE.json's north_star prescribes; torch.distributed is plumbing only
(nccl = RCCL on the GPU box, gloo in the CPU tests)."""
import queue
import threading


class MultiGpuDatabase:
    def __init__(self, contigs, devices, **kw):
        """devices: GPU ordinals, e.g. [0, 1, 2, 3]; a repeated ordinal gives that GPU two independent contexts (tests on one-GPU machines).
        kw: ReferenceDatabase's build options (mode, enable_gapmers, max_query_length, cache_dir)."""
        from . import api
        if not devices:
            raise ValueError("at least one device")
        kw.pop("device", None)
        # what every context will allocate beside its scratch (the command line's --out-mutations: one pile-up per context, 40-48 bytes per reference base)
        per_context_extra = kw.pop("per_context_extra", 0)
        self.devices = [int(d) for d in devices]
        # contexts of one GPU (a repeated ordinal) share the index's tables in its HBM (xm_context_new: no copy) and divide what is free after the
        # index is resident as their scratch (api.divide_scratch; a GPU with room for fewer contexts than asked uses fewer).  Three contexts
        # align ~10 % more reads per second than one (profiles/r02/NOTES.md): the waves of one context's gapped pass leave slots idle
        # that another context's passes fill.
        first = api.ReferenceDatabase(contigs, device=self.devices[0], **kw)
        by_device = {self.devices[0]: first}
        self.replicas = [first]
        for d in self.devices[1:]:
            if d in by_device:
                self.replicas.append(by_device[d].new_context())
            else:
                by_device[d] = first.replicate(d)
                self.replicas.append(by_device[d])
        keep = []
        for d in by_device:
            mine = [r for r, dd in zip(self.replicas, self.devices) if dd == d]
            n = api.divide_scratch(mine, d, per_context_extra=int(per_context_extra))[0] if len(mine) > 1 else 1
            keep += mine[:n]
            for r in mine[n:]:
                r.close()
        order = {id(r): k for k, r in enumerate(self.replicas)}
        keep.sort(key=lambda r: order[id(r)])
        self.devices = [dd for r, dd in zip(self.replicas, self.devices) if any(r is k for k in keep)]
        self.replicas = keep
        self.contigs = first.contigs

    def info(self):
        return self.replicas[0].info()

    def close(self):
        for r in self.replicas:
            r.close()
        self.replicas = []

    def align_stream(self, batches, parameters, depth=2, on_aligned=None):
        """batches: iterable of upload_arrays' six-array tuples; yields their BatchResults in order.  Batch k is aligned by replica k mod N.
        on_aligned(replica, k) is called on the replica's thread while batch k is still resident there."""
        n = len(self.replicas)
        inbox = [queue.Queue(maxsize=depth) for _ in range(n)]    # batches on their way to GPU g
        outbox = [queue.Queue(maxsize=depth) for _ in range(n)]   # results of GPU g, in its own order
        stop = threading.Event()

        def put(q, item):  # a put that gives up when the stream is being torn down (a consumer that went away must not block the producers)
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def feed(g):
            while not stop.is_set():
                try:
                    item = inbox[g].get(timeout=0.1)
                except queue.Empty:
                    continue
                if item is None:
                    return
                yield item

        def worker(g):
            try:
                hook = (lambda j: on_aligned(g, g + j * n)) if on_aligned else None
                for r in self.replicas[g].align_stream(feed(g), parameters, **({"on_aligned": hook} if hook else {})):
                    if not put(outbox[g], r):
                        return
                put(outbox[g], None)
            except BaseException as e:  # noqa: BLE001  (handed to the consumer)
                put(outbox[g], e)

        def dealer():
            end = None
            try:
                for k, arrays in enumerate(batches):
                    if not put(inbox[k % n], arrays):
                        break
                    dealt.put(k % n)
            except BaseException as e:  # noqa: BLE001  (reading or packing the queries failed: handed to the consumer)
                end = e
            finally:
                for g in range(n):
                    put(inbox[g], None)
                dealt.put(end)

        dealt = queue.Queue()
        threads = [threading.Thread(target=worker, args=(g,), daemon=True) for g in range(n)] + [threading.Thread(target=dealer, daemon=True)]
        for t in threads:
            t.start()
        try:
            while True:
                g = dealt.get()
                if g is None:
                    break
                if isinstance(g, BaseException):
                    raise g
                r = outbox[g].get()
                if isinstance(r, BaseException):
                    raise r
                if r is None:
                    raise RuntimeError("GPU %d stopped before its batch was aligned" % self.devices[g])
                yield r
        finally:
            stop.set()
            for t in threads:
                t.join(timeout=60)

    def align_batches(self, queries, parameters, batch_size, on_aligned=None):
        """Same contract as ReferenceDatabase.align_batches: yields (first query index, BatchResult) per batch, in order."""
        from . import api
        starts = list(range(0, len(queries), max(1, int(batch_size))))
        arrays = (api.ReferenceDatabase.batch_arrays(queries[s:s + batch_size]) for s in starts)
        hook = (lambda g, k: on_aligned(g, starts[k], queries[starts[k]:starts[k] + batch_size])) if on_aligned else None
        for s, r in zip(starts, self.align_stream(arrays, parameters, on_aligned=hook)):
            yield s, r


import numpy as np  # noqa: E402


def shard_range(nq, rank, world):
    """Contiguous, balanced shard [lo, hi) of nq queries for `rank` of `world`."""
    base, extra = divmod(int(nq), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_streams(dist, ints, dbls, int_off, dbl_off, rank, world):
    """Gathers per-rank result streams on rank 0 and concatenates them in rank order (host side)."""
    import torch
    payload = [np.asarray(ints), np.asarray(dbls), np.asarray(int_off), np.asarray(dbl_off)]
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(payload, gathered, dst=0)
    if rank != 0:
        return None
    out_i, out_d, io, do = [], [], [0], [0]
    for p_i, p_d, p_io, p_do in gathered:
        out_i.append(p_i)
        out_d.append(p_d)
        io.extend((np.asarray(p_io[1:]) + io[-1]).tolist())
        do.extend((np.asarray(p_do[1:]) + do[-1]).tolist())
    del torch
    return np.concatenate(out_i), np.concatenate(out_d), np.asarray(io, dtype=np.int64), np.asarray(do, dtype=np.int64)


def reduce_histograms(dist, hist, rank, world):
    """Host-side sum of per-GPU histograms in fixed rank order (float accumulation order is therefore deterministic)."""
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(np.asarray(hist), gathered, dst=0)
    if rank != 0:
        return None
    total = np.zeros_like(gathered[0])
    for h in gathered:
        total = total + h
    return total
