"""In-process multi-GPU driver for the alignment path: the index is built once and replicated to every GPU
(xm_index_replicate: HBM-to-HBM copies), batch k of the query stream goes to GPU k mod N, every GPU has its own host thread and two streams
(the upload of its next batch overlaps the alignment of its current one, ReferenceDatabase.align_stream), and the results come back in batch
order.  No collective: reads are independent (the reference shares one HashBlock_Database between its AlignerWorker threads and hands them
batches of queries the same way, Mapper.java:912-1134, AlignerWorker.java:92-175).

bench.py's --gpus N keeps the one-process-per-GPU form of the benchmark contract (torch.distributed over RCCL); this module is the product's
form of the same sharding, used by `python -m mapper_amd.cli --gpus N`."""
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
        self.devices = [int(d) for d in devices]
        first = api.ReferenceDatabase(contigs, device=self.devices[0], **kw)
        self.replicas = [first] + [first.replicate(d) for d in self.devices[1:]]
        self.contigs = first.contigs

    def info(self):
        return self.replicas[0].info()

    def close(self):
        for r in self.replicas:
            r.close()
        self.replicas = []

    def align_stream(self, batches, parameters, depth=2):
        """batches: iterable of upload_arrays' six-array tuples; yields their BatchResults in order.  Batch k is aligned by replica k mod N."""
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
                for r in self.replicas[g].align_stream(feed(g), parameters):
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

    def align_batches(self, queries, parameters, batch_size):
        """Same contract as ReferenceDatabase.align_batches: yields (first query index, BatchResult) per batch, in order."""
        from . import api
        starts = list(range(0, len(queries), max(1, int(batch_size))))
        arrays = (api.ReferenceDatabase.batch_arrays(queries[s:s + batch_size]) for s in starts)
        for s, r in zip(starts, self.align_stream(arrays, parameters)):
            yield s, r
