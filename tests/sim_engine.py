"""TEST HARNESS ONLY: the part of mapper_amd.api that bench.py drives, over the host simulation of the kernel sources (tests/hostsim_lib.py) instead of
libxmapper_hip.so - so that `bench.py --engine hostsim` can exercise its own multi-rank flow (spawned ranks, rendezvous, barriers, max-over-ranks timing, one
rank-0 line) on a machine without GPUs.  Nothing here is a product path: the package has no CPU alignment path, and a line printed with this engine is marked
as a dry run."""
import threading

import numpy as np

import hostsim_lib
import oracle_lib


class _Result:
    pass


_one_at_a_time = threading.Lock()   # (the host simulation keeps a read's context in static storage: bench.py's context threads take turns)


class SimDatabase:
    def __init__(self, contigs=None, mode="mapper", shared=None, **kw):
        self._sim = shared if shared is not None else hostsim_lib.SimReference([(n, np.ascontiguousarray(c, dtype=np.uint8)) for n, c in contigs], mode=mode)
        self._batch = None

    def info(self):
        mn, mx = self._sim.index_info()
        return {"position_bytes": 4, "index_bytes": 0, "built_on_device": 0, "hash_seconds": 0.0, "duplication_seconds": 0.0, "min_interesting_size": mn,
                "max_hashed_length": mx, "bucket_line_bytes": 0, "total_forward_size": 0, "num_positions": 0}

    def new_context(self):
        return SimDatabase(shared=self._sim)

    def set_scratch(self, nbytes):
        pass

    def upload_arrays(self, mc, mo, ml, codes, exp_in, dev_in):
        self._batch = oracle_lib.QueryBatch.from_arrays(np.ascontiguousarray(mc), np.ascontiguousarray(mo), np.ascontiguousarray(ml), np.ascontiguousarray(codes),
                                                        np.ascontiguousarray(exp_in), np.ascontiguousarray(dev_in))

    def align_resident(self, parameters):
        with _one_at_a_time:
            s = self._sim.align(self._batch, parameters)
        r = _Result()
        r.ints, r.dbls, r.int_off, r.dbl_off = s.ints, s.dbls, s.int_off, s.dbl_off
        r.counters = list(s.counters)
        r.extra = list(s.extra)
        r.kernel_ms, r.kernel_launches, r.d2h_ms, r.h2d_ms = 1.0, 2, 0.0, 0.0
        return r

    def bucket_stats(self):
        return None

    def close(self):
        pass


class Api:
    """mapper_amd.api with ReferenceDatabase and divide_scratch replaced (everything else - AlignmentParameters, the codecs - is the package's)."""

    def __init__(self, real):
        self._real = real
        self.ReferenceDatabase = SimDatabase

    def __getattr__(self, name):
        return getattr(self._real, name)

    @staticmethod
    def divide_scratch(contexts, device, **kw):
        return len(contexts), 1 << 30
