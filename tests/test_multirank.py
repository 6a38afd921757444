"""N>1 path on CPU: two ranks over gloo shard a batch, align their shards and rank 0 reassembles the streams in rank order;
the result must equal the single-process result.  (On the GPU box the same code runs one process per GPU over RCCL.)"""
import os
import sys
import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as o
    import hostsim_lib as hs
    from helpers import se_batch
    from mapper_amd import synth, multi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = synth.synthetic_reference(80_000)
    reads, _, _ = synth.synthetic_single_end(ref, 601)
    lo, hi = multi.shard_range(len(reads), rank, world)
    S = hs.SimReference([("ecoli_syn", ref)])        # index replicated on every rank
    s = S.align(se_batch(reads[lo:hi]), o.make_params())
    hist = np.bincount([len(s.query(q)[0]) for q in range(hi - lo)], minlength=4).astype(np.float64)
    g = multi.gather_streams(dist, s.ints, s.dbls, s.int_off, s.dbl_off, rank, world)
    h = multi.reduce_histograms(dist, hist, rank, world)
    if rank == 0:
        np.savez(tmp, ints=g[0], dbls=g[1], int_off=g[2], dbl_off=g[3], hist=h)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    import hostsim_lib as hs
    from helpers import se_batch
    from mapper_amd import synth, multi
    assert [multi.shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    hs.build()
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, 29517 + os.getpid() % 1000, out), nprocs=2, join=True)
    got = np.load(out)
    ref = synth.synthetic_reference(80_000)
    reads, _, _ = synth.synthetic_single_end(ref, 601)
    want = hs.SimReference([("ecoli_syn", ref)]).align(se_batch(reads), o.make_params())
    assert np.array_equal(got["ints"], want.ints) and np.array_equal(got["dbls"].view(np.int64), want.dbls.view(np.int64))
    assert np.array_equal(got["int_off"], want.int_off) and np.array_equal(got["dbl_off"], want.dbl_off)
    assert got["hist"].sum() == len(reads)
