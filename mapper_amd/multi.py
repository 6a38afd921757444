"""Sharding reads across the GPUs of one node.

Queries are independent (no state is shared between reads: SURVEY.md §8e), so the path partitions by read with the index
replicated on every GPU and NO collective on the data path.  One process per GPU; rank r aligns the contiguous shard
shard_range(nq, r, world).  Per-rank results are concatenated in rank order on the host (gather_streams), which gives a
deterministic output order; per-GPU mutation/depth histograms are summed on the host in fixed rank order
(reduce_histograms) as BASELINE.json's north_star prescribes.  torch.distributed is plumbing only (nccl = RCCL on the GPU
box, gloo in the CPU tests).
"""
import numpy as np


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
