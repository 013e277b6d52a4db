"""Multi-GPU plumbing for the hot path: one process per GPU, reads sharded in contiguous blocks,
one modset per rank, and a single collective — the sum of the 65536-bin depth histograms
(BASELINE.json config 4).  torch.distributed is used only as the RCCL (backend "nccl") / gloo
transport; there is no data-path collective besides this one."""
import numpy as np


def shard_bounds(n_reads, world, rank):
    """contiguous block of reads [lo, hi) owned by `rank` (blocks differ in size by at most 1)"""
    base, extra = divmod(n_reads, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_offsets(offsets, world, rank):
    """offsets[] of this rank's block rebased to 0, plus the base offset of the block"""
    lo, hi = shard_bounds(len(offsets) - 1, world, rank)
    sub = np.asarray(offsets[lo:hi + 1])
    return (sub - sub[0]).astype(offsets.dtype), int(sub[0]), lo, hi


def allreduce_histogram(hist_tensor):
    """in-place SUM over ranks of a 65536-bin int64 histogram tensor (RCCL on GPU, gloo on CPU)"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist_tensor, op=dist.ReduceOp.SUM)
    return hist_tensor


def merge_modsets_in_rank_order(ms, lib):
    """Exact global modset from per-rank modsets built over CONTIGUOUS blocks of reads: every rank's
    (value, depth, info) arrays are gathered and folded into rank 0's modset in rank order with
    modsetMerge semantics (modset.c:106-128).  Because the blocks are contiguous, first-occurrence
    index order and saturated depths equal those of the single-stream build.  Returns on rank 0 the
    merged Modset* (ms itself); other ranks return None.  Uses only torch.distributed object/tensor
    collectives, so it runs over RCCL or gloo alike."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    if hasattr(lib, "modsetSyncToHost"):
        lib.modsetSyncToHost(ms, 0)
    m = ms.contents
    n = m.max
    mine = (np.ctypeslib.as_array(m.value, (n + 1,)).copy(), np.ctypeslib.as_array(m.depth, (n + 1,)).copy(),
            np.ctypeslib.as_array(m.info, (n + 1,)).copy())
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    for r in range(1, world):
        v, d, i = (np.ascontiguousarray(a) for a in gathered[r])
        ok = lib.mgModsetMergeArrays(ms, v.ctypes.data, d.ctypes.data, i.ctypes.data, len(v) - 1)
        if not ok:
            raise RuntimeError("modsets of different hashers cannot be merged")
    return ms
