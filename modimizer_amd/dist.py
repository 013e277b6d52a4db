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


def allreduce_depth(depth_u16):
    """Reads counted per rank against ONE fixed modset that every rank holds (modasm.c:158-174: depth zeroed, ++depth per hit, saturating):
    returns min(65535, sum over ranks) per entry as uint16 -- the counts of the single stream over all reads, since a saturating add is
    associative.  Widened to int32 for the SUM (RCCL on GPU tensors, gloo on CPU).  The C form is mgDepthAllReduce."""
    import torch
    import torch.distributed as dist
    wide = torch.from_numpy(np.ascontiguousarray(depth_u16).astype(np.int32))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            wide = wide.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(wide, op=dist.ReduceOp.SUM)
    return wide.clamp_(max=65535).cpu().numpy().astype(np.uint16)


def merge_modsets_in_rank_order(ms, lib):
    """Exact global modset from per-rank modsets built over CONTIGUOUS blocks of reads: rank by rank, in rank
    order, a rank's (value, depth, info) arrays travel to rank 0 as three tensors (point-to-point send/recv: RCCL
    over xGMI with device tensors, gloo with host tensors) and are folded into rank 0's modset with modsetMerge
    semantics (modset.c:106-128).  Because the blocks are contiguous, first-occurrence index order and saturated
    depths equal those of the single-stream build.  Only one rank's arrays are in flight at a time, so rank 0
    needs room for its own set plus one more, whatever the world size.  Returns on rank 0 the merged Modset*
    (ms itself); other ranks return None."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    if hasattr(lib, "modsetSyncToHost"):
        lib.modsetSyncToHost(ms, 0)
    m = ms.contents
    n = m.max
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[rank] = n
    dist.all_reduce(counts)                                   # everybody learns every rank's entry count
    counts = counts.cpu().tolist()
    if rank != 0:
        v = torch.from_numpy(np.ctypeslib.as_array(m.value, (n + 1,)).view(np.int64).copy()).to(dev)
        d = torch.from_numpy(np.ctypeslib.as_array(m.depth, (n + 1,)).view(np.int16).copy()).to(dev)
        i = torch.from_numpy(np.ctypeslib.as_array(m.info, (n + 1,)).copy()).to(dev)
        for r in range(1, world):                             # rank order: wait for one's turn
            if r == rank:
                for t in (v, d, i):
                    dist.send(t, dst=0)
            dist.barrier()
        return None
    failed = None                                             # a failed merge is reported after the last round: the other ranks sit in the per-round barriers
    for r in range(1, world):
        nr = int(counts[r])
        v = torch.empty(nr + 1, dtype=torch.int64, device=dev)
        d = torch.empty(nr + 1, dtype=torch.int16, device=dev)
        i = torch.empty(nr + 1, dtype=torch.uint8, device=dev)
        for t in (v, d, i):
            dist.recv(t, src=r)
        hv = np.ascontiguousarray(v.cpu().numpy().view(np.uint64))
        hd = np.ascontiguousarray(d.cpu().numpy().view(np.uint16))
        hi = np.ascontiguousarray(i.cpu().numpy())
        if failed is None and not lib.mgModsetMergeArrays(ms, hv.ctypes.data, hd.ctypes.data, hi.ctypes.data, nr):
            failed = r
        del v, d, i
        dist.barrier()
    if failed is not None:
        raise RuntimeError("modsets of different hashers cannot be merged (rank %d's)" % failed)
    return ms
