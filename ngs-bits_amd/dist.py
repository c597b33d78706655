"""Multi-GPU plumbing: one process per GPU, counter vectors combined with ONE collective over RCCL/xGMI.

The hot path shards embarrassingly (one BAM — or one BGZF range of a BAM — per GPU); the only exchange is the reduction
of the ~8 KB int64 counter vector (SUM for counts and histograms, MAX for max_length / paired_end / yx_valid).
`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
import numpy as np

IDX_MAX_LENGTH, IDX_PAIRED_END, IDX_HALF_DEPTH, IDX_YX_VALID = 24, 25, 27, 31
MAX_REDUCED = (IDX_MAX_LENGTH, IDX_PAIRED_END, IDX_HALF_DEPTH, IDX_YX_VALID)


def combine_counters_local(vectors):
    """Reference semantics of the reduction, on host (used by tests and as documentation of the collective)."""
    v = np.stack([np.asarray(x, dtype=np.int64) for x in vectors])
    out = v.sum(axis=0)
    for i in MAX_REDUCED:
        out[i] = v[:, i].max()
    return out


def allreduce_counters(counters, device=None, group=None):
    """All-reduce one rank's NGSQC counter vector across the process group. Returns a numpy int64 array.

    Two tiny collectives on the same tensor layout: SUM over everything, then MAX over the 4 non-additive slots.
    """
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.asarray(counters, dtype=np.int64).copy()
    t = torch.tensor(np.asarray(counters, dtype=np.int64), device=device)   # private copy: the collective runs in place
    mx = t[list(MAX_REDUCED)].clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    t[list(MAX_REDUCED)] = mx
    return t.cpu().numpy()


def shard_blocks(n_blocks, world_size, rank):
    """Contiguous BGZF-member range [b0, b1) of rank `rank` when one BAM is split over `world_size` GPUs."""
    base, rem = divmod(n_blocks, world_size)
    b0 = rank * base + min(rank, rem)
    return b0, b0 + base + (1 if rank < rem else 0)
