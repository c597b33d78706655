"""Multi-GPU plumbing: one process per GPU, counter vectors combined with ONE collective over RCCL/xGMI.

The hot path shards embarrassingly (one BAM — or one BGZF range of a BAM — per GPU); the only exchange is the reduction
of the ~8 KB int64 counter vector (SUM for counts and histograms, MAX for max_length / paired_end / roi_bases / yx_valid).
`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
import numpy as np

IDX_MAX_LENGTH, IDX_PAIRED_END, IDX_ROI_BASES, IDX_HALF_DEPTH, IDX_YX_VALID = 24, 25, 26, 27, 31
MAX_REDUCED = (IDX_MAX_LENGTH, IDX_PAIRED_END, IDX_ROI_BASES, IDX_HALF_DEPTH, IDX_YX_VALID)   # identical or max-like on every rank: not additive


def combine_counters_local(vectors):
    """Reference semantics of the reduction, on host (used by tests and as documentation of the collective)."""
    v = np.stack([np.asarray(x, dtype=np.int64) for x in vectors])
    out = v.sum(axis=0)
    for i in MAX_REDUCED:
        out[i] = v[:, i].max()
    return out


def allreduce_counters(counters, device=None, group=None):
    """All-reduce one rank's NGSQC counter vector across the process group. Returns a numpy int64 array.

    Two tiny collectives on the same tensor layout: SUM over everything, then MAX over the non-additive slots.
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


# ---- one BAM sharded over the ranks (protocol in include/ngsqc.h) ----

class _DeviceInt32:
    """Aliases library-owned device memory as a torch tensor (no copy) through __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False), "version": 3, "strides": None}


def scan_mapping_sharded(handle, mode, device=None, group=None, comm=None, **scan_kw):
    """Rank-side driver: `handle` is this rank's shard (Handle(..., shard=(rank, world))). Every rank returns the same
    (counters, gc_reads) of the WHOLE BAM; afterwards handle.depth_stats()/depth() see the whole BAM's depth array.

    Collectives: all-gather of the 48-byte summaries, SUM/MAX all-reduce of the 8 KB counter vector, SUM all-reduce of
    gc_reads (808 B) and - when there is a target region - one in-place SUM all-reduce of the int32 difference array
    (device memory of the library; RCCL over xGMI with backend "nccl", host copy with "gloo").
    comm: a capi.Comm - the product's own RCCL collectives (include/ngsqc.h ngsqc_comm_*) carry every exchange instead of torch.distributed."""
    import torch
    import torch.distributed as dist
    from .capi import plan_shard_fix

    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if comm is not None:
        world, rank = comm.world, comm.rank
    sites = scan_kw.pop("sites", None); site_params = scan_kw.pop("site_params", (1, 13, False))
    site_counts = None
    if sites is not None:   # MappingQC's contamination pileup rides the same decode (its counts are additive over shards)
        job = handle.run_job(mapping=dict(scan_kw, mode=mode), sites=sites, site_params=site_params, partial=True)
        mine, site_counts = job["summary"], job["site_counts"]
    else:
        mine = handle.scan_mapping_partial(mode, **scan_kw)
    if comm is not None:
        # the whole exchange through the library's communicator: summaries, counters, gc_reads, the difference array in place on the device, site counts
        summaries = comm.allgather_summaries(mine)
        counters, gc = handle.scan_mapping_finish(plan_shard_fix(summaries, rank))
        counters = comm.allreduce_counters(counters); gc = comm.allreduce_f64(gc)
        comm.allreduce_depth(handle)
        handle.depth_finalize()
        if site_counts is not None:
            return counters, gc, summaries, comm.allreduce_i64(site_counts).reshape(np.asarray(site_counts).shape)
        return counters, gc, summaries
    if world > 1:
        t = torch.tensor(mine, dtype=torch.int64, device=device)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        summaries = np.stack([x.cpu().numpy() for x in parts])
    else:
        summaries = mine[None, :]
    fix = plan_shard_fix(summaries, rank)
    counters, gc = handle.scan_mapping_finish(fix)
    counters = allreduce_counters(counters, device=device, group=group)
    ptr, n = handle.depth_device()
    if world > 1:
        g = torch.tensor(gc, dtype=torch.float64, device=device)
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
        gc = g.cpu().numpy()
        if n > 0:
            done = False
            if device is not None and str(device).startswith("cuda"):
                try:
                    d = torch.as_tensor(_DeviceInt32(ptr, n), device=device)
                    dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
                    torch.cuda.synchronize()
                    done = True
                except (TypeError, RuntimeError, ValueError):
                    done = False
            if not done:
                d = torch.from_numpy(handle.depth_diff().copy())
                if device is not None and str(device).startswith("cuda"):
                    d = d.to(device)
                dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
                handle.depth_diff_set(d.cpu().numpy())
    handle.depth_finalize()
    if site_counts is not None:
        if world > 1:
            sc = torch.from_numpy(np.ascontiguousarray(site_counts)).to(device) if device is not None else torch.from_numpy(np.ascontiguousarray(site_counts))
            dist.all_reduce(sc, op=dist.ReduceOp.SUM, group=group)
            site_counts = sc.cpu().numpy()
        return counters, gc, summaries, site_counts
    return counters, gc, summaries


def scan_mapping_sharded_local(handles, mode, **scan_kw):
    """The same protocol inside ONE process for a list of shard handles in file order (tests; several shards on one GPU).
    Returns (counters, gc_reads, summaries); handles[0] ends up holding the whole BAM's finalized depth array."""
    from .capi import plan_shard_fix

    summaries = np.stack([h.scan_mapping_partial(mode, **scan_kw) for h in handles])
    parts, gcs = [], []
    for i, h in enumerate(handles):
        c, g = h.scan_mapping_finish(plan_shard_fix(summaries, i))
        parts.append(c); gcs.append(g)
    counters = combine_counters_local(parts)
    gc = np.sum(np.stack(gcs), axis=0)
    _, n = handles[0].depth_device()
    if n > 0:
        total = handles[0].depth_diff().astype(np.int64)
        for h in handles[1:]:
            total += h.depth_diff()
        handles[0].depth_diff_set(total.astype(np.int32))
    handles[0].depth_finalize()
    return counters, gc, summaries


def scan_depth_sharded_local(handles, regions, **kw):
    """Coverage-tool depth scan over shard handles in one process: handles[0] ends up with the whole BAM's depth array."""
    for h in handles:
        h.scan_depth(regions, partial=True, **kw)
    total = handles[0].depth_diff().astype(np.int64)
    for h in handles[1:]:
        total += h.depth_diff()
    handles[0].depth_diff_set(total.astype(np.int32))
    handles[0].depth_finalize()
    return handles[0]
