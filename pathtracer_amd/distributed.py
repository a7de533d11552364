"""Multi-GPU sharding of one image: one process per GPU, samples-per-pixel split by absolute index.

Every rank renders ALL pixels for its own range of absolute sample indices into a private linear
fp32 sum buffer; the buffers are summed with one all-reduce (RCCL over xGMI when the backend is
"nccl"), and only then is the non-linear finish (mean, sqrt gamma, main.cu:95-98) applied.  The
random stream is keyed by the absolute sample index, so the image does not depend on the number of
ranks except for the fp32 association of the per-pixel sum.
"""


def weak_shard(rank, world, spp_per_rank):
    """Fixed work per rank: rank r renders samples [r*spp, (r+1)*spp); the image gets world*spp."""
    if not (0 <= rank < world) or spp_per_rank <= 0:
        raise ValueError("bad shard request")
    return rank * spp_per_rank, spp_per_rank


def strong_shard(rank, world, spp_total):
    """Fixed total work: [0, spp_total) cut into `world` contiguous ranges that differ by at most one."""
    if not (0 <= rank < world) or spp_total <= 0:
        raise ValueError("bad shard request")
    base, extra = divmod(spp_total, world)
    offset = rank * base + min(rank, extra)
    return offset, base + (1 if rank < extra else 0)


def all_reduce_sum(t):
    """Sums a linear accumulation tensor over all ranks in place (no-op for a single process)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
