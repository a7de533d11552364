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
    """Fixed total work: GPU g of G renders [g*spp/G, (g+1)*spp/G): [0, spp_total) cut into `world` contiguous
    ranges that differ by at most one sample.  Every rank must get at least one sample (pt_render_device rejects
    an empty range)."""
    if not (0 <= rank < world) or spp_total <= 0:
        raise ValueError("bad shard request")
    if spp_total < world:
        raise ValueError("%d samples cannot be split over %d ranks: every rank needs at least one" % (spp_total, world))
    base, extra = divmod(spp_total, world)
    offset = rank * base + min(rank, extra)
    return offset, base + (1 if rank < extra else 0)


def shard(mode, rank, world, spp):
    """(first absolute sample, samples) of `rank` under "weak" (spp per rank) or "strong" (spp in total) scaling."""
    if mode == "weak":
        return weak_shard(rank, world, spp)
    if mode == "strong":
        return strong_shard(rank, world, spp)
    raise ValueError("unknown scaling mode %r" % (mode,))


def all_reduce_sum(t):
    """Sums a linear accumulation tensor over all ranks in place (no-op without a process group)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
