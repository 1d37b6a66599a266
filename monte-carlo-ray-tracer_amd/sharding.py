"""Sample-index sharding across the GPUs of a node.

Camera samples are independent, so rank r of N renders global sample indices
[r*spp, (r+1)*spp) of EVERY pixel; random streams are keyed by the global
sample index (mcrt_rng_seed), which makes the image independent of N up to
summation order.  The only exchange is one sum-reduce of the W*H*3 double
accumulators to rank 0 per pass (RCCL over xGMI on GPUs, gloo in CPU tests).
"""


def rank_sample_range(rank, world, spp_per_rank, first_sample=0):
    """(sample_begin, sample_count) of `rank`; ranges of different ranks never overlap."""
    if not 0 <= rank < world:
        raise ValueError("rank outside the world")
    return first_sample + rank * spp_per_rank, spp_per_rank


def reduce_accumulator(accum, dist=None, dst=0):
    """Sum the per-rank accumulators into rank `dst` (in place).  `dist` is torch.distributed or None."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def total_spp(world, spp_per_rank):
    return world * spp_per_rank
