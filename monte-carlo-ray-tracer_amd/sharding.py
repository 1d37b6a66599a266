"""Sample-index sharding across the GPUs of a node.

Camera samples are independent and their random streams are keyed by the GLOBAL
sample index (mcrt_rng_seed), so which GPU renders which sample does not change
the image beyond summation order.  A job is a list of PASSES; in every pass each
rank renders a contiguous range of global sample indices of EVERY pixel into its
own double accumulator, then one sum-reduce of the W*H*3 doubles moves the
ranks' contributions to rank 0 (RCCL over xGMI on GPUs, gloo in CPU tests),
which resolves.  After a reduce the other ranks clear their accumulators (their
sums now live on rank 0), so a progressive job can go on pass after pass.

  weak scaling    every rank brings `spp` samples of its own: rank r owns global
                  indices [r*spp, (r+1)*spp); the job has world*spp per pixel.
  strong scaling  `spp` is the job's total; every pass window is split over the
                  ranks (the first `window % world` ranks take one sample more).
"""


def split_range(begin, count, parts):
    """`parts` contiguous sub-ranges [(begin_i, count_i)] covering [begin, begin+count); sizes differ by at most one."""
    base, extra = divmod(count, parts)
    out, at = [], begin
    for i in range(parts):
        n = base + (1 if i < extra else 0)
        out.append((at, n))
        at += n
    return out


def rank_sample_range(rank, world, spp_per_rank, first_sample=0):
    """Weak scaling: (sample_begin, sample_count) of `rank`; ranges of different ranks never overlap."""
    if not 0 <= rank < world:
        raise ValueError("rank outside the world")
    return first_sample + rank * spp_per_rank, spp_per_rank


class SamplePlan:
    """What one rank renders, pass by pass.

    passes        [(first global sample index, count)] of THIS rank; every rank has the same number of passes
                  (each pass ends in a collective), a count may be 0 when a window has fewer samples than ranks
    total_spp     samples per pixel of the whole job (all ranks, all passes)
    spp_after(k)  samples per pixel that rank 0's accumulator holds after k passes were reduced
    """

    def __init__(self, rank, world, spp, scaling="weak", pass_spp=None, first_sample=0):
        if not 0 <= rank < world:
            raise ValueError("rank outside the world")
        if scaling not in ("weak", "strong"):
            raise ValueError("scaling is 'weak' or 'strong'")
        self.rank, self.world, self.scaling = rank, world, scaling
        step = int(pass_spp) if pass_spp else int(spp)
        windows = [(b, min(step, spp - b)) for b in range(0, spp, step)]          # per rank (weak) or of the job (strong)
        self.passes, self._after = [], []
        done = 0
        for b, n in windows:
            if scaling == "weak":
                self.passes.append((first_sample + rank * spp + b, n))
                done += n * world
            else:
                self.passes.append(split_range(first_sample + b, n, world)[rank])
                done += n
            self._after.append(done)
        self.total_spp = done

    def spp_after(self, passes_done):
        return self._after[passes_done - 1] if passes_done > 0 else 0


def reduce_accumulator(accum, dist=None, dst=0):
    """Sum the per-rank accumulators into rank `dst` (in place).  `dist` is torch.distributed or None."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def after_reduce(accum, dist=None, dst=0):
    """Progressive jobs: a rank whose sums have just been moved to `dst` starts its next pass from zero."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_rank() != dst:
        accum.zero_()
    return accum


def total_spp(world, spp_per_rank):
    return world * spp_per_rank
