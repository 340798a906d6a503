"""Multi-GPU partition of one image: disjoint sample ranges + one sum-reduce.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm).  Rank g of G renders samples
[g*spp/G, (g+1)*spp/G) of EVERY pixel into its own float4 accumulator (rt_render); the accumulators are summed
onto rank 0 with one reduce; rank 0 resolves.  Paths are keyed by (pixel, sample index), so the set of paths does
not depend on G.  This stands where the reference sums its per-sample images serially (src/main.cpp:720-726).
"""
import torch
import torch.distributed as dist


def sample_range(rank, world, spp):
    """Samples [begin, end) of rank `rank`; the ranges of all ranks tile [0, spp) exactly."""
    if not (0 <= rank < world) or spp < 0:
        raise ValueError(f"sample_range(rank={rank}, world={world}, spp={spp})")
    return rank * spp // world, (rank + 1) * spp // world


def reduce_accumulators(accum, dst=0, group=None, even_alone=False):
    """Sum the [A,B,G,R] accumulators of all ranks onto `dst` (in place there).  A no-op without a process group or with a
    single rank (even_alone=True issues the collective on a one-rank group too: a rehearsal of the RCCL path on one GPU)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or even_alone):
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum


def render_image(render_fn, resolve_fn, spp, rank=None, world=None, group=None):
    """One complete image.  render_fn(sample_begin, sample_end) -> accumulator tensor of this rank;
    resolve_fn(accum) -> 8-bit image, called on rank 0 only.  Returns (accum, image or None)."""
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    s0, s1 = sample_range(rank, world, spp)
    accum = render_fn(s0, s1)
    reduce_accumulators(accum, 0, group)
    image = resolve_fn(accum) if rank == 0 else None
    return accum, image
