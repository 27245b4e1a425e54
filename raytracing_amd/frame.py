"""A frame split over ranks: each rank renders a contiguous range of the global sample indices
of every pixel, the fixed-point accumulation buffers are summed with ONE all-reduce, and the
result is the unsplit frame bit for bit (the sums are integers and every sample draws its
random numbers from (pixel, global sample index)).

This is the data-parallel split the reference's author describes in prose -- run N copies with
different seeds and average the images ("Ray Tracing In a Weekend/README.md":2538-2542) --
made exact.
"""
from __future__ import annotations


def sample_range(spp: int, rank: int, world: int):
    """Samples [begin, begin+count) of rank `rank`: the first spp % world ranks get one extra
    (500 spp over 8 ranks -> 63,63,63,63,62,62,62,62)."""
    if world < 1 or not 0 <= rank < world or spp < 0:
        raise ValueError("bad rank/world/spp")
    base, extra = divmod(spp, world)
    count = base + (1 if rank < extra else 0)
    begin = rank * base + min(rank, extra)
    return begin, count


def render_frame(render_fn, accum, spp: int, rank: int = 0, world: int = 1, all_reduce=None):
    """Zero `accum`, add this rank's share with `render_fn(sample_begin, sample_count, accum)`,
    then sum over ranks with `all_reduce(accum)` (skipped when world == 1).

    `accum` is an int64 tensor [H, W, 3] (torch, on the device the renderer writes to);
    `render_fn` enqueues work that ADDS into it (rt_render_device does)."""
    begin, count = sample_range(spp, rank, world)
    accum.zero_()
    if count > 0:
        render_fn(begin, count, accum)
    if world > 1:
        all_reduce(accum)
    return accum
