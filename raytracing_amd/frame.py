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
    (500 spp over 8 ranks -> 63,63,63,63,62,62,62,62).  The same arithmetic as the library's rt_split_samples."""
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


class FramePipeline:
    """Frames in flight on several slots, each slot = (accumulation buffer, stream, renderer).

    Frame k runs on slot k % n.  `submit(k)` zeroes the slot's buffer and enqueues this rank's share of
    frame k; with several ranks it then issues the all-reduce of the PREVIOUS frame (n > 1) or of this
    frame (n == 1), after draining that frame's stream on the host.  Draining first means the collective
    never waits on an unfinished kernel through a cross-stream event, and with n > 1 the host wait
    happens while the next frame is already queued on another stream, so the long-path tail of one
    frame and its all-reduce overlap the start of the next.  `flush()` issues what is still pending.

    A slot needs: `.accum` (int64 tensor), `.render(begin, count, accum, k)` (enqueues work that adds),
    `.stream()` (context manager making the slot's stream current) and `.drain()` (host wait for the
    slot's stream).  Every rank must call submit/flush in the same order.
    """

    def __init__(self, slots, spp: int, rank: int = 0, world: int = 1, all_reduce=None):
        if not slots:
            raise ValueError("need at least one slot")
        if world > 1 and all_reduce is None:
            raise ValueError("several ranks need an all_reduce")
        self.slots = list(slots)
        self.spp, self.rank, self.world, self.all_reduce = spp, rank, world, all_reduce
        self.begin, self.count = sample_range(spp, rank, world)
        self._pending = []          # slot indices whose frame is rendered (enqueued) but not yet reduced

    def _reduce(self, i):
        slot = self.slots[i]
        slot.drain()
        with slot.stream():
            self.all_reduce(slot.accum)

    def submit(self, k: int):
        i = k % len(self.slots)
        if i in self._pending:      # only with a caller that skips frames: never overwrite an unreduced frame
            self.flush()
        slot = self.slots[i]
        with slot.stream():
            slot.accum.zero_()
            if self.count > 0:
                slot.render(self.begin, self.count, slot.accum, k)
        if self.world > 1:
            while self._pending:
                self._reduce(self._pending.pop(0))
            if len(self.slots) == 1:
                self._reduce(i)
            else:
                self._pending.append(i)
        return slot.accum

    def flush(self):
        while self._pending:
            self._reduce(self._pending.pop(0))
