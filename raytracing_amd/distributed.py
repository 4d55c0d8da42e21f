"""Multi-GPU rendering: one process per GPU, samples-per-pixel sharding, one all-reduce.

The render path shards by sample index: rank r of n renders, for EVERY pixel, the contiguous sample
range ``shard_samples(total, r, n)`` into its own device accumulation buffer; the per-rank sums are
then added with a single ``torch.distributed.all_reduce`` (RCCL over xGMI when the backend is
``nccl``) of the (W, H, 3) float32 buffer - 24.9 MB at 1080p.  Because the random stream of a path
is keyed by (pixel, GLOBAL sample index), the merged image does not depend on the number of ranks
beyond float32 summation order.  No other data-path communication exists.
"""
from __future__ import annotations

from typing import Optional, Tuple

__all__ = ["shard_samples", "DistributedRenderer"]


def shard_samples(total: int, rank: int, world: int) -> Tuple[int, int]:
    """(first sample, count) of ``rank``: contiguous ranges whose sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


class DistributedRenderer:
    """Drives one backend per process and merges with ``torch.distributed``.

    ``backend`` needs ``render(W, H, sample_begin, sample_count, max_bounces, seed, accum_ptr=, stream=)``
    and ``synchronize(stream=)``; ``accum`` is a (W, H, 3) float32 torch tensor on the backend's device
    that the backend adds into (``HipBackend`` with ``accum_ptr=accum.data_ptr()``).
    """

    def __init__(self, backend, accum, process_group=None) -> None:
        import torch.distributed as dist

        self.backend = backend
        self.accum = accum
        self.group = process_group
        self.dist = dist
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.library_merge = False
        self.library_merge_error = None

    def render_local(self, width: int, height: int, total_samples: int, max_bounces: int, seed: int,
                     sample_offset: int = 0, stream: int = 0) -> Tuple[int, int]:
        begin, count = shard_samples(total_samples, self.rank, self.world)
        if count:
            self.backend.render(width, height, sample_offset + begin, count, max_bounces, seed,
                                accum_ptr=self.accum.data_ptr(), stream=stream)
        return begin, count

    def use_library_merge(self) -> bool:
        """Give the backend its own RCCL communicator (``rt_comm_init``), so that ``merge`` is the
        library's ``rt_merge`` and PyTorch is left with the tensor hand-off only.  Rank 0 makes the
        communicator id; ``torch.distributed`` merely carries its 128 bytes to the other ranks (any
        channel would do).  Returns False, and leaves ``merge`` on ``torch.distributed``, when the
        backend has no such entry points or RCCL cannot be set up.

        Every rank makes the same sequence of collectives whatever fails where: (1) rank 0 ALWAYS
        broadcasts - the id, or None when it could not make one; (2) every rank reports whether it holds
        an id and can open RCCL, and ``comm_init`` - itself a collective inside RCCL - is entered only
        when ALL ranks can; (3) the ranks agree on whether all of them came out of it."""
        if not hasattr(self.backend, "comm_init"):
            return False
        if self.world == 1:
            try:
                self.backend.comm_init(1, 0, None)
                self.library_merge = True
            except Exception as e:   # noqa: BLE001
                self.library_merge_error = repr(e)
                self.library_merge = False
            return self.library_merge
        box = [None]
        if self.rank == 0:
            try:
                box[0] = type(self.backend).comm_unique_id()
            except Exception as e:   # noqa: BLE001 - the other ranks are waiting in the broadcast: it must happen
                self.library_merge_error = repr(e)
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        ready = box[0] is not None
        if ready and hasattr(self.backend, "comm_ready"):
            try:
                ready = bool(self.backend.comm_ready())
            except Exception as e:   # noqa: BLE001
                self.library_merge_error = repr(e)
                ready = False
        flags = [None] * self.world
        self.dist.all_gather_object(flags, ready, group=self.group)
        if not all(flags):
            if self.library_merge_error is None:
                self.library_merge_error = f"ranks {[r for r, f in enumerate(flags) if not f]} cannot set RCCL up"
            self.library_merge = False
            return False
        ok = True
        try:
            self.backend.comm_init(self.world, self.rank, box[0])
        except Exception as e:   # noqa: BLE001 - whatever went wrong, the torch path still works
            self.library_merge_error = repr(e)
            ok = False
        self.dist.all_gather_object(flags, ok, group=self.group)   # all ranks or none
        self.library_merge = all(flags)
        return self.library_merge

    def order_collective_behind(self, stream: int = 0):
        """``torch.distributed`` orders a collective on a device tensor against torch's CURRENT stream only.
        When the render was enqueued on another stream (``stream``: a raw hipStream_t, 0 = the default
        stream), make the current stream wait for it - an event, no host synchronisation - and return a
        function that makes ``stream`` wait for the collective in turn.  Nothing to do for host tensors."""
        accum = self.accum
        if not getattr(accum, "is_cuda", False):
            return lambda: None
        import torch

        cur = torch.cuda.current_stream(accum.device)
        if int(stream or 0) == int(cur.cuda_stream):
            return lambda: None
        ext = torch.cuda.ExternalStream(int(stream), device=accum.device) if stream else torch.cuda.default_stream(accum.device)
        done = torch.cuda.Event()
        done.record(ext)
        cur.wait_event(done)

        def back():
            merged = torch.cuda.Event()
            merged.record(cur)
            ext.wait_event(merged)

        return back

    def merge(self, stream: int = 0) -> None:
        """Sum the per-rank buffers in place on every rank: one all-reduce (RCCL over xGMI), issued by
        the library (``rt_merge``) on the render's own stream after ``use_library_merge``, else through
        ``torch.distributed`` with the two streams ordered by events (``order_collective_behind``)."""
        if self.world <= 1:
            return
        if getattr(self, "library_merge", False):
            w, h = int(self.accum.shape[0]), int(self.accum.shape[1])
            self.backend.merge(w, h, accum_ptr=self.accum.data_ptr(), stream=stream)
        else:
            back = self.order_collective_behind(stream)
            self.dist.all_reduce(self.accum, op=self.dist.ReduceOp.SUM, group=self.group)
            back()

    def render(self, width: int, height: int, total_samples: int, max_bounces: int, seed: int,
               sample_offset: int = 0, stream: int = 0):
        self.render_local(width, height, total_samples, max_bounces, seed, sample_offset, stream)
        self.merge(stream)   # on the render's stream: ordered behind the kernels, no host synchronisation in between
        return self.accum
