"""Single-node data parallelism: one process per GPU, replicated weights, SUM all-reduce of the flat gradient arena over
RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm), overlapped with backward.

The reference has no data parallelism (its `torch.distributed` import, M:7, is unused).  Semantics chosen here
(SURVEY 8e): the reference loss is an un-normalised SUM over all pixels (A:264), so the single-process equivalent of a
global batch is the SUM (not the mean) of the per-rank gradients; BatchNorm uses per-replica batch statistics.

Every gradient lives in one contiguous fp32 arena that the plan cuts into buckets of whole tensors (~25 MB,
`dmm_plan_grad_bucket`) listed in the order backward finishes them: head, decoder, block 4 ... stems.  The whole backward is
enqueued asynchronously, so `reduce_overlapped()` only has to enqueue, per bucket, "wait for the bucket's event" followed by
one all-reduce: RCCL then runs each exchange on its own stream as soon as the bucket is final, beside the remaining
data-gradient and weight-gradient kernels.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a few large messages
per step (17-40 MB each for DenseNet-121) keep the ring per-link bound rather than latency bound."""
import torch
import torch.distributed as dist


def bucket_ranges(numel, bucket_elems):
    """Plain equal-size cut of an arena (used when no plan-provided bucket list exists, e.g. a bare tensor)."""
    return [(off, min(bucket_elems, numel - off)) for off in range(0, numel, bucket_elems)]


class GradAllReduce:
    def __init__(self, model_or_arena, bucket_bytes=64 << 20, group=None, force=False):
        self._src = model_or_arena
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.group = group
        self.force = force  # issue the collectives even at world size 1 (exercises the RCCL path on a single GPU)
        self._feeder = None
        self.last_ranges = []

    def _arena(self):
        a = getattr(self._src, "grad_arena", self._src)
        return a() if callable(a) else a

    def _active(self):
        return dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)

    def all_reduce(self, async_op=False):
        """Sum the gradient arena over all ranks after backward, bucket by bucket (not overlapped)."""
        if not self._active():
            return []
        arena = self._arena()
        self.last_ranges = bucket_ranges(arena.numel(), self.bucket_elems)
        return [dist.all_reduce(arena[off:off + n], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
                for off, n in self.last_ranges]

    def reduce_overlapped(self):
        """Call right after model.loss_backward(): enqueues one all-reduce per plan bucket, each behind that bucket's
        readiness event, and returns the work handles; `wait(works)` orders the current stream (Adam) behind them."""
        if not self._active():
            return []
        model = self._src
        arena = self._arena()
        if not hasattr(model, "grad_buckets"):   # a bare arena tensor: no plan, no readiness events - plain asynchronous buckets
            return self.all_reduce(async_op=True)
        ranges = model.grad_buckets()
        self.last_ranges = ranges
        works = []
        if not arena.is_cuda:  # host tensors (gloo tests): nothing to overlap with
            return [dist.all_reduce(arena[off:off + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for off, n in ranges]
        if self._feeder is None:
            self._feeder = torch.cuda.Stream(device=arena.device)
        for i, (off, n) in enumerate(ranges):
            with torch.cuda.stream(self._feeder):
                # the collective is ordered behind the feeder stream, which waits for the bucket's event(s)
                model.grad_bucket_wait(i, self._feeder)
                works.append(dist.all_reduce(arena[off:off + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return works

    @staticmethod
    def wait(works):
        for w in works:
            w.wait()


def broadcast_parameters(model, src=0, group=None, force=False):
    """Make every rank start from rank `src`'s weights and running statistics."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    dist.broadcast(model.param_arena, src=src, group=group)
    dist.broadcast(model._buffer_arena, src=src, group=group)
