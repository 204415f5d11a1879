"""Data-parallel gradient exchange for the flat gradient arena (RCCL over xGMI on MI355X).

Replaces torch DDP's reducer for this engine (reference: run_pretraining_multimae.py:380-387 wraps
the model in DistributedDataParallel; collective inventory SURVEY.md section 2.4 C1-C4).

Design for xGMI (8 GPUs fully connected, 7 links x ~153 GB/s each): few LARGE buckets (default 64 MiB
instead of DDP's 25 MB) over contiguous slices of the gradient arena, launched as soon as the
backward pass has finished the parameters they cover (the encoder stack and each output adapter
report completion), on RCCL's own stream so they overlap the remaining backward kernels.  The
arena is laid out in module registration order, so "encoder layer l done" == "a contiguous range is
final".  Averaging (1/world) is folded into the all-reduce via a pre-scale of the bucket.

Backend-agnostic (torch.distributed): "nccl" == RCCL on ROCm; the world_size-2 gloo CPU test runs the
same bucket logic on CPU tensors.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def plan_buckets(sizes: List[Tuple[str, int, int]], bucket_elems: int) -> List[Tuple[int, int, List[str]]]:
    """sizes: [(name, offset, padded_numel)] in arena order.  Returns contiguous buckets
    (start, end, names) of about bucket_elems elements, never splitting a tensor."""
    buckets, cur, start, end = [], [], None, None
    for name, off, n in sizes:
        if start is None:
            start = off
        cur.append(name)
        end = off + n
        if end - start >= bucket_elems:
            buckets.append((start, end, cur))
            cur, start = [], None
    if cur:
        buckets.append((start, end, cur))
    return buckets


class GradAllReducer:
    """Bucketed, overlap-capable all-reduce(mean) of a flat gradient buffer."""

    def __init__(self, grad: torch.Tensor, sizes: List[Tuple[str, int, int]], bucket_mb: float = 64.0,
                 group: Optional[dist.ProcessGroup] = None):
        self.grad = grad
        self.group = group
        self.join = None
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = plan_buckets(sizes, int(bucket_mb * 1024 * 1024 / 4))
        self._bucket_of: Dict[str, int] = {}
        for i, (_, _, names) in enumerate(self.buckets):
            for n in names:
                self._bucket_of[n] = i
        self._pending: List[int] = []
        self._remaining: List[int] = []
        self._handles = []
        self._launched: List[bool] = []
        self.reset()

    @classmethod
    def for_arena(cls, arena, **kw) -> 'GradAllReducer':
        from .engine import ALIGN
        sizes = [(n, arena.offsets[n], (arena.sizes[n] + ALIGN - 1) // ALIGN * ALIGN) for n in arena.names if arena.trainable[n]]
        r = cls(arena.grad, sizes, **kw)
        from . import engine
        r.join = engine.join_wgrad_streams
        return r

    def reset(self) -> None:
        self._remaining = [len(names) for _, _, names in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._handles = []

    def _launch(self, i: int) -> None:
        if self._launched[i] or self.world == 1:
            self._launched[i] = True
            return
        s, e, _ = self.buckets[i]
        view = self.grad[s:e]
        if self.join is not None:
            self.join()                      # weight-gradient side streams -> current stream
        view.mul_(1.0 / self.world)
        self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._launched[i] = True

    def mark_ready(self, names: List[str]) -> None:
        """The gradients of these parameters are final: launch every bucket that became complete."""
        for n in names:
            i = self._bucket_of.get(n)
            if i is None:
                continue
            self._remaining[i] -= 1
            if self._remaining[i] == 0:
                self._launch(i)

    def mark_prefix_ready(self, prefix: str) -> None:
        self.mark_ready([n for n in self._bucket_of if n.startswith(prefix + '.') or n == prefix])

    def finish(self) -> None:
        """Launch whatever is left (parameters that never reported) and wait for all buckets."""
        for i in range(len(self.buckets)):
            self._launch(i)
        for h in self._handles:
            h.wait()
        self.reset()


def attach(model, reducer: GradAllReducer) -> None:
    """Wire a MultiMAE model's backward-progress callbacks to the reducer (overlap with backward)."""
    model._grad_ready_cb = reducer.mark_prefix_ready


def broadcast_parameters(arena, src: int = 0, group=None) -> None:
    """DDP-constructor semantics (C2): every rank starts from rank src's parameters."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(arena.param, src=src, group=group)
