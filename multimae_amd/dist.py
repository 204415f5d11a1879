"""Data-parallel gradient exchange for the flat gradient arena (RCCL over xGMI on MI355X).

Replaces torch DDP's reducer for this engine (reference: run_pretraining_multimae.py:380-387 wraps
the model in DistributedDataParallel; collective inventory SURVEY.md section 2.4 C1-C4).

Design for xGMI (8 GPUs fully connected, 7 links x ~153 GB/s each):

* The arena is laid out in the order the backward pass FINISHES gradients (engine.ParamArena readiness order: output
  adapters, encoder.L-1 ... encoder.0, then the input adapters and the global token), so "the next unit is done" always
  means "the next contiguous range is final" and the buckets -- few and LARGE, default 64 MiB instead of DDP's 25 MB -- are
  contiguous slices that complete one after another while backward is still running.  The last bucket (input adapters +
  global token, ~8 MB for ViT-B) is cut off on its own: it is the only all-reduce that cannot overlap backward.
* A bucket is SUMMED (no pre-scale pass over the arena); the 1/world_size average is folded into the fused optimiser's
  gradient scale (FusedAdamW.grad_prescale, mmae_opt_step) -- 392 MB less read-modify-write per step than a mul_.
* Stream ordering: every readiness report records an event on the stream that ran that unit's backward (the main stream or
  an output adapter's stream) and on its weight-gradient side stream; a bucket's all-reduce is issued from a dedicated launch
  stream that first waits for the events of EVERY unit inside the bucket, so it cannot start before any contributing
  gradient kernel has finished, whichever stream wrote it.  The compute streams are never blocked by the exchange;
  finish() makes the calling stream wait for all collectives.
* Readiness callbacks only make sense when backward writes straight into the arena (engine.set_direct_grads(True)); with
  autograd-delivered gradients (AccumulateGrad runs after the node returns) they are ignored and finish() reduces everything.

Backend-agnostic (torch.distributed): "nccl" == RCCL on ROCm; the gloo CPU tests run the same bucket logic on CPU tensors.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def plan_buckets(sizes: List[Tuple[str, int, int]], bucket_elems: int, cut_before: Iterable[str] = ()) -> List[Tuple[int, int, List[str]]]:
    """sizes: [(name, offset, padded_numel)] in arena order.  Returns contiguous buckets (start, end, names) of about
    bucket_elems elements, never splitting a tensor; a bucket also ends right before every name in cut_before."""
    cut = set(cut_before)
    buckets, cur, start, end = [], [], None, None
    for name, off, n in sizes:
        if cur and name in cut:
            buckets.append((start, end, cur))
            cur, start = [], None
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
    """Bucketed, overlap-capable all-reduce(sum) of a flat gradient buffer; average = sum * grad_prescale in the optimiser."""

    def __init__(self, grad: torch.Tensor, sizes: List[Tuple[str, int, int]], bucket_mb: float = 64.0,
                 group: Optional[dist.ProcessGroup] = None, cut_before: Iterable[str] = (), average_in_place: bool = False,
                 bf16_buckets: bool = False, force_collective: bool = False, exchange: str = 'all_reduce'):
        self.grad = grad
        # exchange: how a bucket is summed across the ranks.  'all_reduce' -- one RCCL all_reduce(SUM) (RCCL picks the algorithm: a ring
        # on a fully connected xGMI node).  'rs_ag' -- the direct form SURVEY 8(e) prefers for point-to-point xGMI: reduce_scatter (every
        # rank sums its 1/world shard, each peer's contribution arriving over its own link) followed by all_gather of the summed shards --
        # the same bytes as a ring all-reduce, as TWO collectives whose algorithm is fixed by construction; selectable so that the first
        # 8-GPU run can A/B ring against direct (bench.py --dp-exchange).  Results agree up to the order of the sum.
        if exchange not in ('all_reduce', 'rs_ag'):
            raise ValueError(f"GradAllReducer: exchange must be 'all_reduce' or 'rs_ag', got {exchange!r}")
        self.exchange = exchange
        self._shard_tmp: Dict[int, torch.Tensor] = {}
        self.bucket_mb = bucket_mb
        # bf16_buckets: every bucket travels as bf16 (cast -> all-reduce -> widen back into the fp32 arena): half the bytes on
        # the xGMI links (196 instead of 392 MB for ViT-B) for a bf16-rounded SUM -- the trade torch DDP's bf16_compress_hook
        # makes; off by default (the fp32 sum is what the reference's DDP produces)
        self.bf16_buckets = bf16_buckets
        # force_collective: issue the all-reduces even at world size 1 (bench.py --force-dist: RCCL + the launch-stream / event
        # ordering run on a single GPU; a one-rank all-reduce is a copy)
        self.force_collective = force_collective
        # gemm_cu_reserve: compute units the library's persistent GEMM grids leave free from the first bucket launch of a step
        # until finish() (ops.gemm_cu_reserve).  The ping-pong GEMMs hold one 8-wave workgroup with the whole register file and
        # 147 KiB of LDS on EVERY CU; RCCL's channel kernels cannot co-reside with one, so a bucket in flight either waits for a
        # GEMM to end or -- once its workgroups sit on k CUs -- leaves the next full-chip persistent grid with k workgroups that
        # start only after the first round has finished (a statically strided persistent kernel then takes twice as long).
        # With the reserve the grids are n_cu - k workgroups wide and everything stays resident: k / n_cu of the MFMA rate
        # while buckets fly instead of time-slicing.  0 = off (single GPU).
        self.gemm_cu_reserve = 0
        self._reserved = False
        self.timing = False                             # bench.py: event-time the exposed wait in finish(), every bucket's launch -> done, and
        self._wait_events: List[tuple] = []             # the stretch of backward that ran on the reduced-width GEMM grids
        self._bucket_events: List[tuple] = []           # (bucket, issue event, done event) on the launch stream
        self._reserve_events: List[tuple] = []          # (first bucket launch, finish()) on the compute stream
        self._reserve_ev0 = None
        self._bucket_events_committed = 0
        self._bf16_tmp: Dict[int, torch.Tensor] = {}
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = plan_buckets(sizes, int(bucket_mb * 1024 * 1024 / 4), cut_before)
        self.average_in_place = average_in_place        # tests / callers without the fused optimiser: scale the bucket after the sum
        self.direct_grads = lambda: True                # for_arena() binds engine.direct_grads
        self._bucket_of: Dict[str, int] = {}
        for i, (_, _, names) in enumerate(self.buckets):
            for n in names:
                self._bucket_of[n] = i
        self._remaining: List[int] = []
        self._handles = []
        self._launched: List[bool] = []
        self._events: List[list] = []
        self._launch_stream = None
        self._side_of = None
        self.reset()

    @classmethod
    def for_arena(cls, arena, **kw) -> 'GradAllReducer':
        from . import engine
        from .engine import ALIGN
        sizes = [(n, arena.offsets[n], (arena.sizes[n] + ALIGN - 1) // ALIGN * ALIGN) for n in arena.names if arena.trainable[n]]
        kw.setdefault('cut_before', arena.tail_names()[:1])      # the tail starts its own bucket(s)
        r = cls(arena.grad, sizes, **kw)
        r.direct_grads = engine.direct_grads
        r._side_of = engine.existing_side_stream_of
        return r

    @property
    def grad_prescale(self) -> float:
        """What the optimiser must multiply the (summed) gradients by: FusedAdamW.grad_prescale = reducer.grad_prescale."""
        return 1.0 if self.average_in_place else 1.0 / self.world

    def reset(self) -> None:
        """Back to "no bucket launched".  Also the recovery path after a step that raised between the first bucket launch and
        finish(): the CUs reserved for the collective go back to the GEMMs (ADVICE r4 -- the reserve used to stay for the rest
        of the process)."""
        if getattr(self, '_reserved', False):
            from . import ops
            ops.gemm_cu_reserve(0)
            self._reserved = False
        # an aborted step's half-open timing bracket and the bucket events it recorded: finish() has consumed / committed
        # its own before it calls reset(), so whatever is left beyond the committed mark belongs to a step that never finished
        self._reserve_ev0 = None
        del self._bucket_events[getattr(self, '_bucket_events_committed', 0):]
        self._remaining = [len(names) for _, _, names in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._events = [[] for _ in self.buckets]
        self._handles = []

    # -- stream bookkeeping -------------------------------------------------------------------------------------------
    def _record(self, buckets: Sequence[int]) -> None:
        """Remember "everything enqueued so far on the current stream and its weight-gradient side stream" for these buckets."""
        if not self.grad.is_cuda:
            return
        cur = torch.cuda.current_stream(self.grad.device)
        evs = [cur.record_event()]
        side = self._side_of(cur) if self._side_of is not None else None
        if side is not None:
            evs.append(side.record_event())
        for i in buckets:
            self._events[i].extend(evs)

    def _launch(self, i: int) -> None:
        if self._launched[i]:
            return
        self._launched[i] = True
        if self.world == 1 and not self.force_collective:
            return
        s, e, _ = self.buckets[i]
        view = self.grad[s:e]
        if self.gemm_cu_reserve > 0 and not self._reserved:
            from . import ops
            ops.gemm_cu_reserve(self.gemm_cu_reserve)    # GEMMs enqueued from here on leave room for the collective's kernels
            self._reserved = True
            if self.timing and self.grad.is_cuda:
                self._reserve_ev0 = torch.cuda.Event(enable_timing=True)
                self._reserve_ev0.record()               # on the compute stream: backward from here runs n_cu - k wide

        def exchange():
            buf = view
            if self.bf16_buckets:
                buf = self._bf16_tmp.get(i)
                if buf is None:
                    buf = self._bf16_tmp[i] = torch.empty(e - s, device=self.grad.device, dtype=torch.bfloat16)
                buf.copy_(view)
            n = e - s
            if self.exchange == 'rs_ag' and self.world > 1 and n % self.world == 0:
                # reduce_scatter into a shard buffer of its own (no aliasing of the collective's input), then all_gather of the
                # summed shards back over the whole bucket.  h.wait() orders the launching stream (the dedicated launch stream on a
                # GPU; the host on gloo) behind the first collective, never the compute streams.
                shard = self._shard_tmp.get(i)
                if shard is None or shard.dtype != buf.dtype:
                    shard = self._shard_tmp[i] = torch.empty(n // self.world, device=buf.device, dtype=buf.dtype)
                dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
                return dist.all_gather_into_tensor(buf, shard, group=self.group, async_op=True)
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        if self.grad.is_cuda:
            if self._launch_stream is None:
                self._launch_stream = torch.cuda.Stream(device=self.grad.device)
            self._record([i])                            # whatever the launching stream itself has written
            ls = self._launch_stream
            for ev in self._events[i]:
                ls.wait_event(ev)
            with torch.cuda.stream(ls):
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()                          # every contributing gradient kernel has finished: the bucket could start
                h = exchange()
                if self.timing:
                    h.wait()                             # orders the launch stream (only) behind the collective
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self._bucket_events.append((i, e0, e1))
        else:
            h = exchange()
        self._handles.append((h, i))

    # -- readiness ----------------------------------------------------------------------------------------------------
    def mark_ready(self, names: List[str]) -> None:
        """The gradients of these parameters are final (written by kernels already enqueued on the current stream or its
        side stream): launch every bucket that became complete."""
        if not self.direct_grads():
            return                                       # autograd still has to deliver them: finish() reduces everything
        touched, done = set(), []
        for n in names:
            i = self._bucket_of.get(n)
            if i is None:
                continue
            touched.add(i)
            self._remaining[i] -= 1
            if self._remaining[i] == 0:
                done.append(i)
        if touched:
            self._record(sorted(touched))
        for i in done:
            self._launch(i)

    def mark_prefix_ready(self, prefix: str) -> None:
        self.mark_ready([n for n in self._bucket_of if n.startswith(prefix + '.') or n == prefix])

    def finish(self) -> None:
        """Launch whatever is left (parameters that never reported) and make the current stream wait for all buckets."""
        pending = [i for i in range(len(self.buckets)) if not self._launched[i]]
        active = self.world > 1 or self.force_collective
        if pending and self.grad.is_cuda and active:
            # gradients that were not reported may have been written on any stream: order behind all of them (autograd has
            # joined the streams it ran backward nodes on with the caller's stream; the side streams are joined here)
            self._join_all(pending)
        for i in pending:
            self._launch(i)
        ev0 = None
        if self.timing and self.grad.is_cuda and self._reserve_ev0 is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                                  # end of backward on the compute stream
            self._reserve_events.append((self._reserve_ev0, ev))
            self._reserve_ev0 = None
        if self.timing and self.grad.is_cuda and self._handles:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        for h, i in self._handles:
            h.wait()                                     # the calling stream waits for the collective (no host block on NCCL / RCCL)
            s, e, _ = self.buckets[i]
            if self.bf16_buckets:
                self.grad[s:e].copy_(self._bf16_tmp[i])  # widen the bf16 sum back into the fp32 arena
            if self.average_in_place and self.world > 1:
                self.grad[s:e].mul_(1.0 / self.world)
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self._wait_events.append((ev0, ev1))
        self._bucket_events_committed = len(self._bucket_events)
        self.reset()                                     # also lifts the CU reserve: the optimiser step and the next forward get the whole chip

    def exposed_wait_ms(self) -> float:
        """Sum of the event-timed waits finish() put on the calling stream since the last call (timing = True): the part of the
        gradient exchange that did NOT overlap backward.  Synchronises; call after the timed region."""
        tot = 0.0
        for a, b in self._wait_events:
            b.synchronize()
            tot += a.elapsed_time(b)
        self._wait_events = []
        return tot

    def bucket_timings(self) -> dict:
        """Self-diagnosis of an N > 1 run (timing = True; call after the timed region, synchronises): per bucket the mean time from
        "its last gradient kernel finished" to "its all-reduce finished" on the launch stream (size / that = the algorithm bandwidth the
        bucket saw while backward kept running), and the mean stretch of backward -- first bucket launch to finish() -- that ran on
        GEMM grids `gemm_cu_reserve` CUs narrower.  A bucket whose launch-to-done time approaches that stretch is the one finish()
        ends up waiting for."""
        per: Dict[int, List[float]] = {}
        for i, a, b in self._bucket_events:
            b.synchronize()
            per.setdefault(i, []).append(a.elapsed_time(b))
        self._bucket_events = []
        self._bucket_events_committed = 0
        res = []
        for a, b in self._reserve_events:
            b.synchronize()
            res.append(a.elapsed_time(b))
        self._reserve_events = []
        out = []
        for i in sorted(per):
            s, e, names = self.buckets[i]
            ms = sum(per[i]) / len(per[i])
            mb = (e - s) * (2 if self.bf16_buckets else 4) / 1e6
            out.append(dict(bucket=i, mb=round(mb, 1), first=names[0], launch_to_done_ms=round(ms, 3),
                            algbw_gb_s=round(mb / ms, 1) if ms > 0 else None, n=len(per[i])))
        return dict(buckets=out, backward_ms_on_reduced_width_grids=round(sum(res) / len(res), 3) if res else 0.0,
                    gemm_cu_reserved=self.gemm_cu_reserve)

    def _join_all(self, buckets: Sequence[int]) -> None:
        from . import engine
        evs = engine.all_stream_events(self.grad.device)
        for i in buckets:
            self._events[i].extend(evs)


def attach(model, reducer: GradAllReducer, optimizer=None) -> None:
    """Wire a MultiMAE model's backward-progress callbacks to the reducer (overlap with backward) and, when given, fold the
    1 / world_size average into the optimiser's gradient scale."""
    model._grad_ready_cb = reducer.mark_prefix_ready
    # encoder backward as several library calls, one per gradient bucket's worth of layers, so buckets launch in between
    per_layer = sum(p.numel() for n, p in model.named_parameters() if n.startswith('encoder.0.') and p.requires_grad)
    bucket = max((e - s for s, e, _ in reducer.buckets), default=0)
    model._bwd_chunk_layers = max(1, int(round(bucket / per_layer))) if per_layer else 1
    if optimizer is not None:
        optimizer.grad_prescale = reducer.grad_prescale


def broadcast_parameters(arena, src: int = 0, group=None) -> None:
    """DDP-constructor semantics (C2): every rank starts from rank src's parameters."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(arena.param, src=src, group=group)
