"""Engine state: activation precision, flat parameter arenas and bf16 weight shadows.

Memory layout (MI355X-first): all parameters of a model live in ONE flat fp32 HBM
arena (each tensor 64-element aligned), their gradients in a second arena of the same
layout and -- in bf16 speed mode -- a bf16 *shadow* arena that the GEMMs read.  The
``nn.Parameter`` objects are views into the arenas, so ``state_dict()`` / checkpoints
keep the reference's key/shape layout while

  * zero_grad is one memset, the gradient all-reduce runs over a few large contiguous
    buckets, AdamW + grad-norm are single launches over the arena,
  * the f32 -> bf16 weight refresh is one cast launch per step (or free: the fused
    AdamW writes the shadow itself).
"""
from __future__ import annotations

import contextlib
import weakref
from typing import Dict, Iterable, List, Optional

import torch
from torch import nn

from . import ops

ALIGN = 64  # elements

_state = {
    'act_dtype': torch.bfloat16,     # bf16 speed mode | float32 exact parity mode
    'direct_grads': False,           # backward accumulates straight into p.grad and returns None
    'adapter_streams': False,
    'wgrad_stream': False,           # direct-grad mode: weight-gradient GEMMs / bias column sums on a side stream per compute stream        # run the independent output adapters on separate HIP streams
    'fp32_adapter_gemm': 'h16',      # fp32_output_adapters in bf16 speed mode: 'h16' (fp16 STORAGE: the bf16 pipeline's kernels with TF32's significand) | 'f16' (f32 tensors, fp16 operands) | 'x3' (split bf16) | 'exact'
    'patch_domain_loss': __import__('os').environ.get('MMAE_PATCH_LOSS', '1') != '0',
    'first_write_stores': __import__('os').environ.get('MMAE_FIRST_WRITE', '1') != '0',     # A/B switch of claim_first_write()
    'encoder_fanout': __import__('os').environ.get('MMAE_ENC_FANOUT', '1') != '0',           # A/B switch of multimae._EncoderFanOut
}


def act_dtype() -> torch.dtype:
    return _state['act_dtype']


def set_precision(mode: str) -> None:
    """'bf16' (default): bf16 MFMA operands, fp32 accumulate/residual/softmax/LN/loss.
    'fp32': every operand f32 on the exact-f32 MFMA path (parity mode, 1/16 the rate).
    'mxfp8': as 'bf16', but the forward, dX and (ops.mx_wgrad, default on) weight-gradient products of the encoder blocks run on
    the block-scaled MFMA with OCP MX-fp8 operands (e4m3 elements, one power-of-two scale per 32 contraction elements;
    BASELINE.json configs[4]); attention, the adapters and everything else stay as in 'bf16'.  Needs dim_tokens and the MLP width to be multiples of 256
    (ViT-B / ViT-L); other encoders silently keep their bf16 products."""
    if mode not in ('bf16', 'fp32', 'mxfp8'):
        raise ValueError(mode)
    _state['act_dtype'] = torch.float32 if mode == 'fp32' else torch.bfloat16
    _state['mx_encoder'] = mode == 'mxfp8'


def mx_encoder() -> bool:
    return _state.get('mx_encoder', False)


def precision_mode() -> str:
    return 'fp32' if _state['act_dtype'] == torch.float32 else ('mxfp8' if mx_encoder() else 'bf16')


@contextlib.contextmanager
def precision(mode: str):
    old = precision_mode()
    set_precision(mode)
    try:
        yield
    finally:
        set_precision(old)


def patch_domain_loss() -> bool:
    return _state['patch_domain_loss']


def set_patch_domain_loss(flag: bool) -> None:
    """Masked losses applied to an output adapter's prediction read the adapter's patch rows and hand their gradient back as
    patch rows (no f32 image-domain gradient, no patchify pass); the image is still materialised for the API.  Default on."""
    _state['patch_domain_loss'] = bool(flag)


def lazy_predictions() -> bool:
    return _state.get('lazy_predictions', True)


def set_lazy_predictions(flag: bool) -> None:
    """Output adapters return their (B, C, H, W) prediction as a lazy.LazyPrediction: the image is rearranged out of the patch rows
    the first time anything reads it (the masked losses never do).  Default on; False = written eagerly by the forward."""
    _state['lazy_predictions'] = bool(flag)


def enc_bwd_side_cus() -> int:
    return int(_state.get('enc_bwd_side_cus', 0))


def set_enc_bwd_side_cus(k: int) -> None:
    """A/B switch (default 0): compute units the encoder's backward leaves to its weight-gradient stream (ops.gemm_side_cus)."""
    _state['enc_bwd_side_cus'] = int(k)


def adapter_cu_share() -> int:
    return int(_state.get('adapter_cu_share', 0))


def set_adapter_cu_share(k: int) -> None:
    """Compute units the output adapters' persistent GEMM grids leave free while the adapters run on separate streams (A/B switch,
    default 0; see functions._adapter_cu_share)."""
    _state['adapter_cu_share'] = int(k)


def encoder_fanout() -> bool:
    return _state.get('encoder_fanout', True)


def set_encoder_fanout(flag: bool) -> None:
    """A/B switch (default on; env MMAE_ENC_FANOUT=0): the output adapters' gradients with respect to the shared encoder tokens are summed by ONE
    kernel behind one autograd node (multimae._EncoderFanOut, ops.add_n) instead of autograd's chain of at::native adds."""
    _state['encoder_fanout'] = bool(flag)


def adapter_streams() -> bool:
    return _state['adapter_streams']


def set_adapter_streams(flag: bool) -> None:
    """Run each output adapter (forward and, through autograd's stream replay, backward) on its own HIP stream."""
    _state['adapter_streams'] = bool(flag)


def wgrad_stream() -> bool:
    return _state['wgrad_stream']


def set_wgrad_stream(flag: bool) -> None:
    """Direct-grad mode only: launch dW = dY^T X GEMMs (and bias column sums) on a side stream so they fill the bubbles of
    the latency-bound dX chain.  Every consumer of .grad (FusedAdamW.step, GradAllReducer) calls join_wgrad_streams()."""
    _state['wgrad_stream'] = bool(flag)


_side_streams = {}
_dirty_sides = set()          # side streams that received work since the last join


def side_stream_of(stream: 'torch.cuda.Stream') -> 'torch.cuda.Stream':
    key = (stream.device, stream.cuda_stream)
    s = _side_streams.get(key)
    if s is None:
        s = torch.cuda.Stream(device=stream.device)
        _side_streams[key] = s
    return s


def existing_side_stream_of(stream: 'torch.cuda.Stream') -> Optional['torch.cuda.Stream']:
    """The weight-gradient side stream of `stream` if one was ever created (else None; never creates one)."""
    return _side_streams.get((stream.device, stream.cuda_stream))


def all_stream_events(device=None) -> list:
    """Events marking "everything enqueued so far" on the current stream and on every weight-gradient side stream: a
    consumer that waits for all of them is ordered behind every gradient kernel of a finished backward pass (autograd itself
    joins the streams it ran nodes on with the stream that called backward(); the side streams are ours to join)."""
    evs = [torch.cuda.current_stream(device).record_event()]
    for s in _side_streams.values():
        evs.append(s.record_event())
    return evs


def mark_side_dirty(side: 'torch.cuda.Stream') -> None:
    _dirty_sides.add(side)


_side_keepalive = []


def keep_until_join(obj) -> None:
    """Hold tensors a side stream may still be reading until the compute stream has joined it (then the caching allocator
    may hand their memory out again: anything enqueued afterwards is ordered behind the side stream's work)."""
    _side_keepalive.append(obj)


def join_wgrad_streams() -> None:
    """Make the current stream wait for every weight-gradient side stream that was handed work since the last join.
    (Only those: inside a hipGraph capture a wait on a stream that is not part of the capture is an error.)"""
    if not _dirty_sides:
        _side_keepalive.clear()
        return
    cur = torch.cuda.current_stream()
    for s in list(_dirty_sides):
        cur.wait_stream(s)
    _dirty_sides.clear()
    _side_keepalive.clear()


# ---------------------------------------------------------------------------------------------
# hipGraph capture of a whole training step (graph.StepGraph).  The few values a step takes from the HOST every
# iteration -- the Dirichlet token budget of the mask sampler (CPU generator, multimae.py:185-189) and AdamW's
# step-dependent scalars -- enter the graph through static device tensors that are refreshed before every replay.
# ---------------------------------------------------------------------------------------------
class HostInputs:
    """(static device tensor, host thunk) pairs registered while a step is being captured.

    The device tensors are carved out of a slab allocated BEFORE the capture starts: a tensor allocated inside the
    capture comes from the graph's private pool, where it may alias a temporary that was freed earlier in the same step --
    whose kernels would then overwrite the refreshed value during every replay (seen: AdamW reading garbage scalars)."""

    def __init__(self, device=None, slab_bytes: int = 1 << 20):
        self.items = []          # [device tensor, thunk, pending host value]
        self.slab = torch.empty(slab_bytes, dtype=torch.uint8, device=device) if device is not None else None
        self.used = 0

    def _carve(self, shape, dtype, device) -> torch.Tensor:
        if self.slab is None:
            return torch.empty(shape, dtype=dtype, device=device)
        nbytes = torch.empty((), dtype=dtype).element_size()
        for d in shape:
            nbytes *= d
        off = (self.used + 255) // 256 * 256
        if off + nbytes > self.slab.numel():
            raise RuntimeError('HostInputs: static slab exhausted (raise slab_bytes)')
        self.used = off + nbytes
        return self.slab[off:off + nbytes].view(dtype).view(shape)

    def add(self, thunk, device) -> torch.Tensor:
        val = thunk()
        dev = self._carve(tuple(val.shape), val.dtype, device)
        self.items.append([dev, thunk, val])
        return dev

    def refresh(self) -> None:
        """Draw / compute this step's host values and enqueue their H2D copies (stream-ordered before the replay)."""
        for it in self.items:
            dev, thunk, pending = it
            val = pending if pending is not None else thunk()
            it[2] = None
            dev.copy_(val.pin_memory() if dev.is_cuda else val, non_blocking=True)


_capture: Optional[HostInputs] = None


def capturing() -> Optional[HostInputs]:
    return _capture


def host_input(thunk, device) -> torch.Tensor:
    """Per-step host-computed tensor on the device.  Eager: pinned + non-blocking H2D (a pageable copy would block the
    host until the stream drains, losing the whole launch lead).  While a step graph is being captured: a static device
    tensor that StepGraph refreshes from `thunk` before every replay."""
    if _capture is not None:
        return _capture.add(thunk, device)
    val = thunk()
    if val.device.type == 'cpu' and torch.device(device).type == 'cuda':
        return val.pin_memory().to(device, non_blocking=True)
    return val.to(device)


def fp32_adapter_gemm() -> str:
    return _state['fp32_adapter_gemm']


def set_fp32_adapter_gemm(mode: str) -> None:
    """How fp32_output_adapters run in the bf16 speed mode (residual stream, LayerNorm statistics, softmax, losses and every parameter
    gradient are f32 in all of them):
    'h16' (default, round 4): fp16 STORAGE (mmae.h MMAE_F16).  The reference runs these adapters with autocast off, i.e. f32 tensors whose
          matmul inputs the A100 rounded to TF32's 11-bit significand (torch 1.10: allow_tf32).  Here everything that is only ever a
          matmul / attention input -- LayerNorm outputs, q / k / v, attention output, the MLP's hidden activations, and the matching
          gradient tensors -- is STORED with that significand (IEEE half) instead of rounded on every read, so the adapter runs on the
          bf16 pipeline's kernels (ping-pong GEMMs with fused epilogues, the grouped weight-gradient launch, LDS-DMA attention) with half
          the HBM traffic: -1.5 ms of a 30.9 ms cfg3 step, per-tensor gradient parity unchanged (profiles/r04_h16_*).  Gradients are kept in
          units of S = 2^(4 - floor(log2 m)), m = max_b |d loss / d logit| as the cross-entropy kernel bounds it before writing (any other
          gradient source: its largest element), and multiplied by 1/S where they leave as f32.  Needs the composite adapter call and
          widths that are multiples of 32 (patch row: of 8); anything else runs as 'f16'.
    'f16': f32 tensors in memory, both operands of every Linear product rounded to fp16 -- an 11-bit significand, exactly TF32's, what the reference's fp32
          on their way into ONE MFMA per tile step, fp32 accumulation.  Gradient operands are
          pre-scaled by a power of two taken from the loss gradient's largest element (written by the masked-loss backward
          kernel), so fp16's exponent range is no limit; when no such amax exists (a loss on the image tensor instead of the
          adapter's patch rows) the gradient products run as 'x3'.  The attention cores stay 'x3'.
    'x3': f32 operands multiplied as split bf16 (a_hi.b_hi + a_hi.b_lo + a_lo.b_hi, fp32 accumulate; ~16 mantissa bits,
          above TF32), three MFMAs per tile step.
    'exact': f32-input MFMA (bit-level fmaf chain, 1/16 the bf16 rate)."""
    assert mode in ('h16', 'f16', 'x3', 'exact')
    _state['fp32_adapter_gemm'] = mode


def direct_grads() -> bool:
    return _state['direct_grads']


def set_direct_grads(flag: bool) -> None:
    """True: the hand-written backward accumulates weight gradients directly into the
    arena-backed ``p.grad`` (no autograd AccumulateGrad pass, no temporaries) and reports
    None to autograd.  Use with ParamArena + FusedAdamW / GradAllReducer (not with
    torch DDP, whose hooks need autograd-delivered gradients)."""
    _state['direct_grads'] = bool(flag)


class ParamArena:
    """Flat fp32 parameter / gradient arenas (+ optional bf16 shadow) for one module tree.

    ``groups``: name prefixes in the order the backward pass finishes their gradients (for MultiMAE: output adapters,
    encoder.L-1 ... encoder.0); parameters are laid out group by group, everything not named by a prefix last (the
    ``tail``: input adapters + global token, whose gradients are final only when backward ends).  A data-parallel reducer then
    sees contiguous ranges complete one after another (dist.GradAllReducer).  state_dict() order is unaffected."""

    def __init__(self, module: nn.Module, device: Optional[torch.device] = None, groups: Optional[List[str]] = None):
        params = [(n, p) for n, p in module.named_parameters()]
        if not params:
            raise ValueError('module has no parameters')
        # registration order (named_parameters()): what torch.optim state dicts index by -- the arena layout below is the
        # backward-readiness order and must never leak into a checkpoint (checkpoint.py iterates param_order)
        self.param_order: List[str] = [n for n, _ in params]
        self._tail: List[str] = []
        if groups:
            taken, ordered_all = set(), []
            for pre in groups:
                for n, p in params:
                    if n not in taken and (n == pre or n.startswith(pre)):
                        taken.add(n)
                        ordered_all.append((n, p))
            tail = [(n, p) for n, p in params if n not in taken]
            self._tail = [n for n, p in tail if p.requires_grad]
            params = ordered_all + tail
        device = device or params[0][1].device
        self.device = device
        self.names: List[str] = []
        self.offsets: Dict[str, int] = {}
        self.sizes: Dict[str, int] = {}
        self.trainable: Dict[str, bool] = {}
        # trainable tensors first (so optimiser / all-reduce touch one contiguous prefix)
        ordered = [(n, p) for n, p in params if p.requires_grad] + [(n, p) for n, p in params if not p.requires_grad]
        off = 0
        for n, p in ordered:
            self.names.append(n)
            self.offsets[n] = off
            self.sizes[n] = p.numel()
            self.trainable[n] = p.requires_grad
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            if p.requires_grad:
                self.n_trainable = off
        self.numel = off
        self.param = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.grad = torch.zeros(self.n_trainable, device=device, dtype=torch.float32)
        self.shadow: Optional[torch.Tensor] = None
        self._shadow_token = False
        self._scope = 0
        self._views: Dict[int, torch.Tensor] = {}
        self._shadow_views: Dict[int, torch.Tensor] = {}
        self._params: Dict[str, nn.Parameter] = {}
        self.zero_epoch = 0
        with torch.no_grad():
            for n, p in ordered:
                o, s = self.offsets[n], self.sizes[n]
                view = self.param[o:o + s].view(p.shape)
                view.copy_(p.data)
                p.data = view
                if p.requires_grad:
                    p.grad = self.grad[o:o + s].view(p.shape)
                    p._mmae_arena_ref, p._mmae_written = weakref.ref(self), -1
                self._params[n] = p
                self._views[id(p)] = view
        module._mmae_arena = self
        # load_state_dict() copies new values into the arena views: the bf16 shadow the optimiser vouched for is stale then
        if hasattr(module, 'register_load_state_dict_post_hook'):
            module.register_load_state_dict_post_hook(lambda m, incompatible: setattr(self, '_shadow_token', False))

    def tail_names(self) -> List[str]:
        """Trainable parameters laid out after every readiness group (their bucket is cut off on its own, dist.plan_buckets)."""
        return list(self._tail)

    # -- integrity -----------------------------------------------------------------
    def intact(self) -> bool:
        """False if someone re-materialised the parameters (e.g. module.to(...)) after the arena
        was built.  Called several times per step (arena_of): three sentinel parameters are compared every time -- a
        re-materialisation moves them all -- and the full walk over the 351 tensors runs on every 64th call."""
        base = self.param.data_ptr()
        self._intact_calls = getattr(self, '_intact_calls', 0) + 1
        if self._intact_calls % 64 != 1:
            sent = getattr(self, '_sentinels', None)
            if sent is None:
                items = list(self._params.items())
                sent = self._sentinels = [items[0], items[len(items) // 2], items[-1]] if items else []
            return all(p.data_ptr() == base + 4 * self.offsets[n] for n, p in sent)
        for n, p in self._params.items():
            if p.data_ptr() != base + 4 * self.offsets[n]:
                return False
        return True

    def rebind_grads(self) -> None:
        """(re)point every p.grad at the gradient arena (after zero_grad(set_to_none=True)).  The views are made once; a step
        whose gradients are still bound costs one identity check per parameter."""
        views = getattr(self, '_grad_views', None)
        if views is None or self._grad_views_base != self.grad.data_ptr():
            views = self._grad_views = {}
            self._grad_views_base = self.grad.data_ptr()
        for n, p in self._params.items():
            if not p.requires_grad:
                continue
            v = views.get(n)
            if v is None:                # the gradient arena covers the trainable tensors only
                v = views[n] = self.grad[self.offsets[n]:self.offsets[n] + self.sizes[n]].view(p.shape)
            if p.grad is not v:
                p.grad = v

    def zero_grad(self) -> None:
        self.grad.zero_()
        self.zero_epoch += 1              # every gradient of this arena is exactly zero again: see claim_first_write()

    # -- shadows -------------------------------------------------------------------
    def refresh_shadow(self) -> None:
        """bf16 copy of the whole parameter arena (one cast launch)."""
        if self._scope > 0 and self.shadow is not None:
            return                      # inside a model forward: refreshed once when the scope was entered
        if self.shadow is None:
            self.shadow = torch.empty(self.numel, device=self.device, dtype=torch.bfloat16)
        if self._shadow_token:          # produced by the fused optimiser for exactly these values
            self._shadow_token = False
            return
        ops.cast_into(self.param, self.shadow)

    def mark_shadow_fresh(self) -> None:
        self._shadow_token = True

    @contextlib.contextmanager
    def forward_scope(self):
        """One model forward: the shadow is brought up to date once on entry and every WeightCache created inside (embed,
        encoder stack, each output adapter) reuses it -- without the scope each of them re-cast the whole 98 M-element
        arena, 5 full passes per step."""
        if self._scope == 0:
            self.refresh_shadow()
        self._scope += 1
        try:
            yield
        finally:
            self._scope -= 1

    def weight(self, p: nn.Parameter, dtype: torch.dtype) -> torch.Tensor:
        """act-dtype view of parameter p (the f32 master itself in fp32 mode)."""
        if dtype == torch.float32:
            return p.data
        key = id(p)
        v = self._shadow_views.get(key)
        if v is None or v.data_ptr() != self.shadow.data_ptr() + 2 * self._off_of(p):
            o = self._off_of(p)
            v = self.shadow[o:o + p.numel()].view(p.shape)
            self._shadow_views[key] = v
        return v

    def _off_of(self, p: nn.Parameter) -> int:
        return (p.data_ptr() - self.param.data_ptr()) // 4


@contextlib.contextmanager
def forward_scope(module: nn.Module):
    """Shadow-refresh scope of one forward pass of `module` (no-op without an arena or outside bf16 mode)."""
    a = arena_of(module)
    if a is None or act_dtype() != torch.bfloat16:
        yield
        return
    with a.forward_scope():
        yield


def claim_first_write(params: Iterable[Optional[torch.Tensor]]) -> bool:
    """True when NONE of these (arena-bound, trainable) parameters has received a gradient since its arena's last ``zero_grad()`` --
    the composite backward call that is about to write them may then STORE its results instead of accumulating onto known zeros:
    the weight-gradient reduction drops its read of the destination (a quarter of its traffic, ~0.2 ms of a cfg3 step).  Marks the
    parameters written, so a second backward before the next ``zero_grad()`` (gradient accumulation) accumulates as before; one
    parameter already written (a tensor shared between two calls) keeps the whole call accumulating.  ``set_first_write_stores(False)``
    turns the shortcut off."""
    ps = [p for p in params if p is not None and p.requires_grad]
    if not _state.get('first_write_stores', True) or not ps:
        return False
    fresh = True
    for p in ps:
        ref = getattr(p, '_mmae_arena_ref', None)
        a = ref() if ref is not None else None
        if a is None or p._mmae_written == a.zero_epoch or a.zero_epoch == 0:
            fresh = False                                # (epoch 0: zero_grad() has not run yet -- whoever fills .grad by hand keeps accumulate semantics)
        if a is not None:
            p._mmae_written = a.zero_epoch
    return fresh


def mark_written(p: Optional[torch.Tensor]) -> None:
    """A per-kernel gradient writer (functions.GradSink: the dropout / drop-path fall-back paths, ``MlpFn``, ``block_bwd`` off the
    composite path) is about to ACCUMULATE into this parameter's arena-bound gradient: record it, so that a composite backward that
    touches the same parameter later in the same ``zero_grad()`` epoch accumulates too instead of storing over the contribution
    (``claim_first_write`` only saw marks that composites had set)."""
    if p is None:
        return
    ref = getattr(p, '_mmae_arena_ref', None)
    a = ref() if ref is not None else None
    if a is not None:
        p._mmae_written = a.zero_epoch


def set_first_write_stores(flag: bool) -> None:
    _state['first_write_stores'] = bool(flag)


def arena_of(module: nn.Module) -> Optional[ParamArena]:
    a = getattr(module, '_mmae_arena', None)
    if a is not None and not a.intact():
        return None
    return a


class WeightCache:
    """Per-forward provider of act-dtype weights.

    With an arena: one cast for everything.  Without (stand-alone sub-modules, tests):
    cast the individual parameter on demand and memoise for the duration of one forward."""

    def __init__(self, arena: Optional[ParamArena], dtype: torch.dtype):
        self.arena, self.dtype = arena, dtype
        self._memo: Dict[int, torch.Tensor] = {}
        if arena is not None and dtype == torch.bfloat16:
            arena.refresh_shadow()

    def __call__(self, p: torch.Tensor) -> torch.Tensor:
        if self.dtype == torch.float32:
            return p.detach()
        if self.arena is not None and id(p) in self.arena._views:
            return self.arena.weight(p, self.dtype)
        k = id(p)
        w = self._memo.get(k)
        if w is None:
            w = ops.cast(p.detach().contiguous(), self.dtype)
            self._memo[k] = w
        return w
