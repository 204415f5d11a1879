"""Hand-scheduled forward/backward of the MultiMAE hot path as module-level autograd Functions.

Each Function launches a fixed sequence of HIP kernels (see ops.py) for its forward and a
hand-written sequence for its backward -- torch autograd only links the few module-level
nodes together (embed -> encoder stack -> 4 output adapters -> losses), it never
differentiates through individual ops.  Activations are kept (never recomputed): at ViT-B,
B=256 they total ~8 GB of the 288 GB HBM.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import engine, ops
from .ops import AttnView, EPI_DGELU, EPI_GELU

Tensor = torch.Tensor


def _bound_grad(p: Tensor) -> Optional[Tensor]:
    """p.grad of a LEAF parameter (the arena view, in direct mode), else None.  A non-leaf tensor in a parameter slot -- e.g. the
    mask_token + task_embedding sum of the mask-token-query branch -- has no bound gradient, and reading .grad on it warns."""
    return p.grad if p.is_leaf else None


class GradSink:
    """Where parameter gradients go: straight into the arena-backed p.grad (direct mode, returns
    None to autograd) or into fresh tensors handed back to autograd."""

    def __init__(self, direct: bool):
        self.direct = direct
        self.side = None
        if direct and engine.wgrad_stream():
            self.main = torch.cuda.current_stream()
            self.side = engine.side_stream_of(self.main)

    def _on_side(self, fn, *tensors):
        """run fn() on the weight-gradient side stream, ordered after everything enqueued so far on the compute stream"""
        self.side.wait_stream(self.main)
        engine.mark_side_dirty(self.side)
        with torch.cuda.stream(self.side):
            fn()
        for t in tensors:
            t.record_stream(self.side)          # keep the operands alive until the side stream is done with them

    def _target(self, p: Tensor):
        g = _bound_grad(p) if self.direct else None
        if g is not None:
            engine.mark_written(p)              # a later composite backward of this zero_grad() epoch must accumulate, not store
            return g, True
        return torch.empty(p.shape, device=p.device, dtype=torch.float32), False

    def weight(self, p: Tensor, dy: Tensor, x: Tensor, x_off: int = 0, ldx: Optional[int] = None, K: Optional[int] = None):
        """dW[N,K] = dy[M,N]^T x[M,K]"""
        if not p.requires_grad:
            return None
        tgt, acc = self._target(p)
        M, N = dy.shape
        K = K if K is not None else x.shape[1]

        def run():
            ops.linear_dw(dy, x, tgt, acc, x_off=x_off, ldx=ldx, K=K)
        if self.side is not None and acc:
            self._on_side(run, dy, x)
        else:
            run()
        return None if acc else tgt

    def linear(self, w: Tensor, b: Optional[Tensor], dy: Tensor, x: Tensor):
        """(dW, db) of y = x W^T + b in ONE launch where the GEMM kernel can also sum dy's columns."""
        if b is None or not b.requires_grad or not w.requires_grad:
            return self.weight(w, dy, x), (self.bias(b, dy) if b is not None else None)
        wt, wacc = self._target(w)
        bt, bacc = self._target(b)

        def run():
            ops.linear_dw(dy, x, wt, wacc, db=bt.view(-1), db_accumulate=bacc)
        if self.side is not None and wacc and bacc:
            self._on_side(run, dy, x)
        else:
            run()
        return (None if wacc else wt), (None if bacc else bt)

    def bias(self, p: Tensor, dy: Tensor):
        if not p.requires_grad:
            return None
        tgt, acc = self._target(p)
        if self.side is not None and acc:
            self._on_side(lambda: ops.colsum(dy, tgt, acc), dy)
        else:
            ops.colsum(dy, tgt, acc)
        return None if acc else tgt

    def colsums(self, part: Tensor, seg_w: int, params: Sequence):
        """Column sums of part [rows, len(params) * seg_w] -> segment i is (added to) the gradient of params[i].
        An entry is a Parameter (seg_w == numel), a (Parameter, k) pair (slice k of a gradient of several segments) or None
        (dropped).  Direct mode: ONE reduction scatters straight into the arena-backed .grad views -- on the weight-gradient
        side stream when there is one, so the parameter-gradient reductions leave the dX critical path; returns None per
        entry.  Otherwise returns the gradient tensors (one per distinct Parameter, at its first entry)."""
        dsts: List[Optional[Tensor]] = []
        accs: List[bool] = []
        outs: List[Optional[Tensor]] = [None] * len(params)
        fresh = {}
        for i, e in enumerate(params):
            p, k = (e if isinstance(e, tuple) else (e, None))
            if p is None or not p.requires_grad:
                dsts.append(None)
                accs.append(False)
                continue
            if self.direct and _bound_grad(p) is not None:
                flat, acc = p.grad.view(-1), True
                engine.mark_written(p)
            else:
                acc = False
                flat = fresh.get(id(p))
                if flat is None:
                    flat = (torch.zeros if k is not None else torch.empty)((p.numel(),), device=part.device, dtype=torch.float32)
                    fresh[id(p)] = flat
                    outs[i] = flat.view(p.shape)
            accs.append(acc)
            dsts.append(flat if k is None else flat[k * seg_w:(k + 1) * seg_w])
        live = [a for a, d in zip(accs, dsts) if d is not None]
        if not live:
            return outs
        if all(live) or not any(live):
            acc = live[0]
            if self.side is not None and acc:
                self._on_side(lambda: ops.colsum_scatter(part, seg_w, dsts, True), part)
            else:
                ops.colsum_scatter(part, seg_w, dsts, acc)
            return outs
        # mixed bound / unbound gradients (some .grad views missing): reduce once, hand out per parameter
        tmp = torch.empty((part.shape[1],), device=part.device, dtype=torch.float32)
        ops.colsum(part, tmp, False)
        for i, (d, a) in enumerate(zip(dsts, accs)):
            if d is None:
                continue
            seg = tmp[i * seg_w:i * seg_w + d.numel()]
            if a:
                ops.axpy_(d, seg.contiguous(), 1.0)
            else:
                d.copy_(seg)
        return outs

    def vec(self, p: Tensor, g: Tensor):
        """g already holds the f32 gradient (any shape with p.numel() elements)."""
        if not p.requires_grad:
            return None
        if self.direct and _bound_grad(p) is not None:
            engine.mark_written(p)
            g = g.contiguous()
            if self.side is not None:       # same stream as the GEMM accumulations into the arena (no cross-stream races on .grad)
                self._on_side(lambda: ops.axpy_(p.grad, g, 1.0), g)
            else:
                ops.axpy_(p.grad, g, 1.0)
            return None
        return g.reshape(p.shape)


def _new(shape, like: Tensor, dtype):
    return torch.empty(shape, device=like.device, dtype=dtype)


# ------------------------------------------------------------------------------------------
# transformer block (multimae_utils.py:217-232) on 2-D [B*N, D] activations
# ------------------------------------------------------------------------------------------
BLOCK_PARAMS = ('norm1.weight', 'norm1.bias', 'attn.qkv.weight', 'attn.qkv.bias', 'attn.proj.weight', 'attn.proj.bias',
                'norm2.weight', 'norm2.bias', 'mlp.fc1.weight', 'mlp.fc1.bias', 'mlp.fc2.weight', 'mlp.fc2.bias')


def block_fwd(x: Tensor, P: Sequence[Tensor], wc, heads: int, eps: float, act, B: int, N: int, save: bool, dp=(None, None), drop=(0., 0.)):
    """dp = (s1, s2): per-sample stochastic-depth scales f32 [B] of the attention / MLP branch (None = branch kept).
    drop = (attn_drop, drop): the rates of Block(drop=, attn_drop=) in training mode (multimae_utils.py:217-227): nn.Dropout on the attention
    probabilities (:177), after the attention's proj (:181) and after fc2 (:154; the reference's Mlp has its dropout behind the activation
    commented out, :151-152) -- masks drawn in that (the reference's) order through ops._dropout_keep; the three sites leave the fused
    paths (materialised probabilities, un-fused residual adds)."""
    n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b = P
    R, D = x.shape
    hd = D // heads
    s1, s2 = dp
    p_attn, p_drop = drop
    dropping = p_attn > 0. or p_drop > 0.
    if s1 is None and s2 is None and not dropping and ops.block_composite_ok(x, act, heads, N):   # the whole block as one library call (same kernels)
        return ops.block_fwd_composite(x, P, (wc(qkvw), wc(projw), wc(fc1w), wc(fc2w)), heads, eps, act, B, N)
    inv = 1.0 / (1.0 - p_drop) if p_drop > 0. else 1.0
    ln1, mean1, rstd1 = ops.layernorm_fwd(x, n1w, n1b, eps, act)
    qkv = ops.linear_fwd(ln1, wc(qkvw), qkvb, _new((R, 3 * D), x, act))
    ao = _new((R, D), x, act)
    Pm = ops.attention_fwd(AttnView(qkv, 0, 3 * D, N), AttnView(qkv, D, 3 * D, N), AttnView(qkv, 2 * D, 3 * D, N),
                           AttnView(ao, 0, D, N), B, heads, hd, hd ** -0.5, drop_p=p_attn)
    k_proj = k_out = None
    if s1 is None and p_drop == 0.:
        x1 = ops.linear_fwd(ao, wc(projw), projb, _new((R, D), x, torch.float32), resid=x)
    else:                                                   # x1 = x + s1[b] * proj_drop(attn(..))   (multimae_utils.py:181, 229)
        y = ops.linear_fwd(ao, wc(projw), projb, _new((R, D), x, torch.float32))
        if p_drop > 0.:
            k_proj = ops._dropout_keep((R, D), p_drop, x.device)
            x1 = ops.dropout_apply(y, k_proj, inv, resid=x, row_scale=s1, per=N * D)
        else:
            x1 = ops.rowscale_add(x, y, s1, N)
    ln2, mean2, rstd2 = ops.layernorm_fwd(x1, n2w, n2b, eps, act)
    Hd = fc1w.shape[0]
    hpre = _new((R, Hd), x, act)
    hact = ops.linear_fwd(ln2, wc(fc1w), fc1b, _new((R, Hd), x, act), aux=hpre, epi=EPI_GELU)
    if s2 is None and p_drop == 0.:
        x2 = ops.linear_fwd(hact, wc(fc2w), fc2b, _new((R, D), x, torch.float32), resid=x1)
    else:
        y = ops.linear_fwd(hact, wc(fc2w), fc2b, _new((R, D), x, torch.float32))
        if p_drop > 0.:                                     # Mlp.drop behind fc2 (:154)
            k_out = ops._dropout_keep((R, D), p_drop, x.device)
            x2 = ops.dropout_apply(y, k_out, inv, resid=x1, row_scale=s2, per=N * D)
        else:
            x2 = ops.rowscale_add(x1, y, s2, N)
    saved = (x, ln1, mean1, rstd1, qkv, Pm, ao, x1, ln2, mean2, rstd2, hpre, hact) if save else None
    if save and dropping:
        saved = saved + ((k_proj, k_out, inv),)
    return x2, saved


def block_bwd(dx: Tensor, dx_act: Tensor, fc2b_done: bool, saved, P: Sequence[Tensor], wc, sink: GradSink, heads: int, act,
              B: int, N: int, cs_param: Optional[Tensor] = None, dp=(None, None)):
    """dx f32 [R,D] (+ its act-dtype copy) -> (dx0, dx0_act, g_cs, 12 parameter grads).
    fc2b_done: the bias gradient of fc2 (column sums of dx) was already delivered by the producer of dx.
    cs_param: the parameter whose gradient is colsum(dx0) -- the bias of the Linear that produced this block's input
    (the previous block's fc2); its gradient rides along with the LayerNorm-1 reduction and comes back as g_cs."""
    n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b = P
    k_proj = k_out = None
    inv = 1.0
    if len(saved) == 14:                                    # a forward with dropout: the keep masks of its two element-wise sites
        k_proj, k_out, inv = saved[13]
        saved = saved[:13]
    x0, ln1, mean1, rstd1, qkv, Pm, ao, x1, ln2, mean2, rstd2, hpre, hact = saved
    R, D = x0.shape
    hd = D // heads
    Hd = fc1w.shape[0]
    lnact = None if act == torch.float32 else act
    s1, s2 = dp
    dropping = k_proj is not None or len(Pm) > 3
    if s1 is None and s2 is None and not dropping and Pm[0] == 'fused' and ops.block_composite_ok(x0, act, heads, N):
        out = _block_bwd_composite(dx, dx_act, fc2b_done, saved, P, wc, sink, heads, act, B, N, cs_param)
        if out is not None:
            return out
    # MLP: x2 = x1 + s2[b] * mlp(norm2(x1)); the branch sees the per-sample rescaled gradient
    if k_out is not None:                                   # the gradient through fc2's dropout (and the branch's stochastic-depth scale)
        dm_act = ops.dropout_apply(dx, k_out, inv, out_dtype=act, row_scale=s2, per=N * D)
    else:
        dm_act = dx_act if s2 is None else ops.rowscale_cast(dx, s2, N, act)
    assert not ((s2 is not None or k_out is not None) and fc2b_done)
    part_h = _new(ops.dx_colsum_part_shape(R, Hd), dx, torch.float32) if fc1b.requires_grad else None
    # a composite forward with bf16 activations left GELU'(pre-activation) in hpre: multiply, do not differentiate again
    epi_h = ops.EPI_MUL if (Pm[0] == 'fused' and len(Pm) > 2 and Pm[2]) else EPI_DGELU
    d_hpre = ops.linear_dx(dm_act, wc(fc2w), _new(hpre.shape, dx, act), aux=hpre, epi=epi_h, colsum_part=part_h)
    if fc2b_done:
        g_fc2w, g_fc2b = sink.weight(fc2w, dm_act, hact), None
    else:
        g_fc2w, g_fc2b = sink.linear(fc2w, fc2b, dm_act, hact)
    d_ln2 = ops.linear_dx(d_hpre, wc(fc1w), _new((R, D), dx, act))
    g_fc1w = sink.weight(fc1w, d_hpre, ln2)
    g_fc1b = sink.colsums(part_h, Hd, [fc1b])[0] if part_h is not None else None
    dx1, dx1_act, part2 = ops.layernorm_bwd_part(d_ln2, x1, n2w, mean2, rstd2, dx, lnact)
    if dx1_act is None:
        dx1_act = dx1
    # attention: x1 = x0 + s1[b] * attn(norm1(x0))
    if s1 is None and k_proj is None:
        g_n2w, g_n2b, g_projb = sink.colsums(part2, D, [n2w, n2b, projb])
        da_act = dx1_act
        g_projw = None
    else:
        g_n2w, g_n2b, _ = sink.colsums(part2, D, [n2w, n2b, None])
        da_act = ops.dropout_apply(dx1, k_proj, inv, out_dtype=act, row_scale=s1, per=N * D) if k_proj is not None else ops.rowscale_cast(dx1, s1, N, act)
    d_ao = ops.linear_dx(da_act, wc(projw), _new((R, D), dx, act))
    if s1 is None and k_proj is None:
        g_projw = sink.weight(projw, da_act, ao)
    else:
        g_projw, g_projb = sink.linear(projw, projb, da_act, ao)
    d_qkv = _new((R, 3 * D), dx, act)
    ops.attention_bwd(AttnView(qkv, 0, 3 * D, N), AttnView(qkv, D, 3 * D, N), AttnView(qkv, 2 * D, 3 * D, N), Pm,
                      AttnView(ao, 0, D, N), AttnView(d_ao, 0, D, N), AttnView(d_qkv, 0, 3 * D, N), AttnView(d_qkv, D, 3 * D, N),
                      AttnView(d_qkv, 2 * D, 3 * D, N), B, heads, hd, hd ** -0.5)
    d_ln1 = ops.linear_dx(d_qkv, wc(qkvw), _new((R, D), dx, act))
    g_qkvw, g_qkvb = sink.linear(qkvw, qkvb, d_qkv, ln1)
    dx0, dx0_act, part1 = ops.layernorm_bwd_part(d_ln1, x0, n1w, mean1, rstd1, dx1, lnact)
    if dx0_act is None:
        dx0_act = dx0
    g_n1w, g_n1b, g_cs = sink.colsums(part1, D, [n1w, n1b, cs_param])
    grads = (g_n1w, g_n1b, g_qkvw, g_qkvb, g_projw, g_projb, g_n2w, g_n2b, g_fc1w, g_fc1b, g_fc2w, g_fc2b)
    return dx0, dx0_act, g_cs, grads


def _block_bwd_composite(dx, dx_act, fc2b_done, saved, P, wc, sink: GradSink, heads, act, B, N, cs_param):
    """block_bwd through ONE mmae_block_bwd call.  A mix of bound and unbound .grad views gets fresh tensors for ALL parameters
    (autograd accumulates them), as _grad_targets does: after a composite forward the per-op path must never run -- with bf16
    activations that forward left GELU'(pre-activation) in `hpre`, which the per-op dGELU epilogue would differentiate again
    (ADVICE r3)."""
    n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b = P
    dsts, acc = _grad_targets(sink, list(P) + [cs_param])
    if acc:                                             # accumulating straight into the arena: later composites of this epoch must accumulate too
        for p in list(P) + [cs_param]:
            if p is not None and p.requires_grad:
                engine.mark_written(p)
    use_side = sink.side is not None and acc
    dx0, dx0_act, keep = ops.block_bwd_composite(dx, dx_act, fc2b_done, saved, P, (wc(qkvw), wc(projw), wc(fc1w), wc(fc2w)), dsts[:12],
                                                 dsts[12], acc, heads, act, B, N, sink.side.cuda_stream if use_side else None)
    if use_side:
        engine.mark_side_dirty(sink.side)
        engine.keep_until_join(keep)
    if acc:
        return dx0, dx0_act, None, (None,) * 12
    if fc2b_done:
        dsts[11] = None
    return dx0, dx0_act, dsts[12], tuple(dsts[:12])


class _Cfg:
    """bag of static (non-tensor) arguments for a Function call"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _grad_targets(sink: GradSink, params: Sequence[Optional[Tensor]]):
    """Gradient destinations for a composite backward call: every wanted gradient goes either into the arena-backed .grad
    (accumulate) or into a fresh tensor handed to autograd.  A mix of bound and unbound parameters falls back to fresh tensors
    for all of them (autograd then accumulates), so one flag describes the whole call."""
    live = [p for p in params if p is not None and p.requires_grad]
    direct = sink.direct and all(_bound_grad(p) is not None for p in live)
    dsts: List[Optional[Tensor]] = []
    for p in params:
        if p is None or not p.requires_grad:
            dsts.append(None)
        elif direct:
            dsts.append(p.grad)
        else:
            dsts.append(torch.empty(p.shape, device=p.device, dtype=torch.float32))
    return dsts, direct


class EncoderStackFn(torch.autograd.Function):
    """L transformer blocks in one autograd node.  forward(cfg, x[B,N,D] f32, *params(12 per layer))

    cfg: act, wc, heads, eps, all_layers, on_layer_done; optional dp (2 L per-sample stochastic-depth scale tensors or None
    entries), bwd_chunk (layers per backward library call when a gradient reducer wants per-layer progress)."""

    @staticmethod
    def forward(ctx, cfg: _Cfg, x: Tensor, *params: Tensor):
        ops._require_gpu(x, 'encoder input')
        B, N, D = x.shape
        L = len(params) // 12
        save = any(ctx.needs_input_grad)
        h = x.contiguous().view(B * N, D)
        dp = getattr(cfg, 'dp', None)
        drops = getattr(cfg, 'drops', None)                 # per block (attn_drop, drop) in training mode, or None: nn.Dropout sites (round 5)
        ctx.cfg, ctx.params, ctx.shape = cfg, params, (B, N, D)
        ctx.stack, ctx.saved = None, None
        if drops is None and ops.stack_composites() and L <= 64 and ops.block_composite_ok(h, cfg.act, cfg.heads, N):
            outs, state = ops.stack_fwd(h, params, cfg.wc, cfg.heads, cfg.eps, cfg.act, B, N, dp, mx=bool(getattr(cfg, 'mx', False)))
            ctx.stack = state if save else None
            if cfg.all_layers:
                return tuple(o.view(B, N, D) for o in outs)
            return outs[-1].view(B, N, D)
        saved, outs = [], []
        for l in range(L):
            dpl = (None, None) if dp is None else (dp[2 * l], dp[2 * l + 1])
            h, s = block_fwd(h, params[12 * l:12 * l + 12], cfg.wc, cfg.heads, cfg.eps, cfg.act, B, N, save, dpl,
                             drop=(0., 0.) if drops is None else drops[l])
            saved.append(s)
            if cfg.all_layers:
                outs.append(h.view(B, N, D))
        ctx.saved = saved
        if cfg.all_layers:
            return tuple(outs)
        return h.view(B, N, D)

    @staticmethod
    def backward(ctx, *douts: Tensor):
        cfg, params = ctx.cfg, ctx.params
        B, N, D = ctx.shape
        L = len(params) // 12
        sink = GradSink(engine.direct_grads())
        dp = getattr(cfg, 'dp', None)
        if ctx.stack is not None:
            d_outs: List[Optional[Tensor]] = [None] * L
            for l in range(L):
                dl = douts[l] if cfg.all_layers else (douts[0] if l == L - 1 else None)
                if dl is not None:
                    d_outs[l] = dl.contiguous().view(B * N, D)
            if d_outs[L - 1] is None:                       # nothing reached the last block: zero gradient there
                d_outs[L - 1] = torch.zeros((B * N, D), device=params[0].device, dtype=torch.float32)
            dsts, acc = _grad_targets(sink, params)
            use_side = sink.side is not None and acc
            # first write since zero_grad(): the library stores instead of accumulating onto zeros (engine.claim_first_write)
            lib_acc = acc and not engine.claim_first_write(params)
            chunks, on_chunk = None, None
            # readiness reports only when the gradients really are final in the arena at that point: with fresh tensors handed to
            # autograd (acc False) AccumulateGrad writes them later, and a bucket reduced now would miss them
            if cfg.on_layer_done is not None and acc:
                c = max(1, int(getattr(cfg, 'bwd_chunk', 1) or 1))
                chunks = [(max(0, hi - c), hi) for hi in range(L, 0, -c)]

                def on_chunk(lo, hi):
                    for l in reversed(range(lo, hi)):
                        cfg.on_layer_done(l)
            side_cus = engine.enc_bwd_side_cus() if use_side else 0       # A/B switch: dX chain and weight gradients side by side
            prev_side = ops.gemm_side_cus(side_cus) if side_cus > 0 else 0
            try:
                dx, keep = ops.stack_bwd(ctx.stack, d_outs, dsts, lib_acc, sink.side.cuda_stream if use_side else None, chunks, on_chunk)
            finally:
                if side_cus > 0:
                    ops.gemm_side_cus(prev_side)
            ctx.stack = None
            if use_side:
                engine.mark_side_dirty(sink.side)
                engine.keep_until_join(keep)
            grads = [None] * (12 * L) if acc else dsts
            return (None, dx.view(B, N, D), *grads)
        grads: List[Optional[Tensor]] = [None] * (12 * L)
        dx, dx_act = None, None
        fc2b_done, g_cs = False, None
        for l in reversed(range(L)):
            dl = douts[l] if cfg.all_layers else (douts[0] if l == L - 1 else None)
            if dl is not None:
                dl = dl.contiguous().view(B * N, D)
                dx = dl if dx is None else ops.axpy_(dx, dl, 1.0)
                dx_act = None
            if dx is None:
                continue
            if dx_act is None:
                dx_act = ops.cast(dx, cfg.act)
            # the bias gradient of block l-1's fc2 (column sums of this block's input gradient) rides along with this
            # block's LayerNorm-1 reduction -- unless more gradient is added to dx before block l-1 sees it (all_layers) or
            # block l-1's MLP branch was rescaled per sample (stochastic depth)
            below_scaled = dp is not None and l > 0 and dp[2 * (l - 1) + 1] is not None
            drops = getattr(cfg, 'drops', None)
            below_dropped = drops is not None and l > 0 and drops[l - 1][1] > 0.       # block l-1's fc2 output went through nn.Dropout
            cs_param = params[12 * (l - 1) + 11] if (l > 0 and not cfg.all_layers and not below_scaled and not below_dropped) else None
            dpl = (None, None) if dp is None else (dp[2 * l], dp[2 * l + 1])
            dx, dx_act, g_cs_next, g = block_bwd(dx, dx_act, fc2b_done, ctx.saved[l], params[12 * l:12 * l + 12], cfg.wc, sink,
                                                 cfg.heads, cfg.act, B, N, cs_param, dpl)
            grads[12 * l:12 * l + 12] = g
            if fc2b_done:
                grads[12 * l + 11] = g_cs
            fc2b_done, g_cs = cs_param is not None and cs_param.requires_grad, g_cs_next
            ctx.saved[l] = None
            if cfg.on_layer_done is not None and all(x is None for x in grads[12 * l:12 * l + 12]):
                cfg.on_layer_done(l)                         # every gradient of the layer went straight into the arena
        return (None, dx.view(B, N, D), *grads)


# ------------------------------------------------------------------------------------------
# gather-first patch embedding (input adapters + token select + global tokens)
# ------------------------------------------------------------------------------------------
class AvgClassEmbFn(torch.autograd.Function):
    """SemSegInputAdapter(interpolate_class_emb=True), input_adapters.py:192-198 + :223: class ids [B, H, W] -> the class-embedding
    image resized by 1 / patch, f32 [B, E, nh, nw] (fed to the patch embedding as a 1 x 1-patch image)."""

    @staticmethod
    def forward(ctx, x: Tensor, class_emb: Tensor, ph: int, pw: int, pad_idx: int):
        x = x.contiguous()
        ctx.x, ctx.emb, ctx.geom = x, class_emb, (ph, pw, pad_idx)
        return ops.semseg_avg_emb_fwd(x, class_emb.detach().contiguous(), ph, pw)

    @staticmethod
    def backward(ctx, d_img: Tensor):
        ph, pw, pad_idx = ctx.geom
        sink = GradSink(engine.direct_grads())
        buf = ops.semseg_avg_emb_bwd(d_img.contiguous(), ctx.x, ctx.emb.shape[0], ph, pw, pad_idx)
        return None, sink.vec(ctx.emb, buf), None, None, None


class EmbedFn(torch.autograd.Function):
    """forward(cfg, sel[B,n_sel] int64, global_tokens|None, *per task: data, proj.weight, proj.bias, class_emb|None, pos)

    pos: the task's position table [n_patches, D] f32 (the adapter's pos_emb resized to the token grid); it receives a gradient
    only when the adapter was built with learnable_pos_emb=True.

    cfg.tasks: list of dicts (kind, C, H, W, ph, pw, k_off, K, n_patches, pos [n_patches, D] f32)
    cfg.task_offsets, cfg.D, cfg.G, cfg.wc, cfg.act
    """

    @staticmethod
    def forward(ctx, cfg: _Cfg, sel: Tensor, global_tokens: Optional[Tensor], *tens: Tensor):
        T = len(cfg.tasks)
        B, n_sel = sel.shape
        D, G, act, wc = cfg.D, cfg.G, cfg.act, cfg.wc
        data, ws, bs, embs, poss = tens[0::5], tens[1::5], tens[2::5], tens[3::5], tens[4::5]
        Ktot = sum(t['K'] for t in cfg.tasks)
        if act == torch.bfloat16 and any(t['K'] % 8 for t in cfg.tasks):
            raise NotImplementedError('bf16 patch embedding needs C*P_H*P_W to be a multiple of 8 for every modality')
        srcs = []
        for t, d, e in zip(cfg.tasks, data, embs):
            ops._require_gpu(d, 'input tensor')
            srcs.append(dict(data=d.contiguous(), emb=e, kind=t['kind'], C=t['C'], H=t['H'], W=t['W'], ph=t['ph'], pw=t['pw'],
                             k_off=t['k_off']))
        sel = sel.contiguous()
        gt = global_tokens.detach().reshape(G, D) if G > 0 else None
        if act == torch.bfloat16 and D <= 768 and ops.patch_embed_supported(srcs, n_sel, D):   # (ViT-L width, 196 kept tokens: measured slower than the three passes)
            # ONE kernel (csrc/embed.hip): gather of the kept patches, bf16 MFMA against the task's projection, + bias + pos-emb, token
            # rows out; the zero-padded bf16 patch rows the weight-gradient products contract over are written on the side
            tok, rows = ops.patch_embed_fwd(srcs, [wc(w).view(D, t['K']) for t, w in zip(cfg.tasks, ws)], [b.detach() for b in bs],
                                            [p.detach() for p in poss], cfg.task_offsets, sel, gt, B, n_sel, G, D, Ktot,
                                            want_rows=any(ctx.needs_input_grad))
        else:
            rows = ops.patch_rows(srcs, cfg.task_offsets, sel, B, n_sel, Ktot, act)
            proj = torch.empty((B * n_sel, D), device=sel.device, dtype=torch.float32)
            for i, (t, w) in enumerate(zip(cfg.tasks, ws)):
                ops.gemm(rows, wc(w).view(D, t['K']), proj, B * n_sel, D, t['K'], lda=Ktot, ldb=t['K'], ldc=D, a_off=t['k_off'],
                         accumulate=(i > 0))
            tok = ops.tokens_assemble(proj, [b.detach() for b in bs], [p.detach() for p in poss], cfg.task_offsets, sel, gt,
                                      B, n_sel, G, D)
        ctx.cfg, ctx.sel, ctx.rows = cfg, sel, rows
        ctx.tens, ctx.gt = tens, global_tokens
        ctx.srcs = srcs
        return tok

    @staticmethod
    def backward(ctx, d_tok: Tensor):
        cfg, sel, rows, tens = ctx.cfg, ctx.sel, ctx.rows, ctx.tens
        T = len(cfg.tasks)
        B, n_sel = sel.shape
        D, G, act, wc = cfg.D, cfg.G, cfg.act, cfg.wc
        Ktot = rows.shape[1]
        sink = GradSink(engine.direct_grads())
        d_proj, part = ops.tokens_assemble_bwd(d_tok.contiguous(), cfg.task_offsets, sel, B, n_sel, G, D, act, raw=True)
        # one reduction of the per-workgroup partials feeds every projection bias and the global tokens
        gt_p = ctx.gt if (G > 0 and ctx.gt is not None) else None
        g_small = sink.colsums(part, D, [tens[5 * i + 2] for i in range(T)] + [(gt_p, g) for g in range(G)])
        d_pos = None
        if any(tens[5 * i + 4].requires_grad for i in range(T)):          # learnable positional embeddings
            d_pos = ops.pos_emb_bwd(d_tok.contiguous(), sel, cfg.task_offsets[-1], B, n_sel, G, D)
        out: List[Optional[Tensor]] = []
        for i, t in enumerate(cfg.tasks):
            w, b, emb = tens[5 * i + 1], tens[5 * i + 2], tens[5 * i + 3]
            gw = sink.weight(w, d_proj, rows, x_off=t['k_off'], ldx=Ktot, K=t['K'])
            if gw is not None:
                gw = gw.view(w.shape)
            gb = g_small[i]
            ge = None
            gdata = None
            if t['kind'] == 0 and tens[5 * i].requires_grad:        # an image-like input that wants its gradient (e.g. AvgClassEmbFn's output)
                d_rows = ops.linear_dx(d_proj, wc(w).view(D, t['K']), torch.empty((B * n_sel, t['K']), device=sel.device, dtype=torch.float32))
                gdata = ops.rows_to_image(d_rows, 0, sel, B, n_sel, cfg.task_offsets[i], t['n_patches'], t['C'], t['H'], t['W'], t['ph'], t['pw'])
            if emb is not None and emb.requires_grad:
                d_rows = ops.linear_dx(d_proj, wc(w).view(D, t['K']), torch.empty((B * n_sel, t['K']), device=sel.device, dtype=act))
                ge_buf = torch.empty(emb.shape, device=sel.device, dtype=torch.float32)      # stored, not accumulated: no zero-fill launch
                s = ctx.srcs[i]
                ops.semseg_emb_bwd(d_rows, s['data'], sel, ge_buf, B=B, H=t['H'], W=t['W'], E=t['C'], ph=t['ph'], pw=t['pw'],
                                   n_sel=n_sel, k_off=0, tok_off=cfg.task_offsets[i], n_patches=t['n_patches'], n_cls=emb.shape[0], accumulate=False)
                if t.get('pad_idx') is not None:              # nn.Embedding(padding_idx=...): that row receives no gradient
                    ge_buf[t['pad_idx']].zero_()
                ge = sink.vec(emb, ge_buf)
            gp = d_pos[cfg.task_offsets[i]:cfg.task_offsets[i + 1]] if (d_pos is not None and tens[5 * i + 4].requires_grad) else None
            out += [gdata, gw, gb, ge, gp]
        g_glob = g_small[T] if G > 0 else None
        # The embedding parameters are reported ready only if nothing is still on its way through autograd: a learnable pos_emb's
        # gradient (gp) reaches pos_emb.grad through F.interpolate's backward + AccumulateGrad, interpolate_class_emb's class
        # table through AvgClassEmbFn.backward (gdata), and without direct gradients everything does.  Otherwise the reducer's
        # finish() takes the tail bucket after backward has ended (the early report raced with those later nodes).
        late = g_glob is not None or any(x is not None for x in out)
        if getattr(cfg, 'on_done', None) is not None and not late:
            cfg.on_done()
        return (None, None, g_glob, *out)


# ------------------------------------------------------------------------------------------
# SpatialOutputAdapter (output_adapters.py:236-282)
# ------------------------------------------------------------------------------------------
def _lin_fwd(x, w, b, wc, out_dtype, **kw):
    return ops.linear_fwd(x, wc(w), b, torch.empty((x.shape[0], w.shape[0]), device=x.device, dtype=out_dtype), **kw)


class PatHandle:
    """Side channel between an output adapter and a masked loss evaluated on its prediction (patch-domain loss): the adapter's
    patch rows `pat` (f32 [B*n_q, C*ph*pw], out_proj's output before the image rearrangement), the geometry, the autograd
    `token` that links the loss node to the adapter node, and -- filled in by the loss's backward -- the gradient rows `d_pat`
    in the adapter's activation dtype."""
    __slots__ = ('pat', 'token', 'C', 'nh', 'nw', 'ph', 'pw', 'act', 'd_pat', 'dy_amax')

    def __init__(self, pat, C, nh, nw, ph, pw, act):
        self.pat, self.C, self.nh, self.nw, self.ph, self.pw, self.act = pat, C, nh, nw, ph, pw, act
        self.token, self.d_pat = None, None
        # f32 adapter with fp16-operand products (engine.set_fp32_adapter_gemm('f16')): a zeroed device scalar the loss's backward
        # kernel raises to the largest |element| of d_pat; the adapter's backward scales its gradient operands by its power of two
        self.dy_amax = None

    def matches(self, img: Tensor, patch: int) -> bool:
        return (self.pat is not None and self.token is not None and self.ph == patch and self.pw == patch
                and tuple(img.shape) == (self.pat.shape[0] // (self.nh * self.nw), self.C, self.nh * self.ph, self.nw * self.pw))

    def ld(self) -> int:
        return ops.round_up(self.C * self.ph * self.pw, 8)


import contextlib as _contextlib


@_contextlib.contextmanager
def _adapter_cu_share():
    """Experiment switch (engine.set_adapter_cu_share(k), default 0 = off): while the output adapters run on their own streams, launch
    their persistent GEMM grids n_cu - k wide, so that the kernels of two adapters are resident side by side (one's store-bound
    epilogue under the other's K loop) instead of time-slicing full-chip grids."""
    k = engine.adapter_cu_share() if engine.adapter_streams() else 0
    if k <= 0:
        yield
        return
    prev = ops.gemm_cu_share(k)                      # its own slot: the gradient reducer's reserve is untouched
    try:
        yield
    finally:
        ops.gemm_cu_share(prev)


def _f32_mode(cfg) -> str:
    """ops.f32_gemm_mode of an adapter call: 'h16' (fp16 storage) is a property of the composite call; whatever runs outside it (a geometry
    the composite does not take, single launches) uses the fp16-operand products on f32 tensors"""
    m = getattr(cfg, 'f32_gemm', 'exact')
    return 'f16' if m == 'h16' else m


class SpatialAdapterFn(torch.autograd.Function):
    """forward(cfg, enc[B,NC,Denc] f32, ids_keep, ids_restore, *params)

    params order: mask_token, task_emb_0..T-1 (one per INPUT task, None if absent), then
      q.w,q.b, kv.w,kv.b, proj.w,proj.b, ctxn.w,ctxn.b, qn.w,qn.b, outn.w,outn.b,
      fc1.w,fc1.b, fc2.w,fc2.b, [12 per transformer block]*depth, out_proj.w,out_proj.b,
      proj_context.w, proj_context.b
    cfg: act, wc, heads, eps, task_offsets, q_task, G, D, pos [n_q, D], depth, C, nh, nw, ph, pw
    """

    @staticmethod
    def forward(ctx, cfg: _Cfg, enc: Tensor, ids_keep: Tensor, ids_restore: Tensor, *params):
        """returns (img, token): token is a 1-element tensor whose only purpose is to connect a patch-domain loss node to this
        node in the autograd graph (see PatHandle); cfg.handle is set when the patch rows are available."""
        ctx.set_materialize_grads(False)
        cfg.handle = None
        cfg.lazy_fill = None
        with ops.f32_gemm_mode(_f32_mode(cfg)), _adapter_cu_share():
            img = SpatialAdapterFn._forward(ctx, cfg, enc, ids_keep, ids_restore, *params)
        token = img.new_empty(1)
        if cfg.handle is not None:
            cfg.handle.token = token
        ctx.handle = cfg.handle
        return img, token

    @staticmethod
    def backward(ctx, d_img: Optional[Tensor], d_token: Optional[Tensor] = None):
        with ops.f32_gemm_mode(_f32_mode(ctx.cfg)), _adapter_cu_share():
            return SpatialAdapterFn._backward(ctx, d_img)

    @staticmethod
    def _forward(ctx, cfg: _Cfg, enc: Tensor, ids_keep: Tensor, ids_restore: Tensor, *params):
        ops._require_gpu(enc, 'encoder tokens')
        act, wc, D, G, heads, eps = cfg.act, cfg.wc, cfg.D, cfg.G, cfg.heads, cfg.eps
        T = len(cfg.task_offsets) - 1
        B, NC, Denc = enc.shape
        n_keep = NC - G
        n_q = cfg.nh * cfg.nw if cfg.q_task < 0 else cfg.task_offsets[cfg.q_task + 1] - cfg.task_offsets[cfg.q_task]
        mask_token = params[0]
        temb = params[1:1 + T]
        (qw, qb, kvw, kvb, pw_, pb, cnw, cnb, qnw, qnb, onw, onb, f1w, f1b, f2w, f2b) = params[1 + T:17 + T]
        blocks = params[17 + T:17 + T + 12 * cfg.depth]
        ow, ob, pcw, pcb = params[17 + T + 12 * cfg.depth:]
        dev = enc.device
        save = any(ctx.needs_input_grad)

        enc2 = enc.contiguous().view(B * NC, Denc)
        enc_act = getattr(cfg, 'enc_act', None)      # act-dtype copy shared by all adapters of one forward (MultiMAE.forward)
        if enc_act is None or enc_act.dtype != act or enc_act.shape != enc2.shape:
            enc_act = None
        ctx.comp = None
        KP = cfg.C * cfg.ph * cfg.pw
        xattn = bool(getattr(cfg, 'use_xattn', True))
        # fp32 adapter, engine.set_fp32_adapter_gemm('h16'): activations / saved tensors / gradients of the composite call stored as fp16
        # (mmae.h MMAE_F16) -- the bf16 pipeline's kernels and traffic with TF32's significand; residual stream, statistics, losses f32
        h16 = (act == torch.float32 and getattr(cfg, 'f32_gemm', 'exact') == 'h16' and xattn
               and ops.adapter_h16_ok(enc2, heads, D, f1w.shape[0], n_q, NC, cfg.depth, T, KP))
        ctx.h16 = h16
        # stochastic depth in decoder_transformer (SpatialOutputAdapter(drop_path_rate > 0), training): per-sample scales [(attention
        # branch, MLP branch)] * depth; the one-call adapter has no such option, the per-block sequence below folds them into the residual adds
        dps = getattr(cfg, 'dp', None)
        dp_of = (lambda l: (dps[2 * l], dps[2 * l + 1])) if dps is not None else (lambda l: (None, None))
        # nn.Dropout sites (SpatialOutputAdapter(drop_rate / attn_drop_rate > 0), training): cfg.drops = (attn_drop, drop) -- the cross
        # attention's probabilities and proj output (output_adapters.py:118-119), the three sites of every decoder block (:131-132)
        drops = getattr(cfg, 'drops', None)
        p_attn, p_drop = drops if drops is not None else (0., 0.)
        if dps is not None or drops is not None:
            h16 = ctx.h16 = False
        if xattn and dps is None and drops is None and (h16 or ops.adapter_composite_ok(enc2, act, heads, D, n_q, NC, cfg.depth, T, KP)):
            # the whole adapter as ONE library call (same kernels, same order)
            w_list = [wc(qw), wc(kvw), wc(pw_), wc(f1w), wc(f2w)] + [wc(blocks[12 * l + i]) for l in range(cfg.depth) for i in (2, 4, 8, 10)] \
                + [wc(ow), wc(pcw)]
            store = None
            if h16:
                store, enc_act = torch.float16, None
                w_list = list(ops.h16_weights([w.detach() for w in w_list]).refresh())
            p_list = [qb, kvb, pb, cnw, cnb, qnw, qnb, onw, onb, f1b, f2b] \
                + [blocks[12 * l + i] for l in range(cfg.depth) for i in (0, 1, 3, 5, 6, 7, 9, 11)] + [ob, pcb]
            # the image is an API output nothing on the training path reads (the losses work on the patch rows): allocate it, write
            # it on first use (lazy.LazyPrediction, wrapped around the result by SpatialOutputAdapter.forward)
            lazy_img = engine.lazy_predictions() and engine.capturing() is None
            img, state = ops.adapter_fwd(enc.contiguous(), enc_act, ids_keep, ids_restore, cfg, w_list, p_list, mask_token.detach().reshape(D),
                                         [None if t is None else t.detach().reshape(D) for t in temb], want_img=not lazy_img, store=store)
            if lazy_img:
                img = torch.empty((B, cfg.C, cfg.nh * cfg.ph, cfg.nw * cfg.pw), device=enc.device, dtype=torch.float32)
                ev = torch.cuda.current_stream().record_event() if enc.is_cuda else None
                pat, geom = state.pat, (B, cfg.C, cfg.nh, cfg.nw, cfg.ph, cfg.pw)

                def fill(dst, pat=pat, ev=ev, geom=geom):     # holds the prediction ROWS (their own allocation), not the activation slab
                    if ev is not None:
                        cur = torch.cuda.current_stream()
                        cur.wait_event(ev)
                        pat.record_stream(cur)               # written on the adapter's stream, read here on the consumer's
                    ops.unpatchify_into(pat, dst, *geom)
                cfg.lazy_fill = fill
            ctx.comp = state if save else None
            ctx.cfg, ctx.params, ctx.ids = cfg, params, (ids_keep, ids_restore)
            ctx.dims = (B, NC, Denc, n_keep, n_q, T)
            if save and engine.patch_domain_loss():
                cfg.handle = PatHandle(state.pat, cfg.C, cfg.nh, cfg.nw, cfg.ph, cfg.pw, torch.float16 if h16 else act)
                if h16 or (act == torch.float32 and getattr(cfg, 'f32_gemm', 'exact') == 'f16'):
                    cfg.handle.dy_amax = torch.zeros(1, device=enc.device, dtype=torch.float32)
            return img
        if enc_act is None:
            enc_act = ops.cast(enc2, act)
        ctx_tok = _lin_fwd(enc_act, pcw, pcb, wc, torch.float32)                        # :258
        te = torch.zeros((T, D), device=dev, dtype=torch.float32)
        for i, t in enumerate(temb):
            if t is not None:
                te[i].copy_(t.detach().reshape(D))
        queries, context = ops.decoder_build(ctx_tok, ids_keep, ids_restore, mask_token.detach().reshape(D), te, cfg.pos,
                                             cfg.task_offsets, cfg.q_task, B, n_keep, G, D, n_q)   # :183-234
        if not xattn:                                # use_xattn=False (output_adapters.py:264-268): x = queries
            h, bsaved = queries, []
            for l in range(cfg.depth):
                h, s = block_fwd(h, blocks[12 * l:12 * l + 12], wc, heads, eps, act, B, n_q, save, dp=dp_of(l), drop=(p_attn, p_drop))
                bsaved.append(s)
            h_act = ops.cast(h, act)
            pat = _lin_fwd(h_act, ow, ob, wc, torch.float32)
            img = ops.unpatchify(pat, B, cfg.C, cfg.nh, cfg.nw, cfg.ph, cfg.pw)
            if save:
                ctx.saved = (enc_act, context.shape, h_act, bsaved)
            ctx.cfg, ctx.params, ctx.ids = cfg, params, (ids_keep, ids_restore)
            ctx.dims = (B, NC, Denc, n_keep, n_q, T)
            return img
        qn, qmean, qrstd = ops.layernorm_fwd(queries, qnw, qnb, eps, act)
        cn, cmean, crstd = ops.layernorm_fwd(context, cnw, cnb, eps, act)
        q = _lin_fwd(qn, qw, qb, wc, act)
        kv = _lin_fwd(cn, kvw, kvb, wc, act)
        hd = D // heads
        xo = torch.empty((B * n_q, D), device=dev, dtype=act)
        Pm = ops.attention_fwd(AttnView(q, 0, D, n_q), AttnView(kv, 0, 2 * D, NC), AttnView(kv, D, 2 * D, NC),
                               AttnView(xo, 0, D, n_q), B, heads, hd, hd ** -0.5, drop_p=p_attn)
        x = _lin_fwd(xo, pw_, pb, wc, torch.float32)                                    # :265, no residual
        k_x = None
        if p_drop > 0.:                                                                 # CrossAttention.proj_drop (multimae_utils.py:213)
            k_x = ops._dropout_keep((B * n_q, D), p_drop, dev)
            x = ops.dropout_apply(x, k_x, 1.0 / (1.0 - p_drop), out=x)
        on, omean, orstd = ops.layernorm_fwd(x, onw, onb, eps, act)
        hpre = torch.empty((B * n_q, f1w.shape[0]), device=dev, dtype=act)
        hact = _lin_fwd(on, f1w, f1b, wc, act, aux=hpre, epi=EPI_GELU)
        x1 = _lin_fwd(hact, f2w, f2b, wc, torch.float32, resid=x)                       # :266
        h, bsaved = x1, []
        for l in range(cfg.depth):
            h, s = block_fwd(h, blocks[12 * l:12 * l + 12], wc, heads, eps, act, B, n_q, save, dp=dp_of(l), drop=(p_attn, p_drop))   # :271
            bsaved.append(s)
        h_act = ops.cast(h, act)
        pat = _lin_fwd(h_act, ow, ob, wc, torch.float32)                                # :274
        img = ops.unpatchify(pat, B, cfg.C, cfg.nh, cfg.nw, cfg.ph, cfg.pw)             # :277-280
        if save:
            ctx.saved = (enc_act, queries, context, qn, qmean, qrstd, cn, cmean, crstd, q, kv, Pm, xo, x, on, omean, orstd,
                         hpre, hact, h_act, bsaved, k_x)
        ctx.cfg, ctx.params, ctx.ids = cfg, params, (ids_keep, ids_restore)
        ctx.dims = (B, NC, Denc, n_keep, n_q, T)
        return img

    @staticmethod
    def _backward(ctx, d_img: Tensor):
        cfg, params = ctx.cfg, ctx.params
        ids_keep, ids_restore = ctx.ids
        B, NC, Denc, n_keep, n_q, T = ctx.dims
        act, wc, D, G, heads = cfg.act, cfg.wc, cfg.D, cfg.G, cfg.heads
        lnact = None if act == torch.float32 else act
        if getattr(ctx, 'comp', None) is not None:
            sink = GradSink(engine.direct_grads())
            dsts, acc = _grad_targets(sink, params)
            use_side = sink.side is not None and acc
            h = getattr(ctx, 'handle', None)
            d_pat = h.d_pat if h is not None else None
            dy_amax = h.dy_amax if (h is not None and d_pat is not None and d_img is None) else None
            if d_pat is not None:
                if d_img is not None:
                    raise NotImplementedError('this prediction received gradient both through a patch-domain loss and through the image '
                                              'tensor; turn the patch-domain losses off (engine.set_patch_domain_loss(False))')
                d_pat.record_stream(torch.cuda.current_stream())     # produced by the loss's backward on another stream
                h.d_pat, h.pat = None, None
            elif d_img is None:                                      # nothing reached this adapter
                d_img = torch.zeros((B, cfg.C, cfg.nh * cfg.ph, cfg.nw * cfg.pw), device=params[0].device, dtype=torch.float32)
            if getattr(ctx, 'h16', False) and (d_pat is None or d_pat.dtype != torch.float16):
                # the gradient did not come from a loss kernel that writes scaled fp16 rows (image-domain gradient, a pixel loss,
                # several losses added up): bring it into the adapter's units here -- rows, their largest element, one scaled cast
                rows = d_pat if d_pat is not None else ops.patchify(d_img.contiguous().float(), cfg.C, cfg.nh, cfg.nw, cfg.ph, cfg.pw, torch.float32)
                dy_amax = rows.detach().abs().amax().reshape(1).float()
                d_pat, d_img = ops.cast_f16(rows, scale_amax=dy_amax), None
            lib_acc = acc and not engine.claim_first_write(params)      # first write since zero_grad(): store, do not accumulate
            d_enc, keep = ops.adapter_bwd(ctx.comp, d_img, d_pat, [None if t is None else t.view(-1) for t in dsts], lib_acc,
                                          sink.side.cuda_stream if use_side else None, dy_amax=dy_amax)
            ctx.comp = None
            if use_side:
                engine.mark_side_dirty(sink.side)
                engine.keep_until_join(keep)
            if cfg.on_done is not None and acc:
                cfg.on_done()
            return (None, d_enc, None, None, *([None] * len(params) if acc else dsts))
        xattn = bool(getattr(cfg, 'use_xattn', True))
        if xattn:
            (enc_act, queries, context, qn, qmean, qrstd, cn, cmean, crstd, q, kv, Pm, xo, x, on, omean, orstd, hpre, hact, h_act,
             bsaved, k_x) = ctx.saved
        else:
            enc_act, ctx_shape, h_act, bsaved = ctx.saved
        if d_img is None:
            d_img = torch.zeros((B, cfg.C, cfg.nh * cfg.ph, cfg.nw * cfg.pw), device=params[0].device, dtype=torch.float32)
        mask_token = params[0]
        temb = params[1:1 + T]
        (qw, qb, kvw, kvb, pw_, pb, cnw, cnb, qnw, qnb, onw, onb, f1w, f1b, f2w, f2b) = params[1 + T:17 + T]
        blocks = params[17 + T:17 + T + 12 * cfg.depth]
        ow, ob, pcw, pcb = params[17 + T + 12 * cfg.depth:]
        dev = d_img.device
        sink = GradSink(engine.direct_grads())
        hd = D // heads

        d_pat = ops.patchify(d_img, cfg.C, cfg.nh, cfg.nw, cfg.ph, cfg.pw, act)         # [B*n_q, C*ph*pw]
        g_ow, g_ob = sink.linear(ow, ob, d_pat, h_act)
        dh_act = ops.linear_dx(d_pat, wc(ow), torch.empty((B * n_q, D), device=dev, dtype=act))
        dh = ops.cast(dh_act, torch.float32)
        bgrads: List[Optional[Tensor]] = [None] * (12 * cfg.depth)
        fc2b_done, g_cs = False, None
        dps = getattr(cfg, 'dp', None)
        drops = getattr(cfg, 'drops', None)
        for l in reversed(range(cfg.depth)):
            cs_param = blocks[12 * (l - 1) + 11] if l > 0 else (f2b if xattn else None)
            if dps is not None and l > 0 and dps[2 * (l - 1) + 1] is not None:
                cs_param = None                      # the block below rescales its MLP branch: colsum(dx0) is not its fc2 bias gradient
            if l > 0 and drops is not None and drops[1] > 0.:
                cs_param = None                      # ... or drops elements of its fc2 output
            dh, dh_act, g_cs_next, g = block_bwd(dh, dh_act, fc2b_done, bsaved[l], blocks[12 * l:12 * l + 12], wc, sink, heads, act, B,
                                                 n_q, cs_param, dp=(dps[2 * l], dps[2 * l + 1]) if dps is not None else (None, None))
            bgrads[12 * l:12 * l + 12] = g
            if fc2b_done:
                bgrads[12 * l + 11] = g_cs
            fc2b_done, g_cs = (cs_param is not None and cs_param.requires_grad), g_cs_next
        if not xattn:                                # the queries fed the blocks directly: no cross-attention / MLP parameters
            d_ctx, part_b = ops.decoder_build_bwd(dh, torch.zeros(ctx_shape, device=dev, dtype=torch.float32), ids_keep, ids_restore,
                                                  cfg.task_offsets, cfg.q_task, B, n_keep, G, D, n_q, raw=True)
            g_build = sink.colsums(part_b, D, list(temb) + [mask_token])
            d_ctx_act = ops.cast(d_ctx, act)
            g_pcw, g_pcb = sink.linear(pcw, pcb, d_ctx_act, enc_act)
            d_enc = ops.linear_dx(d_ctx_act, wc(pcw), torch.empty((B * NC, Denc), device=dev, dtype=torch.float32))
            ctx.saved = None
            if cfg.on_done is not None and all(x is None for x in (g_build[T], *g_build[:T], *bgrads, g_ow, g_ob, g_pcw, g_pcb)):
                cfg.on_done()
            return (None, d_enc.view(B, NC, Denc), None, None, g_build[T], *g_build[:T], *([None] * 16), *bgrads, g_ow, g_ob, g_pcw, g_pcb)
        # x1 = x + mlp(out_norm(x))
        Hd = f1w.shape[0]
        part_h = torch.empty(ops.dx_colsum_part_shape(B * n_q, Hd), device=dev, dtype=torch.float32)
        d_hpre = ops.linear_dx(dh_act, wc(f2w), torch.empty(hpre.shape, device=dev, dtype=act), aux=hpre, epi=EPI_DGELU, colsum_part=part_h)
        if fc2b_done:
            g_f2w, g_f2b = sink.weight(f2w, dh_act, hact), g_cs
        else:
            g_f2w, g_f2b = sink.linear(f2w, f2b, dh_act, hact)
        d_on = ops.linear_dx(d_hpre, wc(f1w), torch.empty((B * n_q, D), device=dev, dtype=act))
        g_f1w, g_f1b = sink.weight(f1w, d_hpre, on), sink.colsums(part_h, Hd, [f1b])[0]
        dx, dx_act, part_o = ops.layernorm_bwd_part(d_on, x, onw, omean, orstd, dh, lnact)
        if dx_act is None:
            dx_act = dx
        # x = proj_drop(proj(attn(q, k, v)))
        if k_x is None:
            g_onw, g_onb, g_pb = sink.colsums(part_o, D, [onw, onb, pb])
            g_pw = sink.weight(pw_, dx_act, xo)
        else:                                        # the bias gradient is the column sums of the gradient BEHIND the dropout
            g_onw, g_onb, _ = sink.colsums(part_o, D, [onw, onb, None])
            dx_act = ops.dropout_apply(dx, k_x, 1.0 / (1.0 - drops[1]), out_dtype=act)
            g_pw, g_pb = sink.linear(pw_, pb, dx_act, xo)
        d_xo = ops.linear_dx(dx_act, wc(pw_), torch.empty((B * n_q, D), device=dev, dtype=act))
        d_q = torch.empty((B * n_q, D), device=dev, dtype=act)
        d_kv = torch.empty((B * NC, 2 * D), device=dev, dtype=act)
        ops.attention_bwd(AttnView(q, 0, D, n_q), AttnView(kv, 0, 2 * D, NC), AttnView(kv, D, 2 * D, NC), Pm,
                          AttnView(xo, 0, D, n_q), AttnView(d_xo, 0, D, n_q), AttnView(d_q, 0, D, n_q), AttnView(d_kv, 0, 2 * D, NC),
                          AttnView(d_kv, D, 2 * D, NC), B, heads, hd, hd ** -0.5)
        d_qn = ops.linear_dx(d_q, wc(qw), torch.empty((B * n_q, D), device=dev, dtype=act))
        g_qw, g_qb = sink.linear(qw, qb, d_q, qn)
        d_cn = ops.linear_dx(d_kv, wc(kvw), torch.empty((B * NC, D), device=dev, dtype=act))
        g_kvw, g_kvb = sink.linear(kvw, kvb, d_kv, cn)
        d_queries, _, part_q = ops.layernorm_bwd_part(d_qn, queries, qnw, qmean, qrstd, None, None)
        g_qnw, g_qnb, _ = sink.colsums(part_q, D, [qnw, qnb, None])
        d_context, _, part_c = ops.layernorm_bwd_part(d_cn, context, cnw, cmean, crstd, None, None)
        g_cnw, g_cnb, _ = sink.colsums(part_c, D, [cnw, cnb, None])
        d_ctx, part_b = ops.decoder_build_bwd(d_queries, d_context, ids_keep, ids_restore, cfg.task_offsets, cfg.q_task, B, n_keep, G, D,
                                              n_q, raw=True)
        g_build = sink.colsums(part_b, D, list(temb) + [mask_token])
        d_ctx_act = ops.cast(d_ctx, act)
        g_pcw, g_pcb = sink.linear(pcw, pcb, d_ctx_act, enc_act)
        d_enc = ops.linear_dx(d_ctx_act, wc(pcw), torch.empty((B * NC, Denc), device=dev, dtype=torch.float32))
        g_mask = g_build[T]
        g_temb = g_build[:T]
        ctx.saved = None
        grads = [g_mask, *g_temb, g_qw, g_qb, g_kvw, g_kvb, g_pw, g_pb, g_cnw, g_cnb,
                 g_qnw, g_qnb, g_onw, g_onb, g_f1w, g_f1b, g_f2w, g_f2b,
                 *bgrads, g_ow, g_ob, g_pcw, g_pcb]
        if cfg.on_done is not None and all(x is None for x in grads):
            cfg.on_done()
        return (None, d_enc.view(B, NC, Denc), None, None, *grads)


# ------------------------------------------------------------------------------------------
# generic single ops as autograd nodes (stand-alone Linear / LayerNorm / attention modules)
# ------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg: _Cfg, x: Tensor, w: Tensor, b: Optional[Tensor]):
        ops._require_gpu(x, 'linear input')
        shp = x.shape
        x2 = ops.cast(x.contiguous().view(-1, shp[-1]), cfg.act)
        y = ops.linear_fwd(x2, cfg.wc(w), b.detach() if b is not None else None,
                           torch.empty((x2.shape[0], w.shape[0]), device=x.device, dtype=torch.float32))
        ctx.cfg, ctx.x2, ctx.w, ctx.b, ctx.shp = cfg, x2, w, b, shp
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy: Tensor):
        cfg, x2, w, b = ctx.cfg, ctx.x2, ctx.w, ctx.b
        sink = GradSink(engine.direct_grads())
        dy2 = ops.cast(dy.contiguous().view(-1, w.shape[0]), cfg.act)
        dx = ops.linear_dx(dy2, cfg.wc(w), torch.empty((dy2.shape[0], w.shape[1]), device=dy.device, dtype=torch.float32))
        gw, gb = sink.linear(w, b, dy2, x2)
        return None, dx.view(ctx.shp), gw, gb


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Tensor, eps: float):
        ops._require_gpu(x, 'layernorm input')
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1]).float()
        y, mean, rstd = ops.layernorm_fwd(x2, w.detach(), b.detach(), eps, torch.float32)
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.shp, ctx.wb = shp, (w, b)
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2, w, mean, rstd = ctx.saved_tensors
        sink = GradSink(engine.direct_grads())
        dx, _, dg, db, _ = ops.layernorm_bwd(dy.contiguous().view(x2.shape).float(), x2, w.detach(), mean, rstd, None, None)
        return dx.view(ctx.shp), sink.vec(ctx.wb[0], dg), sink.vec(ctx.wb[1], db), None


class TokenMeanFn(torch.autograd.Function):
    """(B, N, D) -> (B, D) mean over tokens (LinearOutputAdapter's pooling, output_adapters.py:346-347)."""

    @staticmethod
    def forward(ctx, x: Tensor):
        from . import _lib
        ops._require_gpu(x, 'token mean input')
        B, N, D = x.shape
        x = x.contiguous().float()
        y = torch.empty((B, D), device=x.device, dtype=torch.float32)
        ops.check(_lib.load().mmae_token_mean_fwd(x.data_ptr(), y.data_ptr(), B, N, D, ops._stream()), 'token_mean_fwd')
        ctx.dims = (B, N, D)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        from . import _lib
        B, N, D = ctx.dims
        dy = dy.contiguous().float()
        dx = torch.empty((B, N, D), device=dy.device, dtype=torch.float32)
        ops.check(_lib.load().mmae_token_mean_bwd(dy.data_ptr(), dx.data_ptr(), B, N, D, ops._stream()), 'token_mean_bwd')
        return dx


class AttentionCoreFn(torch.autograd.Function):
    """softmax(q k^T scale) v on packed activations; used by the stand-alone Attention / CrossAttention modules."""

    @staticmethod
    def forward(ctx, q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float, act, drop_p: float = 0.):
        B, Nq, D = q.shape
        Nk = k.shape[1]
        hd = D // heads
        qa, ka, va = (ops.cast(t.contiguous().view(-1, D), act) for t in (q, k, v))
        o = torch.empty((B * Nq, D), device=q.device, dtype=act)
        Pm = ops.attention_fwd(AttnView(qa, 0, D, Nq), AttnView(ka, 0, D, Nk), AttnView(va, 0, D, Nk), AttnView(o, 0, D, Nq), B, heads,
                               hd, scale, drop_p=drop_p)
        ctx.saved = (qa, ka, va, Pm, o)
        ctx.dims = (B, Nq, Nk, D, heads, hd, scale, act)
        return ops.cast(o, torch.float32).view(B, Nq, D)

    @staticmethod
    def backward(ctx, d_o: Tensor):
        qa, ka, va, Pm, o = ctx.saved
        B, Nq, Nk, D, heads, hd, scale, act = ctx.dims
        do = ops.cast(d_o.contiguous().view(-1, D), act)
        dq = torch.empty((B * Nq, D), device=d_o.device, dtype=act)
        dk = torch.empty((B * Nk, D), device=d_o.device, dtype=act)
        dv = torch.empty((B * Nk, D), device=d_o.device, dtype=act)
        ops.attention_bwd(AttnView(qa, 0, D, Nq), AttnView(ka, 0, D, Nk), AttnView(va, 0, D, Nk), Pm, AttnView(o, 0, D, Nq), AttnView(do, 0, D, Nq),
                          AttnView(dq, 0, D, Nq), AttnView(dk, 0, D, Nk), AttnView(dv, 0, D, Nk), B, heads, hd, scale)
        f = torch.float32
        return ops.cast(dq, f).view(B, Nq, D), ops.cast(dk, f).view(B, Nk, D), ops.cast(dv, f).view(B, Nk, D), None, None, None, None


class MlpFn(torch.autograd.Function):
    """fc1 -> exact-erf GELU (fused epilogue) -> fc2, multimae_utils.py:147-155."""

    @staticmethod
    def forward(ctx, cfg: _Cfg, x: Tensor, f1w: Tensor, f1b: Tensor, f2w: Tensor, f2b: Tensor):
        ops._require_gpu(x, 'mlp input')
        shp = x.shape
        x2 = ops.cast(x.contiguous().view(-1, shp[-1]), cfg.act)
        hpre = torch.empty((x2.shape[0], f1w.shape[0]), device=x.device, dtype=cfg.act)
        hact = _lin_fwd(x2, f1w, f1b.detach(), cfg.wc, cfg.act, aux=hpre, epi=EPI_GELU)
        p = float(getattr(cfg, 'drop', 0.))              # Mlp(drop=) in training mode: behind fc2 only (:154; :151-152 is commented out)
        k_y = None
        y = _lin_fwd(hact, f2w, f2b.detach(), cfg.wc, torch.float32)
        if p > 0.:
            k_y = ops._dropout_keep(tuple(y.shape), p, x.device)
            y = ops.dropout_apply(y, k_y, 1.0 / (1.0 - p), out=y)
        ctx.cfg, ctx.saved, ctx.params, ctx.shp = cfg, (x2, hpre, hact, k_y), (f1w, f1b, f2w, f2b), shp
        return y.view(*shp[:-1], f2w.shape[0])

    @staticmethod
    def backward(ctx, dy: Tensor):
        cfg = ctx.cfg
        x2, hpre, hact, k_y = ctx.saved
        f1w, f1b, f2w, f2b = ctx.params
        sink = GradSink(engine.direct_grads())
        inv = 1.0 / (1.0 - float(getattr(cfg, 'drop', 0.)))
        if k_y is not None:
            dy2 = ops.dropout_apply(dy.contiguous().view(-1, f2w.shape[0]).float(), k_y, inv, out_dtype=cfg.act)
        else:
            dy2 = ops.cast(dy.contiguous().view(-1, f2w.shape[0]), cfg.act)
        d_hpre = ops.linear_dx(dy2, cfg.wc(f2w), torch.empty(hpre.shape, device=dy.device, dtype=cfg.act), aux=hpre, epi=EPI_DGELU)
        dx = ops.linear_dx(d_hpre, cfg.wc(f1w), torch.empty((dy2.shape[0], f1w.shape[1]), device=dy.device, dtype=torch.float32))
        g1w, g1b = sink.linear(f1w, f1b, d_hpre, x2)
        g2w, g2b = sink.linear(f2w, f2b, dy2, hact)
        return None, dx.view(ctx.shp), g1w, g1b, g2w, g2b


# ------------------------------------------------------------------------------------------
# masked losses (criterion.py)
# ------------------------------------------------------------------------------------------
def _loss_bufs(B: int, dev):
    from . import _lib
    ns = _lib.load().mmae_loss_split()
    f = torch.float32
    return (torch.empty((B, ns), device=dev, dtype=f), torch.empty((B, 2), device=dev, dtype=f),
            torch.empty((2,), device=dev, dtype=f))


class MaskedPixelLossFn(torch.autograd.Function):
    """MaskedMSELoss / MaskedL1Loss (criterion.py:84-114,141-171).  kind 0 = MSE, 1 = L1."""

    @staticmethod
    def forward(ctx, pred: Tensor, target: Tensor, mask: Tensor, kind: int, norm_pix: bool, patch: int):
        from . import _lib
        ops._require_gpu(pred, 'loss input')
        pred = pred.contiguous().float()
        target = target.contiguous().float()
        mask = mask.contiguous().long()
        B, C, H, W = pred.shape
        partial, per_sample, loss = _loss_bufs(B, pred.device)
        np_ = (H // patch) * (W // patch)
        stats = torch.empty((B, np_, 2), device=pred.device, dtype=torch.float32) if norm_pix else None
        ops.check(_lib.load().mmae_masked_pixel_loss_fwd(pred.data_ptr(), target.data_ptr(), mask.data_ptr(), kind, int(norm_pix), B, C, H,
                                                         W, patch, ops._p(stats), partial.data_ptr(), per_sample.data_ptr(),
                                                         loss.data_ptr(), ops._stream()), 'masked_pixel_loss_fwd')
        ctx.saved = (pred, target, mask, stats, per_sample, loss, (pred._version, target._version, mask._version))
        ctx.args = (kind, norm_pix, patch)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g: Tensor):
        from . import _lib
        pred, target, mask, stats, per_sample, loss, versions = ctx.saved
        ctx.saved = None
        if (pred._version, target._version, mask._version) != versions:      # what ctx.save_for_backward would have caught
            raise RuntimeError('a tensor saved for the masked-loss backward was modified in place')
        kind, norm_pix, patch = ctx.args
        B, C, H, W = pred.shape
        d_pred = torch.empty_like(pred)
        up = g.contiguous().float().reshape(1)
        ops.check(_lib.load().mmae_masked_pixel_loss_bwd(pred.data_ptr(), target.data_ptr(), mask.data_ptr(), kind, int(norm_pix), B, C, H,
                                                         W, patch, ops._p(stats), per_sample.data_ptr(), loss.data_ptr(), up.data_ptr(),
                                                         d_pred.data_ptr(), ops._stream()), 'masked_pixel_loss_bwd')
        return d_pred, None, None, None, None, None


class MaskedCEFn(torch.autograd.Function):
    """MaskedCrossEntropyLoss (criterion.py:37-57)."""

    @staticmethod
    def forward(ctx, logits: Tensor, target: Tensor, mask: Tensor, patch: int, smooth: float = 0.0):
        from . import _lib
        ops._require_gpu(logits, 'loss input')
        logits = logits.contiguous().float()
        target = target.contiguous().long()
        mask = mask.contiguous().long()
        B, C, H, W = logits.shape
        partial, per_sample, loss = _loss_bufs(B, logits.device)
        lse = torch.empty((B, H, W), device=logits.device, dtype=torch.float32)
        ops.check(_lib.load().mmae_masked_ce_fwd(logits.data_ptr(), target.data_ptr(), mask.data_ptr(), B, C, H, W, patch, float(smooth), lse.data_ptr(),
                                                 partial.data_ptr(), per_sample.data_ptr(), loss.data_ptr(), ops._stream()), 'masked_ce_fwd')
        ctx.saved = (logits, target, mask, lse, per_sample, loss, (logits._version, target._version, mask._version))
        ctx.patch, ctx.smooth = patch, float(smooth)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g: Tensor):
        from . import _lib
        logits, target, mask, lse, per_sample, loss, versions = ctx.saved
        ctx.saved = None
        if (logits._version, target._version, mask._version) != versions:
            raise RuntimeError('a tensor saved for the masked-loss backward was modified in place')
        B, C, H, W = logits.shape
        d = torch.empty_like(logits)
        up = g.contiguous().float().reshape(1)
        ops.check(_lib.load().mmae_masked_ce_bwd(logits.data_ptr(), target.data_ptr(), mask.data_ptr(), B, C, H, W, ctx.patch, ctx.smooth,
                                                 lse.data_ptr(), per_sample.data_ptr(), loss.data_ptr(), up.data_ptr(), d.data_ptr(),
                                                 ops._stream()), 'masked_ce_bwd')
        return d, None, None, None, None


def _h16_rows_to_f32(h: PatHandle) -> None:
    """gradient rows an fp16-storage adapter already received (scaled fp16, from the cross-entropy kernel) back to plain f32"""
    if h.d_pat is not None and h.d_pat.dtype == torch.float16:
        h.d_pat = ops.cast_f16(h.d_pat, scale_amax=h.dy_amax)


class MaskedPixelLossPatFn(torch.autograd.Function):
    """MaskedMSELoss / MaskedL1Loss on an output adapter's patch rows (PatHandle): same per-pixel arithmetic as
    MaskedPixelLossFn, gradient handed to the adapter as patch rows in its activation dtype."""

    @staticmethod
    def forward(ctx, token: Tensor, h: PatHandle, target: Tensor, mask: Tensor, kind: int, norm_pix: bool, patch: int):
        from . import _lib
        pat = h.pat
        target = target.contiguous().float()
        mask = mask.contiguous().long()
        B, C, H, W = target.shape[0], h.C, h.nh * h.ph, h.nw * h.pw
        partial, per_sample, loss = _loss_bufs(B, pat.device)
        stats = torch.empty((B, h.nh * h.nw, 2), device=pat.device, dtype=torch.float32) if norm_pix else None
        ops.check(_lib.load().mmae_masked_pixel_loss_pat_fwd(pat.data_ptr(), target.data_ptr(), mask.data_ptr(), kind, int(norm_pix), B, C, H, W,
                                                             patch, ops._p(stats), partial.data_ptr(), per_sample.data_ptr(), loss.data_ptr(),
                                                             ops._stream()), 'masked_pixel_loss_pat_fwd')
        ctx.saved = (h, target, mask, stats, per_sample, loss, (target._version, mask._version))
        ctx.args = (kind, norm_pix, patch, B, C, H, W)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g: Tensor):
        from . import _lib
        h, target, mask, stats, per_sample, loss, versions = ctx.saved
        ctx.saved = None
        kind, norm_pix, patch, B, C, H, W = ctx.args
        if (target._version, mask._version) != versions:
            raise RuntimeError('a tensor saved for the masked-loss backward was modified in place')
        # (fp16-storage adapter: f32 rows here, the adapter's backward scales them into its units -- only the cross entropy bounds its
        # gradient before computing it)
        row_t = torch.float32 if h.act == torch.float16 else h.act
        _h16_rows_to_f32(h)
        d_pat = torch.empty((h.pat.shape[0], h.ld()), device=h.pat.device, dtype=row_t)
        up = g.contiguous().float().reshape(1)
        ops.check(_lib.load().mmae_masked_pixel_loss_pat_bwd(h.pat.data_ptr(), target.data_ptr(), mask.data_ptr(), kind, int(norm_pix), B, C, H, W,
                                                             patch, ops._p(stats), per_sample.data_ptr(), loss.data_ptr(), up.data_ptr(),
                                                             d_pat.data_ptr(), ops.dcode(row_t), d_pat.stride(0),
                                                             None if h.act == torch.float16 else ops._p(h.dy_amax), ops._stream()),
                  'masked_pixel_loss_pat_bwd')
        h.d_pat = d_pat if h.d_pat is None else h.d_pat.add_(d_pat)        # several losses on one prediction: their gradients add
        return up.new_empty(1), None, None, None, None, None, None


class MaskedCEPatFn(torch.autograd.Function):
    """MaskedCrossEntropyLoss on an output adapter's patch rows."""

    @staticmethod
    def forward(ctx, token: Tensor, h: PatHandle, target: Tensor, mask: Tensor, patch: int, smooth: float = 0.0):
        from . import _lib
        pat = h.pat
        target = target.contiguous().long()
        mask = mask.contiguous().long()
        B, C, H, W = target.shape[0], h.C, h.nh * h.ph, h.nw * h.pw
        partial, per_sample, loss = _loss_bufs(B, pat.device)
        lse = torch.empty((pat.shape[0], patch * patch), device=pat.device, dtype=torch.float32)
        ops.check(_lib.load().mmae_masked_ce_pat_fwd(pat.data_ptr(), target.data_ptr(), mask.data_ptr(), B, C, H, W, patch, float(smooth), lse.data_ptr(),
                                                     partial.data_ptr(), per_sample.data_ptr(), loss.data_ptr(), ops._stream()), 'masked_ce_pat_fwd')
        ctx.saved = (h, target, mask, lse, per_sample, loss, (target._version, mask._version))
        ctx.args = (patch, B, C, H, W, float(smooth))
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g: Tensor):
        from . import _lib
        h, target, mask, lse, per_sample, loss, versions = ctx.saved
        ctx.saved = None
        patch, B, C, H, W, smooth = ctx.args
        if (target._version, mask._version) != versions:
            raise RuntimeError('a tensor saved for the masked-loss backward was modified in place')
        # fp16-storage adapter: the first (normally the only) loss writes scaled fp16 rows and the bound they are scaled by (dy_amax);
        # a second loss on the same prediction continues in f32 rows
        row_t = h.act
        if h.act == torch.float16 and h.d_pat is not None:
            _h16_rows_to_f32(h)
            row_t = torch.float32
        d_pat = torch.empty((h.pat.shape[0], h.ld()), device=h.pat.device, dtype=row_t)
        up = g.contiguous().float().reshape(1)
        ops.check(_lib.load().mmae_masked_ce_pat_bwd(h.pat.data_ptr(), target.data_ptr(), mask.data_ptr(), B, C, H, W, patch, smooth, lse.data_ptr(),
                                                     per_sample.data_ptr(), loss.data_ptr(), up.data_ptr(), d_pat.data_ptr(), ops.dcode(row_t),
                                                     d_pat.stride(0), None if (h.act == torch.float16 and row_t == torch.float32) else ops._p(h.dy_amax),
                                                     ops._stream()), 'masked_ce_pat_bwd')
        h.d_pat = d_pat if h.d_pat is None else h.d_pat.add_(d_pat)
        return up.new_empty(1), None, None, None, None, None
