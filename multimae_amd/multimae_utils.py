"""Building blocks of the MI355X-native MultiMAE engine (mirror of the reference's
``multimae/multimae_utils.py`` API: same class names, constructor arguments, parameter names
and registration order -- so ``state_dict`` keys and seeded initialisation are identical --
but every forward/backward is a hand-written HIP kernel sequence, see functions.py).

Reference: multimae/multimae_utils.py:29-45 (sincos), :48-102 (trunc_normal_), :105-135
(DropPath), :138-155 (Mlp), :158-182 (Attention), :185-214 (CrossAttention), :217-232 (Block).
"""
from __future__ import annotations

import math
import warnings

import torch
from typing import Optional
from torch import nn

from . import engine
from .functions import (AttentionCoreFn, EncoderStackFn, LayerNormFn, LinearFn, MlpFn, _Cfg)


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def build_2d_sincos_posemb(h, w, embed_dim=1024, temperature=10000.):
    """(1, embed_dim, h, w) fixed sin-cos table (MoCo-v3 layout as used by the reference:
    channel quarters [sin(r w_k), cos(r w_k), sin(c w_k), cos(c w_k)] for row r / column c on
    square grids; for h != w the reference's flatten-then-reshape order is reproduced)."""
    assert embed_dim % 4 == 0, 'Embed dimension must be divisible by 4 for 2D sin-cos position embedding'
    q = embed_dim // 4
    omega = 1. / (temperature ** (torch.arange(q, dtype=torch.float32) / q))
    a = torch.arange(w, dtype=torch.float32).repeat_interleave(h)     # flat f = i*h + j -> i
    b = torch.arange(h, dtype=torch.float32).repeat(w)                # -> j
    oa, ob = a[:, None] * omega[None, :], b[:, None] * omega[None, :]
    emb = torch.cat([torch.sin(oa), torch.cos(oa), torch.sin(ob), torch.cos(ob)], dim=1)
    return emb.reshape(h, w, embed_dim).permute(2, 0, 1).unsqueeze(0).contiguous()


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """Truncated normal by inverse-CDF sampling (the timm/PyTorch recipe the reference vendors:
    uniform in [2*Phi(l)-1, 2*Phi(u)-1] -> erfinv -> scale -> clamp); same RNG consumption, so a
    seeded construction reproduces the reference's initial weights bit for bit."""
    def cdf(x):
        return (1. + math.erf(x / math.sqrt(2.))) / 2.
    if (mean < a - 2 * std) or (mean > b + 2 * std):
        warnings.warn('mean is more than 2 std from [a, b] in trunc_normal_', stacklevel=2)
    with torch.no_grad():
        lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
        tensor.uniform_(2 * lo - 1, 2 * hi - 1)
        tensor.erfinv_()
        tensor.mul_(std * math.sqrt(2.))
        tensor.add_(mean)
        tensor.clamp_(min=a, max=b)
    return tensor


def set_root(root: nn.Module) -> None:
    """Let every sub-module find the model that owns the parameter arena (weak reference, stored
    outside nn.Module's attribute registry)."""
    import weakref
    ref = weakref.ref(root)
    for m in root.modules():
        object.__setattr__(m, '_mmae_root', ref)


def root_of(module: nn.Module) -> nn.Module:
    ref = getattr(module, '_mmae_root', None)
    r = ref() if ref is not None else None
    return r if r is not None else module


def _cfg(module: nn.Module, act=None, **kw) -> _Cfg:
    act = act or engine.act_dtype()
    arena = engine.arena_of(root_of(module))
    return _Cfg(act=act, wc=engine.WeightCache(arena, act), **kw)


class Linear(nn.Linear):
    """nn.Linear parameter container whose forward is the MFMA GEMM kernel (fp32 in / fp32 out
    at the module boundary; the engine's fused paths bypass this and feed bf16 directly)."""

    def forward(self, x):
        return LinearFn.apply(_cfg(self), x, self.weight, self.bias)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameter container; forward = wave-per-row HIP kernel."""

    def forward(self, x):
        return LayerNormFn.apply(x, self.weight, self.bias, self.eps)


class DropoutFn(torch.autograd.Function):
    """nn.Dropout on the HIP element-wise kernel (mmae_dropout): y = keep * x / (1 - p), the same pass on the gradient."""

    @staticmethod
    def forward(ctx, x, p: float):
        from . import ops
        ops._require_gpu(x, 'dropout input')
        xc = x.contiguous().float()
        keep = ops._dropout_keep(tuple(xc.shape), p, xc.device)
        ctx.keep, ctx.inv = keep, 1.0 / (1.0 - p)
        return ops.dropout_apply(xc, keep, ctx.inv)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        return ops.dropout_apply(dy.contiguous().float(), ctx.keep, ctx.inv), None


def dropout(x, p: float, training: bool):
    """nn.Dropout(p)(x) (multimae_utils.py:152-154, 177, 181): identity in eval mode and at the pre-training default p = 0."""
    if not training or p == 0.:
        return x
    return DropoutFn.apply(x, float(p))


def _drop_path_rand(shape, device) -> torch.Tensor:
    """The uniform draw behind a stochastic-depth mask (multimae_utils.py:117: torch.rand(shape, device=x.device)).  Tests
    replace this hook to inject a fixed draw on both sides of a comparison."""
    return torch.rand(shape, dtype=torch.float32, device=device)


def drop_path_scale(batch: int, drop_prob: float, device) -> torch.Tensor:
    """Per-sample scale of a residual branch under stochastic depth: floor(keep + u) / keep, u ~ U[0,1), f32 [B]
    (multimae_utils.py:115-120: ``x.div(keep_prob) * floor(keep_prob + rand)``)."""
    keep = 1.0 - drop_prob
    u = _drop_path_rand((batch,), device)
    return ((keep + u).floor_() / keep).contiguous()


def drop_path(x, drop_prob: float = 0., training: bool = False):
    """Stochastic depth per sample (multimae_utils.py:105-122).  Identity at the pre-training default (rate 0); otherwise
    one row-scale kernel (and the same kernel on the gradient)."""
    if drop_prob == 0. or not training:
        return x
    return DropPathFn.apply(x, drop_path_scale(x.shape[0], drop_prob, x.device))


class DropPathFn(torch.autograd.Function):
    """y[b] = s[b] * x[b] on the HIP row-scale kernel (stand-alone DropPath modules; inside Blocks the scale is folded into the
    residual add, functions.block_fwd)."""

    @staticmethod
    def forward(ctx, x, s):
        from . import ops
        ops._require_gpu(x, 'drop_path input')
        shp = x.shape
        x2 = x.contiguous().float().view(shp[0], -1)
        ctx.s, ctx.shp = s, shp
        return ops.rowscale_cast(x2, s, 1, torch.float32).view(shp)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        d2 = dy.contiguous().float().view(ctx.shp[0], -1)
        return ops.rowscale_cast(d2, ctx.s, 1, torch.float32).view(ctx.shp), None


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)

    def extra_repr(self) -> str:
        return 'p={}'.format(self.drop_prob)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU:
            raise NotImplementedError('the HIP engine fuses exact-erf GELU into the fc1 epilogue; other activations are not built')
        self.fc1 = Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        p = float(self.drop.p) if self.training else 0.
        return MlpFn.apply(_cfg(self, drop=p), x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        if not qkv_bias:          # the fused kernels always add a bias row: a frozen zero vector outside the state_dict stands in
            self.register_buffer('_zero_qkv_bias', torch.zeros(dim * 3), persistent=False)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        o = AttentionCoreFn.apply(q, k, v, self.num_heads, self.scale, engine.act_dtype(), float(self.attn_drop.p) if self.training else 0.)
        return dropout(self.proj(o), self.proj_drop.p, self.training)


class CrossAttention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.q = Linear(dim, dim, bias=qkv_bias)
        self.kv = Linear(dim, dim * 2, bias=qkv_bias)
        if not qkv_bias:
            self.register_buffer('_zero_q_bias', torch.zeros(dim), persistent=False)
            self.register_buffer('_zero_kv_bias', torch.zeros(dim * 2), persistent=False)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x, context):
        C = x.shape[-1]
        q = self.q(x)
        kv = self.kv(context)
        o = AttentionCoreFn.apply(q, kv[..., :C], kv[..., C:], self.num_heads, self.scale, engine.act_dtype(),
                                  float(self.attn_drop.p) if self.training else 0.)
        return dropout(self.proj(o), self.proj_drop.p, self.training)


def bias_or_zero(lin: nn.Linear, zero: Optional[torch.Tensor]) -> torch.Tensor:
    """the Linear's bias, or the owner's frozen zero vector when it was built with bias=False (qkv_bias=False)"""
    return lin.bias if lin.bias is not None else zero


def block_params(blk: 'Block'):
    return [blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, bias_or_zero(blk.attn.qkv, getattr(blk.attn, '_zero_qkv_bias', None)),
            blk.attn.proj.weight, blk.attn.proj.bias,
            blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias]


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = _as_hip_norm(norm_layer, dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = _as_hip_norm(norm_layer, dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)

    def forward(self, x):
        return run_blocks([self], x)


def _as_hip_norm(norm_layer, dim):
    """Instantiate the caller's norm_layer and make sure it is the HIP LayerNorm container."""
    m = norm_layer(dim)
    if isinstance(m, LayerNorm):
        return m
    if isinstance(m, nn.LayerNorm):
        h = LayerNorm(dim, eps=m.eps)
        return h
    raise NotImplementedError(f'norm layer {type(m)} is not built in the HIP engine')


def _stack_drop_path(blocks, batch: int, device):
    """Per-block stochastic-depth scales [(attention branch, MLP branch)] * L, drawn in the reference's order AMONG THEMSELVES (two draws
    per block, block by block, multimae_utils.py:229-232); None when no block drops paths (eval mode, rate 0).  All of them are drawn here,
    up front: with drop_path_rate > 0 AND drop_rate / attn_drop_rate > 0 together the reference interleaves attn_drop, proj_drop, DropPath,
    mlp.drop, DropPath per block on one generator, so under a shared seed its random STREAM differs from this engine's (same distributions,
    same sites; each kind of mask keeps its own order -- ADVICE r5)."""
    if not any((not isinstance(b.drop_path, nn.Identity)) and b.training and (b.drop_path.drop_prob or 0.) > 0. for b in blocks):
        return None
    dp = []
    for b in blocks:
        rate = 0. if isinstance(b.drop_path, nn.Identity) or not b.training else float(b.drop_path.drop_prob or 0.)
        dp += [drop_path_scale(batch, rate, device), drop_path_scale(batch, rate, device)] if rate > 0. else [None, None]
    return dp


def _stack_dropout(blocks):
    """Per-block (attn_drop, drop) rates of Block(drop=, attn_drop=) -- Attention.attn_drop / .proj_drop and Mlp.drop share `drop`
    (multimae_utils.py:221-227) -- or None when no block drops anything (eval mode, the pre-training default 0)."""
    rates = [((float(b.attn.attn_drop.p), float(b.mlp.drop.p)) if b.training else (0., 0.)) for b in blocks]
    for b, r in zip(blocks, rates):
        assert not b.training or float(b.attn.proj_drop.p) == r[1], 'Block: attn.proj_drop and mlp.drop carry one rate (multimae_utils.py:221-227)'
    return rates if any(a > 0. or d > 0. for a, d in rates) else None


def run_blocks(blocks, x, root=None, all_layers=False, on_layer_done=None, bwd_chunk: int = 1, mx: bool = False):
    """Run a sequence of Blocks as ONE autograd node (the encoder / decoder_transformer fast path), stochastic depth
    included: the per-sample scales are drawn here and folded into the blocks' residual adds.  Blocks with dropout > 0 (training
    mode) run their per-kernel sequence with the three nn.Dropout sites as element-wise passes (with bf16 products also in MX-fp8 mode:
    the scaled-MFMA products are a property of the one-call stack)."""
    blocks = list(blocks)
    if not blocks:
        return [] if all_layers else x
    b0 = blocks[0]
    cfg = _cfg(b0 if root is None else root, heads=b0.attn.num_heads, eps=b0.norm1.eps, all_layers=all_layers,
               on_layer_done=on_layer_done, dp=_stack_drop_path(blocks, x.shape[0], x.device), bwd_chunk=bwd_chunk, mx=mx,
               drops=_stack_dropout(blocks))
    params = [p for b in blocks for p in block_params(b)]
    out = EncoderStackFn.apply(cfg, x, *params)
    return list(out) if all_layers else out
