"""CPU oracle for the MultiMAE pre-training hot path -- TEST INFRASTRUCTURE ONLY.

This file is a functional, state-dict driven, fp32 CPU restatement of the
reference algorithm.  It is the *checker* for the HIP engine in
``multimae_amd/``; nothing in the product path may import it.  Allowed users:
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.

Parity status: PINNED.  ``tests/golden/make_golden.py`` runs the reference's own
classes (imported from /root/reference in the build container) and this oracle
on identical weights / inputs / RNG stream and stores the reference outputs in
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks the oracle
against those vectors wherever the suite runs (the GPU box has no reference).

Every function cites the reference file:line it restates (paths relative to
/root/reference).  Floating-point work is plain torch fp32 on CPU (the
reference's arithmetic *is* ATen); integer work (mask/ids) is exact.

The oracle takes parameters as a flat ``{name: tensor}`` dict with the
reference's ``state_dict`` key layout so a reference
checkpoint, a golden fixture and the engine's own ``state_dict()`` are all
directly usable.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# configuration                                                               #
# --------------------------------------------------------------------------- #
@dataclass
class DomainSpec:
    """One modality.  kind: 'image' (PatchedInputAdapter) or 'semseg'."""
    name: str
    kind: str                 # 'image' | 'semseg'
    channels: int             # image channels, or number of classes for semseg
    stride_level: int = 1
    dim_class_emb: int = 64   # semseg only


@dataclass
class OracleConfig:
    """Static description of a MultiMAE instance (run_pretraining_multimae.py:243-293)."""
    in_domains: List[DomainSpec]
    out_tasks: List[Tuple[str, str]]          # (adapter key, task it decodes) e.g. ('norm_rgb','rgb')
    patch_size: int = 16
    image_size: int = 224
    dim_tokens: int = 768
    depth: int = 12
    num_heads: int = 12
    num_global_tokens: int = 1
    dec_dim: int = 256
    dec_depth: int = 2
    dec_heads: int = 8
    ln_eps: float = 1e-6
    domains_by_name: Dict[str, DomainSpec] = field(default_factory=dict)
    out_only_domains: List[DomainSpec] = field(default_factory=list)      # decoded but not fed to the encoder (--in_domains rgb --out_domains rgb-depth)

    def __post_init__(self):
        self.domains_by_name = {d.name: d for d in list(self.in_domains) + list(self.out_only_domains)}

    def patch_hw(self, d: DomainSpec) -> Tuple[int, int]:
        p = max(1, self.patch_size // d.stride_level)     # input_adapters.py:61-62
        return p, p


# --------------------------------------------------------------------------- #
# positional embedding                                                        #
# --------------------------------------------------------------------------- #
def sincos_posemb_2d(h: int, w: int, dim: int, temperature: float = 10000.0) -> Tensor:
    """Fixed 2-D sin/cos table, (1, dim, h, w).  multimae_utils.py:29-45.

    Channel blocks of dim/4 (square grids): [sin(row*w_k), cos(row*w_k), sin(col*w_k),
    cos(col*w_k)] with w_k = temperature**(-k/(dim/4)).  (The reference's "grid_w" ends
    up indexing rows because of meshgrid's ij order followed by the '(h w)' reshape.)
    """
    assert dim % 4 == 0
    q = dim // 4
    omega = 1.0 / (temperature ** (torch.arange(q, dtype=torch.float32) / q))
    # the reference meshgrid(w, h) (ij indexing) + flatten makes the flat index
    # run over h fastest: flat = x * h + y, later reinterpreted as (h w) with
    # h = outer.  For h == w this is pos[(r, c)] <- (x = r, y = c).
    gx, gy = torch.meshgrid(torch.arange(w, dtype=torch.float32),
                            torch.arange(h, dtype=torch.float32), indexing='ij')
    ox = gx.flatten()[:, None] * omega[None, :]
    oy = gy.flatten()[:, None] * omega[None, :]
    emb = torch.cat([torch.sin(ox), torch.cos(ox), torch.sin(oy), torch.cos(oy)], dim=1)  # (h*w, dim)
    return emb.reshape(h, w, dim).permute(2, 0, 1)[None].contiguous()


def resized_posemb_tokens(pos_emb: Tensor, nh: int, nw: int, mode: str) -> Tensor:
    """(1,D,h,w) parameter -> (nh*nw, D) token table.  input_adapters.py:113-114,
    235-236; output_adapters.py:172-173.  ATen's interpolate is the arithmetic."""
    if mode == 'bicubic':
        t = F.interpolate(pos_emb, size=(nh, nw), mode='bicubic', align_corners=False)
    else:
        t = F.interpolate(pos_emb, size=(nh, nw), mode='bilinear', align_corners=False)
    return t[0].flatten(1).t().contiguous()


# --------------------------------------------------------------------------- #
# input adapters                                                              #
# --------------------------------------------------------------------------- #
def patchify_rows(x: Tensor, ph: int, pw: int) -> Tensor:
    """(B,C,H,W) -> (B, N, C*ph*pw) with the Conv2d weight's (c, i, j) column order."""
    B, C, H, W = x.shape
    nh, nw = H // ph, W // pw
    x = x.reshape(B, C, nh, ph, nw, pw).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, nh * nw, C * ph * pw)


def image_tokens(x: Tensor, sd: Dict[str, Tensor], prefix: str, ph: int, pw: int) -> Tensor:
    """PatchedInputAdapter.forward, input_adapters.py:97-119: strided conv == per-patch
    linear, plus the bicubic-resized positional table."""
    B, C, H, W = x.shape
    assert H % ph == 0 and W % pw == 0
    w = sd[prefix + 'proj.weight']
    rows = patchify_rows(x, ph, pw)
    tok = rows @ w.reshape(w.shape[0], -1).t() + sd[prefix + 'proj.bias']
    pos = resized_posemb_tokens(sd[prefix + 'pos_emb'], H // ph, W // pw, 'bicubic')
    return tok + pos[None]


def semseg_tokens(x: Tensor, sd: Dict[str, Tensor], prefix: str, ph: int, pw: int, interpolate_class_emb: bool = False,
                  padding_idx: Optional[int] = None) -> Tensor:
    """SemSegInputAdapter.forward, input_adapters.py:215-241: class-id -> embedding ->
    per-patch linear (ph x pw conv) + bilinear-resized positional table.
    interpolate_class_emb (input_adapters.py:192-198): the embedding image is bilinearly resized by 1 / patch (nn.Upsample used as
    a down-sampler: for even patch sizes the mean of the 2 x 2 centre pixels) and projected by a 1 x 1 conv (`proj.1.*`).
    padding_idx (:186): that embedding row receives no gradient."""
    B, H, W = x.shape
    assert H % ph == 0 and W % pw == 0
    emb = F.embedding(x, sd[prefix + 'class_emb.weight'], padding_idx=padding_idx)   # (B,H,W,E)
    emb = emb.permute(0, 3, 1, 2)                            # (B,E,H,W)
    if interpolate_class_emb:
        w = sd[prefix + 'proj.1.weight']
        small = F.interpolate(emb, scale_factor=(1.0 / ph, 1.0 / pw), mode='bilinear')
        tok = small.flatten(2).transpose(1, 2) @ w.reshape(w.shape[0], -1).t() + sd[prefix + 'proj.1.bias']
    else:
        w = sd[prefix + 'proj.weight']
        rows = patchify_rows(emb, ph, pw)
        tok = rows @ w.reshape(w.shape[0], -1).t() + sd[prefix + 'proj.bias']
    pos = resized_posemb_tokens(sd[prefix + 'pos_emb'], H // ph, W // pw, 'bilinear')
    return tok + pos[None]


# --------------------------------------------------------------------------- #
# mask sampling (integer path)                                                #
# --------------------------------------------------------------------------- #
def samples_per_task_from_dirichlet(dist: Tensor, num_encoded_tokens: int) -> Tensor:
    """multimae.py:189 -- round-half-even of p*n, int64."""
    return (dist * num_encoded_tokens).round().long()


def masks_from_noise(samples_per_task: Tensor, task_noise: Sequence[Tensor], all_noise: Tensor,
                     num_encoded_tokens: int):
    """Deterministic core of MultiMAE.generate_random_masks (multimae.py:191-216), with
    the random draws passed in.  Returns (mask_all (B,Ntot) int64 0=visible,
    ids_keep (B,n) int64, ids_restore (B,Ntot) int64).

    Ties in the noise have probability ~0 for the per-task draw; the global key
    (mask + u) is an fp32 sum and can tie with p~1e-2 per sample -- the reference's
    argsort is then implementation-defined; the oracle (and the HIP kernel) break
    ties by lower index first (stable).
    """
    B = all_noise.shape[0]
    pre = []
    for t, noise in enumerate(task_noise):
        n = noise.shape[1]
        order = torch.argsort(noise, dim=1, stable=True)          # order[j] = index of j-th smallest
        # reference quirk: position j is visible iff order[j] < k (not rank[j] < k)
        pre.append(torch.where(order < samples_per_task[:, t:t + 1], 0, 1))
    pre = torch.cat(pre, dim=1)
    keys = pre + all_noise                                            # int64 + f32 -> f32
    ids_shuffle = torch.argsort(keys, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    ids_keep = ids_shuffle[:, :num_encoded_tokens]
    mask_all = torch.ones_like(pre)
    mask_all[:, :num_encoded_tokens] = 0
    mask_all = torch.gather(mask_all, 1, ids_restore)
    return mask_all, ids_keep, ids_restore


def draw_mask_randoms(B: int, tokens_per_task: Sequence[int], alphas, generator=None):
    """Issue the reference's RNG call sequence (multimae.py:187,195,204) on the CPU
    generator: Dirichlet(alphas).sample((B,)), then one rand(B,n_t) per task, then
    rand(B, sum n_t).  With the same seed this reproduces a CPU reference run bit
    for bit."""
    from torch.distributions.dirichlet import Dirichlet
    a = [alphas] * len(tokens_per_task) if isinstance(alphas, float) else list(alphas)
    dist = Dirichlet(torch.Tensor(a)).sample((B,))
    task_noise = [torch.rand(B, n) for n in tokens_per_task]
    all_noise = torch.rand(B, sum(tokens_per_task))
    return dist, task_noise, all_noise


# --------------------------------------------------------------------------- #
# transformer pieces                                                          #
# --------------------------------------------------------------------------- #
def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)                   # biased
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def self_attention(x: Tensor, sd, prefix: str, heads: int) -> Tensor:
    """Attention.forward, multimae_utils.py:170-182.  qkv rows are [q|k|v], head-major."""
    B, N, C = x.shape
    d = C // heads
    qkv = x @ sd[prefix + 'qkv.weight'].t() + sd[prefix + 'qkv.bias']
    qkv = qkv.reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)) * (d ** -0.5)
    a = a.softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return o @ sd[prefix + 'proj.weight'].t() + sd[prefix + 'proj.bias']


def cross_attention(xq: Tensor, ctx: Tensor, sd, prefix: str, heads: int) -> Tensor:
    """CrossAttention.forward, multimae_utils.py:199-214.  kv rows are [k|v]."""
    B, N, C = xq.shape
    M = ctx.shape[1]
    d = C // heads
    q = (xq @ sd[prefix + 'q.weight'].t() + sd[prefix + 'q.bias']).reshape(B, N, heads, d).permute(0, 2, 1, 3)
    kv = (ctx @ sd[prefix + 'kv.weight'].t() + sd[prefix + 'kv.bias']).reshape(B, M, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    a = ((q @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return o @ sd[prefix + 'proj.weight'].t() + sd[prefix + 'proj.bias']


def mlp(x: Tensor, sd, prefix: str) -> Tensor:
    """Mlp.forward, multimae_utils.py:147-155 (dropout p=0)."""
    h = gelu_erf(x @ sd[prefix + 'fc1.weight'].t() + sd[prefix + 'fc1.bias'])
    return h @ sd[prefix + 'fc2.weight'].t() + sd[prefix + 'fc2.bias']


def drop_path(x: Tensor, u: Optional[Tensor], drop_prob: float) -> Tensor:
    """drop_path, multimae_utils.py:105-120, with the uniform draw u (B,) supplied by the caller:
    x.div(keep_prob) * floor(keep_prob + u).  u None or drop_prob 0 = identity."""
    if u is None or drop_prob == 0.0:
        return x
    keep = 1.0 - drop_prob
    mask = (keep + u.to(x.dtype)).floor().view(-1, *([1] * (x.ndim - 1)))
    return x.div(keep) * mask


def block(x: Tensor, sd, prefix: str, heads: int, eps: float, drop_prob: float = 0.0, u=(None, None)) -> Tensor:
    """Block.forward, multimae_utils.py:229-232; u = the two uniform draws (attention branch, MLP branch) of this
    block's DropPath when drop_prob > 0."""
    x = x + drop_path(self_attention(layer_norm(x, sd[prefix + 'norm1.weight'], sd[prefix + 'norm1.bias'], eps),
                                     sd, prefix + 'attn.', heads), u[0], drop_prob)
    x = x + drop_path(mlp(layer_norm(x, sd[prefix + 'norm2.weight'], sd[prefix + 'norm2.bias'], eps), sd, prefix + 'mlp.'), u[1], drop_prob)
    return x


def drop_path_rates(drop_path_rate: float, depth: int):
    """Per-block rates, multimae.py:93: torch.linspace(0, drop_path_rate, depth)."""
    return [float(v) for v in torch.linspace(0, drop_path_rate, depth)]


def encoder(x: Tensor, sd, cfg: OracleConfig, return_all_layers: bool = False, drop_path_rate: float = 0.0, drop_path_u=None):
    """drop_path_u: per block a pair of (B,) uniform draws (or None); rates follow drop_path_rates()."""
    outs = []
    rates = drop_path_rates(drop_path_rate, cfg.depth)
    for i in range(cfg.depth):
        u = (None, None) if drop_path_u is None or drop_path_u[i] is None else drop_path_u[i]
        x = block(x, sd, f'encoder.{i}.', cfg.num_heads, cfg.ln_eps, rates[i], u)
        outs.append(x)
    return outs if return_all_layers else x


# --------------------------------------------------------------------------- #
# output adapter (SpatialOutputAdapter)                                       #
# --------------------------------------------------------------------------- #
def spatial_adapter(enc: Tensor, sd, cfg: OracleConfig, key: str, task: str,
                    tokens_per_task: Dict[str, int], ids_keep: Tensor, ids_restore: Tensor,
                    image_hw: Tuple[int, int], use_xattn: bool = True, use_task_queries: bool = True) -> Tensor:
    """SpatialOutputAdapter.forward + get_queries_and_context, output_adapters.py:183-282.  Queries: the task's own rows of
    the unshuffled context (use_task_queries and the task is an encoder input, :209-212) or mask_token + pos-emb
    (+ the task's embedding, if the adapter has one) on every grid position (:213-220).  use_xattn=False skips the
    cross-attention layer and its MLP (:264-268: x = queries)."""
    p = f'output_adapters.{key}.'
    dom = cfg.domains_by_name[task]
    ph, pw = cfg.patch_hw(dom)
    H, W = image_hw
    nh, nw = H // (dom.stride_level * ph), W // (dom.stride_level * pw)
    B = enc.shape[0]
    G = cfg.num_global_tokens
    D = cfg.dec_dim

    ctx = enc @ sd[p + 'proj_context.weight'].t() + sd[p + 'proj_context.bias']   # :258
    ctx_vis = ctx[:, :-G] if G > 0 else ctx
    n_total = sum(tokens_per_task.values())
    fill = sd[p + 'mask_token'].expand(B, n_total - ctx_vis.shape[1], D)           # :196-198
    full = torch.cat([ctx_vis, fill], dim=1)
    full = torch.gather(full, 1, ids_restore[:, :, None].expand(-1, -1, D))         # :201-202
    # context embeddings: task embedding + bilinear pos-emb, per input task (:160-181)
    embs = []
    pe = resized_posemb_tokens(sd[p + 'pos_emb'], nh, nw, 'bilinear')
    for t, n in tokens_per_task.items():
        assert pe.shape[0] == n
        if p + f'task_embeddings.{t}' in sd:                                         # :166-169 (zeros for a task without an embedding)
            embs.append((sd[p + f'task_embeddings.{t}'].reshape(1, 1, D) + pe[None]).expand(B, n, D))
        else:
            embs.append(pe[None].expand(B, n, D))
    full = full + torch.cat(embs, dim=1)                                             # :207
    if use_task_queries and task in tokens_per_task:
        start = 0
        for t, n in tokens_per_task.items():
            if t == task:
                break
            start += n
        queries = full[:, start:start + tokens_per_task[task]]                       # :209-212
    else:
        queries = sd[p + 'mask_token'].reshape(1, 1, D) + pe[None].expand(B, nh * nw, D)     # :214-217
        if p + f'task_embeddings.{task}' in sd:
            queries = queries + sd[p + f'task_embeddings.{task}'].reshape(1, 1, D)           # :218-220
    ctx2 = torch.gather(full, 1, ids_keep[:, :, None].expand(-1, -1, D))            # :224-225
    if G > 0:
        ctx2 = torch.cat([ctx2, ctx[:, -G:]], dim=1)                                 # :228-230

    eps = cfg.ln_eps
    if use_xattn:
        qn = layer_norm(queries, sd[p + 'query_norm.weight'], sd[p + 'query_norm.bias'], eps)
        cn = layer_norm(ctx2, sd[p + 'context_norm.weight'], sd[p + 'context_norm.bias'], eps)
        x = cross_attention(qn, cn, sd, p + 'decoder.', cfg.dec_heads)               # :265 (no residual)
        x = x + mlp(layer_norm(x, sd[p + 'out_norm.weight'], sd[p + 'out_norm.bias'], eps), sd, p + 'mlp.')  # :266
    else:
        x = queries                                                                  # :268
    for i in range(cfg.dec_depth):
        x = block(x, sd, p + f'decoder_transformer.{i}.', cfg.dec_heads, eps)        # :271
    x = x @ sd[p + 'out_proj.weight'].t() + sd[p + 'out_proj.bias']                  # :274
    C = dom.channels
    x = x.reshape(B, nh, nw, C, ph, pw).permute(0, 3, 1, 4, 2, 5).reshape(B, C, nh * ph, nw * pw)  # :277-280
    return x


# --------------------------------------------------------------------------- #
# losses                                                                      #
# --------------------------------------------------------------------------- #
def _upsampled_mask(mask: Tensor, H: int, W: int, scale: int) -> Tensor:
    nh, nw = H // scale, W // scale
    m = mask.reshape(-1, nh, nw).float()
    return m.repeat_interleave(scale, 1).repeat_interleave(scale, 2)   # nearest upsample by integer factor


def _norm_pix_target(target: Tensor, scale: int) -> Tensor:
    """criterion.py:88-95: per-patch mean / unbiased var over (p1 p2 c), eps in the sqrt."""
    B, C, H, W = target.shape
    nh, nw = H // scale, W // scale
    t = target.reshape(B, C, nh, scale, nw, scale).permute(0, 2, 4, 3, 5, 1).reshape(B, nh * nw, -1)
    mean = t.mean(-1, keepdim=True)
    var = t.var(-1, keepdim=True)
    t = (t - mean) / torch.sqrt(var + 1e-6)
    return t.reshape(B, nh, nw, scale, scale, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, H, W)


def _masked_reduce(per_pixel: Tensor, mask: Optional[Tensor], scale: int) -> Tensor:
    if mask is None:
        return per_pixel.mean()
    if int(mask.sum()) == 0:
        return torch.tensor(0)                                            # criterion.py:100-101 (int64, no grad)
    H, W = per_pixel.shape[-2:]
    m = _upsampled_mask(mask, H, W, scale)
    per_sample = (per_pixel * m).flatten(1).sum(1) / m.flatten(1).sum(1)
    return per_sample.nanmean()


def masked_mse(pred: Tensor, target: Tensor, mask: Optional[Tensor], patch_size: int, stride: int = 1,
               norm_pix: bool = False) -> Tensor:
    """MaskedMSELoss.forward, criterion.py:84-114."""
    scale = patch_size // stride
    if norm_pix:
        target = _norm_pix_target(target, scale)
    err = (pred - target) ** 2
    if mask is None:
        return err.mean()
    return _masked_reduce(err.mean(1), mask, scale)


def masked_l1(pred: Tensor, target: Tensor, mask: Optional[Tensor], patch_size: int, stride: int = 1,
              norm_pix: bool = False) -> Tensor:
    """MaskedL1Loss.forward, criterion.py:141-171."""
    scale = patch_size // stride
    if norm_pix:
        target = _norm_pix_target(target, scale)
    err = (pred - target).abs()
    if mask is None:
        return err.mean()
    return _masked_reduce(err.mean(1), mask, scale)


def masked_ce(logits: Tensor, target: Tensor, mask: Optional[Tensor], patch_size: int, stride: int = 1,
              label_smoothing: float = 0.0) -> Tensor:
    """MaskedCrossEntropyLoss.forward, criterion.py:37-57.  label_smoothing as F.cross_entropy defines it (:47):
    (1 - eps) * nll(target) + eps * mean over the C classes of -log_softmax."""
    scale = patch_size // stride
    lse = torch.logsumexp(logits, dim=1)
    nll = lse - torch.gather(logits, 1, target[:, None]).squeeze(1)
    if label_smoothing:
        nll = (1.0 - label_smoothing) * nll + label_smoothing * (lse - logits.mean(dim=1))
    return _masked_reduce(nll, mask, scale)


# --------------------------------------------------------------------------- #
# whole model                                                                 #
# --------------------------------------------------------------------------- #
def all_input_tokens(x: Dict[str, Tensor], sd, cfg: OracleConfig) -> Dict[str, Tensor]:
    toks = {}
    for d in cfg.in_domains:
        if d.name not in x:
            continue
        ph, pw = cfg.patch_hw(d)
        pre = f'input_adapters.{d.name}.'
        toks[d.name] = (semseg_tokens if d.kind == 'semseg' else image_tokens)(x[d.name], sd, pre, ph, pw)
    return toks


def image_hw_of(x: Dict[str, Tensor], cfg: OracleConfig) -> Tuple[int, int]:
    """multimae.py:298-309."""
    if 'rgb' in x:
        return tuple(x['rgb'].shape[-2:])
    if 'semseg' in x:
        s = cfg.domains_by_name['semseg'].stride_level
        return x['semseg'].shape[-2] * s, x['semseg'].shape[-1] * s
    return tuple(next(iter(x.values())).shape[-2:])


def multimae_forward(x: Dict[str, Tensor], sd, cfg: OracleConfig, ids_keep: Tensor, ids_restore: Tensor,
                     return_intermediates: bool = False, drop_path_rate: float = 0.0, drop_path_u=None):
    """MultiMAE.forward (multimae.py:271-379) with the (ids_keep, ids_restore) pair
    supplied by the caller (see masks_from_noise).  Returns preds keyed like
    cfg.out_tasks; with return_intermediates also the selected input tokens and the
    encoder output."""
    toks = all_input_tokens(x, sd, cfg)
    tokens_per_task = {k: v.shape[1] for k, v in toks.items()}
    B = ids_keep.shape[0]
    cat = torch.cat(list(toks.values()), dim=1)
    sel = torch.gather(cat, 1, ids_keep[:, :, None].expand(-1, -1, cat.shape[2]))          # :343
    g = sd['global_tokens'].expand(B, -1, -1)
    enc_in = torch.cat([sel, g], dim=1)                                                     # :346-347 (global LAST)
    enc = encoder(enc_in, sd, cfg, drop_path_rate=drop_path_rate, drop_path_u=drop_path_u)
    hw = image_hw_of(x, cfg)
    preds = {key: spatial_adapter(enc, sd, cfg, key, task, tokens_per_task, ids_keep, ids_restore, hw)
             for key, task in cfg.out_tasks}
    if return_intermediates:
        return preds, {'enc_in': enc_in, 'enc_out': enc}
    return preds


def multivit_forward_tokens(x: Dict[str, Tensor], sd, cfg: OracleConfig, return_all_layers: bool = False):
    """MultiViT.process_input + encoder (multimae.py:439-487): no masking, all tokens + global."""
    toks = all_input_tokens(x, sd, cfg)
    cat = torch.cat(list(toks.values()), dim=1)
    enc_in = torch.cat([cat, sd['global_tokens'].expand(cat.shape[0], -1, -1)], dim=1)
    return encoder(enc_in, sd, cfg, return_all_layers)


def pretrain_losses(preds: Dict[str, Tensor], targets: Dict[str, Tensor], mask_all: Tensor,
                    cfg: OracleConfig, tokens_per_task: Dict[str, int]) -> Dict[str, Tensor]:
    """Loss wiring of run_pretraining_multimae.py:49-72,321-330,509-520: rgb MSE, depth L1,
    semseg CE, norm_rgb = norm-pix MSE on the rgb target with the rgb mask."""
    masks, s = {}, 0
    for t, n in tokens_per_task.items():
        masks[t] = mask_all[:, s:s + n]
        s += n
    out = {}
    for key, task in cfg.out_tasks:
        d = cfg.domains_by_name[task]
        if d.kind == 'semseg':
            out[key] = masked_ce(preds[key], targets[task], masks[task], cfg.patch_size, d.stride_level)
        elif task == 'depth':
            out[key] = masked_l1(preds[key], targets[task], masks[task], cfg.patch_size, d.stride_level)
        else:
            out[key] = masked_mse(preds[key], targets[task], masks[task], cfg.patch_size, d.stride_level,
                                  norm_pix=(key == 'norm_rgb'))
    return out


def linear_output_adapter(encoder_tokens: Tensor, sd: Dict[str, Tensor], prefix: str = '', use_mean_pooling: bool = True,
                          eps: float = 1e-6) -> Tensor:
    """LinearOutputAdapter.forward (output_adapters.py:341-352): mean over tokens (or the last token), LayerNorm, Linear."""
    x = encoder_tokens.mean(1) if use_mean_pooling else encoder_tokens[:, -1]
    x = F.layer_norm(x, (x.shape[-1],), sd[prefix + 'norm.weight'], sd[prefix + 'norm.bias'], eps)
    return F.linear(x, sd[prefix + 'head.weight'], sd[prefix + 'head.bias'])


def truncated_depth_standardize(depth: Tensor, lo: float = 0.1, hi: float = 0.9, eps: float = 1e-6) -> Tensor:
    """Truncated depth standardisation of the training loop (run_pretraining_multimae.py:487-492): per sample, sort all
    c*h*w values, drop the bottom and top 10 % (slice [int(lo*n), int(hi*n)) of the sorted row), and standardise the whole map
    with that slice's mean and UNBIASED variance:  (x - mean) / sqrt(var + 1e-6)."""
    B = depth.shape[0]
    flat = torch.sort(depth.reshape(B, -1), dim=1)[0]
    n = flat.shape[1]
    trunc = flat[:, int(lo * n):int(hi * n)]
    mean = trunc.mean(dim=1)[:, None, None, None]
    var = trunc.var(dim=1)[:, None, None, None]
    return (depth - mean) / torch.sqrt(var + eps)


def adamw_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], m: Dict[str, Tensor], v: Dict[str, Tensor],
               step: int, lr: float, wd: float, beta1: float = 0.9, beta2: float = 0.95, eps: float = 1e-8):
    """torch.optim.AdamW semantics as the reference uses it (optim_factory.py:138-174:
    decoupled weight decay on EVERY trainable tensor).  In-place on params/m/v."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k, p in params.items():
        g = grads[k]
        p.mul_(1.0 - lr * wd)
        m[k].mul_(beta1).add_(g, alpha=1.0 - beta1)
        v[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m[k], denom, value=-lr / bc1)


# --------------------------------------------------------------------------- #
# convenience: standard configs                                               #
# --------------------------------------------------------------------------- #
def standard_config(domains: Sequence[str], *, patch_size=16, image_size=224, dim_tokens=768, depth=12,
                    num_heads=12, dec_dim=256, dec_depth=2, dec_heads=8, extra_norm_pix=True,
                    num_classes=133, dim_class_emb=64) -> OracleConfig:
    specs = []
    for d in domains:
        if d == 'rgb':
            specs.append(DomainSpec('rgb', 'image', 3, 1))
        elif d == 'depth':
            specs.append(DomainSpec('depth', 'image', 1, 1))
        elif d == 'semseg':
            specs.append(DomainSpec('semseg', 'semseg', num_classes, 4, dim_class_emb))
        else:
            raise ValueError(d)
    outs = [(d, d) for d in domains]
    if extra_norm_pix and 'rgb' in domains:
        outs.append(('norm_rgb', 'rgb'))
    return OracleConfig(specs, outs, patch_size, image_size, dim_tokens, depth, num_heads, 1,
                        dec_dim, dec_depth, dec_heads)
