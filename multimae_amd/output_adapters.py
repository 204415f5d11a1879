"""Output adapters (mirror of the reference's ``multimae/output_adapters.py`` for the
pre-training path).

Reference: SpatialOutputAdapter output_adapters.py:33-282.  Same constructor, same parameter
names and registration order (mask_token, pos_emb, task_embeddings, decoder.{q,kv,proj},
{context,query,out}_norm, mlp, decoder_transformer.*, out_proj, proj_context registered last
in init()).  forward() is ONE autograd node (functions.SpatialAdapterFn) whose forward and
backward are fixed HIP kernel sequences; the (B, N_total, D) mask-token tensor of the
reference is never materialised.

LinearOutputAdapter (output_adapters.py:285-356, the classification head of the MultiViT fine-tuning forward,
SURVEY.md section 8f row 4) is built on the same kernels.  The dense-prediction fine-tuning heads (Segmenter / ConvNeXt /
DPT adapters) are out of scope for the pre-training hot path.
"""
from __future__ import annotations

from functools import partial
from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import nn

from . import engine
from .functions import SpatialAdapterFn, TokenMeanFn
from .multimae_utils import (Block, CrossAttention, LayerNorm, Linear, Mlp, _as_hip_norm, _cfg, bias_or_zero, block_params,
                             build_2d_sincos_posemb, pair, trunc_normal_)


class SpatialOutputAdapter(nn.Module):
    """Cross-attention decoder for spatial outputs (output_adapters.py:33-142)."""

    def __init__(self, num_channels: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens_enc: Optional[int] = None, dim_tokens: int = 256, depth: int = 0, learnable_pos_emb: int = False,
                 image_size: Union[int, Tuple[int]] = 224, mlp_ratio: int = 4.0, num_heads: int = 8, qkv_bias: bool = True,
                 drop_rate: float = 0.0, attn_drop_rate: float = 0.0, drop_path_rate: float = 0.0,
                 norm_layer: nn.Module = partial(nn.LayerNorm, eps=1e-6), use_task_queries: bool = True,
                 task: Optional[str] = None, context_tasks: Optional[list] = None, use_xattn: bool = True):
        super().__init__()
        self.num_channels = num_channels
        self.stride_level = stride_level
        self.patch_size_full = pair(patch_size_full)
        self.dim_tokens_enc = dim_tokens_enc
        self.dim_tokens = dim_tokens
        self.learnable_pos_emb = learnable_pos_emb
        self.image_size = pair(image_size)
        self.use_task_queries = use_task_queries
        self.task = task
        self.use_xattn = use_xattn
        self.num_heads = num_heads
        self.depth = depth
        self._drop_rates = (float(attn_drop_rate), float(drop_rate))
        if learnable_pos_emb:
            # (the reference allocates this table as (1, h, w, D) but resizes it with F.interpolate as if it were (1, D, h, w),
            # output_adapters.py:108-111,172: not a behaviour worth mirroring)
            raise NotImplementedError('learnable decoder positional embeddings are not built in the HIP engine')
        # (drop_rate / attn_drop_rate > 0: the nn.Dropout sites of the cross attention and of the decoder blocks run as element-wise
        # passes between the per-kernel sequence of the adapter; 0 -- every shipped configuration -- keeps the one-call adapter)
        # (drop_path_rate > 0: stochastic depth in decoder_transformer, output_adapters.py:126-132 -- the per-sample scales are drawn in
        # forward() in the reference's order and folded into the blocks' residual adds, as in the encoder)

        self.P_H = max(1, self.patch_size_full[0] // stride_level)
        self.P_W = max(1, self.patch_size_full[1] // stride_level)

        if context_tasks is not None:
            self.task_embeddings = nn.ParameterDict(
                {t: nn.Parameter(torch.zeros(1, 1, self.dim_tokens)) for t in context_tasks})
            for embedding in self.task_embeddings.values():
                trunc_normal_(embedding, std=0.02)
        else:
            self.task_embeddings = None

        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.dim_tokens))

        h_posemb = self.image_size[0] // (self.stride_level * self.P_H)
        w_posemb = self.image_size[1] // (self.stride_level * self.P_W)
        self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h=h_posemb, w=w_posemb, embed_dim=self.dim_tokens),
                                    requires_grad=False)

        self._eps = _as_hip_norm(norm_layer, self.dim_tokens).eps
        if self.use_xattn:           # output_adapters.py:114-123: without it the queries go straight into decoder_transformer
            self.decoder = CrossAttention(dim=self.dim_tokens, num_heads=num_heads, qkv_bias=qkv_bias,
                                          attn_drop=attn_drop_rate, proj_drop=drop_rate)
            self.context_norm = _as_hip_norm(norm_layer, self.dim_tokens)
            self.query_norm = _as_hip_norm(norm_layer, self.dim_tokens)
            self.out_norm = _as_hip_norm(norm_layer, self.dim_tokens)
            mlp_hidden_dim = int(self.dim_tokens * mlp_ratio)
            self.mlp = Mlp(in_features=self.dim_tokens, hidden_features=mlp_hidden_dim)

        if depth > 0:
            dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
            self.decoder_transformer = nn.Sequential(*[
                Block(dim=self.dim_tokens, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                      attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer) for i in range(depth)])
        else:
            self.decoder_transformer = nn.Identity()

        self.dim_patch = self.num_channels * self.P_H * self.P_W
        self.out_proj = Linear(self.dim_tokens, self.dim_patch)

        if self.dim_tokens_enc is not None:
            self.init(dim_tokens_enc=dim_tokens_enc)

    def init(self, dim_tokens_enc: int = 768):
        self.dim_tokens_enc = dim_tokens_enc
        self.proj_context = Linear(self.dim_tokens_enc, self.dim_tokens)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_emb', 'mask_token', 'task_embeddings'}

    def _pos_tokens(self, nh: int, nw: int) -> torch.Tensor:
        """bilinear-resized decoder pos-emb as (nh*nw, D) (output_adapters.py:172-173), cached."""
        p = self.pos_emb
        key = (nh, nw, p._version, p.device, p.data_ptr())
        cache = self.__dict__.setdefault('_pos_cache', {})
        t = cache.get(key)
        if t is None:
            cache.clear()
            with torch.no_grad():
                r = F.interpolate(p.detach(), size=(nh, nw), mode='bilinear', align_corners=False)
                t = r[0].flatten(1).t().contiguous().float()
            cache[key] = t
        return t

    def _params(self, in_tasks):
        te = []
        for t in in_tasks:
            if self.task_embeddings is not None and t in self.task_embeddings:
                te.append(self.task_embeddings[t])
            else:
                te.append(None)
        if not self.use_xattn:
            ps = [self.mask_token, *te] + [None] * 16
            if self.depth > 0:
                for blk in self.decoder_transformer:
                    ps += block_params(blk)
            return ps + [self.out_proj.weight, self.out_proj.bias, self.proj_context.weight, self.proj_context.bias]
        d = self.decoder
        ps = [self.mask_token, *te, d.q.weight, bias_or_zero(d.q, getattr(d, '_zero_q_bias', None)), d.kv.weight,
              bias_or_zero(d.kv, getattr(d, '_zero_kv_bias', None)), d.proj.weight, d.proj.bias,
              self.context_norm.weight, self.context_norm.bias, self.query_norm.weight, self.query_norm.bias,
              self.out_norm.weight, self.out_norm.bias, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight,
              self.mlp.fc2.bias]
        if self.depth > 0:
            for blk in self.decoder_transformer:
                ps += block_params(blk)
        ps += [self.out_proj.weight, self.out_proj.bias, self.proj_context.weight, self.proj_context.bias]
        return ps

    def forward(self, encoder_tokens: torch.Tensor, input_info: Dict, ids_keep: torch.Tensor, ids_restore: torch.Tensor,
                act_dtype: Optional[torch.dtype] = None, on_done=None, f32_gemm: str = 'exact',
                encoder_tokens_act: Optional[torch.Tensor] = None):
        """(B, n_keep+G, D_enc) encoder tokens -> (B, C, H_t, W_t) prediction (output_adapters.py:236-282)."""
        assert self.dim_tokens_enc is not None, 'Need to call init(dim_tokens_enc) function first'
        if self.task_embeddings is None:
            raise AttributeError('SpatialOutputAdapter built with context_tasks=None has no task_embeddings '
                                 '(the reference raises here too, output_adapters.py:166)')
        H, W = input_info['image_size']
        nh = H // (self.stride_level * self.P_H)
        nw = W // (self.stride_level * self.P_W)
        in_tasks = list(input_info['tasks'].keys())
        # queries (output_adapters.py:208-220): the task's own rows of the unshuffled context, or -- when the task is not an
        # encoder input (e.g. --in_domains rgb --out_domains rgb-depth-semseg) or use_task_queries=False -- mask_token + pos
        # (+ the task's embedding if the adapter has one) on every grid position
        task_queries = bool(self.use_task_queries and self.task in input_info['tasks'])
        offs = [0]
        for t in in_tasks:
            n = input_info['tasks'][t]['num_tokens']
            assert n == nh * nw, 'all tasks must share the decoder token grid (output_adapters.py:175)'
            offs.append(offs[-1] + n)
        G = input_info.get('num_global_tokens', 0)
        cfg = _cfg(self, act=act_dtype, heads=self.num_heads, eps=self._eps, use_xattn=self.use_xattn, task_offsets=offs,
                   q_task=in_tasks.index(self.task) if task_queries else -1, G=G, D=self.dim_tokens, pos=self._pos_tokens(nh, nw), depth=self.depth,
                   C=self.num_channels, nh=nh, nw=nw, ph=self.P_H, pw=self.P_W, on_done=on_done, f32_gemm=f32_gemm,
                   enc_act=encoder_tokens_act)
        if self.depth > 0:                                   # two draws per block with a rate > 0, block by block, all up front (see _stack_drop_path on the stream order)
            from .multimae_utils import _stack_drop_path
            cfg.dp = _stack_drop_path(list(self.decoder_transformer), encoder_tokens.shape[0], encoder_tokens.device)
        if self.training and any(r > 0. for r in self._drop_rates):
            cfg.drops = self._drop_rates
        params = self._params(in_tasks)
        if not task_queries and self.task_embeddings is not None and self.task in self.task_embeddings:
            # the query rows are mask_token + task_embeddings[task] + pos: one vector added to every row, passed in the mask-token
            # slot (the kernels use that slot for nothing else in this mode); autograd splits its gradient between the two
            params[0] = self.mask_token + self.task_embeddings[self.task]
        img, token = SpatialAdapterFn.apply(cfg, encoder_tokens, ids_keep.contiguous(), ids_restore.contiguous(), *params)
        if getattr(cfg, 'lazy_fill', None) is not None or cfg.handle is not None:
            # written when first read (the masked losses never read it); already written when the adapter ran eagerly -- the wrapper
            # then only carries the patch-row side channel through clone() (torch DDP's output sink)
            from .lazy import LazyPrediction
            img = LazyPrediction.wrap(img, getattr(cfg, 'lazy_fill', None))
            cfg.lazy_fill = None
        if cfg.handle is not None:
            img._mmae_pat = cfg.handle       # a masked loss applied to exactly this tensor works on the patch rows (criterion.py)
        return img


class LinearOutputAdapter(nn.Module):
    """Linear classification head (output_adapters.py:285-356): mean over the encoder tokens (or the last = global token),
    LayerNorm, Linear.  Same constructor, parameter names (norm.*, head.*) and seeded initialisation as the reference."""

    def __init__(self, num_classes: int, dim_tokens_enc: Optional[int] = None, use_mean_pooling: bool = True,
                 norm_layer: nn.Module = partial(nn.LayerNorm, eps=1e-6), init_scale: float = 1.0):
        super().__init__()
        self.num_classes = num_classes
        self.dim_tokens_enc = dim_tokens_enc
        self.use_mean_pooling = use_mean_pooling
        self.norm_layer = norm_layer
        self.init_scale = init_scale
        if self.dim_tokens_enc is not None:
            self.init(dim_tokens_enc=dim_tokens_enc)

    def init(self, dim_tokens_enc: int = 768):
        self.dim_tokens_enc = dim_tokens_enc
        self.norm = _as_hip_norm(self.norm_layer, self.dim_tokens_enc)
        self.head = Linear(dim_tokens_enc, self.num_classes) if self.num_classes > 0 else nn.Identity()
        self.apply(self._init_weights)
        if self.num_classes > 0:
            self.head.weight.data.mul_(self.init_scale)
            self.head.bias.data.mul_(self.init_scale)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=''):
        self.num_classes = num_classes
        self.init(dim_tokens_enc=self.dim_tokens_enc)

    def forward(self, encoder_tokens: torch.Tensor, **kwargs):
        if self.use_mean_pooling:
            x = TokenMeanFn.apply(encoder_tokens)
        else:
            x = encoder_tokens[:, -1]                       # the global token is appended last (multimae.py:344-347)
        return self.head(self.norm(x))
