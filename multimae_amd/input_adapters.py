"""Input adapters (mirror of the reference's ``multimae/input_adapters.py`` API).

Reference: PatchedInputAdapter input_adapters.py:27-119, SemSegInputAdapter :122-241.
The modules hold the reference's parameters (``proj`` Conv2d weight/bias, ``pos_emb``,
``class_emb``) under the same names; the arithmetic runs in the gather-first HIP path
(functions.EmbedFn): only the tokens that survive masking are ever projected.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import nn

from . import engine
from .functions import AvgClassEmbFn, EmbedFn, _Cfg
from .multimae_utils import _cfg, build_2d_sincos_posemb, pair, trunc_normal_


class _PosEmbMixin:
    """Resized positional table, hoisted off the per-step path: the reference calls
    F.interpolate on every forward (input_adapters.py:113,235); the result only depends on
    (pos_emb values, N_H, N_W), so it is cached per (shape, parameter version, device)."""

    _interp_mode = 'bicubic'

    def pos_tokens(self, nh: int, nw: int) -> torch.Tensor:
        p = self.pos_emb
        if p.requires_grad and torch.is_grad_enabled():      # learnable: resize inside autograd, every forward
            mode = dict(mode='bicubic', align_corners=False) if self._interp_mode == 'bicubic' else dict(mode='bilinear')
            return F.interpolate(p, size=(nh, nw), **mode)[0].flatten(1).t().contiguous().float()
        key = (nh, nw, p._version, p.device, p.data_ptr())
        cache = self.__dict__.setdefault('_pos_cache', {})
        t = cache.get(key)
        if t is None:
            cache.clear()
            with torch.no_grad():
                if self._interp_mode == 'bicubic':
                    r = F.interpolate(p.detach(), size=(nh, nw), mode='bicubic', align_corners=False)
                else:
                    r = F.interpolate(p.detach(), size=(nh, nw), mode='bilinear')
                t = r[0].flatten(1).t().contiguous().float()          # (nh*nw, D)
            cache[key] = t
        return t


class PatchedInputAdapter(nn.Module, _PosEmbMixin):
    """Adapter for spatial inputs (images / feature maps): patches -> tokens.

    Same constructor as the reference (input_adapters.py:41-65)."""
    _interp_mode = 'bicubic'
    kind = 0

    def __init__(self, num_channels: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens: Optional[int] = None, sincos_pos_emb: bool = True, learnable_pos_emb: bool = False,
                 image_size: Union[int, Tuple[int]] = 224):
        super().__init__()
        self.num_channels = num_channels
        self.stride_level = stride_level
        self.patch_size_full = pair(patch_size_full)
        self.dim_tokens = dim_tokens
        self.sincos_pos_emb = sincos_pos_emb
        self.learnable_pos_emb = learnable_pos_emb
        self.image_size = pair(image_size)
        self.num_patches = (self.image_size[0] // patch_size_full) * (self.image_size[1] // patch_size_full)
        self.P_H = max(1, self.patch_size_full[0] // stride_level)
        self.P_W = max(1, self.patch_size_full[1] // stride_level)
        if self.dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def init(self, dim_tokens: int = 768):
        self.dim_tokens = dim_tokens
        h_posemb = self.image_size[0] // (self.stride_level * self.P_H)
        w_posemb = self.image_size[1] // (self.stride_level * self.P_W)
        if self.sincos_pos_emb:
            self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h=h_posemb, w=w_posemb, embed_dim=self.dim_tokens),
                                        requires_grad=self.learnable_pos_emb)
        else:
            self.pos_emb = nn.Parameter(torch.zeros(1, self.dim_tokens, h_posemb, w_posemb))
            trunc_normal_(self.pos_emb, std=0.02)
        # parameter container only (weight (D,C,P_H,P_W) / bias (D,)); never called as a conv
        self.proj = nn.Conv2d(in_channels=self.num_channels, out_channels=self.dim_tokens,
                              kernel_size=(self.P_H, self.P_W), stride=(self.P_H, self.P_W))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_emb'}

    # -- engine interface ----------------------------------------------------------------
    def source_channels(self) -> int:
        return self.num_channels

    def check_input(self, x: torch.Tensor) -> Tuple[int, int]:
        B, C, H, W = x.shape
        assert self.dim_tokens is not None, 'Need to call init(dim_tokens) function first'
        assert (H % self.P_H == 0) and (W % self.P_W == 0), \
            f'Image sizes {H}x{W} must be divisible by patch sizes {self.P_H}x{self.P_W}'
        return H // self.P_H, W // self.P_W

    def embed_args(self, x: torch.Tensor):
        """(task descriptor for EmbedFn, tensors (data, weight, bias, class_emb|None))"""
        nh, nw = self.check_input(x)
        H, W = x.shape[-2:]
        desc = dict(kind=0, C=self.num_channels, H=H, W=W, ph=self.P_H, pw=self.P_W, K=self.num_channels * self.P_H * self.P_W,
                    n_patches=nh * nw, pos=self.pos_tokens(nh, nw))
        return desc, (x.float(), self.proj.weight, self.proj.bias, None, desc['pos'])

    def forward(self, x):
        """All tokens of this modality: (B, N_H*N_W, dim_tokens)  (input_adapters.py:97-119)."""
        return _embed_all(self, x)


class SemSegInputAdapter(nn.Module, _PosEmbMixin):
    """Adapter for semantic-segmentation class maps (input_adapters.py:122-241)."""
    _interp_mode = 'bilinear'
    kind = 1

    def __init__(self, num_classes: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens: Optional[int] = None, sincos_pos_emb: int = True, learnable_pos_emb: int = False,
                 image_size: Union[int, Tuple[int]] = 224, dim_class_emb: int = 64, interpolate_class_emb: bool = False,
                 emb_padding_idx: int = None):
        super().__init__()
        self.num_classes = num_classes
        self.stride_level = stride_level
        self.patch_size_full = pair(patch_size_full)
        self.dim_tokens = dim_tokens
        self.sincos_pos_emb = sincos_pos_emb
        self.learnable_pos_emb = learnable_pos_emb
        self.image_size = pair(image_size)
        self.dim_class_emb = dim_class_emb
        self.interpolate_class_emb = interpolate_class_emb
        self.emb_padding_idx = emb_padding_idx
        if self.emb_padding_idx is not None:
            self.num_classes += 1
        self.P_H = max(1, self.patch_size_full[0] // stride_level)
        self.P_W = max(1, self.patch_size_full[1] // stride_level)
        if self.dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def init(self, dim_tokens: int = 768):
        self.dim_tokens = dim_tokens
        h_posemb = self.image_size[0] // (self.stride_level * self.P_H)
        w_posemb = self.image_size[1] // (self.stride_level * self.P_W)
        if self.sincos_pos_emb:
            self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h=h_posemb, w=w_posemb, embed_dim=self.dim_tokens),
                                        requires_grad=bool(self.learnable_pos_emb))
        else:
            self.pos_emb = nn.Parameter(torch.zeros(1, self.dim_tokens, h_posemb, w_posemb))
            trunc_normal_(self.pos_emb, std=0.02)
        self.class_emb = nn.Embedding(num_embeddings=self.num_classes, embedding_dim=self.dim_class_emb,
                                      padding_idx=self.emb_padding_idx)
        trunc_normal_(self.class_emb.weight, std=0.02)
        if self.interpolate_class_emb:
            # parameter containers only, as in the reference (input_adapters.py:192-198): proj.1.weight (D, E, 1, 1) / proj.1.bias
            self.proj = nn.Sequential(nn.Upsample(scale_factor=(1 / self.P_H, 1 / self.P_W), mode='bilinear'),
                                      nn.Conv2d(in_channels=self.dim_class_emb, out_channels=self.dim_tokens, kernel_size=1, stride=1))
        else:
            self.proj = nn.Conv2d(in_channels=self.dim_class_emb, out_channels=self.dim_tokens,
                                  kernel_size=(self.P_H, self.P_W), stride=(self.P_H, self.P_W))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_emb', 'class_emb'}

    def check_input(self, x: torch.Tensor) -> Tuple[int, int]:
        B, H, W = x.shape
        assert self.dim_tokens is not None, 'Need to call init(dim_tokens) function first'
        assert (H % self.P_H == 0) and (W % self.P_W == 0), \
            f'Image sizes {H}x{W} must be divisible by patch sizes {self.P_H}x{self.P_W}'
        return H // self.P_H, W // self.P_W

    def embed_args(self, x: torch.Tensor):
        nh, nw = self.check_input(x)
        H, W = x.shape[-2:]
        pad = -1 if self.emb_padding_idx is None else int(self.emb_padding_idx)
        if self.interpolate_class_emb:
            # the resized class-embedding image enters the patch embedding as an E-channel image with 1 x 1 patches
            img = AvgClassEmbFn.apply(x.long(), self.class_emb.weight, self.P_H, self.P_W, pad)
            desc = dict(kind=0, C=self.dim_class_emb, H=nh, W=nw, ph=1, pw=1, K=self.dim_class_emb, n_patches=nh * nw, pos=self.pos_tokens(nh, nw))
            return desc, (img, self.proj[1].weight, self.proj[1].bias, None, desc['pos'])
        desc = dict(kind=1, C=self.dim_class_emb, H=H, W=W, ph=self.P_H, pw=self.P_W, K=self.dim_class_emb * self.P_H * self.P_W,
                    n_patches=nh * nw, pos=self.pos_tokens(nh, nw), pad_idx=self.emb_padding_idx)
        return desc, (x.long(), self.proj.weight, self.proj.bias, self.class_emb.weight, desc['pos'])

    def forward(self, x):
        return _embed_all(self, x)


def embed_tokens(owner: nn.Module, adapters: Dict[str, nn.Module], x: Dict[str, torch.Tensor], sel: torch.Tensor,
                 global_tokens: Optional[torch.Tensor], on_done=None) -> torch.Tensor:
    """Gather-first embedding of the selected tokens of several modalities (+ global tokens last).
    sel: (B, n_sel) int64 indices into the concatenated token axis (= ids_keep)."""
    tasks, tens, offs, k_off = [], [], [0], 0
    D = None
    for name, ad in adapters.items():
        desc, t = ad.embed_args(x[name])
        desc['k_off'] = k_off
        k_off += desc['K']
        offs.append(offs[-1] + desc['n_patches'])
        tasks.append(desc)
        tens += list(t)
        D = ad.dim_tokens
    G = 0 if global_tokens is None else global_tokens.shape[1]
    cfg = _cfg(owner, tasks=tasks, task_offsets=offs, D=D, G=G, on_done=on_done)
    return EmbedFn.apply(cfg, sel, global_tokens, *tens)


def _embed_all(adapter: nn.Module, x: torch.Tensor) -> torch.Tensor:
    nh, nw = adapter.check_input(x)
    B = x.shape[0]
    sel = torch.arange(nh * nw, device=x.device, dtype=torch.int64).unsqueeze(0).expand(B, -1).contiguous()
    return embed_tokens(adapter, {'_': adapter}, {'_': x}, sel, None)
