"""MultiMAE / MultiViT (mirror of the reference's ``multimae/multimae.py`` API) on the
MI355X-native engine.

Reference: MultiMAE multimae/multimae.py:40-379 (init :61-116, Dirichlet sampler :148-218,
forward :271-379), factories :382-416, MultiViT :419-502, factories :505-539.

Differences in HOW (not in results):
  * gather-first: masks are drawn first, then only the kept patches are embedded
    (the reference embeds all 3x196 patches, then torch.gather's 98 of them);
  * the mask sampler is one kernel (the reference: 5 argsorts + gathers);
  * the encoder stack and each output adapter are single autograd nodes with hand-written
    backward; all weights live in one flat HBM arena (engine.ParamArena).
"""
from __future__ import annotations

import itertools
import math
from collections import OrderedDict
from functools import partial
from typing import Dict, List, Optional, Union

import torch
from torch import nn
from torch.distributions.dirichlet import Dirichlet

from . import engine, ops
from .input_adapters import embed_tokens
from .multimae_utils import Block, LayerNorm, run_blocks, set_root, trunc_normal_
from .registry import register_model

__all__ = ['pretrain_multimae_base', 'pretrain_multimae_large', 'multivit_base', 'multivit_large']


class MultiMAE(nn.Module):
    """MultiMAE: Multi-task Multi-modal Masked Autoencoder (masking forward pass).

    Constructor arguments as in the reference (multimae.py:61-73)."""

    def __init__(self, input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]],
                 num_global_tokens: int = 1, dim_tokens: int = 768, depth: int = 12, num_heads: int = 12,
                 mlp_ratio: float = 4.0, qkv_bias: bool = True, drop_rate: float = 0.0, attn_drop_rate: float = 0.0,
                 drop_path_rate: float = 0.0, norm_layer: nn.Module = partial(nn.LayerNorm, eps=1e-6)):
        super().__init__()
        for adapter in input_adapters.values():
            adapter.init(dim_tokens=dim_tokens)
        self.input_adapters = nn.ModuleDict(input_adapters)
        if output_adapters is not None:
            for adapter in output_adapters.values():
                adapter.init(dim_tokens_enc=dim_tokens)
            self.output_adapters = nn.ModuleDict(output_adapters)
        else:
            self.output_adapters = None

        self.num_global_tokens = num_global_tokens
        self.global_tokens = nn.Parameter(torch.zeros(1, num_global_tokens, dim_tokens))
        trunc_normal_(self.global_tokens, std=0.02)

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.encoder = nn.Sequential(*[
            Block(dim=dim_tokens, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer) for i in range(depth)])

        # same initialisation walk as the reference (multimae.py:100-125): xavier on every Linear,
        # then per-matrix fan for fused qkv / kv, xavier on the flattened patch projections
        self.apply(self._init_weights)
        for name, m in self.named_modules():
            if isinstance(m, nn.Linear):
                if 'qkv' in name:
                    val = math.sqrt(6. / float(m.weight.shape[0] // 3 + m.weight.shape[1]))
                    nn.init.uniform_(m.weight, -val, val)
                elif 'kv' in name:
                    val = math.sqrt(6. / float(m.weight.shape[0] // 2 + m.weight.shape[1]))
                    nn.init.uniform_(m.weight, -val, val)
            if isinstance(m, nn.Conv2d):
                if '.proj' in name:
                    w = m.weight.data
                    nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        set_root(self)
        self._grad_ready_cb = None

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_num_layers(self):
        return len(self.encoder)

    @torch.jit.ignore
    def no_weight_decay(self):
        no_wd_set = {'global_tokens'}
        for task, adapter in self.input_adapters.items():
            if hasattr(adapter, 'no_weight_decay'):
                no_wd_set |= {f'input_adapters.{task}.{name}' for name in adapter.no_weight_decay()}
        for task, adapter in self.output_adapters.items():
            if hasattr(adapter, 'no_weight_decay'):
                no_wd_set |= {f'output_adapters.{task}.{name}' for name in adapter.no_weight_decay()}
        return no_wd_set

    # -- engine management ------------------------------------------------------------------
    def build_arena(self) -> 'engine.ParamArena':
        """Move every parameter into one flat HBM arena (idempotent; call after .to(device))."""
        a = engine.arena_of(self)
        if a is None:
            # gradient-readiness order (see engine.ParamArena): autograd runs the output adapters' backward in reverse
            # forward order, then the encoder from its last block down; the embedding gradients come last
            groups = []
            if self.output_adapters is not None:
                groups += [f'output_adapters.{d}.' for d in reversed(list(self.output_adapters))]
            groups += [f'encoder.{l}.' for l in reversed(range(len(self.encoder)))]
            a = engine.ParamArena(self, groups=groups)
        return a

    # -- mask sampling ----------------------------------------------------------------------
    def sample_alphas(self, B: int, n_tasks: int, alphas: float = 1.0, eps: float = 1e-5):
        """Uniformly choose a non-empty task subset per sample, Dirichlet over it (multimae.py:148-162)."""
        choices = torch.Tensor([list(i) for i in itertools.product([0, 1], repeat=n_tasks)][1:])
        pick = torch.randint(0, len(choices), (B,))
        return torch.index_select(choices, 0, pick) * torch.tensor(alphas) + eps

    def generate_random_masks(self, input_tokens: Dict[str, Union[torch.Tensor, int]], num_encoded_tokens: int,
                              alphas: Union[float, List[float]] = 1.0, sample_tasks_uniformly: bool = False,
                              batch_size: Optional[int] = None, device=None):
        """Dirichlet token sampling (multimae.py:164-218).

        ``input_tokens`` maps task -> token tensor (only B and the token count are read, as in the
        reference) or task -> token count (then pass batch_size/device).  RNG call sequence is the
        reference's: Dirichlet on the CPU generator, then one torch.rand(B, n_t) per task and one
        torch.rand(B, N_total) on the device; the argsort/gather chain runs as ONE kernel."""
        vals = list(input_tokens.values())
        if isinstance(vals[0], torch.Tensor):
            B, device = vals[0].shape[0], vals[0].device
            counts = [v.shape[1] for v in vals]
        else:
            B, counts = batch_size, [int(v) for v in vals]
        alphas = [alphas] * len(counts) if isinstance(alphas, float) else alphas

        def draw():                     # host side of the sampler: Dirichlet on the CPU generator -> tokens per task
            if sample_tasks_uniformly:
                dist = Dirichlet(self.sample_alphas(B, len(counts), alphas=alphas)).sample()
            else:
                dist = Dirichlet(torch.Tensor(alphas)).sample((B,))
            return (dist * num_encoded_tokens).round().long().contiguous()       # :189
        samples_per_task = engine.host_input(draw, device)
        task_noise = torch.cat([torch.rand(B, n, device=device) for n in counts], dim=1)   # :195
        all_noise = torch.rand(B, sum(counts), device=device)                    # :204
        offs = [0]
        for n in counts:
            offs.append(offs[-1] + n)
        mask_all, ids_keep, ids_restore = ops.mask_sample(samples_per_task, task_noise, all_noise, offs, num_encoded_tokens)
        task_masks = {d: m for d, m in zip(input_tokens.keys(), torch.split(mask_all, counts, dim=1))}
        return task_masks, ids_keep, ids_restore

    @staticmethod
    def make_mask(N_H, N_W, xy_idxs, full_tasks=[], indicate_visible=True, flatten=True, device='cuda'):
        """Task masks from lists of un-masked (x, y) patch coordinates (multimae.py:220-248)."""
        out = {}
        for k, v in xy_idxs.items():
            m = torch.ones(N_H, N_W, device=device)
            v = torch.as_tensor(v, dtype=torch.long)
            if len(v) > 0:
                m[v[:, 1], v[:, 0]] = 0
            out[k] = m
        for t in full_tasks:
            out[t][:] = 0
        if not indicate_visible:
            out = {k: 1 - v for k, v in out.items()}
        if flatten:
            out = {k: v.flatten().unsqueeze(0) for k, v in out.items()}
        return out

    def generate_input_info(self, input_task_tokens, image_size):
        """Per-task token ranges (multimae.py:250-269).  Values may be tensors or token counts."""
        info = OrderedDict()
        i = 0
        info['tasks'] = {}
        for domain, t in input_task_tokens.items():
            n = t.shape[1] if isinstance(t, torch.Tensor) else int(t)
            info['tasks'][domain] = {'num_tokens': n, 'has_2d_posemb': True, 'start_idx': i, 'end_idx': i + n}
            i += n
        info['image_size'] = image_size
        info['num_task_tokens'] = i
        info['num_global_tokens'] = self.num_global_tokens
        return info

    # -- forward ----------------------------------------------------------------------------
    def _image_size(self, x):
        if 'rgb' in x:
            B, _, H, W = x['rgb'].shape
        elif 'semseg' in x:
            B, H, W = x['semseg'].shape
            H *= self.input_adapters['semseg'].stride_level
            W *= self.input_adapters['semseg'].stride_level
        else:
            B, _, H, W = list(x.values())[0].shape
        return B, H, W

    def _token_counts(self, x):
        counts = OrderedDict()
        for d, t in x.items():
            if d in self.input_adapters:
                nh, nw = self.input_adapters[d].check_input(t)
                counts[d] = nh * nw
        return counts

    def forward(self, x: Union[Dict[str, torch.Tensor], torch.Tensor], mask_inputs: bool = True,
                task_masks: Dict[str, torch.Tensor] = None, num_encoded_tokens: int = 128,
                alphas: Union[float, List[float]] = 1.0, sample_tasks_uniformly: bool = False,
                fp32_output_adapters: List[str] = []):
        """(preds: {task: (B,C,H,W)}, task_masks: {task: (B,N) int64, 0 = visible}); without output
        adapters (encoder_tokens, task_masks).  Signature and semantics: multimae.py:271-379."""
        with engine.forward_scope(self):
            return self._forward(x, mask_inputs, task_masks, num_encoded_tokens, alphas, sample_tasks_uniformly, fp32_output_adapters)

    def _forward(self, x, mask_inputs, task_masks, num_encoded_tokens, alphas, sample_tasks_uniformly, fp32_output_adapters):
        x = {'rgb': x} if isinstance(x, torch.Tensor) else x
        B, H, W = self._image_size(x)
        counts = self._token_counts(x)
        dev = next(iter(x.values())).device
        ops._require_gpu(next(iter(x.values())), 'model input')
        input_info = self.generate_input_info(counts, image_size=(H, W))

        if mask_inputs:
            if num_encoded_tokens is None:
                num_encoded_tokens = self.num_encoded_tokens            # AttributeError, as in the reference (:322)
        else:
            num_encoded_tokens = sum(counts.values())

        if task_masks is None:
            task_masks, ids_keep, ids_restore = self.generate_random_masks(
                counts, num_encoded_tokens, alphas=alphas, sample_tasks_uniformly=sample_tasks_uniformly,
                batch_size=B, device=dev)
        else:
            # notebook path (multimae.py:335-338): 0/1 masks -> stable ranks; number kept = zeros in the
            # whole batch (reference quirk: only meaningful for B = 1)
            mask_all = torch.cat([task_masks[t] for t in counts], dim=1).to(dev)
            n_keep = int((mask_all == 0).sum())
            offs = [0, mask_all.shape[1]]
            zeros = torch.zeros(mask_all.shape, device=dev, dtype=torch.float32)
            _, ids_keep, ids_restore = ops.mask_sample(torch.zeros((B, 1), dtype=torch.int64), zeros, mask_all.float(), offs, n_keep)

        adapters = OrderedDict((d, self.input_adapters[d]) for d in counts)
        tokens = embed_tokens(self, adapters, x, ids_keep, self.global_tokens if self.num_global_tokens > 0 else None,
                              on_done=self._embed_done_cb())

        encoder_tokens = run_blocks(self.encoder, tokens, root=self, on_layer_done=self._layer_done_cb(),
                                    bwd_chunk=getattr(self, '_bwd_chunk_layers', 1), mx=engine.mx_encoder())

        if self.output_adapters is None:
            return encoder_tokens, task_masks

        preds = {}
        # The output adapters are independent: run each on its own HIP stream so their many small kernels
        # (K = 256 GEMMs, 2048-head attention, LayerNorms) fill each other's fill/drain bubbles.  autograd
        # replays each node's backward on its forward stream, so the backward passes overlap the same way.
        streams = self._adapter_streams(len(self.output_adapters)) if engine.adapter_streams() else None
        main = torch.cuda.current_stream() if streams is not None else None
        # every bf16 adapter starts from the same bf16 copy of the encoder output: cast it once, before the streams fork
        enc_bf16 = None
        if engine.act_dtype() == torch.bfloat16 and any(d not in fp32_output_adapters for d in self.output_adapters):
            Be, Ne, De = encoder_tokens.shape
            enc_bf16 = ops.cast(encoder_tokens.detach().contiguous().view(Be * Ne, De), torch.bfloat16)
        # one autograd node in front of the adapters: their gradients with respect to the shared encoder tokens are summed in ONE pass
        # (ops.add_n, index order) instead of autograd's chain of n - 1 at::native adds (round 6)
        enc_for = _EncoderFanOut.apply(encoder_tokens, len(self.output_adapters)) if (
            engine.encoder_fanout() and encoder_tokens.is_cuda and encoder_tokens.requires_grad and len(self.output_adapters) > 1 and torch.is_grad_enabled()) else None
        for i, domain in enumerate(self.output_adapters):
            # reference: adapters listed in fp32_output_adapters run with autocast disabled (:367-377);
            # here they run on the exact-f32 MFMA path
            # here: f32 activations; in bf16 speed mode their GEMMs run with fp16 operands ("f16": TF32's 11-bit significand -- what
            # the reference's fp32 adapters got on A100 with torch 1.10's allow_tf32 default) or as split-bf16 ("x3", ~16 bits),
            # engine.set_fp32_adapter_gemm; in the fp32 parity mode on the exact-f32 MFMA path
            fp32 = domain in fp32_output_adapters
            speed = engine.act_dtype() == torch.bfloat16
            kw = dict(encoder_tokens=encoder_tokens if enc_for is None else enc_for[i], input_info=input_info, ids_keep=ids_keep, ids_restore=ids_restore,
                      act_dtype=torch.float32 if fp32 else None, on_done=self._adapter_done_cb(domain),
                      encoder_tokens_act=None if fp32 else enc_bf16,
                      f32_gemm=engine.fp32_adapter_gemm() if (fp32 and speed and engine.fp32_adapter_gemm() in ('x3', 'f16', 'h16')) else 'exact')
            if streams is None:
                preds[domain] = self.output_adapters[domain](**kw)
            else:
                st = streams[i]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    preds[domain] = self.output_adapters[domain](**kw)
                # encoder_tokens / enc_bf16 need no record_stream: main waits for every adapter stream below, so whatever
                # reuses their memory later is ordered behind the adapters' reads.  (Recording would be harmful: the tokens
                # are a view of the stack's 8 GB activation slab, and a block with a recorded foreign stream is not reusable
                # until the GPU has passed the recorded point -- with the host a few steps ahead every step allocated a fresh
                # slab: 55 instead of 38.5 ms per step, measured.)
                preds[domain].record_stream(main)
        if streams is not None:
            for st in streams:
                main.wait_stream(st)
        return preds, task_masks

    def _adapter_streams(self, n):
        ss = self.__dict__.setdefault('_mmae_streams', [])
        while len(ss) < n:
            ss.append(torch.cuda.Stream())
        return ss[:n]

    # hooks used by dist.GradAllReducer to launch bucketed all-reduces while backward is still running
    def _layer_done_cb(self):
        cb = self._grad_ready_cb
        return None if cb is None else (lambda l: cb(f'encoder.{l}'))

    def _adapter_done_cb(self, domain):
        cb = self._grad_ready_cb
        return None if cb is None else (lambda: cb(f'output_adapters.{domain}'))

    def _embed_done_cb(self):
        cb = self._grad_ready_cb
        if cb is None:
            return None

        def done():                     # arena order of the tail: registration order (global_tokens, then the input adapters)
            cb('global_tokens')
            cb('input_adapters')
        return done


class _EncoderFanOut(torch.autograd.Function):
    """encoder_tokens -> n aliases of it (one per output adapter, multimae.py:352-381); backward: the adapters' n gradients summed by one kernel."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        if all(g.dtype == torch.float32 and g.is_cuda and g.numel() % 4 == 0 for g in gs) and len(gs) <= 8:
            return ops.add_n(gs), None
        out = gs[0]
        for g in gs[1:]:
            out = out + g
        return out, None


@register_model
def pretrain_multimae_base(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], **kwargs):
    return MultiMAE(input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=768, depth=12, num_heads=12,
                    mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


@register_model
def pretrain_multimae_large(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], **kwargs):
    return MultiMAE(input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=1024, depth=24, num_heads=16,
                    mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


class MultiViT(MultiMAE):
    """MultiViT: MultiMAE without masking (multimae.py:419-502)."""

    def process_input(self, x):
        x = {'rgb': x} if isinstance(x, torch.Tensor) else x
        B, H, W = self._image_size(x)
        counts = self._token_counts(x)
        dev = next(iter(x.values())).device
        input_info = self.generate_input_info(counts, image_size=(H, W))
        n = sum(counts.values())
        sel = torch.arange(n, device=dev, dtype=torch.int64).unsqueeze(0).expand(B, -1).contiguous()
        adapters = OrderedDict((d, self.input_adapters[d]) for d in counts)
        tokens = embed_tokens(self, adapters, x, sel, self.global_tokens if self.num_global_tokens > 0 else None)
        return tokens, input_info

    def forward(self, x: Union[Dict[str, torch.Tensor], torch.Tensor], return_all_layers=False, **kwargs):
        with engine.forward_scope(self):
            return self._forward_vit(x, return_all_layers)

    def _forward_vit(self, x, return_all_layers):
        input_tokens, input_info = self.process_input(x)
        encoder_tokens = run_blocks(self.encoder, input_tokens, root=self, all_layers=return_all_layers, mx=engine.mx_encoder())
        if self.output_adapters is None:
            return encoder_tokens
        return {domain: self.output_adapters[domain](encoder_tokens=encoder_tokens, input_info=input_info)
                for domain in self.output_adapters}


@register_model
def multivit_base(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], **kwargs):
    return MultiViT(input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=768, depth=12, num_heads=12,
                    mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


@register_model
def multivit_large(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], **kwargs):
    return MultiViT(input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=1024, depth=24, num_heads=16,
                    mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
