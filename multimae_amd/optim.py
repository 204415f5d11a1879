"""Fused optimiser step on the flat arenas ("next" row #1 of SURVEY.md section 8f).

Replaces, for the pre-training recipe, utils/native_scaler.py:14-62 (grad-norm, clip / skip) and
torch.optim.AdamW as configured by utils/optim_factory.py:138-174: ONE param group, decoupled
weight decay on EVERY trainable tensor (biases, LayerNorm, tokens included -- the reference's
dict branch never consults no_weight_decay()), betas (0.9, 0.95).

bf16 needs no loss scaling, so there is no GradScaler; a non-finite gradient norm skips the update
on the device (no host sync).  lr / weight_decay are read from ``param_groups[0]`` each step so the
reference's per-iteration cosine tables (run_pretraining_multimae.py:474-480) plug in unchanged.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import nn

from . import engine, ops


class FusedAdamW:
    def __init__(self, model: nn.Module, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.95), eps: float = 1e-8,
                 weight_decay: float = 0.05, clip_grad: Optional[float] = None, skip_grad: Optional[float] = None):
        self.arena = engine.arena_of(model) or engine.ParamArena(model)
        a = self.arena
        n = a.n_trainable
        self.m = torch.zeros(n, device=a.device, dtype=torch.float32)
        self.v = torch.zeros(n, device=a.device, dtype=torch.float32)
        self.param_groups = [dict(lr=lr, weight_decay=weight_decay, lr_scale=1.0, betas=betas, eps=eps)]
        self.clip_grad, self.skip_grad = clip_grad, skip_grad
        self.step_count = 0
        self._sumsq = torch.zeros(1, device=a.device, dtype=torch.float32)
        self._ws = torch.empty(1024, device=a.device, dtype=torch.float32)
        self._scale = torch.ones(1, device=a.device, dtype=torch.float32)
        self._skip = torch.zeros(1, device=a.device, dtype=torch.int32)
        self.grad_norm = torch.zeros(1, device=a.device, dtype=torch.float32)

    def zero_grad(self, set_to_none: bool = False) -> None:
        engine.join_wgrad_streams()
        self.arena.zero_grad()
        self.arena.rebind_grads()

    @torch.no_grad()
    def step(self) -> torch.Tensor:
        """One AdamW update; returns the (device) gradient 2-norm, as the reference's loss_scaler does."""
        a, g = self.arena, self.param_groups[0]
        n = a.n_trainable
        engine.join_wgrad_streams()
        ops.sumsq(a.grad, self._sumsq, self._ws)
        # scalar bookkeeping on 1-element device tensors (no host sync)
        torch.sqrt(self._sumsq, out=self.grad_norm)
        finite = torch.isfinite(self.grad_norm)
        self._skip.copy_((~finite).to(torch.int32))
        if self.skip_grad is not None:
            self._skip.add_((self.grad_norm >= self.skip_grad).to(torch.int32))
        if self.clip_grad is not None:
            torch.clamp(self.clip_grad / (self.grad_norm + 1e-6), max=1.0, out=self._scale)
        b1, b2 = g['betas']
        shadow = a.shadow[:n] if a.shadow is not None else None
        cap = engine.capturing()
        if cap is not None:
            # hipGraph capture: the step-dependent scalars come from HBM, refreshed by the host before every replay
            def hyper():
                self.step_count += 1
                gg = self.param_groups[0]
                return torch.tensor([gg['lr'] * gg.get('lr_scale', 1.0), gg['weight_decay'], 1.0 - b1 ** self.step_count,
                                     math.sqrt(1.0 - b2 ** self.step_count)], dtype=torch.float32)
            ops.adamw_dev(a.param[:n], a.grad, self.m, self.v, cap.add(hyper, a.device), beta1=b1, beta2=b2, eps=g['eps'],
                          grad_scale=self._scale if self.clip_grad is not None else None, skip_flag=self._skip, shadow=shadow)
        else:
            self.step_count += 1
            ops.adamw(a.param[:n], a.grad, self.m, self.v, lr=g['lr'] * g.get('lr_scale', 1.0), beta1=b1, beta2=b2, eps=g['eps'],
                      weight_decay=g['weight_decay'], step=self.step_count,
                      grad_scale=self._scale if self.clip_grad is not None else None, skip_flag=self._skip, shadow=shadow)
        if shadow is not None:
            a.mark_shadow_fresh()
        return self.grad_norm

    def state_dict(self):
        return dict(m=self.m, v=self.v, step=self.step_count, param_groups=self.param_groups)

    def load_state_dict(self, sd):
        self.m.copy_(sd['m']); self.v.copy_(sd['v'])
        self.step_count = sd['step']
        self.param_groups = sd['param_groups']
