"""Fused optimiser step on the flat arenas ("next" row #1 of SURVEY.md section 8f).

Replaces, for the pre-training recipe, utils/native_scaler.py:14-62 (grad-norm, clip / skip) and
torch.optim.AdamW as configured by utils/optim_factory.py:138-174: ONE param group, decoupled
weight decay on EVERY trainable tensor (biases, LayerNorm, tokens included -- the reference's
dict branch never consults no_weight_decay()), betas (0.9, 0.95).

bf16 needs no loss scaling, so there is no GradScaler; a non-finite gradient norm (or loss) skips the update
on the device (no host sync) and does not advance Adam's step counter, which lives on the device too.  lr / weight_decay are read from ``param_groups[0]`` each step so the
reference's per-iteration cosine tables (run_pretraining_multimae.py:474-480) plug in unchanged.
"""
from __future__ import annotations

import math
import time
from typing import Optional, Tuple

import torch
from torch import nn

from . import engine, ops


class FusedAdamW(torch.optim.Optimizer):
    """A ``torch.optim.Optimizer`` (one param group holding every trainable tensor in ``named_parameters()`` order, exactly the
    group the reference's dict branch builds, utils/optim_factory.py:138-149), so the reference's loop services drive it
    unchanged: ``for g in optimizer.param_groups: g['lr'] = table[it] * g['lr_scale']`` (run_pretraining_multimae.py:474-480),
    ``GradScaler.unscale_(optimizer)`` / ``GradScaler.step(optimizer)`` / ``clip_grad_norm_`` as utils/native_scaler.py:20-40
    calls them -- ``step()`` then runs the one fused library call over the arena.  The parameters' ``.grad`` must be the arena
    views (``zero_grad()`` re-binds them; AccumulateGrad / DDP write into them in place).

    state (device, mmae_opt_desc): ``_state`` f32[8] = [sum of squares, gradient norm, applied gradient scale, lr, weight
    decay, 1 - beta1^t, sqrt(1 - beta2^t), -]; ``_istate`` i32[8] = [skip flag of the last step, t = updates applied, steps
    with a non-finite loss, skipped steps, steps skipped on GradScaler's found_inf (AMP overflow), steps with a non-finite
    gradient norm, -, -].  The step counter lives on the device: a skipped iteration (non-finite gradient
    norm or loss, skip_grad) does not advance Adam's t -- the reference never calls optimizer.step() for it
    (utils/native_scaler.py:27-31, GradScaler.step)."""

    # torch.amp.GradScaler.step(): hands over `found_inf` / `grad_scale` as DEVICE tensors (attributes set around the step() call) instead
    # of deciding on the host with found_inf.item() -- the one synchronisation per iteration of the reference's loss-scaler sequence
    # (utils/native_scaler.py:33) that kept the host from running ahead of the GPU through the drop-in path
    _step_supports_amp_scaling = True

    def __init__(self, model: nn.Module, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.95), eps: float = 1e-8,
                 weight_decay: float = 0.05, clip_grad: Optional[float] = None, skip_grad: Optional[float] = None):
        self.arena = engine.arena_of(model) or engine.ParamArena(model)
        a = self.arena
        n = a.n_trainable
        params = [a._params[nm] for nm in a.param_order if a.trainable[nm]]
        super().__init__([dict(params=params, lr_scale=1.0)], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.m = torch.zeros(n, device=a.device, dtype=torch.float32)
        self.v = torch.zeros(n, device=a.device, dtype=torch.float32)
        self.clip_grad, self.skip_grad = clip_grad, skip_grad
        self.grad_prescale = 1.0          # 1 / world_size when the gradient arena holds rank SUMS (dist.GradAllReducer sets it)
        self._state = torch.zeros(8, device=a.device, dtype=torch.float32)
        self._istate = torch.zeros(8, device=a.device, dtype=torch.int32)
        self._ws = torch.empty(1024, device=a.device, dtype=torch.float32)
        self.grad_norm = self._state[1:2]
        self.max_steps_in_flight = 2      # the host may enqueue at most this many steps ahead of the GPU (see step())
        self._step_events = []
        self.host_wait_s = 0.0            # time the host spent blocked by that bound (not launch work; bench.py subtracts it)

    # Adam's t.  Reading it synchronises with the device: checkpoints and tests only, never inside the step.
    @property
    def step_count(self) -> int:
        return int(self._istate[1])

    @step_count.setter
    def step_count(self, t: int) -> None:
        self._istate[1] = int(t)

    def counters(self, detail: bool = False) -> dict:
        """{'steps', 'nonfinite_loss', 'skipped'} (one host read; poll at logging time to mirror the reference's
        isfinite(loss) exit, run_pretraining_multimae.py:529-531, without a per-step synchronisation).  ``detail=True`` adds
        'amp_overflow' (steps GradScaler's found_inf skipped -- NOT a diverged loss) and 'nonfinite_grad' (steps
        whose gradient norm was inf / NaN: an fp16-storage adapter that overflowed lands here)."""
        c = self._istate.tolist()
        out = dict(steps=c[1], nonfinite_loss=c[2], skipped=c[3])
        if detail:
            out.update(amp_overflow=c[4], nonfinite_grad=c[5])
        return out

    def zero_grad(self, set_to_none: bool = False) -> None:
        """One memset of the gradient arena; ``p.grad`` stays (or becomes again) the arena view, whatever ``set_to_none`` says."""
        engine.join_wgrad_streams()
        self.arena.zero_grad()
        self.arena.rebind_grads()

    @torch.no_grad()
    def step(self, closure=None, *, loss: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One AdamW update in ONE library call (mmae_opt_step); returns the (device) gradient 2-norm, as the reference's
        loss_scaler does.  ``torch.optim.Optimizer`` signature: ``step(closure)`` re-evaluates the model first.  lr /
        weight_decay are read from ``param_groups[0]`` AS THEY ARE: the reference loop already writes
        ``g['lr'] = table[it] * g['lr_scale']`` (run_pretraining_multimae.py:474-480), so ``lr_scale`` is not applied again
        here.  ``loss``: optional device scalar (keyword, or -- the engine-native loop -- the first positional
        argument as a Tensor); a non-finite value skips the update and is counted (``counters()``)."""
        if isinstance(closure, torch.Tensor):            # FusedAdamW.step(loss)
            loss, closure = closure, None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        a, g = self.arena, self.param_groups[0]
        n = a.n_trainable
        engine.join_wgrad_streams()
        b1, b2 = g['betas']
        shadow = a.shadow[:n] if a.shadow is not None else None
        cap = engine.capturing()
        lrwd = None
        if cap is not None:
            # hipGraph capture: the schedule values come from HBM, refreshed by the host before every replay
            def hyper():
                gg = self.param_groups[0]
                return torch.tensor([gg['lr'], gg['weight_decay']], dtype=torch.float32)
            lrwd = cap.add(hyper, a.device)
        if loss is not None:
            loss = loss.detach().float().reshape(1)
        # torch.amp.GradScaler.step() (_step_supports_amp_scaling): device scalars, consumed by the library call itself -- the overflow
        # verdict skips the update under its own counter, the loss scale (set only when unscale_() was not called, i.e. the arena still
        # holds scaled gradients) is folded into the step's gradient multiply instead of a pass over the arena
        found_inf, grad_scale = getattr(self, 'found_inf', None), getattr(self, 'grad_scale', None)
        if found_inf is not None:
            found_inf = found_inf.to(a.grad.device).float().reshape(1)
        if grad_scale is not None:
            grad_scale = grad_scale.to(a.grad.device).float().reshape(1)
        ops.opt_step(a.param[:n], a.grad, self.m, self.v, self._state, self._istate, self._ws, lr=g['lr'],
                     weight_decay=g['weight_decay'], beta1=b1, beta2=b2, eps=g['eps'], clip_grad=self.clip_grad, skip_grad=self.skip_grad,
                     grad_prescale=self.grad_prescale, lrwd_dev=lrwd, loss_dev=loss, shadow=shadow, found_inf_dev=found_inf,
                     grad_scale_dev=grad_scale)
        if shadow is not None:
            a.mark_shadow_fresh()
        if cap is None and a.param.is_cuda and self.max_steps_in_flight:
            # bound the host's lead: nothing in the step synchronises, so without this the host runs as far ahead as the launch
            # queues allow and every step's activation slabs (tens of GB) stay allocated at once
            ev = torch.cuda.Event()
            ev.record()
            self._step_events.append(ev)
            if len(self._step_events) > self.max_steps_in_flight:
                t0 = time.perf_counter()
                self._step_events.pop(0).synchronize()
                self.host_wait_s += time.perf_counter() - t0
        return self.grad_norm

    def state_dict(self):
        """Flat native form (checkpoint.optimizer_state_to_torch gives the torch.optim.AdamW layout of reference checkpoints)."""
        return dict(m=self.m, v=self.v, step=self.step_count,
                    param_groups=[{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups])

    def load_state_dict(self, sd):
        if 'm' not in sd:                                # a torch.optim.AdamW state dict (reference checkpoint)
            from .checkpoint import optimizer_state_from_torch
            optimizer_state_from_torch(self, sd)
            return
        self.m.copy_(sd['m']); self.v.copy_(sd['v'])
        self.step_count = sd['step']
        for g, new in zip(self.param_groups, sd['param_groups']):
            g.update({k: v for k, v in new.items() if k != 'params'})
