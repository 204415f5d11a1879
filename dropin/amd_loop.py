"""The three services of run_pretraining_multimae.py that keep the reference's loop host-bound on the MI355X engine, replaced by
the engine's own -- so that the UNMODIFIED ``train_one_epoch`` (run_pretraining_multimae.py:472-560) runs at the native step time.

Why: through the plain drop-in seam (``dropin/multimae``) the reference loop is functionally right but 25 % slower than the native
loop (profiles/r04_dropin_host_profile.txt: 38 ms of host work per 29 ms GPU step): 351 ``AccumulateGrad`` nodes + DDP's reducer
hooks (13.4 ms), the 345 ``torch.norm`` calls of ``get_grad_norm_`` (9.9 ms), DDP's pre-forward (5.8 ms), ``GradScaler.unscale_``
(1.8 ms).  All four are per-PARAMETER services; the engine's gradient arena makes each of them one pass / one call:

  reference (file:line)                                         here
  ------------------------------------------------------------  ---------------------------------------------------------------
  DistributedDataParallel(model, find_unused_parameters=True)    wrap_model(model, args): model.build_arena() (flat, gradient-
      run_pretraining_multimae.py:380-382                        readiness-ordered arenas), direct gradients, dist.GradAllReducer
                                                                 + dist.attach: buckets all-reduced over RCCL while backward runs
  create_optimizer(args, {'model': .., 'balancer': ..})          create_optimizer(args, model): optim.FusedAdamW -- the ONE group
      :389-390, utils/optim_factory.py:138-174                   the dict branch builds (weight decay on every trainable tensor)
  NativeScaler()  (GradScaler + unscale_ + get_grad_norm_)       LossScaler(): same call signature and return value (the gradient
      :391, utils/native_scaler.py:14-62                         2-norm); backward, reducer.finish(), ONE fused library call for
                                                                 norm + clip / skip + non-finite guard + AdamW.  bf16 needs no
                                                                 loss scale: state_dict() reports scale 1.0

The patch that wires them in is ``dropin/run_pretraining_multimae.patch`` (three call sites).  Everything else in the script --
argument parsing, cosine tables, the per-iteration ``param_group['lr']`` assignment (:474-480), ``train_one_epoch`` itself,
logging, ``utils.save_model`` -- stays the reference's.  ``tests/test_dropin_loop_gpu.py`` runs the loop body with these services at
the bench geometry against the native loop.
"""
from __future__ import annotations

from typing import Optional

import torch

import multimae_amd as M
from multimae_amd import dist as mdist
from multimae_amd.optim import FusedAdamW


def wrap_model(model: torch.nn.Module, args=None, bucket_mb: float = 64.0):
    """Replaces the DistributedDataParallel wrap (run_pretraining_multimae.py:380-382).  Returns ``(model, reducer)``: the model is
    NOT wrapped (``model_without_ddp is model``; checkpoints keep the reference's keys), the reducer is None outside
    torch.distributed.  Rank 0's parameters are broadcast as DDP's constructor does (the script seeds every rank differently, :300-302)."""
    arena = model.build_arena()
    M.engine.set_direct_grads(True)              # backward writes straight into the gradient arena: no AccumulateGrad per parameter
    on_gpu = arena.param.is_cuda                 # (a CPU dry run of the host logic has no streams)
    M.engine.set_adapter_streams(on_gpu)
    M.engine.set_wgrad_stream(on_gpu)
    reducer = None
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        mdist.broadcast_parameters(arena)
        reducer = mdist.GradAllReducer.for_arena(arena, bucket_mb=bucket_mb)
        reducer.gemm_cu_reserve = 16
    model._amd_reducer = reducer
    return model, reducer


def create_optimizer(args, model: torch.nn.Module, reducer=None) -> FusedAdamW:
    """Replaces utils.optim_factory.create_optimizer(args, {'model': model, 'balancer': balancer}) for the pre-training recipe
    (``--opt adamw``, ``--task_balancer none``: the balancer group is empty, utils/optim_factory.py:138-155)."""
    if getattr(args, 'opt', 'adamw').lower().split('_')[-1] != 'adamw':
        raise ValueError('amd_loop.create_optimizer: the fused step is AdamW (--opt adamw); use the reference factory for other optimisers')
    if getattr(args, 'task_balancer', 'none') != 'none':
        raise ValueError('amd_loop.create_optimizer: --task_balancer uncertainty has trainable balancer weights outside the arena; '
                         'use the reference factory (the drop-in seam still works, at the reference loop\'s speed)')
    kw = dict(lr=args.lr, weight_decay=args.weight_decay)
    if getattr(args, 'opt_eps', None) is not None:
        kw['eps'] = args.opt_eps
    if getattr(args, 'opt_betas', None) is not None:
        kw['betas'] = tuple(args.opt_betas)
    opt = FusedAdamW(model, **kw)
    reducer = reducer if reducer is not None else getattr(model, '_amd_reducer', None)
    if reducer is not None:
        mdist.attach(model, reducer, opt)
    opt._amd_reducer = reducer
    return opt


class LossScaler:
    """``utils.NativeScalerWithGradNormCount`` for the engine: same ``__call__`` signature and return value.

    utils/native_scaler.py:20-40 is ``scale(loss).backward(); unscale_; norm = clip_grad_norm_ | get_grad_norm_ (skip if >= skip_grad);
    step; update`` -- clip and skip exclusive, clip first.  Here: ``loss.backward()`` (bf16 exponent range: no loss scale), the
    gradient exchange's ``finish()``, then ``FusedAdamW.step(loss)``: norm, the same clip / skip rule, the non-finite guards and
    AdamW in one library call with every decision on the device.  The returned norm is a device scalar (``metric_logger.update``
    reads it when it prints)."""
    state_dict_key = 'amp_scaler'

    def __init__(self, enabled: bool = True):
        self.enabled = enabled

    def __call__(self, loss, optimizer: FusedAdamW, clip_grad: Optional[float] = None, skip_grad: Optional[float] = None,
                 parameters=None, create_graph: bool = False, update_grad: bool = True):
        loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        reducer = getattr(optimizer, '_amd_reducer', None)
        if reducer is not None:
            reducer.finish()
        optimizer.clip_grad, optimizer.skip_grad = clip_grad, skip_grad
        return optimizer.step(loss)

    def state_dict(self):
        return {'scale': 1.0, 'growth_factor': 2.0, 'backoff_factor': 0.5, 'growth_interval': 2000, '_growth_tracker': 0}

    def load_state_dict(self, state_dict):
        pass
