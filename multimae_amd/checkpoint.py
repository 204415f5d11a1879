"""Checkpoint compatibility with the reference.

* save / auto-resume in the reference's file format (utils/checkpoint.py:75-152): ``checkpoint-<epoch>.pth`` holding
  ``{'model', 'optimizer', 'epoch', 'scaler', 'args'}`` -- the model under the reference's 351 state-dict keys (the engine
  keeps them), the optimiser in ``torch.optim.AdamW.state_dict()`` layout (state index i = i-th trainable
  parameter in ``named_parameters()`` order, which is how utils/optim_factory.py:138-149 builds the pre-training optimiser),
  so a run can move between the reference and this engine at a checkpoint boundary in either direction;
* the MultiMAE <-> timm-ViT key maps of tools/multimae2vit_converter.py:14-51 and tools/vit2multimae_converter.py:14-32;
* position-embedding resizing on load (utils/pos_embed.py:44-58).

State-dict plumbing on host/any-device tensors -- none of this is on the training hot path.
"""
from __future__ import annotations

import glob
import math
import os
import re
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------ key maps
def multimae_to_vit(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """MultiMAE -> timm ViT (one global token becomes cls_token; its pos-embed slot is zero).  tools/multimae2vit_converter.py:14-32."""
    out = {}
    for k, v in sd.items():
        if k == 'global_tokens':
            out['cls_token'] = v
        elif k == 'input_adapters.rgb.pos_emb':
            pe = v.flatten(2).transpose(1, 2)                          # b d h w -> b (h w) d
            out['pos_embed'] = F.pad(pe, (0, 0, 1, 0, 0, 0), mode='constant', value=0.0)
        elif k == 'input_adapters.rgb.proj.weight':
            out['patch_embed.proj.weight'] = v
        elif k == 'input_adapters.rgb.proj.bias':
            out['patch_embed.proj.bias'] = v
        elif 'encoder' in k:
            out[k.replace('encoder', 'blocks')] = v
    return out


def multimae_to_vitmultimae(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """MultiMAE -> timm-style ViTMultiMAE (any number of global tokens).  tools/multimae2vit_converter.py:34-51."""
    out = {}
    for k, v in sd.items():
        if k == 'global_tokens':
            out['global_tokens'] = v
        elif k == 'input_adapters.rgb.pos_emb':
            out['pos_embed'] = v.flatten(2).transpose(1, 2)
        elif k == 'input_adapters.rgb.proj.weight':
            out['patch_embed.proj.weight'] = v
        elif k == 'input_adapters.rgb.proj.bias':
            out['patch_embed.proj.bias'] = v
        elif 'encoder' in k:
            out[k.replace('encoder', 'blocks')] = v
    return out


def vit_to_multimae(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """timm ViT -> MultiMAE; the cls token absorbs its position embedding.  tools/vit2multimae_converter.py:14-32.
    (Like the reference, the sum is made in place on the tensor the input dict holds under 'cls_token'.)"""
    out = {'global_tokens': sd['cls_token']}
    for k, v in sd.items():
        if k == 'pos_embed':
            n = int(math.sqrt(v.shape[1]))
            out['global_tokens'] += v[:, 0]
            out['input_adapters.rgb.pos_emb'] = v[:, 1:].reshape(v.shape[0], n, n, v.shape[2]).permute(0, 3, 1, 2)
        elif k == 'patch_embed.proj.weight':
            out['input_adapters.rgb.proj.weight'] = v
        elif k == 'patch_embed.proj.bias':
            out['input_adapters.rgb.proj.bias'] = v
        elif 'blocks.' in k:
            out[k.replace('blocks.', 'encoder.')] = v
    return out


def interpolate_pos_embed_multimae(model, checkpoint_model: Dict[str, Tensor]) -> None:
    """Bicubic-resize every ``input_adapters.<domain>.pos_emb`` of the checkpoint to the model's grid, in place
    (utils/pos_embed.py:44-58)."""
    pat = re.compile(r'input_adapters\.(.*)\.pos_emb')
    for key in [k for k in checkpoint_model if pat.match(k)]:
        domain = pat.match(key).group(1)
        adapter = getattr(model.input_adapters, domain, None) if hasattr(model, 'input_adapters') else None
        if adapter is None:
            continue
        pe = checkpoint_model[key]
        new_h, new_w = adapter.pos_emb.shape[-2:]
        if pe.shape[-2:] != (new_h, new_w):
            checkpoint_model[key] = F.interpolate(pe, size=(new_h, new_w), mode='bicubic', align_corners=False)


# ------------------------------------------------------------------------------------------ optimiser state
def _trainable(opt):
    """Trainable tensors in ``named_parameters()`` order -- the order ``torch.optim.AdamW`` numbers its state by when the
    reference builds it from ``[p for n, p in model.named_parameters() if p.requires_grad]`` (utils/optim_factory.py:138-149).
    The arena itself is laid out in backward-readiness order (engine.ParamArena(groups=...)); offsets do the translation."""
    a = opt.arena
    return [(n, a.offsets[n], a.sizes[n], a._params[n].shape) for n in a.param_order if a.trainable[n]]


def optimizer_state_to_torch(opt) -> dict:
    """FusedAdamW moments -> ``torch.optim.AdamW.state_dict()`` layout (one param group, state per trainable tensor)."""
    g = opt.param_groups[0]
    state = {}
    for i, (n, off, size, shape) in enumerate(_trainable(opt)):
        state[i] = {'step': torch.tensor(float(opt.step_count)), 'exp_avg': opt.m[off:off + size].view(shape).clone(),
                    'exp_avg_sq': opt.v[off:off + size].view(shape).clone()}
    group = {'lr': g['lr'], 'betas': tuple(g['betas']), 'eps': g['eps'], 'weight_decay': g['weight_decay'], 'amsgrad': False,
             'lr_scale': g.get('lr_scale', 1.0), 'params': list(range(len(state)))}
    return {'state': state, 'param_groups': [group]}


def optimizer_state_from_torch(opt, sd: dict) -> None:
    """Load a ``torch.optim.AdamW`` state dict (e.g. from a reference checkpoint) into FusedAdamW's flat moments.  Extra param
    groups (the reference's loss-balancer group) are ignored; the schedule values are taken from group 0."""
    tr = _trainable(opt)
    ids = sd['param_groups'][0]['params']
    if len(ids) != len(tr):
        raise ValueError(f'optimizer state has {len(ids)} tensors in group 0, the model has {len(tr)} trainable tensors')
    step = 0
    with torch.no_grad():
        for pid, (n, off, size, shape) in zip(ids, tr):
            st = sd['state'].get(pid)
            if st is None:
                opt.m[off:off + size].zero_(); opt.v[off:off + size].zero_()
                continue
            if tuple(st['exp_avg'].shape) != tuple(shape):
                raise ValueError(f'{n}: moment shape {tuple(st["exp_avg"].shape)} != parameter shape {tuple(shape)}')
            opt.m[off:off + size].copy_(st['exp_avg'].reshape(-1))
            opt.v[off:off + size].copy_(st['exp_avg_sq'].reshape(-1))
            step = max(step, int(float(st['step'])))
    opt.step_count = step
    g0 = sd['param_groups'][0]
    for k in ('lr', 'weight_decay', 'eps', 'lr_scale'):
        if k in g0:
            opt.param_groups[0][k] = g0[k]
    if 'betas' in g0:
        opt.param_groups[0]['betas'] = tuple(g0['betas'])


# ------------------------------------------------------------------------------------------ files
def save_checkpoint(output_dir: str, epoch, model, optimizer, args=None, loss_balancer=None, is_main_process: bool = True) -> Optional[str]:
    """utils/checkpoint.py:75-97 (torch.amp branch): ``<output_dir>/checkpoint-<epoch>.pth`` written by the main process.
    'scaler' holds an empty dict -- bf16 training has no GradScaler (the reference's disabled scaler also saves {})."""
    if not is_main_process:
        return None
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, f'checkpoint-{epoch}.pth')
    to_save = {'model': {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
               'optimizer': _cpu(optimizer_state_to_torch(optimizer) if hasattr(optimizer, 'arena') else optimizer.state_dict()),
               'epoch': epoch, 'scaler': {}, 'args': args}
    if loss_balancer is not None:
        to_save['loss_balancer'] = loss_balancer.state_dict()
    torch.save(to_save, path)
    return path


def _cpu(o):
    if isinstance(o, torch.Tensor):
        return o.detach().cpu()
    if isinstance(o, dict):
        return {k: _cpu(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_cpu(v) for v in o)
    return o


def latest_checkpoint(output_dir: str) -> Optional[str]:
    """The auto-resume rule of utils/checkpoint.py:103-113: the highest numeric ``checkpoint-<N>.pth``."""
    best = -1
    for ckpt in glob.glob(os.path.join(output_dir, 'checkpoint-*.pth')):
        t = ckpt.split('-')[-1].split('.')[0]
        if t.isdigit():
            best = max(best, int(t))
    return os.path.join(output_dir, f'checkpoint-{best}.pth') if best >= 0 else None


def load_checkpoint(path: str, model, optimizer=None, resize_pos_emb: bool = True) -> int:
    """Resume (utils/checkpoint.py:115-131): model weights, and -- if present -- optimiser state and epoch.  Accepts
    checkpoints written by the reference or by save_checkpoint.  Returns the epoch to start from."""
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    sd = ckpt['model'] if 'model' in ckpt else ckpt
    if resize_pos_emb:
        interpolate_pos_embed_multimae(model, sd)
    model.load_state_dict(sd)
    start = 0
    if optimizer is not None and 'optimizer' in ckpt and 'epoch' in ckpt:
        if hasattr(optimizer, 'arena'):
            optimizer_state_from_torch(optimizer, ckpt['optimizer'])
        else:
            optimizer.load_state_dict(ckpt['optimizer'])
        start = ckpt['epoch'] + 1
    return start
