"""Shared test helpers: golden loading, model builders for the engine and the oracle config."""
import json
import os

import numpy as np
import torch

import multimae_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_mini():
    z = np.load(os.path.join(GOLD, 'mini_fwd_bwd.npz'))
    g = lambda pre: {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    out = dict(sd=g('sd/'), grad=g('grad/'), x=g('x/'), pred=g('pred/'), loss={k: float(v) for k, v in g('loss/').items()},
               noise=g('noise/'), mask=g('mask/'))
    for k in ('dirichlet', 'noise_all', 'ids_keep', 'ids_restore', 'enc_in', 'enc_out'):
        out[k] = torch.from_numpy(z[k])
    return out


def load_masks_base():
    z = np.load(os.path.join(GOLD, 'masks_base.npz'))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def load_scalars():
    with open(os.path.join(GOLD, 'scalars.json')) as f:
        return json.load(f)


MINI = dict(doms=['rgb', 'depth', 'semseg'], P=8, S=32, B=3, nvis=12, dim=128, depth=2, heads=2, dec_dim=64, dec_depth=1,
            dec_heads=2, class_emb=16)


def mini_oracle_cfg():
    m = MINI
    return orc.standard_config(m['doms'], patch_size=m['P'], image_size=m['S'], dim_tokens=m['dim'], depth=m['depth'],
                               num_heads=m['heads'], dec_dim=m['dec_dim'], dec_depth=m['dec_depth'], dec_heads=m['dec_heads'],
                               dim_class_emb=m['class_emb'])


def build_engine_model(doms, P, S, *, enc=None, dec_dim=256, dec_depth=2, dec_heads=8, class_emb=64, posemb_size=None,
                       factory='pretrain_multimae_base', learnable_pos=False):
    """Engine model built with the same call sequence as tests/golden/make_golden.py::build_ref."""
    import multimae_amd as M
    from functools import partial
    from torch import nn
    Sg = posemb_size or S
    ins = {}
    for d in doms:
        if d == 'semseg':
            ins[d] = M.SemSegInputAdapter(num_classes=133, dim_class_emb=class_emb, interpolate_class_emb=False, stride_level=4,
                                          patch_size_full=P, image_size=Sg, learnable_pos_emb=learnable_pos)
        else:
            ins[d] = M.PatchedInputAdapter(num_channels=3 if d == 'rgb' else 1, stride_level=1, patch_size_full=P, image_size=Sg,
                                           learnable_pos_emb=learnable_pos)
    outs = {}
    for key, task in [(d, d) for d in doms] + ([('norm_rgb', 'rgb')] if 'rgb' in doms else []):
        ch = {'rgb': 3, 'depth': 1, 'semseg': 133}[task]
        outs[key] = M.SpatialOutputAdapter(num_channels=ch, stride_level=4 if task == 'semseg' else 1, patch_size_full=P,
                                           dim_tokens=dec_dim, depth=dec_depth, num_heads=dec_heads, use_task_queries=True,
                                           task=task, context_tasks=list(doms), use_xattn=True, image_size=Sg)
    if enc is None:
        return M.create_model(factory, input_adapters=ins, output_adapters=outs, num_global_tokens=1, drop_path_rate=0.0).train()
    return M.MultiMAE(ins, outs, num_global_tokens=1, dim_tokens=enc[0], depth=enc[1], num_heads=enc[2], mlp_ratio=4,
                      qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6)).train()


def build_mini_engine(learnable_pos=False):
    m = MINI
    return build_engine_model(m['doms'], m['P'], m['S'], enc=(m['dim'], m['depth'], m['heads']), dec_dim=m['dec_dim'],
                              dec_depth=m['dec_depth'], dec_heads=m['dec_heads'], class_emb=m['class_emb'], learnable_pos=learnable_pos)


def make_inputs(doms, B, S):
    x = {}
    if 'rgb' in doms:
        x['rgb'] = torch.randn(B, 3, S, S)
    if 'depth' in doms:
        x['depth'] = torch.randn(B, 1, S, S)
    if 'semseg' in doms:
        x['semseg'] = torch.randint(0, 133, (B, S // 4, S // 4))
    return x


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
