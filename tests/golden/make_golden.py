"""Generate the golden fixtures in tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's own classes (SURVEY.md Appendix B recipe: a stub
``utils`` package so ``utils/__init__.py`` -- which needs torchvision -- never
runs), executes them on CPU fp32, cross-checks ``oracle/multimae_oracle.py``
against them on the spot (the oracle is thereby *pinned*), and stores

  mini_fwd_bwd.npz   weights, inputs, RNG draws, ids, preds, losses, grads of a
                     3-modality mini MultiMAE (dim 128, depth 2, dec 64)
  masks_base.npz     Dirichlet/noise draws + (mask, ids_keep, ids_restore) of the
                     reference sampler at the real token geometry (3x196, 98 kept)
  scalars.json       seeded-init known answers (losses, grad norm, state-dict sum)
                     for ViT-Tiny cfg1 and ViT-B cfg3 at B=4

Nothing under /root/reference is copied; only its outputs are recorded.
"""
import json
import os
import sys
import types
from functools import partial

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'


def import_reference():
    pkg = types.ModuleType('utils')
    pkg.__path__ = [os.path.join(REF, 'utils')]
    sys.modules['utils'] = pkg
    sys.path.insert(0, REF)
    import multimae.multimae as rm
    import multimae.input_adapters as ria
    import multimae.output_adapters as roa
    import multimae.criterion as rc
    sys.path.pop(0)
    return rm, ria, roa, rc


def load_oracle():
    import importlib.util
    spec = importlib.util.spec_from_file_location('multimae_oracle', os.path.join(ROOT, 'oracle', 'multimae_oracle.py'))
    m = importlib.util.module_from_spec(spec)
    sys.modules['multimae_oracle'] = m
    spec.loader.exec_module(m)
    return m


def build_ref(rm, ria, roa, doms, P, S, *, enc=None, dec_dim=256, dec_depth=2, dec_heads=8, class_emb=64,
              posemb_size=None):
    """posemb_size: image_size the adapters build their pos-emb grid for (default: S)."""
    S_in, S = S, (posemb_size or S)
    ins = {}
    for d in doms:
        if d == 'semseg':
            ins[d] = ria.SemSegInputAdapter(num_classes=133, dim_class_emb=class_emb, interpolate_class_emb=False,
                                            stride_level=4, patch_size_full=P, image_size=S)
        else:
            ins[d] = ria.PatchedInputAdapter(num_channels=3 if d == 'rgb' else 1, stride_level=1,
                                             patch_size_full=P, image_size=S)
    outs = {}
    for key, task in [(d, d) for d in doms] + ([('norm_rgb', 'rgb')] if 'rgb' in doms else []):
        ch = {'rgb': 3, 'depth': 1, 'semseg': 133}[task]
        outs[key] = roa.SpatialOutputAdapter(num_channels=ch, stride_level=4 if task == 'semseg' else 1,
                                             patch_size_full=P, dim_tokens=dec_dim, depth=dec_depth,
                                             num_heads=dec_heads, use_task_queries=True, task=task,
                                             context_tasks=list(doms), use_xattn=True, image_size=S)
    if enc is None:
        model = rm.pretrain_multimae_base(ins, outs, num_global_tokens=1, drop_path_rate=0.0)
    else:
        model = rm.MultiMAE(ins, outs, num_global_tokens=1, dim_tokens=enc[0], depth=enc[1], num_heads=enc[2],
                            mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6))
    return model.train()


def make_inputs(doms, B, S):
    x = {}
    if 'rgb' in doms:
        x['rgb'] = torch.randn(B, 3, S, S)
    if 'depth' in doms:
        x['depth'] = torch.randn(B, 1, S, S)
    if 'semseg' in doms:
        x['semseg'] = torch.randint(0, 133, (B, S // 4, S // 4))
    return x


def ref_step(model, rc, x, P, nvis, seed):
    """One fwd + losses + backward exactly as train_one_epoch wires it; returns everything."""
    captured = {}
    orig = model.generate_random_masks

    def spy(*a, **k):
        out = orig(*a, **k)
        captured['ids'] = out
        return out
    model.generate_random_masks = spy
    torch.manual_seed(seed)
    preds, masks = model(x, num_encoded_tokens=nvis, alphas=1.0)
    model.generate_random_masks = orig
    masks = dict(masks)
    if 'rgb' in masks:
        masks['norm_rgb'] = masks['rgb']
    fns = {'rgb': rc.MaskedMSELoss(P, 1), 'depth': rc.MaskedL1Loss(P, 1), 'semseg': rc.MaskedCrossEntropyLoss(P, 4),
           'norm_rgb': rc.MaskedMSELoss(P, 1, norm_pix=True)}
    tgt = dict(x)
    if 'rgb' in x:
        tgt['norm_rgb'] = x['rgb']
    losses = {k: fns[k](preds[k].float(), tgt[k], mask=masks[k]) for k in preds}
    model.zero_grad()
    sum(losses.values()).backward()
    return preds, masks, losses, captured['ids']


def grad_norm(model):
    return float(torch.norm(torch.stack([p.grad.norm() for p in model.parameters() if p.grad is not None])))


def main():
    torch.set_num_threads(8)
    rm, ria, roa, rc = import_reference()
    orc = load_oracle()
    os.makedirs(HERE, exist_ok=True)

    # ------------------------------------------------------------------ mini --
    doms, P, S, B, nvis = ['rgb', 'depth', 'semseg'], 8, 32, 3, 12
    torch.manual_seed(0)
    model = build_ref(rm, ria, roa, doms, P, S, enc=(128, 2, 2), dec_dim=64, dec_depth=1, dec_heads=2, class_emb=16)
    # give mask_token / biases / LN affine non-trivial values so their gradients and
    # forward contributions are exercised (the reference inits them to 0 / 1)
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad and (n.endswith('bias') or 'mask_token' in n):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            if p.requires_grad and 'norm' in n and n.endswith('weight'):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    x = make_inputs(doms, B, S)
    preds, masks, losses, (tm, ids_keep, ids_restore) = ref_step(model, rc, x, P, nvis, seed=1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    # replay the RNG stream to expose the random draws as explicit inputs
    torch.manual_seed(1)
    npt = [(S // P) ** 2] * len(doms)
    dist, task_noise, all_noise = orc.draw_mask_randoms(B, npt, 1.0)
    spt = orc.samples_per_task_from_dirichlet(dist, nvis)
    m_all, k_o, r_o = orc.masks_from_noise(spt, task_noise, all_noise, nvis)
    assert torch.equal(k_o, ids_keep) and torch.equal(r_o, ids_restore), 'oracle sampler != reference'
    assert torch.equal(m_all, torch.cat([tm[d] for d in doms], 1))

    # oracle forward/backward on the same weights
    cfg = orc.standard_config(doms, patch_size=P, image_size=S, dim_tokens=128, depth=2, num_heads=2,
                              dec_dim=64, dec_depth=1, dec_heads=2, dim_class_emb=16)
    sdo = {k: v.clone().requires_grad_(model.state_dict(keep_vars=True)[k].requires_grad) for k, v in sd.items()}
    po, inter = orc.multimae_forward(x, sdo, cfg, ids_keep, ids_restore, return_intermediates=True)
    tpt = {d: n for d, n in zip(doms, npt)}
    lo = orc.pretrain_losses(po, x, m_all, cfg, tpt)
    sum(lo.values()).backward()
    for k in preds:
        err = (po[k] - preds[k]).abs().max().item()
        assert err < 2e-5, (k, err)
        assert abs(float(lo[k]) - float(losses[k])) < 1e-5, (k, float(lo[k]), float(losses[k]))
    for n, gref in grads.items():
        go = sdo[n].grad
        rel = (go - gref).norm() / (gref.norm() + 1e-12)
        assert rel < 1e-4, (n, float(rel))
    print('mini: oracle == reference  (preds, losses, %d grads)' % len(grads))

    out = {}
    for k, v in sd.items():
        out['sd/' + k] = v.numpy()
    for k, v in grads.items():
        out['grad/' + k] = v.numpy()
    for k, v in x.items():
        out['x/' + k] = v.numpy()
    for k, v in preds.items():
        out['pred/' + k] = v.detach().numpy()
    for k, v in losses.items():
        out['loss/' + k] = np.float32(float(v))
    out['dirichlet'] = dist.numpy()
    for i, d in enumerate(doms):
        out['noise/' + d] = task_noise[i].numpy()
        out['mask/' + d] = tm[d].numpy()
    out['noise_all'] = all_noise.numpy()
    out['ids_keep'] = ids_keep.numpy()
    out['ids_restore'] = ids_restore.numpy()
    out['enc_in'] = inter['enc_in'].detach().numpy()
    out['enc_out'] = inter['enc_out'].detach().numpy()
    np.savez_compressed(os.path.join(HERE, 'mini_fwd_bwd.npz'), **out)

    # ------------------------------------------------- sampler at real geometry --
    torch.manual_seed(7)
    Bm, toks = 16, {'rgb': torch.zeros(16, 196, 1), 'depth': torch.zeros(16, 196, 1), 'semseg': torch.zeros(16, 196, 1)}
    tmk, ik, ir = model.generate_random_masks(toks, 98, alphas=1.0)
    torch.manual_seed(7)
    dist, task_noise, all_noise = orc.draw_mask_randoms(Bm, [196] * 3, 1.0)
    spt = orc.samples_per_task_from_dirichlet(dist, 98)
    m2, k2, r2 = orc.masks_from_noise(spt, task_noise, all_noise, 98)
    assert torch.equal(k2, ik) and torch.equal(r2, ir)
    keys = torch.cat([(tmk[d]) for d in toks], 1)
    np.savez_compressed(os.path.join(HERE, 'masks_base.npz'), dirichlet=dist.numpy(),
                        samples_per_task=spt.numpy(), noise_rgb=task_noise[0].numpy(),
                        noise_depth=task_noise[1].numpy(), noise_semseg=task_noise[2].numpy(),
                        noise_all=all_noise.numpy(), mask_all=keys.numpy(), ids_keep=ik.numpy(),
                        ids_restore=ir.numpy())
    print('masks_base: oracle sampler == reference')

    # ------------------------------------------------- seeded-init known answers --
    scal = {}
    # tiny = BASELINE.json configs[0]: adapters keep the default 224 pos-emb grid (28x28 at
    # patch 8) and interpolate it to 8x8 at run time (bicubic in, bilinear in the decoders)
    for name, doms_, P_, S_, nv_, enc_, pes_ in [('tiny_rgb', ['rgb'], 8, 64, 49, (192, 12, 3), 224),
                                                   ('base_rgb_depth_semseg', ['rgb', 'depth', 'semseg'], 16, 224, 98, None, None)]:
        torch.manual_seed(0)
        mdl = build_ref(rm, ria, roa, doms_, P_, S_, enc=enc_, posemb_size=pes_)
        xx = make_inputs(doms_, 4, S_)
        pr, mk, ls, (tm_, ik_, ir_) = ref_step(mdl, rc, xx, P_, nv_, seed=1)
        sdsum = float(sum(v.double().sum() for v in mdl.state_dict().values()))
        scal[name] = {
            'losses': {k: float(v) for k, v in ls.items()},
            'total': float(sum(ls.values())),
            'grad_norm': grad_norm(mdl),
            'visible_sample0': {d: int((tm_[d][0] == 0).sum()) for d in doms_},
            'state_dict_sum': sdsum,
            'num_keys': len(mdl.state_dict()),
            'num_trainable': int(sum(p.numel() for p in mdl.parameters() if p.requires_grad)),
            'ids_keep_checksum': int(ik_.sum()),
            'recipe': 'torch.manual_seed(0); build; inputs randn/randint in order rgb,depth,semseg; '
                      'torch.manual_seed(1); forward(num_encoded_tokens, alphas=1.0); B=4',
            'posemb_image_size': pes_ or S_,
        }
        print(name, json.dumps(scal[name]['losses']), scal[name]['grad_norm'])
    scal['torch_version'] = torch.__version__
    with open(os.path.join(HERE, 'scalars.json'), 'w') as f:
        json.dump(scal, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
