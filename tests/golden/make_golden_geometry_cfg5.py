"""Per-tensor gradient golden at the cfg5 geometry (BASELINE.json configs[4]), generated FROM THE REFERENCE ITSELF:

  cfg5  ViT-L (24 blocks, dim 1024, 16 heads: pretrain_multimae_large, multimae/multimae.py:400-416), RGB+depth+semseg, 224^2,
        patch 16, 196 visible tokens, 4 decoders, B=2

Same seeded recipe and storage as make_golden_geometry.py (seed 0 -> model, inputs; seed 1 -> forward; small gradients whole,
large ones and predictions as sketches), in its own file so that the ViT-B fixture stays byte-identical.  The oracle is pinned
against the reference on this very step before anything is written.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_geometry_cfg5.py      ->  tests/golden/geometry_grads_cfg5.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from sketch import store  # noqa: E402


def build_ref_large(rm, ria, roa, doms, P, S):
    """mg.build_ref with the reference's ViT-L factory"""
    ins = {}
    for d in doms:
        if d == 'semseg':
            ins[d] = ria.SemSegInputAdapter(num_classes=133, dim_class_emb=64, interpolate_class_emb=False, stride_level=4, patch_size_full=P, image_size=S)
        else:
            ins[d] = ria.PatchedInputAdapter(num_channels=3 if d == 'rgb' else 1, stride_level=1, patch_size_full=P, image_size=S)
    outs = {}
    for key, task in [(d, d) for d in doms] + [('norm_rgb', 'rgb')]:
        ch = {'rgb': 3, 'depth': 1, 'semseg': 133}[task]
        outs[key] = roa.SpatialOutputAdapter(num_channels=ch, stride_level=4 if task == 'semseg' else 1, patch_size_full=P, dim_tokens=256, depth=2,
                                             num_heads=8, use_task_queries=True, task=task, context_tasks=list(doms), use_xattn=True, image_size=S)
    return rm.pretrain_multimae_large(ins, outs, num_global_tokens=1, drop_path_rate=0.0).train()


def main():
    torch.set_num_threads(8)
    rm, ria, roa, rc = mg.import_reference()
    orc = mg.load_oracle()
    doms, P, S, B, nvis = ['rgb', 'depth', 'semseg'], 16, 224, 2, 196
    torch.manual_seed(0)
    model = build_ref_large(rm, ria, roa, doms, P, S)
    assert len(model.encoder) == 24 and model.encoder[0].attn.qkv.weight.shape == (3072, 1024)
    x = mg.make_inputs(doms, B, S)
    preds, masks, losses, (tm, ids_keep, ids_restore) = mg.ref_step(model, rc, x, P, nvis, seed=1)
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    print('cfg5', {k: round(float(v), 6) for k, v in losses.items()}, 'grad_norm', mg.grad_norm(model), flush=True)

    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sdo = {k: v.clone().requires_grad_(model.state_dict(keep_vars=True)[k].requires_grad) for k, v in sd.items()}
    cfg = orc.standard_config(doms, dim_tokens=1024, depth=24, num_heads=16)
    po = orc.multimae_forward(x, sdo, cfg, ids_keep, ids_restore)
    m_all = torch.cat([tm[d] for d in doms], 1)
    lo = orc.pretrain_losses(po, x, m_all, cfg, {d: 196 for d in doms})
    sum(lo.values()).backward()
    for k in preds:
        assert (po[k] - preds[k]).abs().max().item() < 2e-4, (k, (po[k] - preds[k]).abs().max().item())
        assert abs(float(lo[k]) - float(losses[k])) < 2e-5, (k, float(lo[k]), float(losses[k]))
    worst = 0.0
    for n, g in grads.items():
        rel = float((sdo[n].grad - g).norm() / (g.norm() + 1e-30))
        worst = max(worst, rel)
        assert rel < 5e-4, (n, rel)
    print(f'cfg5: oracle == reference (preds, losses, {len(grads)} grads; worst grad rel {worst:.2e})', flush=True)

    out = {}
    pre = 'cfg5/'
    for k, v in losses.items():
        out[pre + 'loss/' + k] = np.float64(float(v))
    out[pre + 'grad_norm'] = np.float64(mg.grad_norm(model))
    out[pre + 'ids_keep_checksum'] = np.int64(int(ids_keep.sum()))
    for k, v in preds.items():
        store(out, pre + 'pred/' + k, v)
    for n, g in grads.items():
        store(out, pre + 'grad/' + n, g)
    n_whole = sum(1 for g in grads.values() if g.numel() <= 4096)
    print(f'cfg5: {n_whole} gradients stored whole, {len(grads) - n_whole} as sketches', flush=True)
    np.savez_compressed(os.path.join(HERE, 'geometry_grads_cfg5.npz'), **out)
    print('wrote geometry_grads_cfg5.npz', os.path.getsize(os.path.join(HERE, 'geometry_grads_cfg5.npz')), 'bytes')


if __name__ == '__main__':
    main()
