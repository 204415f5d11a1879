"""Decoder stochastic-depth golden FROM THE REFERENCE: the 3-modality mini MultiMAE of make_golden.py whose four SpatialOutputAdapters are built
with depth = 3, drop_path_rate = 0.3 (output_adapters.py:126-132: rates linspace(0, 0.3, 3) = 0, 0.15, 0.3 over decoder_transformer), encoder
without stochastic depth; forward + losses + backward on CPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dec_droppath.py

The reference draws with torch.rand on the CPU generator: replaying it (mask-sampler draws, then -- adapter by adapter in the order
MultiMAE.forward runs them, multimae.py:366-378 -- two (B,1,1) draws per block with a rate > 0) recovers the uniforms, which are stored as
explicit inputs; the engine test feeds them through multimae_utils._drop_path_rand."""
import os
import sys
from functools import partial

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from sketch import store  # noqa: E402

RATE, DEC_DEPTH = 0.3, 3


def main():
    torch.set_num_threads(8)
    rm, ria, roa, rc = mg.import_reference()
    orc = mg.load_oracle()
    doms, P, S, B, nvis = ['rgb', 'depth', 'semseg'], 8, 32, 3, 12
    torch.manual_seed(0)
    ins = {}
    for d in doms:
        if d == 'semseg':
            ins[d] = ria.SemSegInputAdapter(num_classes=133, dim_class_emb=16, interpolate_class_emb=False, stride_level=4,
                                            patch_size_full=P, image_size=S)
        else:
            ins[d] = ria.PatchedInputAdapter(num_channels=3 if d == 'rgb' else 1, stride_level=1, patch_size_full=P, image_size=S)
    outs = {}
    keys = [(d, d) for d in doms] + [('norm_rgb', 'rgb')]
    for key, task in keys:
        ch = {'rgb': 3, 'depth': 1, 'semseg': 133}[task]
        outs[key] = roa.SpatialOutputAdapter(num_channels=ch, stride_level=4 if task == 'semseg' else 1, patch_size_full=P,
                                             dim_tokens=64, depth=DEC_DEPTH, num_heads=2, use_task_queries=True, task=task,
                                             context_tasks=list(doms), use_xattn=True, image_size=S, drop_path_rate=RATE)
    model = rm.MultiMAE(ins, outs, num_global_tokens=1, dim_tokens=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True,
                        drop_path_rate=0.0, norm_layer=partial(nn.LayerNorm, eps=1e-6)).train()
    g = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad and (n.endswith('bias') or 'mask_token' in n):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    x = mg.make_inputs(doms, B, S)
    seed = 11
    preds, masks, losses, (tm, ids_keep, ids_restore) = mg.ref_step(model, rc, x, P, nvis, seed=seed)
    assert list(preds.keys()) == [k for k, _ in keys], 'adapter execution order'
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    torch.manual_seed(seed)
    npt = [(S // P) ** 2] * len(doms)
    dist, task_noise, all_noise = orc.draw_mask_randoms(B, npt, 1.0)
    spt = orc.samples_per_task_from_dirichlet(dist, nvis)
    m_all, k_o, r_o = orc.masks_from_noise(spt, task_noise, all_noise, nvis)
    assert torch.equal(k_o, ids_keep) and torch.equal(r_o, ids_restore)
    rates = [float(v) for v in torch.linspace(0, RATE, DEC_DEPTH)]
    us = []                                       # [adapter][block] -> (u_attn, u_mlp) or None
    for _ in keys:
        row = []
        for l in range(DEC_DEPTH):
            row.append((torch.rand((B, 1, 1)).view(B), torch.rand((B, 1, 1)).view(B)) if rates[l] > 0 else None)
        us.append(row)
    kept = [float((1 - rates[l] + u).floor().sum()) for row in us for l, pr in enumerate(row) if pr is not None for u in pr]
    assert 0 < sum(kept) < B * len(kept), 'choose a seed where some but not all paths drop'
    # the replayed draws are the ones the reference used: re-run its forward with torch.rand patched to hand them out again
    flat = [u.view(B, 1, 1) for row in us for pr in row if pr is not None for u in pr]
    it = iter(flat)
    real_rand = torch.rand
    model.generate_random_masks = lambda *a, **k: (tm, ids_keep, ids_restore)
    torch.rand = lambda *a, **k: next(it)
    try:
        p2, _ = model(x, num_encoded_tokens=nvis, alphas=1.0)
    finally:
        torch.rand = real_rand
    assert next(it, None) is None
    for k in preds:
        assert torch.equal(p2[k], preds[k]), k
    print('decoder drop-path mini: replayed draws reproduce the reference forward bit for bit; kept paths', kept)

    sd = model.state_dict()
    out = {'state_dict_sum': np.float64(float(sum(v.double().sum() for v in sd.values())))}
    for k, v in grads.items():
        store(out, 'grad/' + k, v)
    for k, v in preds.items():
        store(out, 'pred/' + k, v)
    for k, v in losses.items():
        out['loss/' + k] = np.float32(float(v))
    for i, d in enumerate(doms):
        out['mask/' + d] = tm[d].numpy()
    out['ids_keep'] = ids_keep.numpy()
    out['ids_restore'] = ids_restore.numpy()
    out['rate'] = np.float32(RATE)
    out['dec_depth'] = np.int32(DEC_DEPTH)
    out['u'] = np.stack([u.numpy() for u in [w.view(B) for w in flat]])
    np.savez_compressed(os.path.join(HERE, 'mini_dec_droppath.npz'), **out)
    print('wrote mini_dec_droppath.npz', os.path.getsize(os.path.join(HERE, 'mini_dec_droppath.npz')))


if __name__ == '__main__':
    main()
