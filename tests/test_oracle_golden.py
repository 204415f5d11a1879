"""CPU: the oracle reproduces the reference-generated golden vectors (pins the oracle)."""
import numpy as np
import pytest
import torch

import multimae_oracle as orc
from helpers import MINI, load_masks_base, load_mini, load_scalars, mini_oracle_cfg


def test_oracle_masks_base_bit_exact():
    g = load_masks_base()
    spt = orc.samples_per_task_from_dirichlet(g['dirichlet'], 98)
    assert torch.equal(spt, g['samples_per_task'])
    m, k, r = orc.masks_from_noise(spt, [g['noise_rgb'], g['noise_depth'], g['noise_semseg']], g['noise_all'], 98)
    assert torch.equal(m, g['mask_all']) and torch.equal(k, g['ids_keep']) and torch.equal(r, g['ids_restore'])
    # invariants: exactly 98 visible; ids_keep marks exactly mask == 0
    assert (m == 0).sum(1).eq(98).all()
    assert torch.gather(m, 1, k).sum() == 0
    assert torch.equal(torch.argsort(r, 1)[:, :98], k)


def test_oracle_mini_forward_losses_grads():
    g = load_mini()
    cfg = mini_oracle_cfg()
    sd = {k: v.clone().requires_grad_(k in g['grad']) for k, v in g['sd'].items()}
    preds, inter = orc.multimae_forward(g['x'], sd, cfg, g['ids_keep'], g['ids_restore'], return_intermediates=True)
    assert torch.allclose(inter['enc_in'], g['enc_in'], atol=1e-5)
    assert torch.allclose(inter['enc_out'], g['enc_out'], atol=2e-5)
    for k, v in g['pred'].items():
        assert torch.allclose(preds[k], v, atol=2e-5), k
    mask_all = torch.cat([g['mask'][d] for d in MINI['doms']], 1)
    tpt = {d: g['mask'][d].shape[1] for d in MINI['doms']}
    losses = orc.pretrain_losses(preds, g['x'], mask_all, cfg, tpt)
    for k, v in g['loss'].items():
        assert abs(float(losses[k]) - v) < 1e-5, (k, float(losses[k]), v)
    sum(losses.values()).backward()
    for n, gr in g['grad'].items():
        rel = (sd[n].grad - gr).norm() / (gr.norm() + 1e-12)
        assert rel < 1e-4, (n, float(rel))


def test_oracle_mask_replay_from_seed():
    g = load_mini()
    torch.manual_seed(1)
    dist, tn, an = orc.draw_mask_randoms(MINI['B'], [16, 16, 16], 1.0)
    assert torch.equal(dist, g['dirichlet']) and torch.equal(an, g['noise_all'])
    spt = orc.samples_per_task_from_dirichlet(dist, MINI['nvis'])
    _, k, r = orc.masks_from_noise(spt, tn, an, MINI['nvis'])
    assert torch.equal(k, g['ids_keep']) and torch.equal(r, g['ids_restore'])


def test_oracle_sincos_layout():
    pe = orc.sincos_posemb_2d(14, 14, 768)
    q = 768 // 4
    omega = 1.0 / (10000.0 ** (torch.arange(q, dtype=torch.float32) / q))
    r, c = 5, 9
    assert torch.allclose(pe[0, :q, r, c], torch.sin(r * omega))
    assert torch.allclose(pe[0, 2 * q:3 * q, r, c], torch.sin(c * omega))
    # 14 -> 14 interpolation is the identity
    assert torch.allclose(orc.resized_posemb_tokens(pe, 14, 14, 'bicubic'), pe[0].flatten(1).t(), atol=1e-6)


def test_oracle_adamw_matches_torch():
    torch.manual_seed(0)
    p = {'a': torch.randn(37), 'b': torch.randn(5, 3)}
    ref = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.AdamW(list(ref.values()), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v_ = {k: torch.zeros_like(v) for k, v in p.items()}
    for step in range(1, 4):
        gr = {k: torch.randn_like(x) for k, x in p.items()}
        for k in ref:
            ref[k].grad = gr[k].clone()
        opt.step()
        orc.adamw_step(p, gr, m, v_, step, 1e-3, 0.05)
    for k in p:
        assert torch.allclose(p[k], ref[k].detach(), atol=1e-6)


def test_oracle_truncated_depth_standardize_matches_reference_fixture():
    """oracle.truncated_depth_standardize vs the outputs of the reference's own lines (run_pretraining_multimae.py:487-492)."""
    import numpy as np, os
    import multimae_oracle as orc
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'depth_std.npz'))
    for k in [f[2:] for f in z.files if f.startswith('x/')]:
        x, y = torch.from_numpy(z['x/' + k]), torch.from_numpy(z['y/' + k])
        out = orc.truncated_depth_standardize(x)
        assert float((out - y).abs().max()) <= 1e-6 * float(y.abs().max()), k
