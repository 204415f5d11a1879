"""CPU: the oracle reproduces the reference-generated golden vectors (pins the oracle)."""
import os

import numpy as np
import pytest
import torch

import multimae_oracle as orc
from helpers import GOLD, MINI, load_masks_base, load_mini, load_scalars, mini_oracle_cfg


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


def _seeded_engine_state(doms, P, S, B, enc=None, posemb=None, **kw):
    """Weights and inputs of a golden recipe rebuilt from its seeds: the engine's constructors run on CPU and their seeded
    initialisation is bit-identical to the reference's (test_boundary_cpu.py)."""
    from helpers import build_engine_model, make_inputs
    torch.manual_seed(0)
    model = build_engine_model(doms, P, S, enc=enc, posemb_size=posemb, **kw)
    x = make_inputs(doms, B, S)
    sd = {k: v.detach().clone().requires_grad_(model.state_dict(keep_vars=True)[k].requires_grad) for k, v in model.state_dict().items()}
    return model, sd, x


def test_oracle_tiny_cfg1_known_answers_from_reference():
    """BASELINE configs[0] geometry (ViT-Tiny, 64x64, patch 8, the 28x28 pos-emb grids interpolated to 8x8): the oracle on the
    seeded recipe reproduces the losses and the gradient norm the REFERENCE recorded (scalars.json) -- this pins the oracle at
    the geometry the loss-curve tests run it at."""
    gold = load_scalars()['tiny_rgb']
    _, sd, x = _seeded_engine_state(['rgb'], 8, 64, 4, enc=(192, 12, 3), posemb=224)
    torch.manual_seed(1)
    dist, tn, an = orc.draw_mask_randoms(4, [64], 1.0)
    spt = orc.samples_per_task_from_dirichlet(dist, 49)
    mask_all, ik, ir = orc.masks_from_noise(spt, tn, an, 49)
    assert int(ik.sum()) == gold['ids_keep_checksum']
    cfg = orc.standard_config(['rgb'], patch_size=8, image_size=224, dim_tokens=192, depth=12, num_heads=3)
    preds = orc.multimae_forward(x, sd, cfg, ik, ir)
    losses = orc.pretrain_losses(preds, x, mask_all, cfg, {'rgb': 64})
    for k, v in gold['losses'].items():
        assert abs(float(losses[k].detach()) - v) < 2e-5, (k, float(losses[k].detach()), v)
    sum(losses.values()).backward()
    gn = float(torch.norm(torch.stack([t.grad.norm() for t in sd.values() if t.grad is not None])))
    assert abs(gn - gold['grad_norm']) / gold['grad_norm'] < 1e-5, (gn, gold['grad_norm'])


def test_oracle_drop_path_vs_reference_golden():
    """oracle.drop_path / block(drop_prob, u) against the reference's recorded DropPath step (mini_droppath.npz)."""
    import os
    import sys
    from functools import partial
    from torch import nn
    import multimae_amd as M
    from helpers import GOLD, make_inputs
    sys.path.insert(0, GOLD)
    from sketch import rel_err_vs
    z = np.load(os.path.join(GOLD, 'mini_droppath.npz'))
    doms, P, S, B, nvis, depth = ['rgb', 'depth', 'semseg'], 8, 32, 3, 12, 4
    rate = float(z['rate'])
    torch.manual_seed(0)
    ins = {}
    for d in doms:
        if d == 'semseg':
            ins[d] = M.SemSegInputAdapter(num_classes=133, dim_class_emb=16, interpolate_class_emb=False, stride_level=4, patch_size_full=P,
                                          image_size=S)
        else:
            ins[d] = M.PatchedInputAdapter(num_channels=3 if d == 'rgb' else 1, stride_level=1, patch_size_full=P, image_size=S)
    outs = {}
    for key, task in [(d, d) for d in doms] + [('norm_rgb', 'rgb')]:
        ch = {'rgb': 3, 'depth': 1, 'semseg': 133}[task]
        outs[key] = M.SpatialOutputAdapter(num_channels=ch, stride_level=4 if task == 'semseg' else 1, patch_size_full=P, dim_tokens=64,
                                           depth=1, num_heads=2, use_task_queries=True, task=task, context_tasks=list(doms), use_xattn=True,
                                           image_size=S)
    model = M.MultiMAE(ins, outs, num_global_tokens=1, dim_tokens=128, depth=depth, num_heads=2, mlp_ratio=4, qkv_bias=True,
                       drop_path_rate=rate, norm_layer=partial(nn.LayerNorm, eps=1e-6))
    g = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad and (n.endswith('bias') or 'mask_token' in n):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    x = make_inputs(doms, B, S)
    assert abs(float(sum(v.double().sum() for v in model.state_dict().values())) - float(z['state_dict_sum'])) < 1e-6
    sd = {k: v.detach().clone().requires_grad_(model.state_dict(keep_vars=True)[k].requires_grad) for k, v in model.state_dict().items()}
    us = [None if z['u'][l][0][0] < 0 else (torch.from_numpy(z['u'][l][0]), torch.from_numpy(z['u'][l][1])) for l in range(depth)]
    cfg = orc.standard_config(doms, patch_size=P, image_size=S, dim_tokens=128, depth=depth, num_heads=2, dec_dim=64, dec_depth=1,
                              dec_heads=2, dim_class_emb=16)
    ik, ir = torch.from_numpy(z['ids_keep']), torch.from_numpy(z['ids_restore'])
    preds = orc.multimae_forward(x, sd, cfg, ik, ir, drop_path_rate=rate, drop_path_u=us)
    mask_all = torch.cat([torch.from_numpy(z['mask/' + d]) for d in doms], 1)
    losses = orc.pretrain_losses(preds, x, mask_all, cfg, {d: 16 for d in doms})
    for k in preds:
        assert rel_err_vs(z, 'pred/' + k, preds[k]) < 1e-5, k
        assert abs(float(losses[k].detach()) - float(z['loss/' + k])) < 1e-5, k
    sum(losses.values()).backward()
    for n, t in sd.items():
        if t.requires_grad:
            assert rel_err_vs(z, 'grad/' + n, t.grad) < 1e-4, n


def test_oracle_cfg2_bench_geometry_vs_reference_golden():
    """The oracle at the ViT-B bench geometry (cfg2: RGB only, 224^2, 98 tokens, B = 4) against the reference's per-tensor
    gradients (geometry_grads.npz; the cfg3 case was asserted equal when the fixture was generated and runs on the GPU side)."""
    import os
    import sys
    from helpers import GOLD
    sys.path.insert(0, GOLD)
    from sketch import rel_err_vs
    z = np.load(os.path.join(GOLD, 'geometry_grads.npz'))
    _, sd, x = _seeded_engine_state(['rgb'], 16, 224, 4)
    torch.manual_seed(1)
    dist, tn, an = orc.draw_mask_randoms(4, [196], 1.0)
    spt = orc.samples_per_task_from_dirichlet(dist, 98)
    mask_all, ik, ir = orc.masks_from_noise(spt, tn, an, 98)
    assert int(ik.sum()) == int(z['cfg2/ids_keep_checksum'])
    cfg = orc.standard_config(['rgb'])
    preds = orc.multimae_forward(x, sd, cfg, ik, ir)
    losses = orc.pretrain_losses(preds, x, mask_all, cfg, {'rgb': 196})
    for k in preds:
        assert abs(float(losses[k].detach()) - float(z[f'cfg2/loss/{k}'])) < 2e-5, k
    sum(losses.values()).backward()
    worst = max((rel_err_vs(z, f'cfg2/grad/{n}', t.grad), n) for n, t in sd.items() if t.requires_grad)
    assert worst[0] < 2e-4, worst


def test_oracle_curve_matches_reference_curve_cfg1():
    """First 4 AdamW steps of the cfg1 loss-curve recipe: the oracle (+ oracle AdamW) reproduces the curve the REFERENCE produced
    with torch.optim.AdamW (tests/golden/curve_cfg1.json) -- the chain the GPU loss-curve tests compare the engine with."""
    import json
    import os
    from helpers import GOLD
    gold = json.load(open(os.path.join(GOLD, 'curve_cfg1.json')))['reference_fp32']
    model, sd0, x = _seeded_engine_state(['rgb'], 8, 64, 4, enc=(192, 12, 3), posemb=224)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    sd = {k: v.detach().clone() for k, v in sd0.items()}
    cfg = orc.standard_config(['rgb'], patch_size=8, image_size=224, dim_tokens=192, depth=12, num_heads=3)
    mo = {n: torch.zeros_like(sd[n]) for n in names}
    vo = {n: torch.zeros_like(sd[n]) for n in names}
    for step in range(1, 5):
        torch.manual_seed(1000 + step)
        dist, tn, an = orc.draw_mask_randoms(4, [64], 1.0)
        spt = orc.samples_per_task_from_dirichlet(dist, 49)
        mask_all, ik, ir = orc.masks_from_noise(spt, tn, an, 49)
        sdo = {k: (v.clone().requires_grad_(True) if k in mo else v) for k, v in sd.items()}
        lo = sum(orc.pretrain_losses(orc.multimae_forward(x, sdo, cfg, ik, ir), x, mask_all, cfg, {'rgb': 64}).values())
        lo.backward()
        assert abs(float(lo.detach()) - gold[step - 1]) < 2e-5, (step, float(lo.detach()), gold[step - 1])
        with torch.no_grad():
            orc.adamw_step({n: sd[n] for n in names}, {n: sdo[n].grad for n in names}, mo, vo, step, 1.5e-4, 0.05)


def test_oracle_label_smoothing_is_torch_cross_entropy():
    """oracle.masked_ce(label_smoothing=eps) per pixel == F.cross_entropy(..., label_smoothing=eps) (what criterion.py:47 calls)."""
    import torch.nn.functional as F
    torch.manual_seed(2)
    logits, tgt = torch.randn(2, 11, 8, 8), torch.randint(0, 11, (2, 8, 8))
    for eps in (0.0, 0.1, 0.35):
        ref = F.cross_entropy(logits, tgt, reduction='none', label_smoothing=eps)
        # all-ones mask: masked mean over every pixel, then mean over samples
        got = orc.masked_ce(logits, tgt, torch.ones(2, 4, dtype=torch.long), 4, 1, label_smoothing=eps)
        assert abs(float(got) - float(ref.mean(dim=(1, 2)).mean())) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# MX-fp8 oracle (oracle/mx_oracle.py): pinned against PyTorch's float8_e4m3fn cast and the OCP MX v1.0 tables
# ---------------------------------------------------------------------------------------------------------------------
def test_mx_e4m3_codec_matches_torch_float8():
    from oracle import mx_oracle as mx
    table = mx.e4m3_decode_table()
    ref = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    ok = ~np.isnan(ref)
    assert np.array_equal(table[ok], ref[ok]) and np.isnan(table[~ok]).all()
    assert table[0x7e] == 448.0 and table[0x08] == 2.0 ** -6 and table[0x01] == 2.0 ** -9      # spec table 1: max normal, min normal, min subnormal
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(200000, generator=g) * torch.exp2(torch.randint(-12, 9, (200000,), generator=g).float())).clamp(-448, 448)
    x = torch.cat([x, torch.tensor([0.0, -0.0, 448.0, -448.0, 2.0 ** -10, 3 * 2.0 ** -10, 2.0 ** -9, 0.0009765625 * 1.5, 17.0, 19.0, 464.0, 1e9, -1e9])])
    mine = mx.e4m3_encode(x.numpy())
    theirs = x.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(mine, theirs)
    # ties go to even mantissas: 17 -> 16, 19 -> 20; 2^-10 (half the smallest subnormal) -> 0
    assert table[mx.e4m3_encode(np.float32([17.0, 19.0, 2.0 ** -10]))].tolist() == [16.0, 20.0, 0.0]


def test_mx_block_quantisation_spec_vectors():
    from oracle import mx_oracle as mx
    x = np.zeros((2, 64), dtype=np.float32)
    x[0, :32] = np.linspace(-1.0, 1.0, 32)              # amax 1.0 -> shared exponent 0 - 8 -> scale 2^-8, elements up to 256
    x[0, 32:] = 0.0                                      # empty block -> exponent byte 0, all-zero elements
    x[1, :32] = 1000.0 * np.cos(np.arange(32))           # amax ~1000 -> floor(log2) = 9 -> scale 2^1
    x[1, 32:] = 3e-5 * np.sin(np.arange(32))
    q, e = mx.mx_quantize(x, round_up=False)               # the specification's rule
    assert e[0, 0] == 127 - 8 and e[0, 1] == 0 and e[1, 0] == 127 + 1
    assert not q[0, 32:].any()
    back = mx.mx_dequantize(q, e)
    # relative error of an element against its block's amax is bounded by the e4m3 quantum at the top binade: 2^-4 of 2^floor(log2 amax)
    # ... except above 1.75 x top = 448 x scale, where the spec's floor-based exponent makes the element saturate
    for r in range(2):
        for b in range(2):
            blk = x[r, b * 32:(b + 1) * 32]
            if np.abs(blk).max() == 0:
                continue
            top = 2.0 ** np.floor(np.log2(np.abs(blk).max()))
            err = np.abs(back[r, b * 32:(b + 1) * 32] - blk)
            bound = np.maximum(top / 16, np.abs(blk) - 1.75 * top) + 1e-30
            assert (err <= bound).all()
    # a value in (448, 512) x scale saturates under the spec's rule instead of overflowing to NaN ...
    y = np.zeros((1, 32), dtype=np.float32); y[0, 0] = 500.0; y[0, 1] = 1.0
    q, e = mx.mx_quantize(y, round_up=False)
    assert e[0, 0] == 127 and q[0, 0] == 0x7e
    # ... and is kept (one bit coarser) under the engine's round-up rule: nothing saturates, error <= amax / 16 everywhere
    q, e = mx.mx_quantize(y)
    assert e[0, 0] == 128
    assert abs(mx.mx_dequantize(q, e)[0, 0] - 500.0) <= 500.0 / 16
    q, e = mx.mx_quantize(x)
    assert e[1, 0] == 127 + 2 and e[0, 0] == 127 - 8        # amax 1000 = 1.95 x 512 rounds up; amax 1.0 does not
    back = mx.mx_dequantize(q, e)
    for r in range(2):
        for b in range(2):
            blk = x[r, b * 32:(b + 1) * 32]
            if np.abs(blk).max() > 0:
                assert np.abs(back[r, b * 32:(b + 1) * 32] - blk).max() <= np.abs(blk).max() / 16
    g = np.random.default_rng(0)
    z = (g.standard_normal((64, 256)) * np.exp2(g.integers(-6, 6, (64, 1)))).astype(np.float32)
    q, e = mx.mx_quantize(z)
    zb = z.reshape(64, 8, 32)
    assert (np.abs(zb).max(2) * np.exp2(127.0 - e) <= 448.0).all()           # no block needs saturation
    # packed scale layout: byte j of dword (g, r, h) is block 8 g + 2 j + h
    ex = np.arange(3 * 12, dtype=np.uint8).reshape(3, 12)
    p = mx.pack_scales(ex).reshape(2, 3, 2, 4)
    assert p[0, 1, 1, 2] == ex[1, 2 * 2 + 1] and p[1, 2, 0, 1] == ex[2, 8 + 2] and p[1, 0, 1, 2] == 0


# ---------------------------------------------------------------------------------------------------------------------
# constructor options added in round 2: the oracle's branches against goldens recorded from the REFERENCE's own classes
# (tests/golden/make_golden_options.py)
# ---------------------------------------------------------------------------------------------------------------------
def _opt():
    z = np.load(os.path.join(GOLD, 'options.npz'))
    return lambda pre: {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def test_oracle_output_adapter_without_cross_attention_vs_reference_golden():
    g = _opt()
    sd = {'output_adapters.rgb.' + k: v.clone() for k, v in g('noxattn/sd/').items()}
    grads = g('noxattn/grad/')
    for k in grads:
        sd['output_adapters.rgb.' + k].requires_grad_(True)
    t = g('noxattn/')
    cfg = orc.standard_config(['rgb', 'depth'], patch_size=4, image_size=16, dim_tokens=96, depth=1, num_heads=2, dec_dim=64, dec_depth=1,
                              dec_heads=2, extra_norm_pix=False)
    enc = t['enc'].clone().requires_grad_(True)
    p = orc.spatial_adapter(enc, sd, cfg, 'rgb', 'rgb', {'rgb': 16, 'depth': 16}, t['ids_keep'], t['ids_restore'], (16, 16), use_xattn=False)
    p.backward(t['gout'])
    assert _rel(p.detach(), t['pred']) < 1e-6 and _rel(enc.grad, t['d_enc']) < 1e-5
    for k, v in grads.items():
        assert _rel(sd['output_adapters.rgb.' + k].grad, v) < 1e-5, k


def test_oracle_learnable_pos_emb_vs_reference_golden():
    g = _opt()
    for name, fn, pp in (('rgb', orc.image_tokens, (4, 4)), ('semseg', orc.semseg_tokens, (2, 2))):
        t = g(f'lpos/{name}/')
        sd = {name + '.' + k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in g(f'lpos/{name}/sd/').items()}
        tok = fn(t['x'], sd, name + '.', *pp)
        tok.backward(t['g'])
        assert _rel(tok.detach(), t['tok']) < 1e-6 and _rel(sd[name + '.pos_emb'].grad, t['d_pos']) < 1e-5, name


def test_oracle_blocks_without_qkv_bias_vs_reference_golden():
    g = _opt()
    sd = {k: v.clone().requires_grad_(True) for k, v in g('nobias/sd/').items()}
    grads = g('nobias/grad/')
    t = g('nobias/')
    L = 1 + max(int(k.split('.')[0]) for k in sd)
    D = t['x'].shape[-1]
    full = dict(sd)
    for l in range(L):
        full[f'{l}.attn.qkv.bias'] = torch.zeros(3 * D)
    x = t['x'].clone().requires_grad_(True)
    y = x
    for l in range(L):
        y = orc.block(y, full, f'{l}.', 2, 1e-6)
    y.backward(t['gy'])
    assert _rel(y.detach(), t['y']) < 1e-6 and _rel(x.grad, t['dx']) < 1e-5
    for k, v in grads.items():
        assert _rel(sd[k].grad, v) < 1e-5, k


def test_oracle_semseg_interpolate_and_padding_idx_vs_reference_golden():
    g = _opt()
    for tag, P_, kw in (('interp4', 4, dict(interpolate_class_emb=True)), ('interp2', 2, dict(interpolate_class_emb=True)), ('pad', 4, dict(padding_idx=7))):
        t = g(f'seg/{tag}/')
        sd = {'s.' + k: v.clone().requires_grad_(v.dtype.is_floating_point and k != 'pos_emb') for k, v in g(f'seg/{tag}/sd/').items()}
        tok = orc.semseg_tokens(t['x'], sd, 's.', P_, P_, **kw)
        tok.backward(t['g'])
        assert _rel(tok.detach(), t['tok']) < 1e-6, tag
        for k, v in g(f'seg/{tag}/grad/').items():
            assert _rel(sd['s.' + k].grad, v) < 1e-5, (tag, k)


@pytest.mark.parametrize('tag,task,in_tasks,utq', [('notin', 'depth', ['rgb'], True), ('noq', 'rgb', ['rgb', 'depth'], False)])
def test_oracle_mask_token_queries_vs_reference_golden(tag, task, in_tasks, utq):
    """get_queries_and_context's else-branch (output_adapters.py:213-220): the task is not an encoder input ('notin') or
    use_task_queries=False ('noq'); recorded from the reference's class by tests/golden/make_golden_queries.py."""
    z = np.load(os.path.join(GOLD, 'mask_queries.npz'))
    g = lambda pre: {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    sd = {f'output_adapters.{task}.' + k: v.clone() for k, v in g(f'{tag}/sd/').items()}
    grads = g(f'{tag}/grad/')
    for k in grads:
        sd[f'output_adapters.{task}.' + k].requires_grad_(True)
    t = g(f'{tag}/')
    cfg = orc.standard_config(['rgb', 'depth'], patch_size=4, image_size=16, dim_tokens=96, depth=1, num_heads=2, dec_dim=64, dec_depth=1,
                              dec_heads=2, extra_norm_pix=False)
    enc = t['enc'].clone().requires_grad_(True)
    p = orc.spatial_adapter(enc, sd, cfg, task, task, {d: 16 for d in in_tasks}, t['ids_keep'], t['ids_restore'], (16, 16), use_task_queries=utq)
    p.backward(t['gout'])
    assert _rel(p.detach(), t['pred']) < 1e-6 and _rel(enc.grad, t['d_enc']) < 1e-5
    for k, v in grads.items():
        assert _rel(sd[f'output_adapters.{task}.' + k].grad, v) < 1e-5, k
