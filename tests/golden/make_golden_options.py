"""Goldens FROM THE REFERENCE for the constructor options added in round 2 -- each one runs the reference's own class on CPU,
asserts oracle == reference on the spot and stores weights / inputs / outputs / gradients:

  * SpatialOutputAdapter(use_xattn=False)         output_adapters.py:114-123, 264-268
  * PatchedInputAdapter / SemSegInputAdapter with learnable_pos_emb=True, parameter grid != token grid
                                                   input_adapters.py:75-78, 113-114, 183-186, 235-236
  * Block(qkv_bias=False)                          multimae_utils.py:165-180, 217-232

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_options.py        ->  tests/golden/options.npz
"""
import os
import sys
from functools import partial

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    torch.set_num_threads(8)
    rm, ria, roa, rc = mg.import_reference()
    orc = mg.load_oracle()
    import importlib
    rutil = importlib.import_module('multimae.multimae_utils')
    out = {}

    # ---- output adapter without cross attention
    torch.manual_seed(11)
    B, Denc, D, P, S = 2, 96, 64, 4, 16
    ad = roa.SpatialOutputAdapter(num_channels=3, stride_level=1, patch_size_full=P, dim_tokens=D, depth=1, num_heads=2, use_task_queries=True,
                                  task='rgb', context_tasks=['rgb', 'depth'], use_xattn=False, image_size=S, dim_tokens_enc=Denc).train()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n_, p in ad.named_parameters():
            if p.requires_grad and (n_.endswith('bias') or 'mask_token' in n_):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    n, nkeep, G = 16, 10, 1
    ids_shuffle = torch.argsort(torch.rand(B, 2 * n, generator=g), dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :nkeep]
    enc = torch.randn(B, nkeep + G, Denc, generator=g)
    gout = torch.randn(B, 3, S, S, generator=g)
    info = {'tasks': {'rgb': {'num_tokens': n, 'has_2d_posemb': True, 'start_idx': 0, 'end_idx': n},
                      'depth': {'num_tokens': n, 'has_2d_posemb': True, 'start_idx': n, 'end_idx': 2 * n}},
            'image_size': (S, S), 'num_task_tokens': 2 * n, 'num_global_tokens': G}
    er = enc.clone().requires_grad_(True)
    pr = ad(er, info, ids_keep, ids_restore)
    pr.backward(gout)
    cfg = orc.standard_config(['rgb', 'depth'], patch_size=P, image_size=S, dim_tokens=Denc, depth=1, num_heads=2, dec_dim=D, dec_depth=1,
                              dec_heads=2, extra_norm_pix=False)
    sd = {'output_adapters.rgb.' + k: v.detach().clone() for k, v in ad.state_dict().items()}
    for k, p in ad.named_parameters():
        sd['output_adapters.rgb.' + k].requires_grad_(p.requires_grad)
    eo = enc.clone().requires_grad_(True)
    po = orc.spatial_adapter(eo, sd, cfg, 'rgb', 'rgb', {'rgb': n, 'depth': n}, ids_keep, ids_restore, (S, S), use_xattn=False)
    po.backward(gout)
    assert rel(po, pr) < 1e-6 and rel(eo.grad, er.grad) < 1e-5, (rel(po, pr), rel(eo.grad, er.grad))
    for k, p in ad.named_parameters():
        if p.requires_grad:
            assert rel(sd['output_adapters.rgb.' + k].grad, p.grad) < 1e-5, k
    assert not any(k.startswith(('decoder.', 'query_norm', 'context_norm', 'out_norm', 'mlp.')) for k in ad.state_dict())
    for k, v in ad.state_dict().items():
        out['noxattn/sd/' + k] = v.detach().numpy()
    for k, p in ad.named_parameters():
        if p.requires_grad:
            out['noxattn/grad/' + k] = p.grad.numpy()
    out.update({'noxattn/enc': enc.numpy(), 'noxattn/ids_keep': ids_keep.numpy(), 'noxattn/ids_restore': ids_restore.numpy(),
                'noxattn/gout': gout.numpy(), 'noxattn/pred': pr.detach().numpy(), 'noxattn/d_enc': er.grad.numpy()})

    # ---- learnable positional embeddings on both input adapters (7x7 / 5x5 parameter grids, 4x4 token grids)
    torch.manual_seed(5)
    D = 64
    rgb = ria.PatchedInputAdapter(num_channels=3, stride_level=1, patch_size_full=4, dim_tokens=D, learnable_pos_emb=True, image_size=28).train()
    seg = ria.SemSegInputAdapter(num_classes=11, stride_level=2, patch_size_full=4, dim_tokens=D, learnable_pos_emb=True, image_size=20,
                                 dim_class_emb=16).train()
    assert rgb.pos_emb.requires_grad and seg.pos_emb.requires_grad and rgb.pos_emb.shape[-1] == 7 and seg.pos_emb.shape[-1] == 5
    x_rgb = torch.randn(3, 3, 16, 16, generator=g)
    x_seg = torch.randint(0, 11, (3, 8, 8), generator=g)
    g_rgb = torch.randn(3, 16, D, generator=g)
    g_seg = torch.randn(3, 16, D, generator=g)
    t_rgb = rgb(x_rgb); t_rgb.backward(g_rgb)
    t_seg = seg(x_seg); t_seg.backward(g_seg)
    for name, mod, x, gt, tok, fn, pp in (('rgb', rgb, x_rgb, g_rgb, t_rgb, orc.image_tokens, (4, 4)), ('semseg', seg, x_seg, g_seg, t_seg, orc.semseg_tokens, (2, 2))):
        sdo = {name + '.' + k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in mod.state_dict().items()}
        to = fn(x, sdo, name + '.', *pp)
        to.backward(gt)
        assert rel(to, tok) < 1e-6 and rel(sdo[name + '.pos_emb'].grad, mod.pos_emb.grad) < 1e-5, name
        for k, v in mod.state_dict().items():
            out[f'lpos/{name}/sd/{k}'] = v.detach().numpy()
        out[f'lpos/{name}/x'] = x.numpy(); out[f'lpos/{name}/g'] = gt.numpy(); out[f'lpos/{name}/tok'] = tok.detach().numpy()
        out[f'lpos/{name}/d_pos'] = mod.pos_emb.grad.numpy()

    # ---- SemSegInputAdapter(interpolate_class_emb=True) and (emb_padding_idx=...)       input_adapters.py:186, 192-198
    torch.manual_seed(9)
    for tag, kw, P_ in (('interp4', dict(interpolate_class_emb=True), 4), ('interp2', dict(interpolate_class_emb=True), 2),
                        ('pad', dict(emb_padding_idx=7), 4)):
        mod = ria.SemSegInputAdapter(num_classes=11, stride_level=1, patch_size_full=P_, dim_tokens=D, image_size=16, dim_class_emb=16, **kw).train()
        ncls = 12 if 'emb_padding_idx' in kw else 11
        xs = torch.randint(0, ncls, (2, 16, 16), generator=g)
        ntok = (16 // P_) ** 2
        gs = torch.randn(2, ntok, D, generator=g)
        tk = mod(xs); tk.backward(gs)
        sdo = {'s.' + k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and k != 'pos_emb') for k, v in mod.state_dict().items()}
        to = orc.semseg_tokens(xs, sdo, 's.', P_, P_, interpolate_class_emb=kw.get('interpolate_class_emb', False), padding_idx=kw.get('emb_padding_idx'))
        to.backward(gs)
        assert rel(to, tk) < 1e-6, (tag, rel(to, tk))
        for k, p in mod.named_parameters():
            if p.requires_grad:
                assert rel(sdo['s.' + k].grad, p.grad) < 1e-5, (tag, k)
                out[f'seg/{tag}/grad/{k}'] = p.grad.numpy()
        if 'emb_padding_idx' in kw:
            assert float(mod.class_emb.weight.grad[7].abs().max()) == 0.0 and (xs == 7).any()
        for k, v in mod.state_dict().items():
            out[f'seg/{tag}/sd/{k}'] = v.detach().numpy()
        out[f'seg/{tag}/x'] = xs.numpy(); out[f'seg/{tag}/g'] = gs.numpy(); out[f'seg/{tag}/tok'] = tk.detach().numpy()

    # ---- blocks without qkv bias
    torch.manual_seed(3)
    L, D, heads = 1, 64, 2
    blocks = nn.Sequential(*[rutil.Block(D, heads, mlp_ratio=4, qkv_bias=False, norm_layer=partial(nn.LayerNorm, eps=1e-6)) for _ in range(L)]).train()
    assert not any(k.endswith('qkv.bias') for k in blocks.state_dict())
    x = torch.randn(3, 20, D, generator=g)
    gy = torch.randn(3, 20, D, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = blocks(xr); yr.backward(gy)
    sdo = {k: v.detach().clone().requires_grad_(True) for k, v in blocks.state_dict().items()}
    full = dict(sdo)
    for l in range(L):
        full[f'{l}.attn.qkv.bias'] = torch.zeros(3 * D)
    xo = x.clone().requires_grad_(True)
    yo = xo
    for l in range(L):
        yo = orc.block(yo, full, f'{l}.', heads, 1e-6)
    yo.backward(gy)
    assert rel(yo, yr) < 1e-6 and rel(xo.grad, xr.grad) < 1e-5
    for k, p in blocks.named_parameters():
        assert rel(sdo[k].grad, p.grad) < 1e-5, k
        out['nobias/sd/' + k] = p.detach().numpy(); out['nobias/grad/' + k] = p.grad.numpy()
    out.update({'nobias/x': x.numpy(), 'nobias/gy': gy.numpy(), 'nobias/y': yr.detach().numpy(), 'nobias/dx': xr.grad.numpy()})

    path = os.path.join(HERE, 'options.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB,', len(out), 'arrays')


if __name__ == '__main__':
    main()
