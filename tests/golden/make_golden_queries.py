"""Golden FROM THE REFERENCE for the mask-token-query branch of SpatialOutputAdapter.get_queries_and_context
(output_adapters.py:213-220): the reference's own class on CPU, oracle == reference asserted on the spot, weights / inputs /
prediction / every gradient stored.  Two cases:

  * 'notin':  the adapter's task ('depth') is not among the encoder inputs (run_pretraining_multimae.py --in_domains rgb
              --out_domains rgb-depth builds it with context_tasks = in_domains, :253-283): queries = mask_token + pos_emb
  * 'noq':    use_task_queries=False on an input task: queries = mask_token + pos_emb + task_embeddings[task]

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_queries.py        ->  tests/golden/mask_queries.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from sketch import store  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def e2e(rm, ria, roa, rc, orc, out):
    """The whole model the way run_pretraining_multimae.py:243-293 builds it for --in_domains rgb --out_domains rgb-depth: one input
    adapter, two output adapters with context_tasks = in_domains -- the depth adapter decodes a task the encoder never saw."""
    from functools import partial
    from torch import nn
    P, S, B, nvis = 8, 32, 3, 7
    torch.manual_seed(31)
    ins = {'rgb': ria.PatchedInputAdapter(num_channels=3, stride_level=1, patch_size_full=P, image_size=S)}
    outs = {t: roa.SpatialOutputAdapter(num_channels=c, stride_level=1, patch_size_full=P, dim_tokens=64, depth=1, num_heads=2, use_task_queries=True,
                                        task=t, context_tasks=['rgb'], use_xattn=True, image_size=S) for t, c in (('rgb', 3), ('depth', 1))}
    model = rm.MultiMAE(ins, outs, num_global_tokens=1, dim_tokens=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True,
                        norm_layer=partial(nn.LayerNorm, eps=1e-6)).train()
    x = {'rgb': torch.randn(B, 3, S, S), 'depth': torch.randn(B, 1, S, S)}
    captured = {}
    orig = model.generate_random_masks

    def spy(*a, **k):
        o = orig(*a, **k)
        captured['ids'] = o
        return o
    model.generate_random_masks = spy
    torch.manual_seed(32)
    preds, masks = model({'rgb': x['rgb']}, num_encoded_tokens=nvis, alphas=1.0)
    model.generate_random_masks = orig
    tm, ids_keep, ids_restore = captured['ids']
    fns = {'rgb': rc.MaskedMSELoss(P, 1), 'depth': rc.MaskedL1Loss(P, 1)}
    losses = {k: fns[k](preds[k].float(), x[k], mask=masks.get(k, None)) for k in preds}       # run_pretraining_multimae.py:516-520
    assert 'depth' not in masks
    model.zero_grad()
    sum(losses.values()).backward()
    # oracle on the same step
    cfg = orc.OracleConfig([orc.DomainSpec('rgb', 'image', 3, 1)], [('rgb', 'rgb'), ('depth', 'depth')], P, S, 128, 2, 2, 1, 64, 1, 2,
                           out_only_domains=[orc.DomainSpec('depth', 'image', 1, 1)])
    sdo = {k: v.detach().clone().requires_grad_(model.state_dict(keep_vars=True)[k].requires_grad) for k, v in model.state_dict().items()}
    po = orc.multimae_forward({'rgb': x['rgb']}, sdo, cfg, ids_keep, ids_restore)
    for k in preds:
        assert rel(po[k], preds[k]) < 1e-6, (k, rel(po[k], preds[k]))
    lo = {'rgb': orc.masked_mse(po['rgb'], x['rgb'], tm['rgb'], P), 'depth': orc.masked_l1(po['depth'], x['depth'], None, P)}
    for k in lo:
        assert abs(float(lo[k]) - float(losses[k])) < 1e-5, (k, float(lo[k]), float(losses[k]))
    sum(lo.values()).backward()
    n_g = 0
    for n_, p in model.named_parameters():
        if p.grad is not None:
            assert rel(sdo[n_].grad, p.grad) < 2e-5, (n_, rel(sdo[n_].grad, p.grad))
            store(out, 'e2e/grad/' + n_, p.grad)          # weights are NOT stored: the test re-creates them from the seed (31)
            n_g += 1
    out['e2e/sd_checksum'] = np.float64(sum(float(v.double().sum()) for v in model.state_dict().values()))
    out.update({'e2e/x/rgb': x['rgb'].numpy(), 'e2e/x/depth': x['depth'].numpy(), 'e2e/ids_keep': ids_keep.numpy(), 'e2e/ids_restore': ids_restore.numpy(),
                'e2e/mask/rgb': tm['rgb'].numpy()})
    for k in preds:
        store(out, 'e2e/pred/' + k, preds[k])
        out['e2e/loss/' + k] = np.float64(float(losses[k]))
    print('e2e (in rgb, out rgb + depth): oracle == reference on preds, losses and', n_g, 'gradients')


def main():
    torch.set_num_threads(8)
    rm, ria, roa, rc = mg.import_reference()
    orc = mg.load_oracle()
    out = {}
    B, Denc, D, P, S = 2, 96, 64, 4, 16
    n, G = 16, 1
    for tag, task, in_tasks, ctx_tasks, utq, nkeep in (('notin', 'depth', ['rgb'], ['rgb'], True, 7),
                                                         ('noq', 'rgb', ['rgb', 'depth'], ['rgb', 'depth'], False, 10)):
        torch.manual_seed(21 if tag == 'notin' else 22)
        C = 1 if task == 'depth' else 3
        ad = roa.SpatialOutputAdapter(num_channels=C, stride_level=1, patch_size_full=P, dim_tokens=D, depth=1, num_heads=2,
                                      use_task_queries=utq, task=task, context_tasks=ctx_tasks, image_size=S, dim_tokens_enc=Denc).train()
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():
            for n_, p in ad.named_parameters():
                if p.requires_grad and (n_.endswith('bias') or 'mask_token' in n_):
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
        nt = n * len(in_tasks)
        ids_shuffle = torch.argsort(torch.rand(B, nt, generator=g), dim=1)
        ids_restore = torch.argsort(ids_shuffle, dim=1)
        ids_keep = ids_shuffle[:, :nkeep]
        enc = torch.randn(B, nkeep + G, Denc, generator=g)
        gout = torch.randn(B, C, S, S, generator=g)
        info = {'tasks': {t: {'num_tokens': n, 'has_2d_posemb': True, 'start_idx': i * n, 'end_idx': (i + 1) * n} for i, t in enumerate(in_tasks)},
                'image_size': (S, S), 'num_task_tokens': nt, 'num_global_tokens': G}
        er = enc.clone().requires_grad_(True)
        pr = ad(er, info, ids_keep, ids_restore)
        pr.backward(gout)
        cfg = orc.standard_config(['rgb', 'depth'], patch_size=P, image_size=S, dim_tokens=Denc, depth=1, num_heads=2, dec_dim=D, dec_depth=1,
                                  dec_heads=2, extra_norm_pix=False)
        sd = {f'output_adapters.{task}.' + k: v.detach().clone() for k, v in ad.state_dict().items()}
        for k, p in ad.named_parameters():
            sd[f'output_adapters.{task}.' + k].requires_grad_(p.requires_grad)
        eo = enc.clone().requires_grad_(True)
        po = orc.spatial_adapter(eo, sd, cfg, task, task, {t: n for t in in_tasks}, ids_keep, ids_restore, (S, S), use_task_queries=utq)
        po.backward(gout)
        assert rel(po, pr) < 1e-6 and rel(eo.grad, er.grad) < 1e-5, (tag, rel(po, pr), rel(eo.grad, er.grad))
        for k, p in ad.named_parameters():
            if p.requires_grad:
                if p.grad is None:                         # a task embedding the forward never touched ('notin': rgb's gets context gradient only)
                    continue
                assert rel(sd[f'output_adapters.{task}.' + k].grad, p.grad) < 1e-5, (tag, k)
        for k, v in ad.state_dict().items():
            out[f'{tag}/sd/' + k] = v.detach().numpy()
        for k, p in ad.named_parameters():
            if p.requires_grad and p.grad is not None:
                out[f'{tag}/grad/' + k] = p.grad.numpy()
        out.update({f'{tag}/enc': enc.numpy(), f'{tag}/ids_keep': ids_keep.numpy(), f'{tag}/ids_restore': ids_restore.numpy(),
                    f'{tag}/gout': gout.numpy(), f'{tag}/pred': pr.detach().numpy(), f'{tag}/d_enc': er.grad.numpy()})
        print(tag, 'oracle == reference: pred', rel(po, pr), 'd_enc', rel(eo.grad, er.grad), '| grads stored:',
              sorted(k[len(tag) + 6:] for k in out if k.startswith(f'{tag}/grad/'))[:4], '...')
    e2e(rm, ria, roa, rc, orc, out)
    np.savez_compressed(os.path.join(HERE, 'mask_queries.npz'), **out)
    print('wrote mask_queries.npz', os.path.getsize(os.path.join(HERE, 'mask_queries.npz')), 'bytes')


if __name__ == '__main__':
    main()
