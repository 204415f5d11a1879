"""DropPath golden FROM THE REFERENCE: the 3-modality mini MultiMAE of make_golden.py with drop_path_rate = 0.25
(multimae_utils.py:105-135, Block.forward :229-232), forward + losses + backward on CPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_droppath.py

The reference draws its stochastic-depth masks with torch.rand on the input's device: on CPU that is the seeded CPU
generator, so replaying the generator (mask-sampler draws first, then two (B,1,1) draws per block in execution order)
recovers the very uniforms it used.  They are stored as explicit inputs, the oracle is checked against the reference on the
spot, and the engine test feeds the same draws through multimae_utils._drop_path_rand.
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
from sketch import store  # noqa: E402

RATE = 0.25


def main():
    torch.set_num_threads(8)
    rm, ria, roa, rc = mg.import_reference()
    orc = mg.load_oracle()
    doms, P, S, B, nvis = ['rgb', 'depth', 'semseg'], 8, 32, 3, 12
    depth = 4
    torch.manual_seed(0)
    # build_ref with a deeper encoder and stochastic depth
    ins = {}
    for d in doms:
        if d == 'semseg':
            ins[d] = ria.SemSegInputAdapter(num_classes=133, dim_class_emb=16, interpolate_class_emb=False, stride_level=4,
                                            patch_size_full=P, image_size=S)
        else:
            ins[d] = ria.PatchedInputAdapter(num_channels=3 if d == 'rgb' else 1, stride_level=1, patch_size_full=P, image_size=S)
    outs = {}
    for key, task in [(d, d) for d in doms] + [('norm_rgb', 'rgb')]:
        ch = {'rgb': 3, 'depth': 1, 'semseg': 133}[task]
        outs[key] = roa.SpatialOutputAdapter(num_channels=ch, stride_level=4 if task == 'semseg' else 1, patch_size_full=P,
                                             dim_tokens=64, depth=1, num_heads=2, use_task_queries=True, task=task,
                                             context_tasks=list(doms), use_xattn=True, image_size=S)
    model = rm.MultiMAE(ins, outs, num_global_tokens=1, dim_tokens=128, depth=depth, num_heads=2, mlp_ratio=4, qkv_bias=True,
                        drop_path_rate=RATE, norm_layer=partial(nn.LayerNorm, eps=1e-6)).train()
    g = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad and (n.endswith('bias') or 'mask_token' in n):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    x = mg.make_inputs(doms, B, S)
    # seed 5: both kept and dropped samples occur in the 8 draws (checked below)
    preds, masks, losses, (tm, ids_keep, ids_restore) = mg.ref_step(model, rc, x, P, nvis, seed=5)
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    # replay the generator: sampler draws, then two draws per block
    torch.manual_seed(5)
    npt = [(S // P) ** 2] * len(doms)
    dist, task_noise, all_noise = orc.draw_mask_randoms(B, npt, 1.0)
    spt = orc.samples_per_task_from_dirichlet(dist, nvis)
    m_all, k_o, r_o = orc.masks_from_noise(spt, task_noise, all_noise, nvis)
    assert torch.equal(k_o, ids_keep) and torch.equal(r_o, ids_restore)
    rates = orc.drop_path_rates(RATE, depth)
    us = []
    for l in range(depth):
        if rates[l] > 0:
            us.append((torch.rand((B, 1, 1)).view(B), torch.rand((B, 1, 1)).view(B)))
        else:
            us.append(None)
    kept = [float((1 - rates[l] + u).floor().sum()) for l, pr in enumerate(us) if pr is not None for u in pr]
    assert 0 < sum(kept) < B * len(kept), 'choose a seed where some but not all paths drop'

    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sdo = {k: v.clone().requires_grad_(model.state_dict(keep_vars=True)[k].requires_grad) for k, v in sd.items()}
    cfg = orc.standard_config(doms, patch_size=P, image_size=S, dim_tokens=128, depth=depth, num_heads=2, dec_dim=64, dec_depth=1,
                              dec_heads=2, dim_class_emb=16)
    po = orc.multimae_forward(x, sdo, cfg, ids_keep, ids_restore, drop_path_rate=RATE, drop_path_u=us)
    lo = orc.pretrain_losses(po, x, m_all, cfg, {d: n for d, n in zip(doms, npt)})
    sum(lo.values()).backward()
    for k in preds:
        assert (po[k] - preds[k]).abs().max().item() < 2e-5, k
        assert abs(float(lo[k]) - float(losses[k])) < 1e-5, k
    for n, gref in grads.items():
        rel = float((sdo[n].grad - gref).norm() / (gref.norm() + 1e-12))
        assert rel < 1e-4, (n, rel)
    print('drop-path mini: oracle == reference (preds, losses, %d grads); kept paths %s' % (len(grads), kept))

    # weights and inputs are NOT stored: the test rebuilds them from the same seeds (the engine's seeded initialisation is
    # bit-identical to the reference's, tests/test_boundary_cpu.py); the checksum below guards that reconstruction
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
    out['u'] = np.stack([np.stack([pr[0].numpy(), pr[1].numpy()]) if pr is not None else np.full((2, B), -1.0, np.float32) for pr in us])
    np.savez_compressed(os.path.join(HERE, 'mini_droppath.npz'), **out)
    print('wrote mini_droppath.npz', os.path.getsize(os.path.join(HERE, 'mini_droppath.npz')))


if __name__ == '__main__':
    main()
