"""Per-tensor gradient goldens at the BENCH geometry, generated FROM THE REFERENCE ITSELF.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_geometry.py

Runs the reference's own classes (SURVEY.md Appendix B import recipe) on CPU fp32 for

  cfg3  ViT-B, RGB+depth+semseg, 224^2, patch 16, 98 visible tokens, 4 decoders, B=4   (BASELINE.json configs[2])
  cfg2  ViT-B, RGB only,         224^2, patch 16, 98 visible tokens, 2 decoders, B=4   (BASELINE.json configs[1])

with the seeded recipe of scalars.json (torch.manual_seed(0); build; inputs; torch.manual_seed(1); forward), so a test
re-creates the identical weights / inputs / masks from the seeds and only the reference's OUTPUTS are stored:

  * every gradient tensor with <= 4096 elements whole (all biases, LayerNorm affines, task_embeddings, mask_token,
    global_tokens, ...),
  * every larger gradient (and every prediction) as a sketch (tests/golden/sketch.py): its 2-norm, 8 dot products with fixed
    +-1 vectors (an integer hash of the element index, no RNG) and a strided sample of 2048 elements.
    For a difference d = g_test - g_ref the 8 projections estimate |d|_2 (E[(r.d)^2] = |d|^2), so a test gets the
    relative L2 error of the whole 2-million-element tensor from 8 numbers.

The oracle (oracle/multimae_oracle.py) is cross-checked against the reference on the same step before anything is
written.  Nothing under /root/reference is copied; only its outputs are recorded.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from sketch import store  # noqa: E402


def run_case(rm, ria, roa, rc, orc, name, doms, out):
    P, S, B, nvis = 16, 224, 4, 98
    torch.manual_seed(0)
    model = mg.build_ref(rm, ria, roa, doms, P, S)
    x = mg.make_inputs(doms, B, S)
    preds, masks, losses, (tm, ids_keep, ids_restore) = mg.ref_step(model, rc, x, P, nvis, seed=1)
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    print(name, {k: round(float(v), 6) for k, v in losses.items()}, 'grad_norm', mg.grad_norm(model), flush=True)

    # pin the oracle on this very step (same weights, same ids)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sdo = {k: v.clone().requires_grad_(model.state_dict(keep_vars=True)[k].requires_grad) for k, v in sd.items()}
    cfg = orc.standard_config(doms)
    po = orc.multimae_forward(x, sdo, cfg, ids_keep, ids_restore)
    m_all = torch.cat([tm[d] for d in doms], 1)
    lo = orc.pretrain_losses(po, x, m_all, cfg, {d: 196 for d in doms})
    sum(lo.values()).backward()
    for k in preds:
        assert (po[k] - preds[k]).abs().max().item() < 1e-4, k
        assert abs(float(lo[k]) - float(losses[k])) < 2e-5, (k, float(lo[k]), float(losses[k]))
    worst = 0.0
    for n, g in grads.items():
        rel = float((sdo[n].grad - g).norm() / (g.norm() + 1e-30))
        worst = max(worst, rel)
        assert rel < 2e-4, (n, rel)
    print(f'{name}: oracle == reference (preds, losses, {len(grads)} grads; worst grad rel {worst:.2e})', flush=True)

    pre = name + '/'
    for k, v in losses.items():
        out[pre + 'loss/' + k] = np.float64(float(v))
    out[pre + 'grad_norm'] = np.float64(mg.grad_norm(model))
    out[pre + 'ids_keep_checksum'] = np.int64(int(ids_keep.sum()))
    for k, v in preds.items():
        store(out, pre + 'pred/' + k, v)
    for n, g in grads.items():
        store(out, pre + 'grad/' + n, g)
    n_whole = sum(1 for g in grads.values() if g.numel() <= 4096)
    print(f'{name}: {n_whole} gradients stored whole, {len(grads) - n_whole} as sketches', flush=True)


def main():
    torch.set_num_threads(8)
    rm, ria, roa, rc = mg.import_reference()
    orc = mg.load_oracle()
    out = {}
    run_case(rm, ria, roa, rc, orc, 'cfg3', ['rgb', 'depth', 'semseg'], out)
    run_case(rm, ria, roa, rc, orc, 'cfg2', ['rgb'], out)
    np.savez_compressed(os.path.join(HERE, 'geometry_grads.npz'), **out)
    print('wrote geometry_grads.npz', os.path.getsize(os.path.join(HERE, 'geometry_grads.npz')), 'bytes')


if __name__ == '__main__':
    main()
