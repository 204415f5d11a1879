"""Golden fixture for the API edge paths of MultiMAE.forward / generate_random_masks, made by the reference model itself on the
mini geometry with the weights of mini_fwd_bwd.npz:

  uniform/*   generate_random_masks(sample_tasks_uniformly=True, alphas=[1.0, 0.5, 2.0])  (multimae.py:148-162,182-186):
              every torch.rand draw, and the resulting masks / ids
  given/*     forward(task_masks=...) for B = 1 (the notebook path, multimae.py:335-338): preds and returned masks
  nomask/*    forward(mask_inputs=False) (multimae.py:324-326): preds with every token encoded

Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_api.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import build_ref, import_reference  # noqa: E402


def main():
    torch.set_num_threads(8)
    rm, ria, roa, rc = import_reference()
    z = np.load(os.path.join(HERE, 'mini_fwd_bwd.npz'))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}
    x = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('x/')}
    doms, P, S, nvis = ['rgb', 'depth', 'semseg'], 8, 32, 12
    model = build_ref(rm, ria, roa, doms, P, S, enc=(128, 2, 2), dec_dim=64, dec_depth=1, dec_heads=2, class_emb=16)
    model.load_state_dict(sd)
    model.eval()
    out = {}

    # ---- uniform task sampling with per-task alphas: record the device-noise draws
    draws = []
    real_rand, real_rand_like = torch.rand, torch.rand_like

    def spy_rand(*a, **k):
        t = real_rand(*a, **k)
        draws.append(t.clone())
        return t

    def spy_rand_like(*a, **k):
        t = real_rand_like(*a, **k)
        draws.append(t.clone())
        return t
    torch.manual_seed(5)
    toks = {d: torch.zeros(3, 16, 4) for d in doms}
    torch.rand, torch.rand_like = spy_rand, spy_rand_like
    try:
        tm, ik, ir = model.generate_random_masks(toks, nvis, alphas=[1.0, 0.5, 2.0], sample_tasks_uniformly=True)
    finally:
        torch.rand, torch.rand_like = real_rand, real_rand_like
    assert len(draws) == 4
    for i, t in enumerate(draws):
        out[f'uniform/noise{i}'] = t.numpy()
    out['uniform/ids_keep'], out['uniform/ids_restore'] = ik.numpy(), ir.numpy()
    for d in doms:
        out['uniform/mask/' + d] = tm[d].numpy()

    # ---- given task masks, B = 1
    x1 = {k: v[:1] for k, v in x.items()}
    torch.manual_seed(6)
    given = {d: (torch.rand(1, 16) < 0.7).long() for d in doms}
    with torch.no_grad():
        preds, masks = model(x1, task_masks={d: m.clone() for d, m in given.items()})
    for d in doms:
        out['given/in/' + d] = given[d].numpy()
        out['given/mask/' + d] = masks[d].numpy()
    for k, v in preds.items():
        out['given/pred/' + k] = v.numpy()

    # ---- no masking: every token encoded
    x2 = {k: v[:2] for k, v in x.items()}
    torch.manual_seed(7)
    with torch.no_grad():
        preds, masks = model(x2, mask_inputs=False)
    for k, v in preds.items():
        out['nomask/pred/' + k] = v.numpy()
    for d in doms:
        assert int(masks[d].sum()) == 0
    np.savez_compressed(os.path.join(HERE, 'api_paths.npz'), **out)
    print('wrote api_paths.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
