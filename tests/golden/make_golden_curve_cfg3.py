"""Loss curves of the REFERENCE at the HEADLINE geometry (BASELINE configs[2] / cfg3: ViT-B, RGB + depth + semseg, 224 x 224,
patch 16, 98 of 588 tokens, the four output adapters and their losses), B = 4, 12 AdamW steps (lr 1.5e-4, betas (.9, .95),
wd .05 on every trainable tensor as utils/optim_factory.py:138-155 builds it), mask draws seeded 2000 + step:

  reference_fp32            the reference classes in fp32 -- what the engine's fp32 parity mode must follow to 1e-4 per step
                            (north_star: "loss curve matching CPU reference to 1e-4")
  reference_bf16_autocast   the same run under torch.autocast('cpu', bfloat16) with fp32_output_adapters=['semseg'] (the
                            pre-training YAML's setting, multimae.py:367-377) -- the reference's OWN reduced-precision deviation,
                            against which the engine's bf16 mode (and its fp32-adapter storage forms) are calibrated

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_curve_cfg3.py
"""
import contextlib
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

STEPS = 12
DOMS, P, S, B, NVIS = ['rgb', 'depth', 'semseg'], 16, 224, 4, 98
SEED0 = 2000


def step_masks(orc, step):
    torch.manual_seed(SEED0 + step)
    dist, tn, an = orc.draw_mask_randoms(B, [196, 196, 196], 1.0)
    spt = orc.samples_per_task_from_dirichlet(dist, NVIS)
    return orc.masks_from_noise(spt, tn, an, NVIS)


def run(rm, ria, roa, rc, orc, bf16):
    torch.manual_seed(0)
    model = mg.build_ref(rm, ria, roa, DOMS, P, S)
    x = mg.make_inputs(DOMS, B, S)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05)
    fns = {'rgb': rc.MaskedMSELoss(P, 1), 'depth': rc.MaskedL1Loss(P, 1), 'semseg': rc.MaskedCrossEntropyLoss(P, 4),
           'norm_rgb': rc.MaskedMSELoss(P, 1, norm_pix=True)}
    tgt = dict(x, norm_rgb=x['rgb'])
    total, per_task = [], {k: [] for k in fns}
    for step in range(1, STEPS + 1):
        mask_all, ik, ir = step_masks(orc, step)
        tm = {d: mask_all[:, i * 196:(i + 1) * 196] for i, d in enumerate(DOMS)}
        model.generate_random_masks = lambda *a, **k: (tm, ik, ir)
        with (torch.autocast('cpu', dtype=torch.bfloat16) if bf16 else contextlib.nullcontext()):
            preds, masks = model(x, num_encoded_tokens=NVIS, alphas=1.0, fp32_output_adapters=['semseg'] if bf16 else [])
            mk = dict(masks, norm_rgb=masks['rgb'])
            losses = {k: fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds}
            loss = sum(losses.values())
        opt.zero_grad()
        loss.backward()
        opt.step()
        total.append(float(loss.detach()))
        for k, v in losses.items():
            per_task[k].append(float(v.detach()))
        print(('bf16' if bf16 else 'fp32'), step, total[-1], flush=True)
    return total, per_task


def main():
    torch.set_num_threads(8)
    rm, ria, roa, rc = mg.import_reference()
    orc = mg.load_oracle()
    f32, f32_task = run(rm, ria, roa, rc, orc, False)
    b16, b16_task = run(rm, ria, roa, rc, orc, True)
    dev = [abs(a - b) for a, b in zip(f32, b16)]
    print('max |bf16 - fp32| of the reference itself:', max(dev), 'mean', sum(dev) / len(dev))
    json.dump({'recipe': f'cfg3: ViT-B RGB+depth+semseg 224^2 patch 16, 98 tokens, B={B}, AdamW lr 1.5e-4 (.9,.95) wd .05, mask seed {SEED0}+step, '
                         f'{STEPS} steps; model seed 0 then inputs (tests/golden/make_golden.py: build_ref, make_inputs)',
               'reference_fp32': f32, 'reference_fp32_per_task': f32_task,
               'reference_bf16_autocast': b16, 'reference_bf16_autocast_per_task': b16_task,
               'torch_version': torch.__version__},
              open(os.path.join(HERE, 'curve_cfg3.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
