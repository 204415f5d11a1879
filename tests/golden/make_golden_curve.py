"""Loss curves of the REFERENCE on the cfg1 recipe the GPU loss-curve tests use (ViT-Tiny 192/12/3, RGB 64x64, patch 8, 49 of 64
tokens, B = 4, AdamW lr 1.5e-4 betas (.9,.95) wd .05, mask draws seeded 1000 + step): fp32, and under torch.autocast('cpu',
bfloat16) -- the reference's OWN bf16 deviation per step, against which the engine's bf16 tolerance is calibrated.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_curve.py
"""
import contextlib
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

STEPS = 20


def run(rm, ria, roa, rc, orc, bf16):
    torch.manual_seed(0)
    model = mg.build_ref(rm, ria, roa, ['rgb'], 8, 64, enc=(192, 12, 3), posemb_size=224)
    x = mg.make_inputs(['rgb'], 4, 64)
    opt = torch.optim.AdamW(model.parameters(), lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05)
    fns = {'rgb': rc.MaskedMSELoss(8, 1), 'norm_rgb': rc.MaskedMSELoss(8, 1, norm_pix=True)}
    out = []
    for step in range(1, STEPS + 1):
        torch.manual_seed(1000 + step)
        dist, tn, an = orc.draw_mask_randoms(4, [64], 1.0)
        spt = orc.samples_per_task_from_dirichlet(dist, 49)
        mask_all, ik, ir = orc.masks_from_noise(spt, tn, an, 49)
        model.generate_random_masks = lambda *a, **k: ({'rgb': mask_all}, ik, ir)
        with (torch.autocast('cpu', dtype=torch.bfloat16) if bf16 else contextlib.nullcontext()):
            preds, masks = model(x, num_encoded_tokens=49, alphas=1.0)
            loss = fns['rgb'](preds['rgb'].float(), x['rgb'], mask=masks['rgb']) + fns['norm_rgb'](preds['norm_rgb'].float(), x['rgb'], mask=masks['rgb'])
        opt.zero_grad()
        loss.backward()
        opt.step()
        out.append(float(loss.detach()))
    return out


def main():
    torch.set_num_threads(8)
    rm, ria, roa, rc = mg.import_reference()
    orc = mg.load_oracle()
    fp32 = run(rm, ria, roa, rc, orc, False)
    bf16 = run(rm, ria, roa, rc, orc, True)
    dev = [abs(a - b) for a, b in zip(fp32, bf16)]
    print('max |bf16 - fp32| of the reference itself:', max(dev))
    json.dump({'recipe': 'cfg1: ViT-Tiny RGB 64x64 patch 8, 49 tokens, B=4, AdamW lr 1.5e-4 (.9,.95) wd .05, mask seed 1000+step, 20 steps',
               'reference_fp32': fp32, 'reference_bf16_autocast': bf16, 'torch_version': torch.__version__},
              open(os.path.join(HERE, 'curve_cfg1.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
