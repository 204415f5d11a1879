"""How far does the REFERENCE move when IT runs in bf16?  The tolerance calibration of the bf16-mode parity tests
(tests/test_parity_geometry_gpu.py): the reference's own classes (imported from /root/reference, SURVEY.md Appendix B recipe) run
the seeded golden step once in fp32 and once under torch.autocast('cpu', dtype=torch.bfloat16) -- the CPU form of the autocast the
reference trains under (run_pretraining_multimae.py:500; every adapter in bf16: the reference switches its fp32 adapters off with
torch.cuda.amp.autocast(enabled=False), which does not reach the CPU autocast) -- and every gradient tensor of the bf16 run is
compared with the fp32 run.  Needs the reference checkout, i.e. runs in the build container only; the numbers are
committed under profiles/.

    PYTHONDONTWRITEBYTECODE=1 python tools/reference_bf16_noise.py [cfg3] [cfg5] > profiles/r03_reference_bf16_noise.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as mg  # noqa: E402


def one_case(name, rm, ria, roa, rc):
    doms, P, S = ['rgb', 'depth', 'semseg'], 16, 224
    B, nvis = (4, 98) if name == 'cfg3' else (2, 196)
    runs = {}
    for mode in ('fp32', 'bf16'):
        torch.manual_seed(0)
        if name == 'cfg3':
            model = mg.build_ref(rm, ria, roa, doms, P, S)
        else:
            import make_golden_geometry_cfg5 as g5
            model = g5.build_ref_large(rm, ria, roa, doms, P, S)
        x = mg.make_inputs(doms, B, S)
        if mode == 'bf16':
            with torch.autocast('cpu', dtype=torch.bfloat16):
                preds, masks, losses, _ = mg.ref_step(model, rc, x, P, nvis, seed=1)
        else:
            preds, masks, losses, _ = mg.ref_step(model, rc, x, P, nvis, seed=1)
        runs[mode] = ({n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.grad is not None},
                      {k: float(v) for k, v in losses.items()}, {k: v.detach().double() for k, v in preds.items()})
    g32, l32, p32 = runs['fp32']
    g16, l16, p16 = runs['bf16']
    rows, sq_d, sq_r = [], 0.0, 0.0
    for n, g in g32.items():
        e = float((g16[n] - g).norm() / (g.norm() + 1e-30))
        rows.append((e, g.numel(), n))
        sq_d += float((g16[n] - g).pow(2).sum()); sq_r += float(g.pow(2).sum())
    rows.sort(reverse=True)
    print(f'# {name}: the reference under bf16 autocast (CPU) vs its own fp32 run, B={B}, {nvis} visible tokens, seeds as tests/golden/make_golden_geometry*.py')
    print(f'# losses fp32 {l32}')
    print(f'# losses bf16 {l16}')
    print('# preds rel err', {k: float((p16[k] - p32[k]).norm() / p32[k].norm()) for k in p32})
    print(f'# gradients: global {(sq_d / sq_r) ** 0.5:.3e}  worst tensor {rows[0][0]:.3e}  ({len(rows)} tensors)')
    for e, numel, n in rows[:12]:
        print(f'{e:.3e} {numel:9d} {n}')
    sys.stdout.flush()


def main():
    torch.set_num_threads(8)
    rm, ria, roa, rc = mg.import_reference()
    for name in (sys.argv[1:] or ['cfg3', 'cfg5']):
        one_case(name, rm, ria, roa, rc)


if __name__ == '__main__':
    main()
