"""Golden fixture for the truncated depth standardisation, made BY THE REFERENCE'S OWN LINES.

The reference has no function for this step: it is inline in the training loop (run_pretraining_multimae.py:487-492).  This
script reads those source lines from the read-only checkout at generation time, executes them unmodified on CPU tensors and
records input -> output (nothing is copied into the repo), then cross-checks oracle.truncated_depth_standardize on the spot.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_depth.py          (build container only)

Cases: continuous depth (no ties), and 8-bit quantised depth where the 10 % / 90 % cut points fall inside runs of equal values
(exercises the tie handling of the selection kernel), at 32x32 and at the real 224x224 geometry (one sample).
"""
import os
import sys
import textwrap

import numpy as np
import torch
from einops import rearrange

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SCRIPT = '/root/reference/run_pretraining_multimae.py'


def reference_lines():
    lines = open(REF_SCRIPT).read().splitlines()
    first = next(i for i, l in enumerate(lines) if "if standardize_depth and 'depth' in tasks_dict" in l)
    block = lines[first:first + 5]
    assert 'torch.sort' in block[2] and 'trunc_depth.var' in block[4], block
    return textwrap.dedent('\n'.join(block))


def run_reference(code, depth):
    ns = dict(torch=torch, rearrange=rearrange, standardize_depth=True, tasks_dict={'depth': depth.clone()})
    exec(compile(code, REF_SCRIPT, 'exec'), ns)
    return ns['tasks_dict']['depth']


def main():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import multimae_oracle as orc
    code = reference_lines()
    torch.manual_seed(11)
    cases = {
        'cont32': torch.rand(3, 1, 32, 32) * 9.0 + 0.5,
        'quant32': torch.randint(0, 24, (4, 1, 32, 32)).float() / 8.0,          # ~43 copies of every value: ties at both cuts
        'const32': torch.cat([torch.full((1, 1, 32, 32), 2.5), torch.rand(1, 1, 32, 32)]),   # a constant map: var = 0 -> 1/sqrt(eps)
        'real224': (torch.rand(1, 1, 224, 224) ** 2) * 80.0,
        'quant224': torch.randint(0, 256, (1, 1, 224, 224)).float() / 255.0,
    }
    out = {}
    for k, x in cases.items():
        y = run_reference(code, x)
        yo = orc.truncated_depth_standardize(x)
        assert torch.allclose(yo, y, rtol=0, atol=0) or float((yo - y).abs().max()) < 1e-6 * float(y.abs().max()), k
        out['x/' + k], out['y/' + k] = x.numpy(), y.numpy()
    np.savez_compressed(os.path.join(HERE, 'depth_std.npz'), **out)
    print('wrote depth_std.npz:', {k: tuple(v.shape) for k, v in out.items() if k.startswith('x/')})


if __name__ == '__main__':
    main()
