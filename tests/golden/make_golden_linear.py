"""Golden fixture for LinearOutputAdapter (output_adapters.py:285-356) from the reference class itself: seeded init, forward
and gradients for both pooling modes.  Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_linear.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, load_oracle  # noqa: E402


def main():
    rm, ria, roa, rc = import_reference()
    orc = load_oracle()
    out = {}
    for mode, mean in (('mean', True), ('last', False)):
        torch.manual_seed(31)
        head = roa.LinearOutputAdapter(num_classes=7, dim_tokens_enc=16, use_mean_pooling=mean, init_scale=0.5)
        with torch.no_grad():
            head.head.bias.add_(torch.randn(7) * 0.1)
            head.norm.weight.add_(torch.randn(16) * 0.1)
        x = torch.randn(3, 5, 16, requires_grad=True)
        y = head(x)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        sd = {k: v.detach().clone() for k, v in head.state_dict().items()}
        assert torch.allclose(orc.linear_output_adapter(x.detach(), sd, use_mean_pooling=mean), y.detach(), atol=1e-6)
        for k, v in sd.items():
            out[f'{mode}/sd/{k}'] = v.numpy()
        out[f'{mode}/x'], out[f'{mode}/y'], out[f'{mode}/w'] = x.detach().numpy(), y.detach().numpy(), w.numpy()
        out[f'{mode}/dx'] = x.grad.numpy()
        for n, p in head.named_parameters():
            out[f'{mode}/grad/{n}'] = p.grad.numpy()
    # seeded-init known answer (same RNG consumption as the reference's init walk)
    torch.manual_seed(32)
    h = roa.LinearOutputAdapter(num_classes=10, dim_tokens_enc=32)
    out['init/head.weight'] = h.head.weight.detach().numpy()
    np.savez_compressed(os.path.join(HERE, 'linear_head.npz'), **out)
    print('wrote linear_head.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
