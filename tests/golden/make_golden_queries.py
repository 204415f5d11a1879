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


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


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
    np.savez_compressed(os.path.join(HERE, 'mask_queries.npz'), **out)
    print('wrote mask_queries.npz', os.path.getsize(os.path.join(HERE, 'mask_queries.npz')), 'bytes')


if __name__ == '__main__':
    main()
