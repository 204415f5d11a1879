"""Debug aid: one eager step + one replayed step vs two eager steps on the mini model -- which arena (grad / param / m / v)
and which tensors differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]
import torch
import multimae_amd as M
from multimae_amd.graph import StepGraph
from multimae_amd.optim import FusedAdamW
from helpers import MINI, build_mini_engine, load_mini
DEV = 'cuda'


def run(use_graph, streams=True, n=2):
    g = load_mini()
    model = build_mini_engine(); model.load_state_dict(g['sd']); model.to(DEV)
    arena = model.build_arena()
    opt = FusedAdamW(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    xd = {k: v.to(DEV) for k, v in g['x'].items()}
    tgt = dict(xd, norm_rgb=xd['rgb'])
    P = MINI['P']
    fns = {'rgb': M.MaskedMSELoss(P, 1), 'depth': M.MaskedL1Loss(P, 1), 'semseg': M.MaskedCrossEntropyLoss(P, 4), 'norm_rgb': M.MaskedMSELoss(P, 1, norm_pix=True)}
    tm = {d: g['mask'][d].to(DEV) for d in MINI['doms']}
    ids = (g['ids_keep'].to(DEV), g['ids_restore'].to(DEV))
    model.generate_random_masks = lambda *a, **k: (tm, ids[0], ids[1])
    out = {}
    def step():
        opt.zero_grad()
        preds, masks = model(xd, num_encoded_tokens=MINI['nvis'], alphas=1.0, fp32_output_adapters=['semseg'])
        mk = dict(masks, norm_rgb=masks['rgb'])
        loss = sum(fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds)
        loss.backward()
        opt.step()
        out['loss'] = loss
        return loss
    M.engine.set_direct_grads(True); M.engine.set_adapter_streams(streams); M.engine.set_wgrad_stream(streams)
    step()
    runf = StepGraph(step) if use_graph else step
    for _ in range(n - 1):
        runf()
    torch.cuda.synchronize()
    return dict(loss=float(out['loss']), grad=arena.grad.clone(), param=arena.param.clone(), m=opt.m.clone(), v=opt.v.clone(),
                gn=float(opt.grad_norm), arena=arena)


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

for streams in (True, False):
    e, gr = run(False, streams), run(True, streams)
    print(f'streams={streams}: loss eager {e["loss"]:.5f} graph {gr["loss"]:.5f}  gradnorm {e["gn"]:.4f} / {gr["gn"]:.4f}')
    for k in ('grad', 'param', 'm', 'v'):
        print(f'   {k:6s} rel diff {rel(gr[k], e[k]):.3e}')
    a = e['arena']
    worst = []
    for n in a.names:
        if not a.trainable[n]:
            continue
        o, s = a.offsets[n], a.sizes[n]
        worst.append((rel(gr['grad'][o:o + s], e['grad'][o:o + s]), n))
    for r, n in sorted(worst, reverse=True)[:12]:
        print(f'      grad {n:60s} {r:.3e}')
