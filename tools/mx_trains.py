"""does the MX-fp8 mode TRAIN?  Two measurements on one GPU, written to gpurun_out/mx_trains.txt:

 1. 300 AdamW steps of the cfg3 model (ViT-B, RGB+depth+semseg, 98 visible tokens, B = 32, fixed synthetic batch, lr 1e-4 after a
    20-step warm-up) in mxfp8 mode and in bf16 mode from the same seeded initialisation, with the SAME masks every step (the
    sampler's draws are replayed from one generator seed per step): loss curves, final-loss gap, largest gap of the running means.
 2. the residual stream of the 24-layer ViT-L encoder (cfg5 geometry, B = 2, 197 tokens) after every block in mxfp8, bf16 and
    fp32 mode: relative 2-norm error against fp32 per layer (how the e4m3 rounding accumulates with depth).

    python tools/mx_trains.py [--steps 300] [--batch 32]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'mx_trains.txt'))
args = ap.parse_args()

import multimae_amd as M  # noqa: E402
from multimae_amd.optim import FusedAdamW  # noqa: E402
import bench  # noqa: E402

dev = torch.device('cuda', 0)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def curve(mode):
    torch.manual_seed(0)
    model, doms = bench.build_model('cfg3')
    model.to(dev)
    model.build_arena()
    M.engine.set_precision(mode)
    M.engine.set_direct_grads(True)
    M.engine.set_adapter_streams(True)
    M.engine.set_wgrad_stream(True)
    opt = FusedAdamW(model, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    x = bench.synthetic_batch(doms, args.batch, dev, seed=0)
    tgt = dict(x, norm_rgb=x['rgb'])
    fns = bench.loss_fns()
    out = []
    for it in range(args.steps):
        g = opt.param_groups[0]
        g['lr'] = 1e-4 * min(1.0, (it + 1) / 20)
        torch.manual_seed(1000 + it)                       # the sampler's Dirichlet draw (CPU generator) and its device noise
        torch.cuda.manual_seed(1000 + it)
        opt.zero_grad()
        preds, masks = model(x, num_encoded_tokens=98, alphas=1.0, sample_tasks_uniformly=False, fp32_output_adapters=['semseg'])
        mk = dict(masks, norm_rgb=masks['rgb'])
        loss = sum(fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds)
        loss.backward()
        opt.step(loss)
        out.append(loss.detach())
    torch.cuda.synchronize()
    c = opt.counters()
    M.engine.set_direct_grads(False)
    M.engine.set_precision('bf16')
    return [float(v) for v in out], c


say(f'# MX-fp8 against bf16: {args.steps} AdamW steps of cfg3 (ViT-B, RGB+depth+semseg, 98 visible tokens), B = {args.batch}, identical init / batch / masks')
res = {}
for mode in ('bf16', 'mxfp8'):
    res[mode], cnt = curve(mode)
    say(f'{mode:6s}: loss step 1 {res[mode][0]:.4f}  step {args.steps // 2} {res[mode][args.steps // 2 - 1]:.4f}  last {res[mode][-1]:.4f}  '
        f'mean of the last 20 {sum(res[mode][-20:]) / 20:.4f}  optimizer counters {cnt}')
a, b = res['bf16'], res['mxfp8']
run = lambda v, i, w=20: sum(v[max(0, i - w + 1):i + 1]) / (i - max(0, i - w + 1) + 1)
gaps = [abs(run(a, i) - run(b, i)) for i in range(len(a))]
say(f'final-loss gap (mean of the last 20 steps): {abs(sum(a[-20:]) - sum(b[-20:])) / 20:.4f}   largest gap of the 20-step running means: {max(gaps):.4f} (step {gaps.index(max(gaps)) + 1})   '
    f'largest single-step gap: {max(abs(p - q) for p, q in zip(a, b)):.4f}')
say('step  bf16      mxfp8')
for i in list(range(0, args.steps, max(1, args.steps // 15))) + [args.steps - 1]:
    say(f'{i + 1:4d}  {a[i]:.4f}  {b[i]:.4f}')

# ---- 2. error growth with depth (ViT-L encoder, 24 blocks)
from functools import partial  # noqa: E402
from torch import nn  # noqa: E402
from multimae_amd.multimae_utils import Block, run_blocks  # noqa: E402
torch.manual_seed(0)
L, D, H, N, B = 24, 1024, 16, 197, 2
enc = nn.Sequential(*[Block(D, H, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6)) for _ in range(L)]).to(dev)
x0 = torch.randn(B, N, D, device=dev)
outs = {}
for mode in ('fp32', 'bf16', 'mxfp8'):
    with M.engine.precision(mode), torch.no_grad():
        outs[mode] = [t.float().clone() for t in run_blocks(enc, x0, root=enc, all_layers=True, mx=(mode == 'mxfp8'))]
torch.cuda.synchronize()
say(f'# residual stream after every block of a randomly initialised 24-layer ViT-L encoder (B = {B}, {N} tokens): relative 2-norm error against the fp32 mode')
say('layer   bf16        mxfp8       mxfp8 / bf16')
for l in range(L):
    ref = outs['fp32'][l]
    eb = float((outs['bf16'][l] - ref).norm() / ref.norm())
    em = float((outs['mxfp8'][l] - ref).norm() / ref.norm())
    say(f'{l + 1:5d}   {eb:.3e}   {em:.3e}   {em / max(eb, 1e-12):6.1f}')
os.makedirs(os.path.dirname(args.out), exist_ok=True)
open(args.out, 'w').write('\n'.join(lines) + '\n')
print(json.dumps({'final_gap': abs(sum(a[-20:]) - sum(b[-20:])) / 20, 'max_running_gap': max(gaps)}))
