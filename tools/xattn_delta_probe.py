#!/usr/bin/env python
"""CPU probe for VERDICT r3 "find the 6 % on decoder.q.weight": where does the d = 32 cross-attention backward lose bits at cfg5?

Runs the ORACLE (fp32, CPU) on the seeded cfg5 case of tests/test_parity_geometry_gpu.py (ViT-L, 24 layers, B = 2, 196 visible
tokens), captures q, k, v and dO of one output adapter's cross-attention, and replays the backward of the attention core in
three arithmetics:
  engine : what csrc/attention.hip did in round 3 -- bf16 operands, fp32 S / P / dP, delta = rowsum(dO . O) with the STORED bf16 O
  exact-d: the same, delta = sum_j P_j dP_j (what autograd's softmax backward computes)
  autocast: the reference under bf16 autocast (bf16 S, fp32 softmax, bf16 P / dP / dS)
and reports the error of dQ and of dW_q = dQ^T qn against the fp32 result.

    python tools/xattn_delta_probe.py [task]            # task: rgb (default) | depth | norm_rgb
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import multimae_oracle as orc  # noqa: E402
from helpers import build_engine_model, make_inputs  # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).float()


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    task = sys.argv[1] if len(sys.argv) > 1 else 'rgb'
    doms = ['rgb', 'depth', 'semseg']
    torch.manual_seed(0)
    model = build_engine_model(doms, 16, 224, factory='pretrain_multimae_large')
    x = make_inputs(doms, 2, 224)
    torch.manual_seed(1)
    dist, tn, an = orc.draw_mask_randoms(2, [196] * 3, 1.0)
    spt = orc.samples_per_task_from_dirichlet(dist, 196)
    mask_all, ik, ir = orc.masks_from_noise(spt, tn, an, 196)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = orc.standard_config(doms, dim_tokens=1024, depth=24, num_heads=16)
    cap = {}
    orig = orc.cross_attention

    def hooked(xq, ctx, sd_, prefix, heads):
        if prefix.startswith(f'output_adapters.{task}.'):
            B, N, C = xq.shape
            M = ctx.shape[1]
            d = C // heads
            q = (xq @ sd_[prefix + 'q.weight'].t() + sd_[prefix + 'q.bias']).reshape(B, N, heads, d).permute(0, 2, 1, 3)
            kv = (ctx @ sd_[prefix + 'kv.weight'].t() + sd_[prefix + 'kv.bias']).reshape(B, M, 2, heads, d).permute(2, 0, 3, 1, 4)
            k, v = kv[0], kv[1]
            a = ((q @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
            o = a @ v
            o.retain_grad(); q.retain_grad()
            cap.update(q=q, k=k, v=v, o=o, xq=xq, scale=d ** -0.5)
            o2 = o.transpose(1, 2).reshape(B, N, C)
            return o2 @ sd_[prefix + 'proj.weight'].t() + sd_[prefix + 'proj.bias']
        return orig(xq, ctx, sd_, prefix, heads)
    orc.cross_attention = hooked
    for k_, t in sd.items():
        if t.is_floating_point():
            t.requires_grad_(True)
    preds = orc.multimae_forward(x, sd, cfg, ik, ir)
    losses = orc.pretrain_losses(preds, x, mask_all, cfg, {d: 196 for d in doms})
    sum(losses.values()).backward()
    q, k, v, o, xq, scale = (cap[n] for n in ('q', 'k', 'v', 'o', 'xq', 'scale'))
    dO, dQ_ref = o.grad.detach(), q.grad.detach()
    q, k, v, o, xq = q.detach(), k.detach(), v.detach(), o.detach(), xq.detach()
    B, H, N, d = q.shape
    print(f'{task}: q {tuple(q.shape)} k {tuple(k.shape)}; |q| {float(q.norm()):.3e} |k| {float(k.norm()):.3e} |v| {float(v.norm()):.3e} |dO| {float(dO.norm()):.3e}')
    kbar, vbar = k.mean(2, keepdim=True), v.mean(2, keepdim=True)
    print(f'  common component over the keys: |mean k| / |k - mean| = {float(kbar.norm() * k.shape[2] ** 0.5 / (k - kbar).norm()):.2f}, '
          f'|mean v| / |v - mean| = {float(vbar.norm() * v.shape[2] ** 0.5 / (v - vbar).norm()):.2f}')

    def bwd(mode):
        qb, kb, vb, dob = bf(q), bf(k), bf(v), bf(dO)
        if mode == 'autocast':
            S = bf(bf(qb @ kb.transpose(-2, -1)) * scale)
            P = S.softmax(-1)
            dP = bf(dob @ vb.transpose(-2, -1))
            dS = P * (dP - (P * dP).sum(-1, keepdim=True))
            dS = bf(bf(dS) * scale)
            return bf(dS @ kb)
        S = (qb @ kb.transpose(-2, -1)) * scale
        P = S.softmax(-1)
        dP = dob @ vb.transpose(-2, -1)
        if mode == 'engine':
            ob = bf(bf(P) @ vb)                       # the stored output: bf16 P . V, rounded to bf16
            delta = (dob * ob).sum(-1, keepdim=True)
        else:
            delta = (P * dP).sum(-1, keepdim=True)
        dS = bf(P * (dP - delta) * scale)
        return bf(dS @ kb)
    qn = xq                                            # query_norm output [B, N, C]
    dWq_ref = dQ_ref.permute(0, 2, 1, 3).reshape(B * N, H * d).t() @ qn.reshape(B * N, -1)
    for mode in ('engine', 'exact-d', 'autocast'):
        dQ = bwd(mode)
        dWq = dQ.permute(0, 2, 1, 3).reshape(B * N, H * d).t() @ bf(qn).reshape(B * N, -1)
        print(f'  {mode:9s}: dQ rel err {rel(dQ, dQ_ref):.3e}   dW_q rel err {rel(dWq, dWq_ref):.3e}')


if __name__ == '__main__':
    main()
