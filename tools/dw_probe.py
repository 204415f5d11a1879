"""The four weight gradients of one ViT-B encoder block (rows = 256 x 99 tokens) as the training step launches them: ONE grouped
launch (mmae_gemm_dw_group) -- timed with HIP events over REP launches -- against the rate of a dX product of the same size.
    python tools/dw_probe.py [--split K] [--rep 10]"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument('--split', type=int, default=0)
ap.add_argument('--rep', type=int, default=10)
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--check', action='store_true')
args = ap.parse_args()
R, D, H = 25344, 768, 3072
dev, bf = 'cuda', torch.bfloat16
torch.manual_seed(0)
g = lambda *s: torch.randn(*s, device=dev)
x_ln1, ao, x_ln2, hact = g(R, D).to(bf), g(R, D).to(bf), g(R, D).to(bf), g(R, H).to(bf)
d_qkv, d_x1, d_h, d_x2 = g(R, 3 * D).to(bf), g(R, D).to(bf), g(R, H).to(bf), g(R, D).to(bf)
gw = [torch.zeros(3 * D, D, device=dev), torch.zeros(D, D, device=dev), torch.zeros(H, D, device=dev), torch.zeros(D, H, device=dev)]
gb = [torch.zeros(3 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(H, device=dev), torch.zeros(D, device=dev)]
probs = [(d_qkv, x_ln1, gw[0], gb[0]), (d_x1, ao, gw[1], gb[1]), (d_h, x_ln2, gw[2], gb[2]), (d_x2, hact, gw[3], gb[3])]
flop = sum(2.0 * R * p[0].shape[1] * p[1].shape[1] for p in probs)


def run():
    ops.gemm_dw_group(probs, False, split_k=args.split)


run(); torch.cuda.synchronize()
if args.check:
    worst = 0.0
    for (dy, x, dw, db) in probs:
        ref = dy.float().t() @ x.float()
        e = ((dw - ref).abs().max() / ref.abs().max()).item()
        eb = ((db - dy.float().sum(0)).abs().max() / dy.float().sum(0).abs().max()).item()
        worst = max(worst, e, eb)
        print(f'dW {tuple(dw.shape)}: rel max err {e:.2e}, db {eb:.2e}')
    print('CHECK', 'PASS' if worst < 2e-3 else 'FAIL')
ts = []
for _ in range(args.rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.rep):
        run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / args.rep)
us = sorted(ts)[len(ts) // 2]
print(f'grouped dW (split {args.split or "auto"}): {us:.1f} us per block  {flop / us / 1e6:.1f} TF/s  ({flop / 1e9:.1f} GF)')
