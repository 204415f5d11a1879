"""Fused patch embedding (csrc/embed.hip, one kernel) against the three-pass path (patch_rows -> per-task GEMM -> tokens_assemble) at the
bench geometry (cfg3: RGB + depth + semseg, 98 kept tokens, ViT-B width), B images.   python tools/embed_probe.py [B] [D] [n_sel]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
n_sel = int(sys.argv[3]) if len(sys.argv) > 3 else 98
dev = 'cuda'
torch.manual_seed(0)
tasks = [(0, 3, 224, 224, 16, 0), (0, 1, 224, 224, 16, 0), (1, 64, 56, 56, 4, 133)]
srcs, ws, bs, poss, offs, k_off = [], [], [], [], [0], 0
for kind, C, H, W, P, n_cls in tasks:
    n_p = (H // P) * (W // P)
    data = torch.randn(B, C, H, W, device=dev) if kind == 0 else torch.randint(0, n_cls, (B, H, W), device=dev)
    emb = None if kind == 0 else torch.randn(n_cls, C, device=dev)
    K = C * P * P
    srcs.append(dict(data=data, emb=emb, kind=kind, C=C, H=H, W=W, ph=P, pw=P, k_off=k_off))
    ws.append((torch.randn(D, K, device=dev) * K ** -0.5).to(torch.bfloat16))
    bs.append(torch.randn(D, device=dev)); poss.append(torch.randn(n_p, D, device=dev))
    offs.append(offs[-1] + n_p); k_off += K
Ktot = k_off
sel = torch.stack([torch.randperm(offs[-1], device=dev)[:n_sel] for _ in range(B)])
glob = torch.randn(1, D, device=dev)


def fused(rows=True):
    return ops.patch_embed_fwd(srcs, ws, bs, poss, offs, sel, glob, B, n_sel, 1, D, Ktot, want_rows=rows)


def three():
    rows = ops.patch_rows(srcs, offs, sel, B, n_sel, Ktot, torch.bfloat16)
    proj = torch.empty((B * n_sel, D), device=dev, dtype=torch.float32)
    for i, (s, w) in enumerate(zip(srcs, ws)):
        ops.gemm(rows, w, proj, B * n_sel, D, w.shape[1], lda=Ktot, ldb=w.shape[1], ldc=D, a_off=s['k_off'], accumulate=(i > 0))
    return ops.tokens_assemble(proj, bs, poss, offs, sel, glob, B, n_sel, 1, D), rows


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


flops = 2.0 * B * n_sel * D * Ktot / 3      # useful: every row meets one task's weight (mean K = Ktot / 3 when tasks are hit evenly)
print(f'B={B} D={D} n_sel={n_sel} Ktot={Ktot}')
for name, fn in [('fused, with side rows', lambda: fused(True)), ('fused, no side rows', lambda: fused(False)), ('three-pass', three)]:
    us = timeit(fn)
    print(f'{name:24s} {us:8.1f} us')
