"""HIP-event timing of the fused attention forward / backward at the step's three geometries (encoder 99 x 99 x 12 heads x 64; decoder self 196 x 196 and
cross 196 x 99 x 8 heads x 32), B = 256, bf16; 20 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
from multimae_amd.ops import AttnView
B = 256


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (H, hd, Nq, Nk) in ((12, 64, 99, 99), (8, 32, 196, 196), (8, 32, 196, 99)):
    D = H * hd
    q = torch.randn(B * Nq, D, device='cuda').to(torch.bfloat16)
    k = torch.randn(B * Nk, D, device='cuda').to(torch.bfloat16)
    v = torch.randn(B * Nk, D, device='cuda').to(torch.bfloat16)
    o = torch.empty_like(q); do = torch.randn_like(q)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    V = lambda t_, n: AttnView(t_, 0, D, n)
    st = [None]
    fwd = lambda: st.__setitem__(0, ops.attention_fwd(V(q, Nq), V(k, Nk), V(v, Nk), V(o, Nq), B, H, hd, hd ** -0.5))
    fwd()
    bwd = lambda: ops.attention_bwd(V(q, Nq), V(k, Nk), V(v, Nk), st[0], V(o, Nq), V(do, Nq), V(dq, Nq), V(dk, Nk), V(dv, Nk), B, H, hd, hd ** -0.5)
    print(f'H={H:2d} hd={hd} Nq={Nq:3d} Nk={Nk:3d}   fwd {t(fwd):7.1f} us   bwd {t(bwd):7.1f} us')
