"""Round 6: the output adapters' cross-attention (B = 256 images, 196 queries, 99 context rows, D = 8 x 32 = 256, bf16) as the
three launches the step uses -- q-projection GEMM, kv-projection GEMM, attention core -- against ONE launch with both projections inside the attention
kernel (mmae_xattn_fwd_fused).  HIP-event timed, the two forms interleaved, random operands; results are bit-identical (tests/test_kernels_gpu.py).
    python tools/xattn_fused_ab.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
from multimae_amd.ops import AttnView

B, Nq, Nk, D, H, hd = 256, 196, 99, 256, 8, 32
dev = 'cuda'
torch.manual_seed(0)
qn = torch.randn(B * Nq, D, device=dev).to(torch.bfloat16)
cn = torch.randn(B * Nk, D, device=dev).to(torch.bfloat16)
wq = (torch.randn(D, D, device=dev) * 0.06).to(torch.bfloat16)
wkv = (torch.randn(2 * D, D, device=dev) * 0.06).to(torch.bfloat16)
bq, bkv = torch.randn(D, device=dev), torch.randn(2 * D, device=dev)
q3, kv3, o3 = torch.empty(B * Nq, D, device=dev, dtype=torch.bfloat16), torch.empty(B * Nk, 2 * D, device=dev, dtype=torch.bfloat16), torch.empty(B * Nq, D, device=dev, dtype=torch.bfloat16)


def three():
    ops.linear_fwd(qn, wq, bq, q3)
    ops.linear_fwd(cn, wkv, bkv, kv3)
    ops.attention_fwd(AttnView(q3, 0, D, Nq), AttnView(kv3, 0, 2 * D, Nk), AttnView(kv3, D, 2 * D, Nk), AttnView(o3, 0, D, Nq), B, H, hd, hd ** -0.5)


def fused():
    return ops.xattn_fwd_fused(qn, cn, wq, bq, wkv, bkv, B, Nq, Nk)


def t(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


q, kv, o, lse = fused()
three()
out = {'geometry': dict(B=B, Nq=Nq, Nk=Nk, D=D, heads=H), 'bit_identical': bool(torch.equal(q, q3) and torch.equal(kv, kv3) and torch.equal(o, o3)), 'us': {}}
for rep in range(3):
    out['us'][f'three launches #{rep}'] = round(t(three), 1)
    out['us'][f'fused #{rep}'] = round(t(fused), 1)
flop = 2.0 * B * (Nq * D * D + Nk * D * 2 * D + 2 * H * Nq * Nk * hd)
out['gflop'] = round(flop / 1e9, 2)
print(json.dumps(out, indent=1))
