"""K-sweep of the bf16 GEMM: separates per-tile fixed cost (prologue + epilogue) from the K-loop rate."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
DEV = 'cuda'

def t(M, N, K, tile, out_dtype=torch.bfloat16, iters=30, bias=False):
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16); B = torch.randn(N, K, device=DEV).to(torch.bfloat16)
    C = torch.empty(M, N, device=DEV, dtype=out_dtype)
    bi = torch.randn(N, device=DEV) if bias else None
    f = lambda: ops.gemm(A, B, C, M, N, K, lda=K, ldb=K, ldc=N, tile=tile, bias=bi)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for tile in (5, 3):
    for (M, N) in ((25344, 768), (25344, 2304), (16384, 1024), (32768, 2048)):
        row = []
        for K in (64, 128, 256, 512, 768, 1536, 3072):
            us = t(M, N, K, tile)
            row.append(f'K={K}:{us:6.1f}us')
        print(f'tile={tile} M={M} N={N} tiles={((M+127)//128)*((N+127)//128)}  ' + '  '.join(row), flush=True)
    print('f32 out:', ' '.join(f'K={K}:{t(25344, 768, K, tile, torch.float32):6.1f}us' for K in (64, 768, 3072)), flush=True)
