"""Micro-benchmark of the MFMA GEMM kernels on the shapes of the ViT-B pre-training step
(M = 25344 encoder rows / 50176 decoder rows).  Random (not zero) operands, HIP-event timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops

DEV = 'cuda'


def bench(name, M, N, K, a_trans, b_trans, tile, dtype=torch.bfloat16, iters=20, split=0):
    A = torch.randn((K, M) if a_trans else (M, K), device=DEV).to(dtype)
    B = torch.randn((K, N) if b_trans else (N, K), device=DEV).to(dtype)
    C = torch.empty(M, N, device=DEV, dtype=torch.float32 if (a_trans and b_trans) else dtype)
    f = lambda: ops.gemm(A, B, C, M, N, K, lda=A.shape[1], ldb=B.shape[1], ldc=N, a_trans=a_trans, b_trans=b_trans, tile=tile, split_k=split)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    print(f'{name:28s} M={M:6d} N={N:5d} K={K:6d} at={int(a_trans)} bt={int(b_trans)} tile={tile} {str(dtype)[6:]:9s} {ms:8.3f} ms {tf:8.1f} TF/s', flush=True)
    return tf


if __name__ == '__main__':
    R, Rd = 25344, 50176
    for tile in (3, 5, 6):
        bench('enc qkv fwd', R, 2304, 768, False, False, tile)
        bench('enc proj fwd', R, 768, 768, False, False, tile)
        bench('enc fc1 fwd', R, 3072, 768, False, False, tile)
        bench('enc fc2 fwd', R, 768, 3072, False, False, tile)
        bench('enc fc1 dX (b_trans)', R, 768, 3072, False, True, tile)
        bench('enc fc2 dX (b_trans)', R, 3072, 768, False, True, tile)
        bench('enc fc1 dW (TN)', 3072, 768, R, True, True, tile)
        bench('enc qkv dW (TN)', 2304, 768, R, True, True, tile)
        bench('enc proj dW (TN)', 768, 768, R, True, True, tile)
        bench('dec fc1 fwd', Rd, 1024, 256, False, False, tile)
        bench('dec qkv fwd', Rd, 768, 256, False, False, tile)
        bench('dec fc1 dW (TN)', 1024, 256, Rd, True, True, tile)
    for t in (3, 5, 6):
        bench('square 4096', 4096, 4096, 4096, False, False, t)
        bench('square 8192', 8192, 8192, 8192, False, False, t)
    bench('f32 enc qkv fwd', R, 2304, 768, False, False, 0, torch.float32, 5)
    bench('f32 dec fc1 fwd', Rd, 1024, 256, False, False, 0, torch.float32, 5)
    bench('f32 dec fc1 dW', 1024, 256, Rd, True, True, 0, torch.float32, 5)
    bench('f32 dec out dX', Rd, 256, 2128, False, True, 0, torch.float32, 5)
