"""One GEMM shape x a few kernel variants, for rocprofv3 --pmc runs (stall breakdown of the main loop)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
M, N, K = 25344, 2304, 3072
A = torch.randn(M, K, device='cuda').to(torch.bfloat16); B = torch.randn(N, K, device='cuda').to(torch.bfloat16)
C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
for tile in (3, 9, 10):
    for _ in range(3):
        ops.gemm(A, B, C, M, N, K, lda=K, ldb=K, ldc=N, tile=tile)
torch.cuda.synchronize()
