"""How fast do the vendor libraries (rocBLAS / hipBLASLt through torch.mm) run the encoder's GEMM shapes?  Headroom probe
for the hand-written ping-pong kernel (bench.py never calls these)."""
import torch, time
R, D, H = 25344, 768, 3072
dev = 'cuda'
bf = torch.bfloat16
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, (M, N, K) in dict(qkv=(R, 3 * D, D), proj=(R, D, D), fc1=(R, H, D), fc2=(R, D, H)).items():
    x = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.02).to(bf); dy = torch.randn(M, N, device=dev).to(bf)
    fl = 2.0 * M * N * K
    us = t(lambda: torch.mm(x, w.t()))
    print(f'{name:5s} fwd  y=x.wT   {us:8.1f} us {fl / us / 1e6:7.1f} TF/s')
    us = t(lambda: torch.mm(dy, w))
    print(f'{name:5s} dX   dy.w     {us:8.1f} us {fl / us / 1e6:7.1f} TF/s')
    us = t(lambda: torch.mm(dy.t(), x))
    print(f'{name:5s} dW   dyT.x    {us:8.1f} us {fl / us / 1e6:7.1f} TF/s')
