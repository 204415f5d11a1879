"""HBM ceilings of the box by direction -- write-only, read-only, copy -- with plain ATen kernels (fill_, sum, copy_) on 1 GiB
buffers (far beyond the 256 MiB Infinity Cache), HIP-event timed.  The store-heavy GEMM epilogues are priced against the
WRITE ceiling, not the 8 TB/s specification figure.      python tools/hbm_probe.py"""
import torch
dev = 'cuda'
n = 1 << 28
x = torch.empty(n, device=dev, dtype=torch.float32)
y = torch.empty(n, device=dev, dtype=torch.float32)
h = torch.empty(n, device=dev, dtype=torch.bfloat16)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


for name, fn, byts in [('write only  (fill_ f32, 1 GiB)', lambda: x.fill_(1.0), 4 * n),
                       ('write only  (fill_ bf16, 0.5 GiB)', lambda: h.fill_(1.0), 2 * n),
                       ('read only   (sum f32, 1 GiB)', lambda: x.sum(), 4 * n),
                       ('copy        (copy_ f32, 1 GiB -> 1 GiB)', lambda: y.copy_(x), 8 * n),
                       ('cast        (f32 -> bf16, 1 GiB -> 0.5 GiB)', lambda: h.copy_(x), 6 * n),
                       ('axpy        (y += x, 2 reads + 1 write)', lambda: y.add_(x), 12 * n)]:
    s = t(fn)
    print(f'{name:48s} {s * 1e6:8.1f} us   {byts / s / 1e12:5.2f} TB/s')
