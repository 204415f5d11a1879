"""MX-fp8 vs bf16 GEMM on the transformer shapes of cfg5 (ViT-L, B=128 x 197 rows) and cfg3 (ViT-B, 256 x 99): HIP-event timing."""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multimae_amd import ops

DEV = 'cuda'


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    for name, M, D in (('ViT-L B=128', 128 * 197, 1024), ('ViT-B B=256', 256 * 99, 768)):
        print(f'== {name}: M = {M}')
        x = torch.randn(M, 4 * D, device=DEV).bfloat16()
        for tag, N, K, kind in (('qkv fwd  bias->bf16', 3 * D, D, 'bias'), ('proj fwd bias+resid f32', D, D, 'resid'),
                                ('fc1 fwd  bias+gelu', 4 * D, D, 'gelu'), ('fc2 fwd  bias+resid f32', D, 4 * D, 'resid'),
                                ('dx qkv   bf16', D, 3 * D, 'plain'), ('dx fc1   bf16', D, 4 * D, 'plain'),
                                ('dx fc2   dgelu+cs', 4 * D, D, 'dgelu')):
            a = x[:, :K].contiguous()
            w = (torch.randn(N, K, device=DEV) * 0.03).bfloat16()
            bias = torch.randn(N, device=DEV)
            resid = torch.randn(M, N, device=DEV) if kind == 'resid' else None
            out = torch.empty((M, N), device=DEV, dtype=torch.float32 if kind == 'resid' else torch.bfloat16)
            aux = torch.randn(M, N, device=DEV).bfloat16() if kind in ('gelu', 'dgelu') else None
            part = torch.zeros(((M + 31) // 32, N), device=DEV) if kind == 'dgelu' else None
            qa, qw = ops.mx_quant(a), ops.mx_quant(w)
            kw = dict(bias=bias if kind in ('bias', 'resid', 'gelu') else None, resid=resid, aux=aux,
                      epi=ops.EPI_GELU if kind == 'gelu' else (ops.EPI_DGELU if kind == 'dgelu' else ops.EPI_NONE), colsum_part=part)
            t_mx = timeit(lambda: ops.gemm_mx(qa, qw, out, **kw))
            t_bf = timeit(lambda: ops.gemm(a, w, out, M, N, K, lda=K, ldb=K, ldc=N, ldr=N, ldaux=N, **kw))
            t_q = timeit(lambda: ops.mx_quant(a, qa))
            fl = 2.0 * M * N * K
            print(f'{tag:26s} N={N:5d} K={K:5d}  mxfp8 {t_mx:7.1f} us {fl / t_mx / 1e6:7.1f} TF/s   bf16 {t_bf:7.1f} us {fl / t_bf / 1e6:7.1f} TF/s'
                  f'   quant(A) {t_q:6.1f} us {M * K * 3 / t_q / 1e6:5.2f} TB/s')
        w = (torch.randn(4 * D, D, device=DEV) * 0.03).bfloat16()
        qt = ops.mx_quant_t(w)
        t = timeit(lambda: ops.mx_quant_t(w, qt))
        print(f'quant_t {4 * D} x {D}: {t:.1f} us')


if __name__ == '__main__':
    main()
