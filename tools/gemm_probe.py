"""K-sweep of the bf16 GEMM kernels.  Host launch cost (~30 us through Python/ctypes) hides short
kernels from event timing, so run under rocprofv3 and read true kernel durations from the trace:
    rocprofv3 --kernel-trace -d out -o p --output-format csv -- python tools/gemm_probe.py
    python tools/gemm_probe.py --parse out"""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CONFIGS = [(tile, M, N, K) for tile in (3, 9, 10) for (M, N) in ((25344, 768), (25344, 2304), (25344, 3072), (50176, 1024)) for K in (256, 768, 3072)]
REP = 6
if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    f = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'gemm_bf16' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    assert len(rows) == len(CONFIGS) * REP, (len(rows), len(CONFIGS) * REP)
    for i, (tile, M, N, K) in enumerate(CONFIGS):
        d = sorted(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[i * REP:(i + 1) * REP])
        us = d[len(d) // 2] / 1e3
        print(f'tile={tile} M={M} N={N} K={K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s')
    sys.exit(0)
import torch
from multimae_amd import ops
for (tile, M, N, K) in CONFIGS:
    A = torch.randn(M, K, device='cuda').to(torch.bfloat16); B = torch.randn(N, K, device='cuda').to(torch.bfloat16)
    C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(REP):
        ops.gemm(A, B, C, M, N, K, lda=K, ldb=K, ldc=N, tile=tile)
    torch.cuda.synchronize()
