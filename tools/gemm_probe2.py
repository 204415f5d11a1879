"""Layout / split-K probe of the ping-pong GEMM (run under rocprofv3 --kernel-trace, then --parse <dir>)."""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (M, N, K, a_trans, b_trans, tile, split_k, c_f32)
CONFIGS = [(4096, 4096, 4096, at, bt, 9, 1, False) for (at, bt) in ((0, 0), (0, 1), (1, 0), (1, 1))] + \
          [(5120, 4096, 4096, 0, 0, 10, 1, False), (5120, 4096, 4096, 0, 1, 10, 1, False)] + \
          [(3072, 768, 25344, 1, 1, 9, s, True) for s in (1, 4, 7, 14)] + \
          [(768, 768, 25344, 1, 1, 9, s, True) for s in (14, 28)] + \
          [(3072, 768, 25344, 1, 1, 3, 0, True), (768, 768, 25344, 1, 1, 3, 0, True)]
REP = 5
if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    f = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'FillFunctor<int>' in r['Kernel_Name']][-len(CONFIGS):]
    marks.append(len(rows))
    for c, cfg in enumerate(CONFIGS):
        per = {}
        for r in rows[marks[c] + 1:marks[c + 1]]:
            per.setdefault(r['Kernel_Name'].split('(')[0][-30:], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        us = sum(sorted(v)[len(v) // 2] for v in per.values()) / 1e3
        M, N, K = cfg[:3]
        print(f'{cfg}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s  ' + ' '.join(f'{sorted(v)[len(v) // 2] / 1e3:.1f}' for v in per.values()))
    sys.exit(0)
import torch
from multimae_amd import ops
marker = torch.zeros(1, device='cuda', dtype=torch.int32)
for (M, N, K, at, bt, tile, sk, cf32) in CONFIGS:
    A = torch.randn((K, M) if at else (M, K), device='cuda').to(torch.bfloat16)
    B = torch.randn((K, N) if bt else (N, K), device='cuda').to(torch.bfloat16)
    C = torch.empty(M, N, device='cuda', dtype=torch.float32 if cf32 else torch.bfloat16)
    marker.fill_(1)
    for _ in range(REP):
        ops.gemm(A, B, C, M, N, K, lda=A.shape[1], ldb=B.shape[1], ldc=N, a_trans=bool(at), b_trans=bool(bt), tile=tile, split_k=sk)
    torch.cuda.synchronize()
