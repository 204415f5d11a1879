"""Fused attention fwd/bwd on the encoder (99 tok, 12 heads x 64) and decoder (196 q; 196 or 99 kv; 8 heads x 32) geometries.
Run under rocprofv3 --kernel-trace and --parse <dir>."""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    f = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if 'attn_' in r['Kernel_Name']:
            per.setdefault(r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k, v in per.items():
        v = sorted(v)
        print(f'{k:30s} n={len(v):3d} median {v[len(v) // 2] / 1e3:8.1f} us')
    sys.exit(0)
import torch
from multimae_amd import ops
from multimae_amd.ops import AttnView
B = 256
for (H, hd, Nq, Nk) in ((12, 64, 99, 99), (8, 32, 196, 196), (8, 32, 196, 99)):
    D = H * hd
    q = torch.randn(B * Nq, D, device='cuda').to(torch.bfloat16)
    k = torch.randn(B * Nk, D, device='cuda').to(torch.bfloat16)
    v = torch.randn(B * Nk, D, device='cuda').to(torch.bfloat16)
    o = torch.empty_like(q); do = torch.randn_like(q)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    for _ in range(5):
        st = ops.attention_fwd(AttnView(q, 0, D, Nq), AttnView(k, 0, D, Nk), AttnView(v, 0, D, Nk), AttnView(o, 0, D, Nq), B, H, hd, hd ** -0.5)
        ops.attention_bwd(AttnView(q, 0, D, Nq), AttnView(k, 0, D, Nk), AttnView(v, 0, D, Nk), st, AttnView(o, 0, D, Nq), AttnView(do, 0, D, Nq),
                          AttnView(dq, 0, D, Nq), AttnView(dk, 0, D, Nk), AttnView(dv, 0, D, Nk), B, H, hd, hd ** -0.5)
    torch.cuda.synchronize()
