"""The 12 GEMMs of one ViT-B encoder block (B = 256 x 99 tokens) with their real epilogues, each in isolation.
Host launch cost hides short kernels from event timing, so run under rocprofv3 and read the trace:
    rocprofv3 --kernel-trace -d out -o p --output-format csv -- python tools/encoder_gemms.py
    python tools/encoder_gemms.py --parse out"""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R, D, H = 25344, 768, 3072
REP = int(os.environ.get('MMAE_TABLE_REP', '5'))     # launches per product; the table prints the median
# name, flops
CASES = [('fwd qkv  bias', 2.0 * R * D * 3 * D), ('fwd proj bias+resid f32', 2.0 * R * D * D), ('fwd fc1  bias+gelu, aux=gelu\'', 2.0 * R * D * H),
         ('fwd fc2  bias+resid f32', 2.0 * R * D * H), ('dx  fc2  x aux + colsum', 2.0 * R * D * H), ('dx  fc1', 2.0 * R * D * H),
         ('dx  proj', 2.0 * R * D * D), ('dx  qkv', 2.0 * R * D * 3 * D), ('dw  fc2', 2.0 * R * D * H), ('dw  fc1', 2.0 * R * D * H),
         ('dw  proj', 2.0 * R * D * D), ('dw  qkv', 2.0 * R * D * 3 * D)]
if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    f = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'FillFunctor<int>' in r['Kernel_Name']]      # one marker fill before every case
    marks = marks[-len(CASES):]            # (torch.zeros of the marker itself is a fill too)
    assert len(marks) == len(CASES), (len(marks), len(CASES))
    marks.append(len(rows))
    tot = 0.0
    for c, (name, fl) in enumerate(CASES):
        seg = rows[marks[c] + 1:marks[c + 1]]
        per = {}
        for r in seg:
            k = r['Kernel_Name'].split('(')[0][-40:]
            per.setdefault(k, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        us = sum(sorted(v)[len(v) // 2] for v in per.values()) / 1e3
        tot += us
        detail = '  '.join(f'{k.split("::")[-1]}={sorted(v)[len(v) // 2] / 1e3:.1f}' for k, v in per.items())
        print(f'{name:26s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s   [{detail}]')
    print(f'sum {tot:.1f} us per block -> {12 * tot / 1e3:.2f} ms per 12-block step, {sum(f for _, f in CASES) / tot / 1e6:.1f} TF/s')
    sys.exit(0)
import torch
from multimae_amd import ops
from multimae_amd._lib import EPI_GELU_G, EPI_MUL      # the pair the composite calls use with bf16 activations
dev = 'cuda'
bf = torch.bfloat16
g = lambda *s: torch.randn(*s, device=dev)
x_act, ao, hact = g(R, D).to(bf), g(R, D).to(bf), g(R, H).to(bf)
x_res = g(R, D)
wqkv, wproj, wfc1, wfc2 = (g(3 * D, D) * 0.02).to(bf), (g(D, D) * 0.02).to(bf), (g(H, D) * 0.02).to(bf), (g(D, H) * 0.02).to(bf)
bqkv, bproj, bfc1, bfc2 = g(3 * D), g(D), g(H), g(D)
qkv, hpre, hout = torch.empty(R, 3 * D, device=dev, dtype=bf), torch.empty(R, H, device=dev, dtype=bf), torch.empty(R, H, device=dev, dtype=bf)
x1 = torch.empty(R, D, device=dev)
d_h, d_qkv = g(R, H).to(bf), g(R, 3 * D).to(bf)
d_x = g(R, D).to(bf)
dout_d, dout_h = torch.empty(R, D, device=dev, dtype=bf), torch.empty(R, H, device=dev, dtype=bf)
cs = torch.empty(H, device=dev)
gw = {k: torch.zeros_like(v, dtype=torch.float32) for k, v in dict(qkv=wqkv, proj=wproj, fc1=wfc1, fc2=wfc2).items()}
marker = torch.zeros(1, device=dev, dtype=torch.int32)
fns = [lambda: ops.linear_fwd(x_act, wqkv, bqkv, qkv),
       lambda: ops.linear_fwd(ao, wproj, bproj, x1, resid=x_res),
       lambda: ops.linear_fwd(x_act, wfc1, bfc1, hout, aux=hpre, epi=EPI_GELU_G),
       lambda: ops.linear_fwd(hact, wfc2, bfc2, x1, resid=x_res),
       lambda: ops.linear_dx(d_x, wfc2, dout_h, aux=hpre, epi=EPI_MUL, colsum_out=cs),
       lambda: ops.linear_dx(d_h, wfc1, dout_d),
       lambda: ops.linear_dx(d_x, wproj, dout_d),
       lambda: ops.linear_dx(d_qkv, wqkv, dout_d),
       lambda: ops.linear_dw(d_x, hact, gw['fc2'], True),
       lambda: ops.linear_dw(d_h, x_act, gw['fc1'], True),
       lambda: ops.linear_dw(d_x, ao, gw['proj'], True),
       lambda: ops.linear_dw(d_qkv, x_act, gw['qkv'], True)]
for fn in fns:
    fn()
torch.cuda.synchronize()
for fn in fns:
    marker.fill_(1)
    for _ in range(REP):
        fn()
    torch.cuda.synchronize()
