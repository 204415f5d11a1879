"""The forward / input-gradient GEMMs of one output-adapter transformer block (B = 256 x 196 query tokens, D = 256, Hd = 1024) with their real
epilogues, each alone, on every applicable tile code -- which kernel suits these HBM-bound shapes (contraction of only 256 .. 1024).
    rocprofv3 --kernel-trace -d out -o p --output-format csv -- python tools/decoder_gemms.py
    python tools/decoder_gemms.py --parse out
Per case the HBM roofline of its ALGORITHMIC bytes (operands once, outputs once) at 8 TB/s is printed beside the measured time."""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R, D, H = 50176, 256, 1024
REP = 5
TILES = [9, 10, 1, 2, 6]
_t = [int(a) for a in sys.argv[1:] if a.isdigit()]
if _t:
    TILES = _t
# name, flops, algorithmic bytes
CASES = [('fwd qkv  bias -> bf16', 2.0 * R * D * 3 * D, R * D * 2 + R * 3 * D * 2),
         ('fwd proj bias+resid -> f32', 2.0 * R * D * D, R * D * 2 + R * D * 4 * 2),
         ('fwd fc1  bias+gelu -> 2 x bf16', 2.0 * R * D * H, R * D * 2 + 2 * R * H * 2),
         ('fwd fc2  bias+resid -> f32', 2.0 * R * D * H, R * H * 2 + R * D * 4 * 2),
         ('dx  fc2  x aux + colsum -> bf16', 2.0 * R * D * H, R * D * 2 + 2 * R * H * 2),
         ('dx  fc1  -> bf16', 2.0 * R * D * H, R * H * 2 + R * D * 2),
         ('dx  proj -> bf16', 2.0 * R * D * D, 2 * R * D * 2),
         ('dx  qkv  -> bf16', 2.0 * R * D * 3 * D, R * 3 * D * 2 + R * D * 2)]
if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    f = glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'FillFunctor<int>' in r['Kernel_Name']]
    n = len(CASES) * len(TILES)
    marks = marks[-n:]
    assert len(marks) == n, (len(marks), n)
    marks.append(len(rows))
    i = 0
    for name, fl, by in CASES:
        line = f'{name:34s} roofline {by / 8e6:6.1f} us |'
        for t in TILES:
            seg = rows[marks[i] + 1:marks[i + 1]]
            i += 1
            per = {}
            for r in seg:
                per.setdefault(r['Kernel_Name'].split('(')[0][-40:], []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            us = sum(sorted(v)[len(v) // 2] for v in per.values()) / 1e3
            line += f'  tile {t:2d}: {us:6.1f} us ({by / 8e6 / us * 100 if us else 0:3.0f} %)' if us else f'  tile {t:2d}:   --          '
        print(line)
    sys.exit(0)
import torch
from multimae_amd import ops
from multimae_amd._lib import EPI_GELU_G, EPI_MUL
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
part = torch.empty(ops.dx_colsum_part_shape(R, H), device=dev)
marker = torch.zeros(1, device=dev, dtype=torch.int32)
fns = [lambda t: ops.linear_fwd(x_act, wqkv, bqkv, qkv, tile=t),
       lambda t: ops.linear_fwd(ao, wproj, bproj, x1, resid=x_res, tile=t),
       lambda t: ops.linear_fwd(x_act, wfc1, bfc1, hout, aux=hpre, epi=EPI_GELU_G, tile=t),
       lambda t: ops.linear_fwd(hact, wfc2, bfc2, x1, resid=x_res, tile=t),
       lambda t: ops.linear_dx(d_x, wfc2, dout_h, aux=hpre, epi=EPI_MUL, colsum_part=part, tile=t),
       lambda t: ops.linear_dx(d_h, wfc1, dout_d, tile=t),
       lambda t: ops.linear_dx(d_x, wproj, dout_d, tile=t),
       lambda t: ops.linear_dx(d_qkv, wqkv, dout_d, tile=t)]
for fn in fns:
    for t in TILES:
        try:
            fn(t)
        except Exception as e:
            print('unsupported', t, str(e)[:80])
torch.cuda.synchronize()
for fn in fns:
    for t in TILES:
        marker.fill_(1)
        for _ in range(REP):
            try:
                fn(t)
            except Exception:
                break
        torch.cuda.synchronize()
