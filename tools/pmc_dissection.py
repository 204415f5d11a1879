"""is the ~1.45 PF/s ceiling of the GEMM K loop a POWER limit (MFMA pipe busy ~ 1.0 at a reduced clock) or an
ISSUE limit (busy << 1 at full clock)?  One PMC pass over the production kernel and its dissection builds (experiments library:
make -C multimae_amd/csrc exp; MMAE_LIB=multimae_amd/libmmae_hip_exp.so), each on random AND on zero-filled operands.

  launch side :  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d DIR -o p --output-format csv -- \
                     python tools/pmc_dissection.py --plan DIR/plan.json
  parse side  :  python tools/pmc_dissection.py --parse DIR

Tile codes: 0 = what the planner picks (production ping-pong), 14 = production schedule on 64-wide K tiles, 24 = no DMA in the K loop,
44 = no DMA and no fragment reads, 54 = 44 without workgroup barriers (pure MFMA issue + the real epilogue).  Dissection builds compute garbage.
SQ_VALU_MFMA_BUSY_CYCLES sums over the 1 024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (tools/pmc_mfma.py): busy fraction =
(busy / 1024) / (GUI / 8); effective clock = (GUI / 8) / duration."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument('--plan', default='')
ap.add_argument('--parse', default='')
ap.add_argument('--tiles', default='0,14,24,44,54')
ap.add_argument('--rep', type=int, default=4)
args = ap.parse_args()

if args.parse:
    plan = json.load(open(os.path.join(args.parse, 'plan.json')))
    f = glob.glob(os.path.join(args.parse, '**', '*counter_collection.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    disp = collections.OrderedDict()
    for r in rows:
        if 'gemm' not in r['Kernel_Name']:
            continue
        d = disp.setdefault(int(r['Dispatch_Id']), dict(name=r['Kernel_Name'], c={}))
        d['c'][r['Counter_Name']] = d['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        if r.get('Start_Timestamp') and r.get('End_Timestamp'):
            d['ns'] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    if not all('ns' in d for d in disp.values()):                    # durations from the kernel trace of the same run
        kt = glob.glob(os.path.join(args.parse, '**', '*kernel_trace.csv'), recursive=True)
        if kt:
            for r in csv.DictReader(open(kt[0])):
                d = disp.get(int(r['Dispatch_Id']))
                if d is not None:
                    d['ns'] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    seq = [disp[k] for k in sorted(disp)]
    print(f'{"product":22s} {"tile":>4s} {"data":>6s} {"us":>8s} {"TF/s":>8s} {"MFMA busy":>10s} {"clock GHz":>10s}')
    i = 0
    for e in plan:
        mine = seq[i:i + e['n']]
        i += e['n']
        mine = mine[1:] if len(mine) > 1 else mine                   # first launch of a group: cold
        if not mine or any('ns' not in d for d in mine):
            print(f'{e["product"]:22s} {e["tile"]:4d} {e["data"]:>6s}   (no data)')
            continue
        us = sum(d['ns'] for d in mine) / len(mine) / 1e3
        busy = sum(d['c'].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for d in mine) / len(mine) / 1024.0
        gui = sum(d['c'].get('GRBM_GUI_ACTIVE', 0.0) for d in mine) / len(mine) / 8.0
        print(f'{e["product"]:22s} {e["tile"]:4d} {e["data"]:>6s} {us:8.1f} {e["flop"] / us / 1e6:8.1f} {busy / gui if gui else 0:10.3f} {gui / us / 1e3 if us else 0:10.3f}')
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from multimae_amd import ops  # noqa: E402

R, D, H = 25344, 768, 3072
dev, bf = 'cuda', torch.bfloat16
torch.manual_seed(0)
TILES = [int(t) for t in args.tiles.split(',')]
plan = []


def operands(zero):
    g = (lambda *s: torch.zeros(*s, device=dev)) if zero else (lambda *s: torch.randn(*s, device=dev))
    return dict(x=g(R, D).to(bf), h=g(R, H).to(bf), dq=g(R, 3 * D).to(bf), wqkv=(g(3 * D, D) * 0.02).to(bf), wfc1=(g(H, D) * 0.02).to(bf),
                wfc2=(g(D, H) * 0.02).to(bf), bq=g(3 * D), b2=g(D), res=g(R, D))


out_qkv = torch.empty(R, 3 * D, device=dev, dtype=bf)
out_x1 = torch.empty(R, D, device=dev)
out_dd = torch.empty(R, D, device=dev, dtype=bf)
for zero in (False, True):
    o = operands(zero)
    cases = [('fwd qkv (K=768)', 2.0 * R * D * 3 * D, lambda t: ops.linear_fwd(o['x'], o['wqkv'], o['bq'], out_qkv, tile=t)),
             ('fwd fc2 (K=3072) f32', 2.0 * R * D * H, lambda t: ops.linear_fwd(o['h'], o['wfc2'], o['b2'], out_x1, resid=o['res'], tile=t)),
             ('dx  fc1 (K=3072)', 2.0 * R * D * H, lambda t: ops.linear_dx(o['h'], o['wfc1'], out_dd, tile=t)),
             ('dx  qkv (K=2304)', 2.0 * R * D * 3 * D, lambda t: ops.linear_dx(o['dq'], o['wqkv'], out_dd, tile=t))]
    for name, fl, fn in cases:
        for t in TILES:
            for _ in range(args.rep):
                fn(t)
            torch.cuda.synchronize()
            plan.append(dict(product=name, tile=t, data='zeros' if zero else 'random', n=args.rep, flop=fl))
if args.plan:
    json.dump(plan, open(args.plan, 'w'))
print('launched', sum(e['n'] for e in plan), 'GEMMs')
