"""A/B of the GEMM kernels on the 8 forward / dX products of a ViT-B encoder block (B = 256 x 99 tokens) with their real
epilogues: tile 0 = what the planner picks (ping-pong 256/320 x 256), 9 = ping-pong 256 x 256, 11 = duo (2 x 4 waves per CU,
128 x 256), 12 = the duo schedule on 8 waves / 256 x 256.  First checks every variant against tile 9 and an fp32 reference,
then times them in interleaved rounds (HIP events around REP back-to-back launches).
    python tools/duo_probe.py [--tiles 0,9,11,12] [--rep 20] [--rounds 5]"""
import sys, os, argparse, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops
from multimae_amd._lib import EPI_DGELU, EPI_GELU

ap = argparse.ArgumentParser()
ap.add_argument('--tiles', default='0,9,11,12')
ap.add_argument('--rep', type=int, default=20)
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('--rows', type=int, default=25344)
ap.add_argument('--check-only', action='store_true')
ap.add_argument('--timing-tiles', default='', help='tile codes timed without a correctness check (dissection builds)')
args = ap.parse_args()
TILES = [int(t) for t in args.tiles.split(',')]
TT = [int(t) for t in args.timing_tiles.split(',') if t]
R, D, H = args.rows, 768, 3072
dev, bf = 'cuda', torch.bfloat16
torch.manual_seed(0)
g = lambda *s: torch.randn(*s, device=dev)
x_act, ao, hact = g(R, D).to(bf), g(R, D).to(bf), g(R, H).to(bf)
x_res = g(R, D)
wqkv, wproj, wfc1, wfc2 = (g(3 * D, D) * 0.02).to(bf), (g(D, D) * 0.02).to(bf), (g(H, D) * 0.02).to(bf), (g(D, H) * 0.02).to(bf)
bqkv, bproj, bfc1, bfc2 = g(3 * D), g(D), g(H), g(D)
hpre = g(R, H).to(bf)
d_h, d_qkv, d_x = g(R, H).to(bf), g(R, 3 * D).to(bf), g(R, D).to(bf)


def outs():
    return dict(qkv=torch.empty(R, 3 * D, device=dev, dtype=bf), hpre=torch.empty(R, H, device=dev, dtype=bf), hout=torch.empty(R, H, device=dev, dtype=bf),
                x1=torch.empty(R, D, device=dev), dd=torch.empty(R, D, device=dev, dtype=bf), dh=torch.empty(R, H, device=dev, dtype=bf),
                cs=torch.empty(ops.dx_colsum_part_shape(R, H), device=dev))


CASES = [
    ('fwd qkv  bias', 2.0 * R * D * 3 * D, lambda o, t: ops.linear_fwd(x_act, wqkv, bqkv, o['qkv'], tile=t), ['qkv']),
    ('fwd proj bias+resid f32', 2.0 * R * D * D, lambda o, t: ops.linear_fwd(ao, wproj, bproj, o['x1'], resid=x_res, tile=t), ['x1']),
    ('fwd fc1  bias+gelu+aux', 2.0 * R * D * H, lambda o, t: ops.linear_fwd(x_act, wfc1, bfc1, o['hout'], aux=o['hpre'], epi=EPI_GELU, tile=t), ['hout', 'hpre']),
    ('fwd fc2  bias+resid f32', 2.0 * R * D * H, lambda o, t: ops.linear_fwd(hact, wfc2, bfc2, o['x1'], resid=x_res, tile=t), ['x1']),
    ('dx  fc2  dgelu+colsum', 2.0 * R * D * H, lambda o, t: ops.linear_dx(d_x, wfc2, o['dh'], aux=hpre, epi=EPI_DGELU, colsum_part=o['cs'], tile=t), ['dh', 'cs']),
    ('dx  fc1', 2.0 * R * D * H, lambda o, t: ops.linear_dx(d_h, wfc1, o['dd'], tile=t), ['dd']),
    ('dx  proj', 2.0 * R * D * D, lambda o, t: ops.linear_dx(d_x, wproj, o['dd'], tile=t), ['dd']),
    ('dx  qkv', 2.0 * R * D * 3 * D, lambda o, t: ops.linear_dx(d_qkv, wqkv, o['dd'], tile=t), ['dd']),
]

# ---- correctness: every tile code against tile 9 (bit-level: same MFMA order along K -> expect exact or 1-ulp bf16) ----
ok = True
ref = outs()
for name, fl, fn, keys in CASES:
    for v in ref.values():
        v.fill_(float('nan'))
    fn(ref, 9)
    torch.cuda.synchronize()
    for t in TILES:
        if t in (9,):
            continue
        o = outs()
        for v in o.values():
            v.fill_(float('nan'))
        fn(o, t)
        torch.cuda.synchronize()
        for k in keys:
            a, b = o[k].float(), ref[k].float()
            if not torch.isfinite(a).all():
                print(f'FAIL {name} tile {t} {k}: non-finite output ({(~torch.isfinite(a)).sum().item()} elements)'); ok = False; continue
            err = (a - b).abs().max().item()
            scale = b.abs().max().item()
            tol = (2e-2 if o[k].dtype == bf else 1e-4) * max(scale, 1.0) if k != 'cs' else 2e-3 * max(scale, 1.0)
            flag = 'ok' if err <= tol else 'FAIL'
            if err > tol:
                ok = False
            print(f'{flag} {name:26s} tile {t:2d} {k:5s} max|d| {err:.3e} (ref max {scale:.3e})')
# one fp32 reference for the pp kernel itself (fwd qkv)
o = outs(); CASES[0][2](o, 11 if 11 in TILES else 9); torch.cuda.synchronize()
r32 = x_act[:512].float() @ wqkv.float().t() + bqkv
e = (o['qkv'][:512].float() - r32).abs().max().item()
print(f'fwd qkv vs fp32 torch (512 rows): max|d| {e:.3e}')
ok = ok and e < 5e-2
print('CHECK', 'PASS' if ok else 'FAIL')
if args.check_only or not ok:
    sys.exit(0 if ok else 1)

# ---- timing: interleaved rounds ----
TILES = TILES + TT
o = outs()
res = {}
for name, fl, fn, keys in CASES:
    times = {t: [] for t in TILES}
    for t in TILES:
        fn(o, t)
    torch.cuda.synchronize()
    for rd in range(args.rounds):
        for t in TILES:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.rep):
                fn(o, t)
            e1.record()
            torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) * 1e3 / args.rep)
    res[name] = {t: sorted(v)[len(v) // 2] for t, v in times.items()}
    print(f'{name:26s} ' + '  '.join(f't{t}: {us:7.1f} us {fl / us / 1e6:7.1f} TF' for t, us in res[name].items()), flush=True)
for t in TILES:
    tot = sum(res[n][t] for n in res)
    print(f'tile {t:2d}: sum {tot:8.1f} us   {sum(c[1] for c in CASES) / tot / 1e6:7.1f} TF/s')
print(json.dumps({n: {str(t): round(u, 1) for t, u in r.items()} for n, r in res.items()}))
