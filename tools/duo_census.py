"""Residency census of the duo GEMM (experiments build): which CU ran each workgroup and when.
    python tools/duo_census.py [tile]"""
import sys, os, ctypes, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import ops, _lib
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 11
R, D, H = 25344, 768, 3072
bf = torch.bfloat16
x = torch.randn(R, D, device='cuda').to(bf); w = (torch.randn(H, D, device='cuda') * 0.02).to(bf); b = torch.randn(H, device='cuda')
o = torch.empty(R, H, device='cuda', dtype=bf); a = torch.empty(R, H, device='cuda', dtype=bf)
for _ in range(3):
    ops.linear_fwd(x, w, b, o, aux=a, epi=_lib.EPI_GELU, tile=tile)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 512
buf = (ctypes.c_uint * (4 * n))()
assert lib.mmae_debug_duo_census(buf, 4 * n) == 0
rows = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n)]
t_min = min(r[2] for r in rows)
per_cu = collections.defaultdict(list)
for i, (hw, xcc, t0, t1) in enumerate(rows):
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    per_cu[(xcc & 0xf, se, sh, cu)].append((i, (t0 - t_min) / 100.0, (t1 - t_min) / 100.0))
print('distinct CUs seen:', len(per_cu), ' workgroups:', n)
overl = 0
for k, v in sorted(per_cu.items())[:6]:
    print(k, ['wg %d: %.1f-%.1f us' % e for e in sorted(v, key=lambda e: e[1])])
for k, v in per_cu.items():
    v = sorted(v, key=lambda e: e[1])
    for (i0, a0, b0), (i1, a1, b1) in zip(v, v[1:]):
        if a1 < b0 - 1.0:
            overl += 1
print('pairs of workgroups overlapping in time on one CU:', overl)
hist = collections.Counter(len(v) for v in per_cu.values())
print('workgroups per CU histogram:', dict(hist))
print('end times: max %.1f us' % max((r[3] - t_min) / 100.0 for r in rows))

tb = (ctypes.c_uint * (512 * 8 * 3))()
assert lib.mmae_debug_duo_trace(tb, 512 * 8 * 3) == 0
for k, v in sorted(per_cu.items())[:3]:
    for (i, a0, b0) in sorted(v, key=lambda e: e[1]):
        ev = []
        for t in range(5):
            e = [tb[(i * 8 + t) * 3 + j] for j in range(3)]
            if e[0] == 0:
                break
            ev.append('L %.1f-%.1f E -%.1f' % tuple((x - t_min) / 100.0 for x in e))
        print(k, 'wg', i, ' | '.join(ev))
