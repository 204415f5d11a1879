"""Where does an output tile of the ping-pong GEMM spend its time?  Phase stamps (s_memrealtime = 100 MHz wall time, and s_memtime = shader-clock cycles: their ratio is the clock) of workgroup 0, waves 0 and 4, for the encoder
products with the interesting epilogues -- needs the trace build of the library:
    make -C multimae_amd/csrc trace
    MMAE_LIB=$PWD/multimae_amd/libmmae_hip_trace.so python tools/pp_trace.py [--decoder]
Stamps per output tile: A tile start | B main loop done | C ring drained + barrier | D next tile's first DMA issued | E, F, G the two or
three 64-row store calls done | H stores acknowledged + barrier (gemm_pp_body.h, PP_STAMP)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimae_amd import _lib, ops
from multimae_amd._lib import EPI_GELU_G, EPI_MUL

DEC = '--decoder' in sys.argv                           # the D = 256 block of an output adapter (B x 196 rows) instead of the encoder's
R, D, H = (50176, 256, 1024) if DEC else (25344, 768, 3072)
dev, bf = 'cuda', torch.bfloat16
g = lambda *s: torch.randn(*s, device=dev)
x_act, ao, hact = g(R, D).to(bf), g(R, D).to(bf), g(R, H).to(bf)
x_res = g(R, D)
wqkv, wproj, wfc1, wfc2 = (g(3 * D, D) * 0.02).to(bf), (g(D, D) * 0.02).to(bf), (g(H, D) * 0.02).to(bf), (g(D, H) * 0.02).to(bf)
bqkv, bproj, bfc1, bfc2 = g(3 * D), g(D), g(H), g(D)
qkv, hpre, hout = torch.empty(R, 3 * D, device=dev, dtype=bf), torch.empty(R, H, device=dev, dtype=bf), torch.empty(R, H, device=dev, dtype=bf)
x1 = torch.empty(R, D, device=dev)
d_h, d_x = g(R, H).to(bf), g(R, D).to(bf)
dout_d, dout_h = torch.empty(R, D, device=dev, dtype=bf), torch.empty(R, H, device=dev, dtype=bf)
cs = torch.empty(H, device=dev)
d_qkv = g(R, 3 * D).to(bf)
gw = {k: torch.zeros(v.shape, device=dev) for k, v in dict(qkv=wqkv, proj=wproj, fc1=wfc1, fc2=wfc2).items()}
gb = {k: torch.zeros(v.shape[0], device=dev) for k, v in dict(qkv=wqkv, proj=wproj, fc1=wfc1, fc2=wfc2).items()}
cases = [('fwd qkv  bias -> bf16', lambda: ops.linear_fwd(x_act, wqkv, bqkv, qkv)),
         ('fwd fc1  bias + GELU + GELU\' -> 2 x bf16', lambda: ops.linear_fwd(x_act, wfc1, bfc1, hout, aux=hpre, epi=EPI_GELU_G)),
         ('fwd fc2  bias + residual -> f32', lambda: ops.linear_fwd(hact, wfc2, bfc2, x1, resid=x_res)),
         ('dx  fc2  x aux + column sums -> bf16', lambda: ops.linear_dx(d_x, wfc2, dout_h, aux=hpre, epi=EPI_MUL, colsum_out=cs)),
         ('dx  fc1  -> bf16', lambda: ops.linear_dx(d_h, wfc1, dout_d)),
         ('dW group {fc2, fc1} (the MLP half of a block backward)', lambda: ops.gemm_dw_group([(d_x, hact, gw['fc2'], gb['fc2']), (d_h, x_act, gw['fc1'], gb['fc1'])], False)),
         ('dW group {proj, qkv} (the attention half)', lambda: ops.gemm_dw_group([(d_x, ao, gw['proj'], gb['proj']), (d_qkv, x_act, gw['qkv'], gb['qkv'])], False)),
         ('dW group {fc2, fc1, proj, qkv} in one launch', lambda: ops.gemm_dw_group([(d_x, hact, gw['fc2'], gb['fc2']), (d_h, x_act, gw['fc1'], gb['fc1']),
                                                                                       (d_x, ao, gw['proj'], gb['proj']), (d_qkv, x_act, gw['qkv'], gb['qkv'])], False))]
lib = ctypes.CDLL(_lib.LIB_PATH)
assert hasattr(lib, 'mmae_debug_pp_trace'), 'not a trace build: make -C multimae_amd/csrc trace; MMAE_LIB=.../libmmae_hip_trace.so'
NAMES = 'ABCDEFGH'
for name, fn in cases:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = (ctypes.c_longlong * 256)()
    dwk = name.startswith('dW')                          # the grouped weight-gradient kernel lives in the other translation unit
    assert (lib.mmae_debug_pp_trace_dw if dwk else lib.mmae_debug_pp_trace)(buf) == 0
    print(f'== {name}: {ms * 1e3:.1f} us between events')
    wg = (ctypes.c_longlong * 4096)()
    assert (lib.mmae_debug_pp_wg_dw if dwk else lib.mmae_debug_pp_wg)(wg) == 0
    rows = [(wg[4 * i], wg[4 * i + 1], wg[4 * i + 2], wg[4 * i + 3]) for i in range(256) if wg[4 * i + 1] > wg[4 * i] > 0]
    if rows:
        t0 = min(r[0] for r in rows)
        st = sorted((r[0] - t0) / 100.0 for r in rows)
        en = sorted((r[1] - t0) / 100.0 for r in rows)
        du = sorted((r[1] - r[0]) / 100.0 for r in rows)
        q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
        print(f'  {len(rows)} workgroups: starts 0 .. {st[-1]:.1f} us (median {q(st, .5):.1f}, 90 % {q(st, .9):.1f}); ends {en[0]:.1f} .. {en[-1]:.1f} us (median {q(en, .5):.1f}); '
              f'lifetimes {du[0]:.1f} .. {du[-1]:.1f} us (median {q(du, .5):.1f})')
        by_t = {}
        for r in rows:
            by_t.setdefault(r[3], []).append((r[1] - r[0]) / 100.0)
        print('  lifetime by tiles done: ' + ', '.join(f'{k} tiles x {len(v)}: {min(v):.1f}-{max(v):.1f} us' for k, v in sorted(by_t.items())))
        by_x = {}
        for r in rows:
            by_x.setdefault(r[2], []).append(((r[0] - t0) / 100.0, (r[1] - t0) / 100.0))
        print('  by XCC (first start, last end): ' + ', '.join(f'{k}: {min(a for a, _ in v):.1f}-{max(b for _, b in v):.1f}' for k, v in sorted(by_x.items())))
    for w in range(2):
        n = buf[w * 128] // 2
        t = [buf[w * 128 + 1 + 2 * i] for i in range(n)]            # 100 MHz wall-time counter
        c = [buf[w * 128 + 2 + 2 * i] for i in range(n)]            # shader-clock cycles
        if n < 8:
            print(f'  wave {4 * w}: {n} stamps'); continue
        if n > 24:                                       # many short tiles (decoder shapes): the first, a middle and the last are enough
            keep = [0, (n // 8) // 2, n // 8 - 1]
        else:
            keep = list(range(n // 8))
        print(f'  wave {4 * w}: {n // 8} tiles, workgroup 0 busy for {(t[-1] - t[0]) / 100.0:.1f} us, {c[-1] - c[0]} cycles = {(c[-1] - c[0]) / ((t[-1] - t[0]) / 100.0) / 1e3:.2f} GHz on average')
        for tile in keep:
            s, k = t[tile * 8:tile * 8 + 8], c[tile * 8:tile * 8 + 8]
            d = [(s[i + 1] - s[i]) / 100.0 for i in range(7)]
            loop_ghz = (k[1] - k[0]) / max(d[0], 1e-9) / 1e3
            epi_us = (s[7] - s[1]) / 100.0
            epi_ghz = (k[7] - k[1]) / max(epi_us, 1e-9) / 1e3
            print('    tile %d: ' % tile + '  '.join(f'{NAMES[i]}{NAMES[i + 1]} {d[i]:5.2f}' for i in range(7))
                  + f'   | loop {d[0]:5.2f} us = {k[1] - k[0]} cycles at {loop_ghz:.2f} GHz, epilogue {epi_us:5.2f} us at {epi_ghz:.2f} GHz')
