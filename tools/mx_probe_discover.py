"""Discover the operand / scale lane layout of v_mfma_scale_f32_32x32x64_f8f6f4 with one-hot experiments (mmae_probe_mx_mfma)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from multimae_amd import _lib, ops
from oracle import mx_oracle as mx

DEV = 'cuda'
ONE = 0x38           # e4m3 1.0
table = mx.e4m3_decode_table().astype(np.float64)


def run(a, b, sa, sb, oa=0, ob=0):
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    out = torch.empty((64, 16), device=DEV, dtype=torch.float32)
    a_d, b_d, sa_d, sb_d = t(a), t(b), t(sa), t(sb)
    _lib.check(_lib.load().mmae_probe_mx_mfma(a_d.data_ptr(), b_d.data_ptr(), sa_d.data_ptr(), sb_d.data_ptr(), oa, ob, out.data_ptr(), ops._stream()), 'probe')
    raw = out.cpu().numpy().astype(np.float64)
    D = np.zeros((32, 32))
    for l in range(64):
        for r in range(16):
            D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = raw[l, r]
    return D


def scales(v=127):
    return np.full((64, 4), v, dtype=np.uint8)


def main():
    ones = np.full((64, 32), ONE, dtype=np.uint8)
    zeros = np.zeros((64, 32), dtype=np.uint8)
    # 0. sanity: all ones -> 64 everywhere
    D = run(ones, ones, scales(), scales())
    print('all-ones D unique values:', np.unique(D))
    # 1. C/D layout + row/col of operand lanes: A one-hot lane la (all bytes of that lane), B all ones -> which D rows light up
    for la in (0, 1, 5, 31, 32, 33, 63):
        a = zeros.copy(); a[la, :] = ONE
        D = run(a, ones, scales(), scales())
        nz = np.argwhere(D != 0)
        print(f'A lane {la:2d} all bytes: D nonzero rows {sorted(set(nz[:, 0]))} cols {len(set(nz[:, 1]))} value {np.unique(D[D != 0])}')
    for lb in (0, 1, 31, 32, 63):
        b = zeros.copy(); b[lb, :] = ONE
        D = run(ones, b, scales(), scales())
        nz = np.argwhere(D != 0)
        print(f'B lane {lb:2d} all bytes: D nonzero cols {sorted(set(nz[:, 1]))} rows {len(set(nz[:, 0]))} value {np.unique(D[D != 0])}')
    # 2. k mapping: A one-hot (lane 0 or 32, byte ba); B lanes 0 and 32 carry 64 distinct values (byte codes 0x20 + 32 h + b)
    b = zeros.copy()
    for h in range(2):
        for bb in range(32):
            b[32 * h, bb] = 0x20 + 32 * h + bb
    inv = {table[0x20 + i]: i for i in range(64)}
    kmap = {}
    for h in range(2):
        for ba in range(32):
            a = zeros.copy(); a[32 * h, ba] = ONE
            D = run(a, b, scales(), scales())
            v = D[0, 0]
            kmap[(h, ba)] = inv.get(v, None)
    ident = all(kmap[(h, ba)] == 32 * h + ba for h in range(2) for ba in range(32))
    print('A (half, byte) -> matching B position 32 h + b; identity:', ident)
    if not ident:
        for h in range(2):
            print(' half', h, [kmap[(h, ba)] for ba in range(32)])
    # 3. which lane's scale applies to which k: A one-hot (row 0: lane 0 / 32, byte ba), B all ones; scale_a lane 0 = x2, lane 32 = x4, others x1
    sa = scales(); sa[0, :] = 128; sa[32, :] = 129
    res = {}
    for h in range(2):
        for ba in range(32):
            a = zeros.copy(); a[32 * h, ba] = ONE
            D = run(a, ones, sa, scales())
            res[(h, ba)] = D[0, 0]
    print('scale_a factor seen by A (half 0, byte b):', [res[(0, b)] for b in range(32)])
    print('scale_a factor seen by A (half 1, byte b):', [res[(1, b)] for b in range(32)])
    # other rows' lanes must not matter: scale of lane 1 (row 1) x8
    sa = scales(); sa[1, :] = 130
    a = zeros.copy(); a[0, 0] = ONE
    print('row-0 element with lane-1 scale x8:', run(a, ones, sa, scales())[0, 0])
    # 4. op_sel: byte selection
    for oa in range(4):
        sa = scales(); sa[:, 0] = 127; sa[:, 1] = 128; sa[:, 2] = 129; sa[:, 3] = 130
        D = run(ones, ones, sa, scales(), oa, 0)
        print(f'op_sel_a {oa}: D[0,0] = {D[0, 0]} (64 x factor)')
    for ob in range(4):
        sb = scales(); sb[:, 0] = 127; sb[:, 1] = 128; sb[:, 2] = 129; sb[:, 3] = 130
        D = run(ones, ones, scales(), sb, 0, ob)
        print(f'op_sel_b {ob}: D[0,0] = {D[0, 0]}')
    # 5. scale_b mapping: B one-hot (col 0: lane 0 / 32, byte bb), A all ones; scale_b lane 0 x2, lane 32 x4
    sb = scales(); sb[0, :] = 128; sb[32, :] = 129
    r0 = []; r1 = []
    for bb in range(32):
        b = zeros.copy(); b[0, bb] = ONE
        r0.append(run(ones, b, scales(), sb)[0, 0])
        b = zeros.copy(); b[32, bb] = ONE
        r1.append(run(ones, b, scales(), sb)[0, 0])
    print('scale_b factor seen by B (half 0):', r0)
    print('scale_b factor seen by B (half 1):', r1)


if __name__ == '__main__':
    main()
