"""CPU: the e5m2-for-dY question of the MX-fp8 path, answered at the emulation level.

The engine's 'mxfp8' mode quantises BOTH operands of the encoder's dX (and dW) products to e4m3 blocks (OCP MX: one power-of-two
scale per 32 elements).  Gradients are the classic case for e5m2 (more range, one mantissa bit less) in PER-TENSOR-scaled fp8
recipes; with a scale per 32 elements the range argument mostly disappears.  This file measures it instead of arguing: a stack of
transformer blocks (oracle/mx_oracle.mx_block's operand treatment) is differentiated in three arithmetics --

    f32      exact products                                           (the reference)
    e4m3     activations, weights AND gradient operands as e4m3 blocks (what the engine ships)
    e5m2dy   the same, but every GRADIENT operand (dY of the dX and dW products) as e5m2 blocks

-- and every gradient is compared with the f32 one.  `python tests/test_mx_e5m2_ab_cpu.py` prints the table that is kept as
profiles/r05_mxfp8_e5m2_ab.txt; the test pins the finding: e5m2 for dY is NOT better -- 1-7 % MORE whole-model gradient error, 10-20 %
more on the input gradient -- because the gradient error of an MX-fp8 step is dominated by the forward operands' rounding, which both
variants share; so the kernels keep one element format.  Test infrastructure only (imports oracle/)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle')]

BLOCK = 32


def fake_quant_np(x: np.ndarray, mbits: int, emax: int, emin: int) -> np.ndarray:
    """MX quantise -> dequantise along the last axis (blocks of 32): element format with `mbits` mantissa bits, largest binade 2^emax
    (largest value 1.75 * 2^emax for both e4m3fn -- 448 -- and e5m2 -- 57344), smallest normal 2^emin; shared exponent =
    floor(log2 amax) - emax, plus one when amax's mantissa exceeds 1.75 (the engine's no-saturation rule, oracle/mx_oracle.py)."""
    x = np.asarray(x, dtype=np.float64)
    shp = x.shape
    xb = x.reshape(-1, shp[-1] // BLOCK, BLOCK)
    amax = np.abs(xb).max(axis=2)
    with np.errstate(divide='ignore'):
        fl = np.floor(np.log2(np.where(amax > 0, amax, 1.0)))
    se = fl - emax + ((amax / np.exp2(fl)) > 1.75)
    se = np.where(amax > 0, se, -127.0)
    y = xb / np.exp2(se)[:, :, None]
    a = np.abs(y)
    with np.errstate(divide='ignore'):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, float(emin))
    quantum = np.exp2(e - mbits)
    v = np.minimum(np.rint(a / quantum) * quantum, 1.75 * 2.0 ** emax)
    return (np.sign(y) * v * np.exp2(se)[:, :, None]).reshape(shp)


E4M3 = dict(mbits=3, emax=8, emin=-6)
E5M2 = dict(mbits=2, emax=15, emin=-14)


def fq(t: torch.Tensor, fmt) -> torch.Tensor:
    return torch.from_numpy(fake_quant_np(t.detach().float().numpy(), **fmt).astype(np.float32))


def bf16(t):
    return t.to(torch.bfloat16).float()


def make_linear(mode):
    """mode: 'f32' | 'e4m3' | 'e5m2dy'"""
    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x2, w, b):
            xb = bf16(x2)
            ctx.save_for_backward(xb, w)
            return fq(xb, E4M3) @ fq(w, E4M3).t() + b

        @staticmethod
        def backward(ctx, dy):
            xb, w = ctx.saved_tensors
            dyb = bf16(dy)
            gfmt = E5M2 if mode == 'e5m2dy' else E4M3
            dx = fq(dyb, gfmt) @ fq(w.t().contiguous(), E4M3).t()                       # dY along n_out, W^T along n_out
            # dW = dY^T X: both operands blocked along the ROWS (the engine's mmae_mx_quant_rows_t); pad the rows to whole blocks
            R = dyb.shape[0]
            Rp = (R + BLOCK - 1) // BLOCK * BLOCK
            dyp = torch.zeros(Rp, dyb.shape[1]); dyp[:R] = dyb
            xp = torch.zeros(Rp, xb.shape[1]); xp[:R] = xb
            dw = fq(dyp.t().contiguous(), gfmt) @ fq(xp.t().contiguous(), E4M3).t()
            return dx, dw, dy.sum(0)

    def lin(x, w, b):
        if mode == 'f32':
            return x @ w.t() + b
        shp = x.shape
        return Fn.apply(x.reshape(-1, shp[-1]), w, b).reshape(*shp[:-1], w.shape[0])
    return lin


def block(x, p, heads, lin):
    import multimae_oracle as orc
    B, N, C = x.shape
    d = C // heads
    h = orc.layer_norm(x, p['n1w'], p['n1b'], 1e-6)
    qkv = lin(h, p['qkvw'], p['qkvb']).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    a = ((qkv[0] @ qkv[1].transpose(-2, -1)) * d ** -0.5).softmax(-1)
    x = x + lin((a @ qkv[2]).transpose(1, 2).reshape(B, N, C), p['pw'], p['pb'])
    h = orc.gelu_erf(lin(orc.layer_norm(x, p['n2w'], p['n2b'], 1e-6), p['f1w'], p['f1b']))
    return x + lin(h, p['f2w'], p['f2b'])


def run(mode, params, x0, tgt, heads):
    ps = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in params]
    x = x0.clone().requires_grad_(True)
    h = x
    lin = make_linear(mode)
    for p in ps:
        h = block(h, p, heads, lin)
    loss = ((h - tgt) ** 2).mean()
    loss.backward()
    g = {'x': x.grad}
    for i, p in enumerate(ps):
        for k, v in p.items():
            g[f'{i}.{k}'] = v.grad
    return float(loss), g


def experiment(depth=6, D=256, heads=4, N=64, B=4, seed=0):
    gen = torch.Generator().manual_seed(seed)
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=gen) * sc
    params = []
    for _ in range(depth):
        params.append(dict(n1w=1 + rn(D, sc=0.1), n1b=rn(D, sc=0.1), qkvw=rn(3 * D, D, sc=D ** -0.5), qkvb=rn(3 * D, sc=0.02),
                           pw=rn(D, D, sc=D ** -0.5), pb=rn(D, sc=0.02), n2w=1 + rn(D, sc=0.1), n2b=rn(D, sc=0.1),
                           f1w=rn(4 * D, D, sc=D ** -0.5), f1b=rn(4 * D, sc=0.02), f2w=rn(D, 4 * D, sc=(4 * D) ** -0.5), f2b=rn(D, sc=0.02)))
    x0, tgt = rn(B, N, D), rn(B, N, D)
    ref_loss, ref = run('f32', params, x0, tgt, heads)
    out = {}
    for mode in ('e4m3', 'e5m2dy'):
        loss, g = run(mode, params, x0, tgt, heads)
        errs = {k: float((g[k] - ref[k]).norm() / (ref[k].norm() + 1e-30)) for k in ref}
        glob = float(torch.sqrt(sum(((g[k] - ref[k]) ** 2).sum() for k in ref) / sum((ref[k] ** 2).sum() for k in ref)))
        mats = [v for k, v in errs.items() if k.endswith('w') and not k.endswith(('n1w', 'n2w'))]
        out[mode] = dict(loss=loss, whole=glob, worst=max(errs.values()), worst_name=max(errs, key=errs.get), mean_matrix=sum(mats) / len(mats), dx=errs['x'])
    return ref_loss, out


def test_e5m2_gradient_operands_do_not_beat_e4m3_under_block_scaling():
    _, out = experiment(depth=4, D=128, heads=4, N=32, B=2, seed=0)
    a, b = out['e4m3'], out['e5m2dy']
    assert 0.005 < a['whole'] < 0.2 and 0.005 < b['whole'] < 0.3, out        # fp8-class errors, both finite
    assert b['whole'] >= 0.995 * a['whole'] and b['dx'] > a['dx'], out        # never better; visibly worse where only the backward chain acts
    assert b['whole'] < 1.3 * a['whole'], out                                # ... and not dramatically: the forward rounding dominates both


if __name__ == '__main__':
    print('# MX-fp8 gradient operands: e4m3 (shipped) against e5m2 for every dY operand (dX and dW products), emulated on the CPU with the')
    print("# engine's operand treatment (bf16 -> blocks of 32 with a power-of-two scale, no-saturation exponent rule); relative 2-norm error of")
    print('# the gradients against exact f32 products.  tests/test_mx_e5m2_ab_cpu.py')
    print('# depth  D    seed | e4m3: whole-model  mean matrix  worst tensor  d_input | e5m2 dY: whole-model  mean matrix  worst tensor  d_input | ratio (whole)')
    for depth, D, heads, N, B in ((4, 256, 4, 64, 4), (12, 256, 4, 64, 4), (24, 256, 4, 64, 2)):
        for seed in (0, 1):
            _, o = experiment(depth, D, heads, N, B, seed)
            a, b = o['e4m3'], o['e5m2dy']
            print(f'  {depth:3d}  {D:4d}  {seed:3d}  |   {a["whole"]:.4f}        {a["mean_matrix"]:.4f}      {a["worst"]:.4f}     {a["dx"]:.4f}  |      {b["whole"]:.4f}        {b["mean_matrix"]:.4f}      {b["worst"]:.4f}     {b["dx"]:.4f}  |  {b["whole"] / a["whole"]:.2f}x')
    print('# reading: with a scale per 32 elements the exponent range of e5m2 buys nothing (no block of a gradient tensor spans more than the')
    print('# 2^-6 .. 2^8 of e4m3 after scaling); its missing mantissa bit shows as +1..7 % whole-model gradient error and +10..20 % on the input')
    print('# gradient (the tensor only the backward chain touches).  The difference is small because the gradient error of an MX-fp8 step is')
    print('# dominated by the FORWARD operands (e4m3 activations and weights change what is differentiated), which both variants share.  e5m2 for')
    print('# dY is therefore never better here: the kernels keep e4m3 for every operand; no e5m2 instantiation of the quantisers / scaled MFMA.')
