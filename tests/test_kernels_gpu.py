"""GPU parity tests, kernel level: every HIP kernel (called through the C ABI) against the CPU
oracle / plain fp32 math on the same seeded inputs.  Integer outputs must be bit-exact; fp
tolerances are stated per test."""
import ctypes
import math

import numpy as np
import pytest
import torch

import multimae_oracle as orc
from helpers import load_masks_base, load_mini, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def bf(x):
    return x.to(torch.bfloat16)


# ----------------------------------------------------------------------------------------------
def test_probe_tr16_semantics():
    """Pins the ds_read_b64_tr_b16 lane mapping the k-strided GEMM operands rely on: within a
    16-lane group, lane i supplies the address of 4 contiguous bf16 (row i>>2, cols (i&3)*4..+3 of a
    [4][16] block) and receives column i (rows 0..3)."""
    from multimae_amd import _lib, ops
    img = torch.arange(1024, dtype=torch.int32).to(torch.int16)
    # 4 groups: group g reads block rows 4g..4g+3 of a [16][64]-element row-major image (row stride 128 B)
    addr = torch.zeros(64, dtype=torch.int32)
    for l in range(64):
        g, i = l >> 4, l & 15
        row, col = 4 * g + (i >> 2), (i & 3) * 4
        addr[l] = (row * 64 + col) * 2
    out = torch.zeros(256, dtype=torch.int16, device=DEV)
    img_d, addr_d = img.to(DEV), addr.to(DEV)          # keep the device buffers alive across the launch
    _lib.check(_lib.load().mmae_probe_tr16(img_d.data_ptr(), addr_d.data_ptr(), out.data_ptr(), ops._stream()), 'probe')
    torch.cuda.synchronize()
    out = out.cpu().view(64, 4).int()
    exp = torch.zeros(64, 4, dtype=torch.int32)
    for l in range(64):
        g, i = l >> 4, l & 15
        for j in range(4):
            exp[l, j] = (4 * g + j) * 64 + i
    assert torch.equal(out, exp), f'tr16 mapping differs:\n{out[:20]}\nexpected\n{exp[:20]}'


# ----------------------------------------------------------------------------------------------
def _gemm_case(dtype, M, N, K, a_trans, b_trans, *, tile=0, seed=0):
    from multimae_amd import ops
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    if dtype == torch.bfloat16:
        A, B = bf(A).float(), bf(B).float()
    ref = A.double() @ B.double().t()
    pad = lambda n: (n + 7) // 8 * 8
    if a_trans:
        As = torch.zeros(K, pad(M)); As[:, :M] = A.t(); lda = pad(M)
    else:
        As = torch.zeros(M, pad(K)); As[:, :K] = A; lda = pad(K)
    if b_trans:
        Bs = torch.zeros(K, pad(N)); Bs[:, :N] = B.t(); ldb = pad(N)
    else:
        Bs = torch.zeros(N, pad(K)); Bs[:, :K] = B; ldb = pad(K)
    C = torch.full((M, pad(N)), float('nan'), device=DEV)
    ops.gemm(As.to(DEV, dtype), Bs.to(DEV, dtype), C, M, N, K, lda=lda, ldb=ldb, ldc=pad(N), a_trans=a_trans, b_trans=b_trans, tile=tile)
    torch.cuda.synchronize()
    out = C[:, :N].cpu().double()
    assert torch.isfinite(out).all()
    return float((out - ref).norm() / ref.norm())


@pytest.mark.parametrize('a_trans,b_trans', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('shape', [(256, 256, 128), (128, 384, 64), (300, 200, 136), (99, 99, 64), (196, 99, 32), (520, 72, 1000)])
def test_gemm_bf16_layouts(shape, a_trans, b_trans):
    """bf16 products are exact in fp32, so vs fp64 on the same bf16 inputs only fp32 summation
    error remains: rel-L2 <= 2e-6."""
    M, N, K = shape
    for tile in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
        assert _gemm_case(torch.bfloat16, M, N, K, a_trans, b_trans, tile=tile) < 2e-6, (shape, tile)


@pytest.mark.parametrize('a_trans,b_trans', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('tile', [9, 10])
@pytest.mark.parametrize('shape', [(1000, 520, 32), (1000, 520, 96), (700, 300, 1336), (2048, 768, 776)])
def test_gemm_bf16_pingpong_ring(shape, tile, a_trans, b_trans):
    """8-wave ping-pong kernel (tile codes 9 / 10): several 256/320-row tiles with ragged edges, K from one 32-wide tile
    (shorter than the 4-deep DMA ring) to 42 tiles (ring wraps ten times); repeated to catch ring races."""
    M, N, K = shape
    for seed in range(3):
        assert _gemm_case(torch.bfloat16, M, N, K, a_trans, b_trans, tile=tile, seed=seed) < 2e-6, (shape, tile, seed)


@pytest.mark.parametrize('a_trans,b_trans', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('shape', [(256, 256, 128), (300, 200, 136), (99, 99, 64), (70, 530, 33)])
def test_gemm_f32_layouts(shape, a_trans, b_trans):
    """exact-f32 MFMA == fmaf chain: rel-L2 <= 2e-6 vs fp64."""
    M, N, K = shape
    assert _gemm_case(torch.float32, M, N, K, a_trans, b_trans) < 2e-6


@pytest.mark.parametrize('a_trans,b_trans', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('shape', [(256, 256, 128), (300, 200, 136), (99, 104, 64), (520, 72, 1000)])
def test_gemm_f32x3_layouts(shape, a_trans, b_trans):
    """split-bf16 GEMM on f32 operands: ~16 operand mantissa bits -> rel-L2 <= 2e-5 vs fp64 (TF32 would be ~3e-4,
    plain bf16 ~3e-3)."""
    from multimae_amd import ops
    M, N, K = shape
    with ops.f32_gemm_mode('x3'):
        err = _gemm_case(torch.float32, M, N, K, a_trans, b_trans)
    assert err < 2e-5, err
    assert err > 1e-7            # (sanity: this really is the split path, not the exact-f32 kernel)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_gemm_splitk_workspace(dtype):
    """dW-shaped product (tiny M x N, long K): K slices through the workspace + fixed-order reduce; also
    accumulate-into-C and bit-reproducibility (no atomics)."""
    from multimae_amd import _lib, ops
    torch.manual_seed(12)
    Mr, N, K = 4096, 256, 136                 # dy [Mr, N], x [Mr, K]  ->  dW [N, K]
    dy, x = torch.randn(Mr, N) * 0.1, torch.randn(Mr, K)
    if dtype == torch.bfloat16:
        dy, x = bf(dy).float(), bf(x).float()
    assert _lib.load().mmae_gemm_auto_splitk(N, K, Mr, ops.dcode(dtype)) > 1
    ref = dy.double().t() @ x.double()
    dyd, xd = dy.to(DEV, dtype), x.to(DEV, dtype)
    dw = torch.full((N, K), 2.0, device=DEV)
    ops.linear_dw(dyd, xd, dw, accumulate=True)
    assert rel_err(dw, 2.0 + ref) < 2e-6
    dw2 = torch.empty((N, K), device=DEV)
    ops.linear_dw(dyd, xd, dw2, accumulate=False)
    dw3 = torch.empty((N, K), device=DEV)
    ops.linear_dw(dyd, xd, dw3, accumulate=False)
    assert rel_err(dw2, ref) < 2e-6 and torch.equal(dw2, dw3)


@pytest.mark.parametrize('Mr,N,K', [(12544, 1024, 256), (5000, 256, 136), (50176, 256, 256), (3000, 384, 264)])
def test_gemm_f32x3_split_k_weight_gradient_shapes(Mr, N, K):
    """The fp32 output adapter's weight gradients (split-bf16 x3 products, contraction = all rows, split along it): the kernel hands whole
    K slices to an XCD (slice / tile remap of the flattened workgroup id, plain order for the slices past the last multiple of 8) --
    every (slice, tile) pair must be computed exactly once: against fp64, accumulate-into-C, and bit-reproducible."""
    from multimae_amd import ops
    g = torch.Generator().manual_seed(Mr + N)
    dy, x = torch.randn(Mr, N, generator=g) * 0.1, torch.randn(Mr, K, generator=g)
    ref = dy.double().t() @ x.double()
    with ops.f32_gemm_mode('x3'):
        dw = torch.full((N, K), 2.0, device=DEV)
        ops.linear_dw(dy.to(DEV), x.to(DEV), dw, accumulate=True)
        dw2, dw3 = torch.empty((N, K), device=DEV), torch.empty((N, K), device=DEV)
        ops.linear_dw(dy.to(DEV), x.to(DEV), dw2, accumulate=False)
        ops.linear_dw(dy.to(DEV), x.to(DEV), dw3, accumulate=False)
    assert rel_err(dw2, ref) < 2e-5 and rel_err(dw, 2.0 + ref) < 2e-5 and torch.equal(dw2, dw3)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_gemm_epilogues(dtype):
    from multimae_amd import ops
    from multimae_amd._lib import EPI_DGELU, EPI_GELU
    torch.manual_seed(1)
    M, N, K = 200, 264, 128
    x, w = torch.randn(M, K) * 0.5, torch.randn(N, K) * 0.2
    if dtype == torch.bfloat16:
        x, w = bf(x).float(), bf(w).float()
    bias, resid = torch.randn(N), torch.randn(M, N)
    xd, wd = x.to(DEV, dtype), w.to(DEV, dtype)
    lin = x @ w.t() + bias
    tol = 1e-5 if dtype == torch.float32 else 1e-5
    # bias + residual, f32 out
    out = torch.empty(M, N, device=DEV)
    ops.linear_fwd(xd, wd, bias.to(DEV), out, resid=resid.to(DEV))
    assert rel_err(out, lin + resid) < tol
    # GELU epilogue: aux = pre-activation, C = gelu
    aux = torch.empty(M, N, device=DEV, dtype=dtype)
    act = torch.empty(M, N, device=DEV, dtype=dtype)
    ops.linear_fwd(xd, wd, bias.to(DEV), act, aux=aux, epi=EPI_GELU)
    t2 = 1e-5 if dtype == torch.float32 else 4e-3       # bf16 output rounding 2^-9
    assert rel_err(aux.float(), lin) < t2
    assert rel_err(act.float(), orc.gelu_erf(lin)) < t2
    # dGELU epilogue of the dX product: out = (dy @ W) * gelu'(pre)
    dy = torch.randn(M, N) * 0.3
    if dtype == torch.bfloat16:
        dy = bf(dy).float()
    pre = torch.randn(M, K)
    if dtype == torch.bfloat16:
        pre = bf(pre).float()
    pre_d = pre.to(DEV, dtype)
    dxo = torch.empty(M, K, device=DEV, dtype=dtype)
    cs = torch.empty(K, device=DEV)
    ops.linear_dx(dy.to(DEV, dtype), wd, dxo, aux=pre_d, epi=EPI_DGELU, colsum_out=cs)
    p = pre.clone().requires_grad_(True)
    orc.gelu_erf(p).backward(dy @ w)
    assert rel_err(dxo.float(), p.grad) < t2
    assert rel_err(cs, p.grad.sum(0)) < 1e-4            # bias gradient accumulated in the epilogue
    # the same pair with the derivative taken in the forward epilogue (EPI_GELU_G: aux = gelu'(pre)) and a plain multiply backward (EPI_MUL)
    from multimae_amd._lib import EPI_GELU_G, EPI_MUL
    aux_g = torch.empty(M, N, device=DEV, dtype=dtype)
    act_g = torch.empty(M, N, device=DEV, dtype=dtype)
    ops.linear_fwd(xd, wd, bias.to(DEV), act_g, aux=aux_g, epi=EPI_GELU_G)
    assert torch.equal(act_g, act)
    lp = lin.clone().requires_grad_(True)
    orc.gelu_erf(lp).sum().backward()
    assert rel_err(aux_g.float(), lp.grad) < t2
    dxm = torch.empty(M, K, device=DEV, dtype=dtype)
    csm = torch.empty(K, device=DEV)
    gpre = torch.randn(M, K)
    if dtype == torch.bfloat16:
        gpre = bf(gpre).float()
    ops.linear_dx(dy.to(DEV, dtype), wd, dxm, aux=gpre.to(DEV, dtype), epi=EPI_MUL, colsum_out=csm)
    assert rel_err(dxm.float(), (dy @ w) * gpre) < t2
    assert rel_err(csm, ((dy @ w) * gpre).sum(0)) < 1e-4
    # dW with accumulate
    dw = torch.ones(N, K, device=DEV)
    ops.linear_dw(dy.to(DEV, dtype), xd, dw, accumulate=True)
    assert rel_err(dw, 1.0 + dy.t() @ x) < 1e-5
    # bf16 C, alpha
    c16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(xd, wd, c16, M, N, K, lda=K, ldb=K, ldc=N, alpha=0.5)
    assert rel_err(c16.float(), 0.5 * (x @ w.t())) < 4e-3


@pytest.mark.parametrize('shape', [(25344, 3072, 64), (25344, 2304, 96), (50176, 1024, 256)])
def test_pingpong_tile_boundary_under_memory_load(shape):
    """The flavoured ping-pong kernels leave a tile with stores in flight and take the next tile's first K tiles behind a COUNTED wait (round 6, TAILD:
    a wave's loads, LDS-DMA pieces and stores retire in order).  Short contractions put that boundary under stress -- 3-4 tiles per persistent workgroup,
    two or three K tiles each, so the next tile's operands are requested while most of the previous tile's stores are outstanding.  40 launches of each
    epilogue flavour with a second stream hammering HBM beside them: every launch bit-equal to the first, the first equal to the fp32 product."""
    from multimae_amd import ops
    from multimae_amd._lib import EPI_GELU_G, EPI_MUL
    M_, N_, K_ = shape
    torch.manual_seed(9)
    x = (torch.randn(M_, K_, device=DEV) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N_, K_, device=DEV) * 0.2).to(torch.bfloat16)
    b = torch.randn(N_, device=DEV)
    dy = (torch.randn(M_, N_, device=DEV) * 0.3).to(torch.bfloat16)
    auxg = torch.rand(M_, K_, device=DEV).to(torch.bfloat16)
    noise_a, noise_b = torch.empty(1 << 28, device=DEV, dtype=torch.uint8), torch.empty(1 << 28, device=DEV, dtype=torch.uint8)
    side = torch.cuda.Stream()
    y, aux = torch.empty(M_, N_, device=DEV, dtype=torch.bfloat16), torch.empty(M_, N_, device=DEV, dtype=torch.bfloat16)
    dx = torch.empty(M_, K_, device=DEV, dtype=torch.bfloat16)
    cs = torch.empty(K_, device=DEV)
    cases = {'bias -> bf16': (lambda: ops.linear_fwd(x, w, b, y), lambda: (y,)),
             'bias + GELU pair -> 2 x bf16': (lambda: ops.linear_fwd(x, w, b, y, aux=aux, epi=EPI_GELU_G), lambda: (y, aux)),
             'dX x aux + column sums': (lambda: ops.linear_dx(dy, w, dx, aux=auxg, epi=EPI_MUL, colsum_out=cs), lambda: (dx, cs))}
    ref_lin = x.float() @ w.float().t() + b
    for name, (fn, outs) in cases.items():
        fn()
        torch.cuda.synchronize()
        first = [t.clone() for t in outs()]
        if name == 'bias -> bf16':
            assert rel_err(first[0].float(), ref_lin) < 4e-3, name
        if name.startswith('dX'):
            assert rel_err(first[0].float(), (dy.float() @ w.float()) * auxg.float()) < 6e-3, name
        for it in range(40):
            with torch.cuda.stream(side):
                noise_b.copy_(noise_a)                    # 256 MiB read + 256 MiB written beside the launch
            for t in outs():
                t.zero_()
            fn()
            torch.cuda.synchronize()
            for a, r in zip(outs(), first):
                assert torch.equal(a, r), (name, it, float((a.float() - r.float()).abs().max()))


def test_gemm_timing_books_flops_and_algorithmic_bytes():
    """mmae_gemm_timing_* (bench.py's roofline object): while enabled every GEMM entry point is bracketed by events on its stream and booked with its
    algorithmic FLOPs and -- ABI v7 -- its algorithmic HBM bytes (operands once, C and every epilogue stream once)."""
    import ctypes
    from multimae_amd import ops, _lib
    from multimae_amd._lib import EPI_GELU_G
    lib = _lib.load()
    M_, N_, K_ = 1024, 512, 256
    x = torch.randn(M_, K_, device=DEV).to(torch.bfloat16)
    w = torch.randn(N_, K_, device=DEV).to(torch.bfloat16)
    b = torch.randn(N_, device=DEV)
    y, aux = torch.empty(M_, N_, device=DEV, dtype=torch.bfloat16), torch.empty(M_, N_, device=DEV, dtype=torch.bfloat16)
    y32, res = torch.empty(M_, N_, device=DEV), torch.randn(M_, N_, device=DEV)
    lib.mmae_gemm_timing_enable(1)
    try:
        ops.linear_fwd(x, w, b, y)                                   # A + B + C
        ops.linear_fwd(x, w, b, y, aux=aux, epi=EPI_GELU_G)          # + the aux stream
        ops.linear_fwd(x, w, b, y32, resid=res)                      # f32 C + f32 residual
        torch.cuda.synchronize()
        ms, fl, n = (ctypes.c_double * 3)(), (ctypes.c_double * 3)(), (ctypes.c_int64 * 3)()
        by = (ctypes.c_double * 3)()
        assert lib.mmae_gemm_timing_read(ms, fl, n) == 0 and lib.mmae_gemm_timing_read_bytes(by) == 0
    finally:
        lib.mmae_gemm_timing_enable(0)
    ab = (M_ * K_ + N_ * K_) * 2
    want = (ab + M_ * N_ * 2) + (ab + 2 * M_ * N_ * 2) + (ab + 2 * M_ * N_ * 4)
    assert n[0] == 3 and fl[0] == 3 * 2.0 * M_ * N_ * K_ and ms[0] > 0
    assert by[0] == want, (by[0], want)
    assert n[1] == 0 and by[1] == 0 and by[2] == 0


def test_add_n_sums_in_index_order():
    """mmae_add_n_f32 (round 6: the output adapters' encoder-token gradients in one pass): bit-equal to the left-to-right chain of additions."""
    from multimae_amd import ops
    torch.manual_seed(4)
    xs = [torch.randn(3, 99, 768, device=DEV) * (10.0 ** (i - 2)) for i in range(4)]
    ref = ((xs[0] + xs[1]) + xs[2]) + xs[3]
    assert torch.equal(ops.add_n(xs), ref)
    assert torch.equal(ops.add_n(xs[:1]), xs[0]) and torch.equal(ops.add_n(xs[:2]), xs[0] + xs[1])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_semseg_class_embedding_gradient_is_deterministic(dtype):
    """mmae_semseg_emb_bwd_det (round 6): d_emb[class of every pixel of a selected semseg patch][e] += the patch row's gradient
    (SemSegInputAdapter's nn.Embedding backward, input_adapters.py:215-241) summed in a FIXED order -- against an fp64 index_add, bit-identical
    between two runs and to the accumulate form, where the float-atomic kernel it replaces differs from run to run."""
    from multimae_amd import ops
    B, H, W, E, ph, pw, n_cls, n_sel = 6, 16, 24, 64, 4, 4, 133, 20
    nh, nw = H // ph, W // pw
    n_patches, tok_off = nh * nw, 7                        # the semseg task owns tokens [7, 7 + 24)
    torch.manual_seed(3)
    cls = torch.randint(-1, n_cls + 1, (B, H, W))          # also ids outside [0, n_cls): no gradient
    sel = torch.stack([torch.randperm(tok_off + n_patches + 5)[:n_sel] for _ in range(B)])       # some tokens belong to other tasks
    K = E * ph * pw
    d_rows = torch.randn(B * n_sel, K)
    if dtype == torch.bfloat16:
        d_rows = bf(d_rows)
    ref = torch.zeros(n_cls, E, dtype=torch.float64)
    for b in range(B):
        for r in range(n_sel):
            p = int(sel[b, r]) - tok_off
            if p < 0 or p >= n_patches:
                continue
            py, px = divmod(p, nw)
            row = d_rows[b * n_sel + r].double().view(E, ph, pw)
            c = cls[b, py * ph:(py + 1) * ph, px * pw:(px + 1) * pw]
            ok = (c >= 0) & (c < n_cls)
            ref.index_add_(0, c[ok], row[:, ok].T.contiguous())
    dr, dc, ds = d_rows.to(DEV, dtype), cls.to(DEV), sel.to(DEV)
    kw = dict(B=B, H=H, W=W, E=E, ph=ph, pw=pw, n_sel=n_sel, k_off=0, tok_off=tok_off, n_patches=n_patches, n_cls=n_cls)
    outs = []
    for _ in range(2):
        g = torch.full((n_cls, E), 7.0, device=DEV)         # store mode: whatever the buffer held is overwritten
        ops.semseg_emb_bwd(dr, dc, ds, g, accumulate=False, **kw)
        outs.append(g.clone())
    assert torch.equal(outs[0], outs[1])
    assert (outs[0].double().cpu() - ref).abs().max() < 2e-4 * max(1.0, float(ref.abs().max()))
    g = torch.ones(n_cls, E, device=DEV)
    ops.semseg_emb_bwd(dr, dc, ds, g, accumulate=True, **kw)
    assert torch.equal(g, outs[0] + 1.0)
    ga = torch.zeros(n_cls, E, device=DEV)                 # the atomic form still agrees numerically
    ops.semseg_emb_bwd(dr, dc, ds, ga, accumulate=True, deterministic=False, **kw)
    assert (ga - outs[0]).abs().max() < 1e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('geom', [(3, 196, 99), (2, 196, 128), (2, 50, 33), (1, 1, 1)])
def test_fused_cross_attention_forward(geom):
    """mmae_xattn_fwd_fused (round 6: q-projection + kv-projection + softmax + PV in one launch, CrossAttention.forward of multimae_utils.py:199-214
    at the output adapters' D = 8 x 32 = 256): q, kv, out and lse against the three-launch form it replaces (bias GEMM -> bf16 twice + mmae_attn_fwd:
    the same fp32 accumulation order and the same single rounding -- bit for bit) and against the fp32 formula."""
    from multimae_amd import ops
    from multimae_amd.ops import AttnView
    B, Nq, Nk = geom
    D, H, hd = 256, 8, 32
    torch.manual_seed(11)
    qn, cn = bf(torch.randn(B * Nq, D)).to(DEV, torch.bfloat16), bf(torch.randn(B * Nk, D)).to(DEV, torch.bfloat16)
    wq, wkv = bf(torch.randn(D, D) * 0.06).to(DEV, torch.bfloat16), bf(torch.randn(2 * D, D) * 0.06).to(DEV, torch.bfloat16)
    bq, bkv = torch.randn(D, device=DEV), torch.randn(2 * D, device=DEV)
    q, kv, out, lse = ops.xattn_fwd_fused(qn, cn, wq, bq, wkv, bkv, B, Nq, Nk)
    # the three launches
    q3 = torch.empty_like(q); kv3 = torch.empty_like(kv); out3 = torch.empty_like(out)
    ops.linear_fwd(qn, wq, bq, q3)
    ops.linear_fwd(cn, wkv, bkv, kv3)
    st = ops.attention_fwd(AttnView(q3, 0, D, Nq), AttnView(kv3, 0, 2 * D, Nk), AttnView(kv3, D, 2 * D, Nk), AttnView(out3, 0, D, Nq), B, H, hd, hd ** -0.5)
    assert st[0] == 'fused'
    assert torch.equal(q, q3), (q.float() - q3.float()).abs().max()
    assert torch.equal(kv, kv3), (kv.float() - kv3.float()).abs().max()
    assert torch.equal(out, out3), (out.float() - out3.float()).abs().max()
    assert torch.equal(lse, st[1])
    # the formula, in fp32 on the bf16-rounded projections
    sp = lambda t, n: t.float().cpu().reshape(B, n, H, hd).permute(0, 2, 1, 3)
    qr = bf(qn.float().cpu() @ wq.float().cpu().T + bq.cpu())
    kvr = bf(cn.float().cpu() @ wkv.float().cpu().T + bkv.cpu())
    a = ((sp(qr, Nq) @ sp(kvr[:, :D], Nk).transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    o_ref = (a @ sp(kvr[:, D:], Nk)).transpose(1, 2).reshape(B * Nq, D)
    assert rel_err(q.float().cpu(), qr) < 4e-3 and rel_err(kv.float().cpu(), kvr) < 4e-3
    assert rel_err(out.float().cpu(), o_ref) < 1e-2


@pytest.mark.parametrize('path', ['fused', 'gemm', 'f32', 'f32x3', 'f32f16'])
@pytest.mark.parametrize('geom', [(3, 2, 17, 17, 64), (2, 12, 99, 99, 64), (2, 8, 196, 99, 32), (2, 8, 196, 196, 32), (1, 3, 50, 50, 64),
                                  (2, 16, 197, 197, 64), (1, 2, 256, 256, 32), (2, 3, 1, 33, 64)])
def test_attention_fwd_bwd(path, geom):
    """attention (self and cross) vs the oracle formula, fwd and bwd: the fused single-kernel bf16 path,
    the batched-GEMM + row-softmax bf16 path, the exact-f32 GEMM path, and the fused split-bf16 kernel for f32 activations
    (fp32 output adapters in speed mode; falls back to x3 GEMMs where its backward tiles exceed the LDS)."""
    from multimae_amd import ops
    from multimae_amd.ops import AttnView
    dtype = torch.float32 if path in ('f32', 'f32x3', 'f32f16') else torch.bfloat16
    ops.set_fused_attention(path in ('fused', 'f32x3', 'f32f16'))
    try:
        if path in ('f32x3', 'f32f16'):
            with ops.f32_gemm_mode('x3'):
                _attention_case(dtype, geom, path)
        else:
            _attention_case(dtype, geom, path)
    finally:
        ops.set_fused_attention(True)


def _attention_case(dtype, geom, path):
    from multimae_amd import ops
    from multimae_amd.ops import AttnView
    B, H, Nq, Nk, hd = geom
    D = H * hd
    torch.manual_seed(2)
    q, k, v, do = (torch.randn(B, n, D) for n in (Nq, Nk, Nk, Nq))
    f16 = path == 'f32f16'                                 # fp16-operand products (TF32-class); dO ~1e-7 with the amax pre-scale
    amax = None
    if f16:
        do = do * 2.3e-8
        amax = do.abs().max().reshape(1).to(DEV)
    if dtype == torch.bfloat16:
        q, k, v, do = (bf(t).float() for t in (q, k, v, do))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    sp = lambda t, n: t.reshape(B, n, H, hd).permute(0, 2, 1, 3)
    a = ((sp(qr, Nq) @ sp(kr, Nk).transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    o_ref = (a @ sp(vr, Nk)).transpose(1, 2).reshape(B, Nq, D)
    o_ref.backward(do)
    # packed like the engine: q in its own buffer, k|v packed with ld 2D
    qd = q.reshape(B * Nq, D).to(DEV, dtype)
    kvd = torch.cat([k, v], -1).reshape(B * Nk, 2 * D).to(DEV, dtype)
    od = torch.empty(B * Nq, D, device=DEV, dtype=dtype)
    fits = 4 * ((Nq + 31) // 32 * 32 + (Nk + 31) // 32 * 32) * hd * 2 + 8 * ((Nq + 31) // 32 * 32) <= 160 * 1024
    if f16 and not fits:
        pytest.skip('tiles beyond the LDS: the adapter falls back to GEMMs (covered by the f32x3 case)')
    P = ops.attention_fwd(AttnView(qd, 0, D, Nq), AttnView(kvd, 0, 2 * D, Nk), AttnView(kvd, D, 2 * D, Nk), AttnView(od, 0, D, Nq), B, H, hd,
                          hd ** -0.5, f16=f16)
    assert P[0] == ('fused' if (path == 'fused' or (path in ('f32x3', 'f32f16') and fits)) else 'gemm')
    tol = (1.5e-3 if f16 else 1e-4 if path == 'f32x3' else 2e-5) if dtype == torch.float32 else 1e-2
    assert rel_err(od.float().view(B, Nq, D), o_ref) < tol
    dq = torch.empty(B * Nq, D, device=DEV, dtype=dtype)
    dkv = torch.empty(B * Nk, 2 * D, device=DEV, dtype=dtype)
    dod = do.reshape(B * Nq, D).to(DEV, dtype)
    ops.attention_bwd(AttnView(qd, 0, D, Nq), AttnView(kvd, 0, 2 * D, Nk), AttnView(kvd, D, 2 * D, Nk), P, AttnView(od, 0, D, Nq), AttnView(dod, 0, D, Nq),
                      AttnView(dq, 0, D, Nq), AttnView(dkv, 0, 2 * D, Nk), AttnView(dkv, D, 2 * D, Nk), B, H, hd, hd ** -0.5, f16=f16, dy_amax=amax)
    tol = (3e-3 if f16 else 2e-4 if path == 'f32x3' else 5e-5) if dtype == torch.float32 else 2e-2
    assert rel_err(dq.float().view(B, Nq, D), qr.grad) < tol
    assert rel_err(dkv.float().view(B, Nk, 2 * D)[..., :D], kr.grad) < tol
    assert rel_err(dkv.float().view(B, Nk, 2 * D)[..., D:], vr.grad) < tol


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize('D', [64, 128, 192, 256, 768, 1024])
def test_layernorm_fwd_bwd(D):
    from multimae_amd import ops
    torch.manual_seed(3)
    R = 333
    x = torch.randn(R, D) * 2 + 0.5
    w, b = torch.randn(D), torch.randn(D)
    dy, dxin = torch.randn(R, D), torch.randn(R, D)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = orc.layer_norm(xr, wr, br, 1e-6)
    y_ref.backward(dy)
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, torch.float32)
    assert rel_err(y, y_ref) < 2e-6
    y16, _, _ = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, torch.bfloat16)
    assert rel_err(y16.float(), y_ref) < 4e-3
    dx, dxa, dg, db, cs = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, dxin.to(DEV), torch.bfloat16)
    assert rel_err(dx, xr.grad + dxin) < 5e-6
    assert rel_err(cs, (xr.grad + dxin).sum(0)) < 1e-5
    assert rel_err(dxa.float(), xr.grad + dxin) < 4e-3
    assert rel_err(dg, wr.grad) < 1e-5 and rel_err(db, br.grad) < 1e-5
    dx2, none, _, _, _ = ops.layernorm_bwd(bf(dy).to(DEV), x.to(DEV), w.to(DEV), mean, rstd, None, None)
    assert none is None and rel_err(dx2, xr.grad) < 6e-3


def test_softmax_colsum_cast_transpose():
    from multimae_amd import _lib, ops
    torch.manual_seed(4)
    rows, n, ld = 500, 99, 104
    S = torch.randn(rows, ld) * 3
    P = torch.full((rows, ld), 7.0, device=DEV)
    ops.softmax_fwd(S.to(DEV), P, rows, n, 0.125)
    ref = (S[:, :n] * 0.125).softmax(-1)
    assert rel_err(P[:, :n], ref) < 2e-6 and float(P[:, n:].abs().max()) == 0.0
    dP = torch.randn(rows, ld)
    dS = torch.full((rows, ld), 7.0, device=DEV)
    ops.softmax_bwd(P, dP.to(DEV), dS, rows, n, 0.125)
    sr = S[:, :n].clone().requires_grad_(True)
    (sr * 0.125).softmax(-1).backward(dP[:, :n])
    assert rel_err(dS[:, :n], sr.grad) < 5e-6 and float(dS[:, n:].abs().max()) == 0.0
    # column sums
    for dt in (torch.float32, torch.bfloat16):
        dy = torch.randn(3001, 264)
        dyd = dy.to(DEV, dt)
        out = torch.ones(264, device=DEV)
        ops.colsum(dyd, out, True)
        assert rel_err(out, 1.0 + dyd.float().cpu().sum(0)) < 1e-5
    # ragged widths (a 7-class head): the one-column-per-lane kernel
    dy = torch.randn(1003, 7).to(DEV)
    out = torch.zeros(7, device=DEV)
    ops.colsum(dy, out, False)
    assert rel_err(out, dy.double().sum(0).cpu()) < 1e-5
    # scattered column sums: short/wide partial blocks (the LayerNorm / dGELU reductions), segments to separate
    # destinations, a dropped segment, accumulate on and off
    for M_, seg, nseg in ((512, 768, 3), (396, 3072, 1), (7, 64, 5), (25344, 256, 2)):
        dy = torch.randn(M_, seg * nseg).to(DEV)
        ref = dy.double().sum(0).cpu()
        for acc in (False, True):
            dsts = [torch.full((seg,), 2.0, device=DEV) if i != 1 or nseg == 1 else None for i in range(nseg)]
            ops.colsum_scatter(dy, seg, dsts, acc)
            for i, d in enumerate(dsts):
                if d is not None:
                    exp = ref[i * seg:(i + 1) * seg] + (2.0 if acc else 0.0)
                    assert float((d.double().cpu() - exp).abs().max()) < 1e-3 * max(1.0, M_ ** 0.5 / 8), (M_, seg, nseg, acc, i)
    # casts
    x = torch.randn(100003)
    assert torch.equal(ops.cast(x.to(DEV), torch.bfloat16).cpu(), x.to(torch.bfloat16))
    assert torch.equal(ops.cast(bf(x).to(DEV), torch.float32).cpu(), bf(x).float())
    w = torch.randn(130, 70)
    wt = torch.empty(70, 130, device=DEV, dtype=torch.bfloat16)
    w_d = w.to(DEV)
    _lib.check(_lib.load().mmae_transpose_cast(w_d.data_ptr(), wt.data_ptr(), 1, 130, 70, ops._stream()), 'transpose')
    assert torch.equal(wt.cpu(), bf(w.t().contiguous()))
    y = torch.randn(1001).to(DEV); z = torch.randn(1001).to(DEV); y0 = y.clone()
    ops.axpy_(y, z, 0.5)
    assert torch.allclose(y, y0 + 0.5 * z)


# ----------------------------------------------------------------------------------------------
def test_mask_sampler_bit_exact_vs_reference_golden():
    """integer path: bit-exact against the reference's own sampler output (3x196 tokens, 98 kept)."""
    from multimae_amd import ops
    g = load_masks_base()
    noise = torch.cat([g['noise_rgb'], g['noise_depth'], g['noise_semseg']], 1)
    m, k, r = ops.mask_sample(g['samples_per_task'], noise.to(DEV), g['noise_all'].to(DEV), [0, 196, 392, 588], 98)
    assert torch.equal(m.cpu(), g['mask_all'])
    assert torch.equal(k.cpu(), g['ids_keep'])
    assert torch.equal(r.cpu(), g['ids_restore'])
    gm = load_mini()
    noise = torch.cat([gm['noise'][d] for d in ('rgb', 'depth', 'semseg')], 1)
    spt = orc.samples_per_task_from_dirichlet(gm['dirichlet'], 12)
    m, k, r = ops.mask_sample(spt, noise.to(DEV), gm['noise_all'].to(DEV), [0, 16, 32, 48], 12)
    assert torch.equal(k.cpu(), gm['ids_keep']) and torch.equal(r.cpu(), gm['ids_restore'])


def test_mask_sampler_properties_full_batch():
    """size-independent properties at the bench batch (B=256): permutation, exact count, keep==visible,
    and equality with the oracle on fresh noise (incl. zero-token tasks and ties)."""
    from multimae_amd import ops
    torch.manual_seed(5)
    B = 256
    dist, tn, an = orc.draw_mask_randoms(B, [196] * 3, 1.0)
    spt = orc.samples_per_task_from_dirichlet(dist, 98)
    spt[0] = torch.tensor([98, 0, 0]); spt[1] = torch.tensor([0, 0, 98]); spt[2] = torch.tensor([40, 40, 40])   # edge rows
    an[3, 10] = an[3, 11]                                                                                       # forced tie
    mo, ko, ro = orc.masks_from_noise(spt, tn, an, 98)
    m, k, r = ops.mask_sample(spt, torch.cat(tn, 1).to(DEV), an.to(DEV), [0, 196, 392, 588], 98)
    assert torch.equal(m.cpu(), mo) and torch.equal(k.cpu(), ko) and torch.equal(r.cpu(), ro)
    assert torch.equal(torch.sort(r.cpu(), 1)[0], torch.arange(588).expand(B, -1))
    assert (m == 0).sum(1).eq(98).all() and torch.gather(m, 1, k).sum() == 0


def test_patch_embed_assemble_vs_oracle():
    """gather-first embedding (patch rows -> K-sliced GEMMs -> assemble) == oracle's embed-all + gather."""
    from multimae_amd import engine
    from helpers import MINI, build_mini_engine, mini_oracle_cfg
    g = load_mini()
    cfg = mini_oracle_cfg()
    model = build_mini_engine()
    model.load_state_dict(g['sd'])
    model.to(DEV)
    from multimae_amd.input_adapters import embed_tokens
    x = {k: v.to(DEV) for k, v in g['x'].items()}
    with engine.precision('fp32'):
        tok = embed_tokens(model, model.input_adapters, x, g['ids_keep'].to(DEV), model.global_tokens)
        assert rel_err(tok, g['enc_in']) < 2e-6
        # stand-alone adapters produce ALL tokens (reference adapter API)
        toks = orc.all_input_tokens(g['x'], g['sd'], cfg)
        for d in MINI['doms']:
            assert rel_err(model.input_adapters[d](x[d]), toks[d]) < 2e-6, d
    with engine.precision('bf16'):
        tok = embed_tokens(model, model.input_adapters, x, g['ids_keep'].to(DEV), model.global_tokens)
        assert rel_err(tok, g['enc_in']) < 1e-2


def _embed_case(name):
    """(tasks, D, n_sel, G, B) of a fused-embedding test geometry; each task: kind, C (channels / class-embedding width), H, W, patch, n_cls."""
    return {
        # the bench geometry: RGB + depth 224^2 patch 16, semseg class ids 56^2 patch 4 with 64-wide class embeddings, ViT-B width
        'cfg3': ([(0, 3, 224, 224, 16, 0), (0, 1, 224, 224, 16, 0), (1, 64, 56, 56, 4, 133)], 768, 98, 1, 5),
        # ViT-L width (the four-column-block kernel), 196 kept tokens: up to four 64-row groups per task
        'vitl': ([(0, 3, 224, 224, 16, 0), (0, 1, 224, 224, 16, 0), (1, 64, 56, 56, 4, 133)], 1024, 196, 1, 3),
        # a width that is no multiple of 256 (waves without a column block), K = 192 (no multiple of the 128-element chunk), two global tokens
        'tiny': ([(0, 3, 64, 64, 8, 0), (1, 16, 32, 32, 4, 7)], 192, 49, 2, 4),
        # image patches whose rows are not 8-pixel multiples (element-wise gather), one task that keeps nothing at all
        'p4': ([(0, 4, 32, 32, 4, 0), (0, 4, 48, 48, 12, 0), (0, 4, 16, 16, 4, 0)], 256, 40, 0, 3),
    }[name]


@pytest.mark.parametrize('case', ['cfg3', 'vitl', 'tiny', 'p4'])
def test_fused_patch_embedding_vs_fp32_formula_and_the_three_pass_path(case):
    """mmae_patch_embed_fwd (ONE kernel: gather of the kept patches, bf16 MFMA, + bias + pos-emb; input_adapters.py:97-119, 215-241 and
    multimae.py:340-347) against (a) the formula in f64 on the bf16-rounded operands -- the only rounding the kernel adds is the f32
    accumulation, tolerance 2e-6 of the row scale -- and (b) patch_rows -> per-task GEMM -> tokens_assemble; the side rows are bit-exact."""
    from multimae_amd import ops
    tasks, D, n_sel, G, B = _embed_case(case)
    gen = torch.Generator().manual_seed(11)
    srcs, ws, bs, poss, offs, k_off = [], [], [], [], [0], 0
    for kind, C, H, W, P, n_cls in tasks:
        n_p = (H // P) * (W // P)
        if kind == 0:
            data, emb = torch.randn(B, C, H, W, generator=gen), None
        else:
            data, emb = torch.randint(-1, n_cls + 1, (B, H, W), generator=gen), torch.randn(n_cls, C, generator=gen)   # -1 and n_cls: ids that embed as zeros
        K = C * P * P
        srcs.append(dict(data=data, emb=emb, kind=kind, C=C, H=H, W=W, ph=P, pw=P, k_off=k_off))
        ws.append(torch.randn(D, K, generator=gen) * K ** -0.5)
        bs.append(torch.randn(D, generator=gen))
        poss.append(torch.randn(n_p, D, generator=gen))
        offs.append(offs[-1] + n_p)
        k_off += K
    Ktot = k_off
    ntot = offs[-1]
    sel = torch.stack([torch.randperm(ntot, generator=gen)[:n_sel] for _ in range(B)])
    if case == 'p4':
        sel = torch.stack([torch.randperm(offs[2], generator=gen)[:n_sel] for _ in range(B)])      # nothing from the third task
    if case == 'vitl':
        sel[0] = torch.randperm(offs[1], generator=gen)[:n_sel]                                    # sample 0: all 196 tokens from RGB
    glob = torch.randn(G, D, generator=gen) if G else None

    # (a) f64 formula on bf16-rounded operands
    ref = torch.zeros(B, n_sel + G, D, dtype=torch.float64)
    rows_ref = torch.zeros(B * n_sel, Ktot)
    for b in range(B):
        for r in range(n_sel):
            idx = int(sel[b, r])
            t = max(i for i in range(len(tasks)) if idx >= offs[i])
            kind, C, H, W, P, n_cls = tasks[t]
            p = idx - offs[t]
            py, px = divmod(p, W // P)
            if kind == 0:
                patch = srcs[t]['data'][b, :, py * P:(py + 1) * P, px * P:(px + 1) * P]
            else:
                ids = srcs[t]['data'][b, py * P:(py + 1) * P, px * P:(px + 1) * P]
                ok = (ids >= 0) & (ids < n_cls)
                patch = (srcs[t]['emb'][ids.clamp(0, n_cls - 1)] * ok[..., None]).permute(2, 0, 1)
            v = bf(patch.reshape(-1))
            rows_ref[b * n_sel + r, srcs[t]['k_off']:srcs[t]['k_off'] + v.numel()] = v
            ref[b, r] = bf(ws[t]).double() @ v.double() + bs[t].double() + poss[t][p].double()
        if G:
            ref[b, n_sel:] = glob.double()

    dsrcs = [dict(s, data=s['data'].to(DEV), emb=None if s['emb'] is None else s['emb'].to(DEV)) for s in srcs]
    assert ops.patch_embed_supported(dsrcs, n_sel, D)
    w16 = [w.to(DEV).to(torch.bfloat16).contiguous() for w in ws]
    dbs, dps = [x.to(DEV) for x in bs], [x.to(DEV) for x in poss]
    dsel, dglob = sel.to(DEV), None if glob is None else glob.to(DEV)
    tok, rows = ops.patch_embed_fwd(dsrcs, w16, dbs, dps, offs, dsel, dglob, B, n_sel, G, D, Ktot)
    assert torch.equal(rows.float().cpu(), rows_ref)
    scale = ref[:, :n_sel].abs().max()
    assert float((tok.double().cpu() - ref).abs().max() / scale) < 2e-6
    if G:
        assert torch.equal(tok[:, n_sel:].cpu(), glob.expand(B, G, D))
    tok2, none = ops.patch_embed_fwd(dsrcs, w16, dbs, dps, offs, dsel, dglob, B, n_sel, G, D, Ktot, want_rows=False)
    assert none is None and torch.equal(tok2, tok)                                               # deterministic, with or without the side rows

    # (b) the three-pass path on the same operands
    rows3 = ops.patch_rows(dsrcs, offs, dsel, B, n_sel, Ktot, torch.bfloat16)
    assert torch.equal(rows3, rows)
    proj = torch.empty((B * n_sel, D), device=DEV, dtype=torch.float32)
    for i, (s, w) in enumerate(zip(srcs, w16)):
        ops.gemm(rows3, w, proj, B * n_sel, D, w.shape[1], lda=Ktot, ldb=w.shape[1], ldc=D, a_off=s['k_off'], accumulate=(i > 0))
    tok3 = ops.tokens_assemble(proj, dbs, dps, offs, dsel, dglob, B, n_sel, G, D)
    assert float((tok3 - tok).abs().max() / scale) < 2e-6


@pytest.mark.parametrize('geom', [(3, 5, 4, 6, 2, 4), (2, 3, 3, 5, 16, 16), (2, 133, 2, 3, 4, 4), (3, 2, 4, 6, 3, 2)])
def test_patchify_roundtrip_and_layout(geom):
    """'b (nh nw) (c ph pw) <-> b c (nh ph) (nw pw)' (output_adapters.py:277-280): the vectorised pw % 4 == 0 kernels and the
    scalar fallback, bit-exact copies; bf16 patch rows with a padded leading dimension."""
    from multimae_amd import ops
    torch.manual_seed(6)
    B, C, nh, nw, ph, pw = geom
    pat = torch.randn(B * nh * nw, C * ph * pw)
    img = ops.unpatchify(pat.to(DEV), B, C, nh, nw, ph, pw)
    ref = pat.reshape(B, nh, nw, C, ph, pw).permute(0, 3, 1, 4, 2, 5).reshape(B, C, nh * ph, nw * pw)
    assert torch.equal(img.cpu(), ref)
    back = ops.patchify(img, C, nh, nw, ph, pw, torch.float32)
    assert torch.equal(back.cpu(), pat)
    back16 = ops.patchify(img, C, nh, nw, ph, pw, torch.bfloat16)
    assert back16.shape == pat.shape and torch.equal(back16.float().cpu(), bf(pat))


@pytest.mark.parametrize('S,P', [(32, 8), (24, 6)])
@pytest.mark.parametrize('kind,norm_pix', [(0, False), (0, True), (1, False)])
def test_masked_pixel_losses(kind, norm_pix, S, P):
    """patch 8 takes the float4 backward kernel, patch 6 the scalar one; 16 patches either way."""
    import multimae_amd as M
    torch.manual_seed(7)
    B, C = 5, 3
    pred, tgt = torch.randn(B, C, S, S), torch.randn(B, C, S, S)
    mask = (torch.rand(B, 16) < 0.7).long()
    mask[1] = 0                                           # a sample with no masked token drops out (nanmean)
    pr = pred.clone().requires_grad_(True)
    ref = (orc.masked_mse if kind == 0 else orc.masked_l1)(pr, tgt, mask, P, 1, norm_pix=norm_pix)
    (ref * 1.7).backward()
    fn = (M.MaskedMSELoss if kind == 0 else M.MaskedL1Loss)(P, 1, norm_pix=norm_pix)
    pd = pred.to(DEV).requires_grad_(True)
    out = fn(pd, tgt.to(DEV), mask=mask.to(DEV))
    (out * 1.7).backward()
    assert abs(float(out) - float(ref)) < 2e-6 * max(1.0, abs(float(ref)))
    # sample 1 has no masked token: the reference's autograd yields NaN there (0 * inf through sum/count);
    # the engine yields exact zeros.  Compare the other samples.
    keep = [0, 2, 3, 4]
    assert torch.isnan(pr.grad[1]).all() and float(pd.grad[1].abs().max()) == 0.0
    assert rel_err(pd.grad[keep], pr.grad[keep]) < 1e-5
    # mask=None == plain mean; all-zero mask == 0 with zero grad
    assert abs(float(fn(pred.to(DEV), tgt.to(DEV))) - float((orc.masked_mse if kind == 0 else orc.masked_l1)(pred, tgt, None, P, 1, norm_pix=norm_pix))) < 5e-6
    z = fn(pred.to(DEV).requires_grad_(True), tgt.to(DEV), mask=torch.zeros(B, 16, dtype=torch.long, device=DEV))
    assert float(z) == 0.0


@pytest.mark.parametrize('S,patch', [(8, 8), (16, 16)])
def test_masked_cross_entropy(S, patch):
    """semseg map S x S at stride 4: effective patch 2 (scalar backward kernel) and 4 (the float4 one, as at 224^2 / 16)."""
    import multimae_amd as M
    torch.manual_seed(8)
    B, C = 4, 133
    logits, tgt = torch.randn(B, C, S, S) * 2, torch.randint(0, C, (B, S, S))
    mask = (torch.rand(B, 16) < 0.6).long()
    mask[2] = 0
    lr = logits.clone().requires_grad_(True)
    ref = orc.masked_ce(lr, tgt, mask, patch, 4)
    ref.backward()
    ld = logits.to(DEV).requires_grad_(True)
    out = M.MaskedCrossEntropyLoss(patch, 4)(ld, tgt.to(DEV), mask=mask.to(DEV))
    out.backward()
    assert abs(float(out) - float(ref)) < 5e-6
    keep = [0, 1, 3]                                    # sample 2: no masked token (reference grad is NaN, engine 0)
    assert float(ld.grad[2].abs().max()) == 0.0
    assert rel_err(ld.grad[keep], lr.grad[keep]) < 1e-5


def test_truncated_depth_standardize_vs_reference_golden():
    """run_pretraining_multimae.py:487-492 (fixture produced by executing those reference lines, tests/golden/make_golden_depth.py):
    continuous maps, 8-bit quantised maps whose 10 % / 90 % cuts fall inside runs of equal values, a constant map (var = 0),
    and the real 224 x 224 geometry.  fp tolerance 2e-6 relative to the output range (the reference sums in fp32)."""
    import numpy as np, os
    import multimae_amd as M
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'depth_std.npz'))
    for k in [f[2:] for f in z.files if f.startswith('x/')]:
        x, y = torch.from_numpy(z['x/' + k]), torch.from_numpy(z['y/' + k])
        out = M.truncated_depth_standardize(x.to(DEV)).cpu()
        scale = float(y.abs().max())
        assert float((out - y).abs().max()) <= 2e-6 * scale + 1e-6, (k, float((out - y).abs().max()), scale)
        assert torch.equal(orc.truncated_depth_standardize(x), y) or float((orc.truncated_depth_standardize(x) - y).abs().max()) < 1e-6 * scale
    # size-independent property at the full batch geometry: an affine map of the depth leaves the result unchanged
    torch.manual_seed(12)
    d = (torch.rand(16, 1, 224, 224, device=DEV) ** 2) * 50
    a, b = M.truncated_depth_standardize(d), M.truncated_depth_standardize(d * 3.0 + 7.0)
    assert float((a - b).abs().max()) < 2e-3


def test_adamw_and_sumsq():
    from multimae_amd import ops
    torch.manual_seed(9)
    n = 100003
    p, g = torch.randn(n), torch.randn(n)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd = p.to(DEV), m.to(DEV), v.to(DEV)
    shadow = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    P, Mo, V = {'p': p.clone()}, {'p': m.clone()}, {'p': v.clone()}
    for step in (1, 2, 3):
        g = torch.randn(n)
        ops.adamw(pd, g.to(DEV), md, vd, lr=1e-3, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.05, step=step, shadow=shadow)
        orc.adamw_step(P, {'p': g}, Mo, V, step, 1e-3, 0.05)
    assert rel_err(pd, P['p']) < 1e-6 and rel_err(vd, V['p']) < 1e-6
    assert torch.equal(shadow.cpu(), pd.cpu().to(torch.bfloat16))
    out, ws = torch.zeros(1, device=DEV), torch.empty(1024, device=DEV)
    ops.sumsq(g.to(DEV), out, ws)
    assert abs(float(out) - float((g.double() ** 2).sum())) / float((g.double() ** 2).sum()) < 1e-6


@pytest.mark.parametrize('shape', [(4096, 512, 264), (25344, 768, 768), (5000, 1024, 256)])
@pytest.mark.parametrize('acc', [False, True])
def test_linear_dw_fused_bias_grad(shape, acc):
    """dW product with the bias gradient taken from the dy tiles inside the GEMM (mmae_gemm_desc.a_colsum): the column sums
    are exact bf16 -> f32 additions, so vs fp64 only f32 summation error remains (rel 2e-6); also split-K + accumulate."""
    from multimae_amd import _lib, ops
    import ctypes
    Mr, N, K = shape
    torch.manual_seed(3)
    dy, x = bf(torch.randn(Mr, N) * 0.1 + 0.03).float(), bf(torch.randn(Mr, K)).float()
    dyd, xd = dy.to(DEV, torch.bfloat16), x.to(DEV, torch.bfloat16)
    dw = torch.full((N, K), 1.5 if acc else float('nan'), device=DEV)
    db = torch.full((N,), 0.25 if acc else float('nan'), device=DEV)
    # (the planner must actually put these shapes on the kernel that implements the fused sum)
    d = _lib.GemmDesc(); d.M, d.N, d.K, d.ab_dtype, d.c_dtype, d.a_trans, d.b_trans, d.batch, d.batch_inner, d.alpha = N, K, Mr, 1, 0, 1, 1, 1, 1, 1.0
    d.ldc = K
    t_, s_ = ctypes.c_int(0), ctypes.c_int(0)
    _lib.load().mmae_gemm_plan(ctypes.byref(d), ctypes.byref(t_), ctypes.byref(s_))
    if t_.value != 9:
        pytest.skip('MMAE_GEMM_TILE override active')
    ops.linear_dw(dyd, xd, dw, acc, db=db, db_accumulate=acc)
    base_w, base_b = (1.5, 0.25) if acc else (0.0, 0.0)
    assert rel_err(dw, base_w + dy.double().t() @ x.double()) < 2e-6
    assert rel_err(db, base_b + dy.double().sum(0)) < 2e-6


@pytest.mark.parametrize('acc', [False, True])
@pytest.mark.parametrize('case', ['block', 'ragged', 'single_slice'])
def test_gemm_dw_group_vs_fp32_reference(case, acc):
    """mmae_gemm_dw_group: several dW = dY^T X products (+ bias column sums) in one grid.  'block': the four weight gradients of
    a small transformer block geometry over 5 000 rows (K not a multiple of the 32-row tile; 3 K slices forced); 'ragged': widths
    that are not multiples of the 256 tile (partial tiles in both directions) with the automatic slice count; 'single_slice':
    enough tiles to fill the chip without splitting."""
    from multimae_amd import ops
    g = torch.Generator().manual_seed(7)
    if case == 'block':
        rows, shapes, split = 5000, [(256, 1024, False), (1024, 256, False), (256, 256, True), (768, 256, True)], 3
    elif case == 'ragged':
        rows, shapes, split = 3001, [(264, 136, True), (520, 72, False), (8, 264, True)], 0
    else:
        rows, shapes, split = 700, [(2048, 1024, True), (1024, 2048, False), (2304, 768, True), (768, 3072, False), (3072, 768, True)], 0
    probs, refs = [], []
    for n_out, k_in, want_db in shapes:
        dy = (torch.randn(rows, n_out, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
        x = torch.randn(rows, k_in, generator=g).to(DEV).to(torch.bfloat16)
        dw0 = torch.randn(n_out, k_in, generator=g).to(DEV)
        db0 = torch.randn(n_out, generator=g).to(DEV) if want_db else None
        dw, db = dw0.clone(), (db0.clone() if want_db else None)
        probs.append((dy, x, dw, db))
        rw = dy.float().t() @ x.float() + (dw0 if acc else 0)
        rb = (dy.float().sum(0) + (db0 if acc else 0)) if want_db else None
        refs.append((rw, rb))
    ops.gemm_dw_group(probs, acc, split)
    torch.cuda.synchronize()
    for (dy, x, dw, db), (rw, rb) in zip(probs, refs):
        assert rel_err(dw, rw) < 2e-5, (case, dw.shape, rel_err(dw, rw))
        if rb is not None:
            assert rel_err(db, rb) < 2e-5, (case, rel_err(db, rb))
    # determinism: fixed summation order
    probs2 = [(dy, x, (torch.zeros_like(dw) if not acc else dw.clone()), (None if db is None else db.clone())) for dy, x, dw, db in probs]
    if not acc:
        ops.gemm_dw_group(probs2, False, split)
        torch.cuda.synchronize()
        for (_, _, a, _), (_, _, b, _) in zip(probs, probs2):
            assert torch.equal(a, b)


@pytest.mark.parametrize('act', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', [('mse', 3, 32, 8), ('mse_norm', 3, 32, 8), ('l1', 1, 24, 4), ('mse', 3, 12, 2), ('ce', 133, 8, 2), ('ce', 133, 16, 4), ('ce', 7, 16, 8)])
def test_patch_domain_losses_equal_image_domain(case, act):
    """mmae_masked_*_pat_fwd / _bwd: the masked losses evaluated on patch rows (what out_proj produces) equal the image-domain
    kernels on the rearranged image -- same value, and the gradient rows equal patchify(image-domain gradient) (in the
    adapter's activation dtype, pad columns zero).  Includes a sample without masked tokens and non-multiple-of-8 row widths."""
    from multimae_amd import ops
    from multimae_amd.functions import MaskedCEFn, MaskedCEPatFn, MaskedPixelLossFn, MaskedPixelLossPatFn, PatHandle
    kind_name, C, S, P = case
    B, nh = 3, S // P
    g = torch.Generator().manual_seed(5)
    img = torch.randn(B, C, S, S, generator=g).to(DEV).requires_grad_(True)
    mask = (torch.rand(B, nh * nh, generator=g) < 0.6).long().to(DEV)
    mask[1] = 0                                                   # a sample with no masked token
    if kind_name == 'ce':
        target = torch.randint(0, C, (B, S, S), generator=g).to(DEV)
        ref = MaskedCEFn.apply(img, target, mask, P)
    else:
        target = torch.randn(B, C, S, S, generator=g).to(DEV)
        kind, norm = (1 if kind_name == 'l1' else 0), kind_name.endswith('norm')
        ref = MaskedPixelLossFn.apply(img, target, mask, kind, norm, P)
    ref.backward()
    d_img = img.grad
    pat = ops.patchify(img.detach(), C, nh, nh, P, P, torch.float32).contiguous()
    h = PatHandle(pat, C, nh, nh, P, P, act)
    h.token = torch.zeros(1, device=DEV, requires_grad=True)
    if kind_name == 'ce':
        out = MaskedCEPatFn.apply(h.token, h, target, mask, P)
    else:
        out = MaskedPixelLossPatFn.apply(h.token, h, target, mask, kind, norm, P)
    assert abs(float(out) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), (float(out), float(ref))
    out.backward()
    torch.cuda.synchronize()
    KP = C * P * P
    d_ref = ops.patchify(d_img, C, nh, nh, P, P, torch.float32)
    assert h.d_pat.dtype == act and h.d_pat.shape == (B * nh * nh, (KP + 7) // 8 * 8)
    got = h.d_pat.float()
    assert rel_err(got[:, :KP], d_ref) < (1e-6 if act == torch.float32 else 4e-3), rel_err(got[:, :KP], d_ref)
    assert float(got[:, KP:].abs().sum()) == 0.0
    assert float(got.view(B, nh * nh, -1)[1].abs().sum()) == 0.0  # the sample without masked tokens: zero gradient


@pytest.mark.parametrize('tile', [9, 10])
@pytest.mark.parametrize('M', [2500, 4160])
def test_pingpong_dgelu_colsum_epilogue(tile, M):
    """dX product with the dGELU epilogue and its fused column sums on the ping-pong kernel (256- and 320-row tiles, the
    compile-time-flavoured instantiations): per-32-row-block partials [ceil(M/32)][N], M not a multiple of 32 / of the tile."""
    from multimae_amd import ops
    from multimae_amd._lib import EPI_DGELU
    g = torch.Generator().manual_seed(3)
    Nn, K = 520, 264                                    # out[M, Nn] = dy[M, K] @ w[K, Nn]
    dy = bf(torch.randn(M, K, generator=g) * 0.3)
    w = bf(torch.randn(K, Nn, generator=g) * 0.2)
    pre = bf(torch.randn(M, Nn, generator=g))
    out = torch.empty(M, Nn, device=DEV, dtype=torch.bfloat16)
    part = torch.full(ops.dx_colsum_part_shape(M, Nn), float('nan'), device=DEV)
    ops.gemm(dy.to(DEV), w.to(DEV), out, M, Nn, K, lda=K, ldb=Nn, ldc=Nn, b_trans=True, aux=pre.to(DEV), ldaux=Nn, epi=EPI_DGELU, tile=tile,
             colsum_part=part)
    torch.cuda.synchronize()
    p = pre.float().clone().requires_grad_(True)
    orc.gelu_erf(p).backward(dy.float() @ w.float())
    assert rel_err(out.float(), p.grad) < 4e-3
    assert part.shape[0] == (M + 31) // 32 and not torch.isnan(part).any()
    ref_blocks = torch.stack([p.grad[i:i + 32].sum(0) for i in range(0, M, 32)])
    assert rel_err(part.cpu(), ref_blocks) < 1e-4


@pytest.mark.parametrize('tile,M', [(9, 2100), (10, 2600)])
def test_pingpong_gelu_derivative_pair(tile, M):
    """The MLP pair of the composite calls on the ping-pong flavours: fc1 forward with EPI_GELU_G (C = gelu(pre), aux = gelu'(pre) -- the
    derivative taken where the erf terms are in registers) and fc2's dX with EPI_MUL + column sums (out = (dy @ W) * aux): against
    the fp32 formulas, and the forward output bit-identical to the EPI_GELU flavour's."""
    from multimae_amd import ops
    from multimae_amd._lib import EPI_GELU, EPI_GELU_G, EPI_MUL
    g = torch.Generator().manual_seed(5)
    K, Nn = 264, 520
    x, w, b = bf(torch.randn(M, K, generator=g) * 0.5), bf(torch.randn(Nn, K, generator=g) * 0.2), torch.randn(Nn, generator=g) * 0.1
    h0, a0 = (torch.empty(M, Nn, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    h1, a1 = (torch.empty(M, Nn, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), h0, aux=a0, epi=EPI_GELU, tile=tile)
    ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), h1, aux=a1, epi=EPI_GELU_G, tile=tile)
    assert torch.equal(h0, h1)
    pre = (x.float() @ w.float().t() + b).clone().requires_grad_(True)
    orc.gelu_erf(pre).sum().backward()
    assert rel_err(a1.float(), pre.grad) < 4e-3 and rel_err(a0.float(), pre.detach()) < 4e-3
    dy = bf(torch.randn(M, K, generator=g) * 0.3)           # gradient of the fc2 output; W2 [K, Nn]
    w2 = bf(torch.randn(K, Nn, generator=g) * 0.2)
    out = torch.empty(M, Nn, device=DEV, dtype=torch.bfloat16)
    part = torch.full(ops.dx_colsum_part_shape(M, Nn), float('nan'), device=DEV)
    ops.gemm(dy.to(DEV), w2.to(DEV), out, M, Nn, K, lda=K, ldb=Nn, ldc=Nn, b_trans=True, aux=a1, ldaux=Nn, epi=EPI_MUL, tile=tile,
             colsum_part=part)
    want = (dy.float() @ w2.float()) * a1.float().cpu()
    assert rel_err(out.float(), want) < 4e-3
    ref_blocks = torch.stack([want[i:i + 32].sum(0) for i in range(0, M, 32)])
    e_part = rel_err(part.cpu(), ref_blocks)
    assert not torch.isnan(part).any() and e_part < 1e-4, (e_part, int(torch.isnan(part).sum()), tuple(part.shape), tuple(ref_blocks.shape))


@pytest.mark.parametrize('domain', ['image', 'patch'])
@pytest.mark.parametrize('eps', [0.1, 0.3])
def test_masked_cross_entropy_label_smoothing(eps, domain):
    """label_smoothing > 0 (criterion.py:47, F.cross_entropy semantics) in the image-domain and the patch-domain CE kernels
    against the oracle (whose formula is checked against F.cross_entropy on the CPU side)."""
    import multimae_amd as M
    from multimae_amd import ops
    from multimae_amd.functions import MaskedCEPatFn, PatHandle
    torch.manual_seed(9)
    B, C, S, P = 3, 133, 16, 4
    logits, tgt = torch.randn(B, C, S, S) * 2, torch.randint(0, C, (B, S, S))
    mask = (torch.rand(B, 16) < 0.6).long()
    lr = logits.clone().requires_grad_(True)
    ref = orc.masked_ce(lr, tgt, mask, 16, 4, label_smoothing=eps)
    ref.backward()
    if domain == 'image':
        ld = logits.to(DEV).requires_grad_(True)
        out = M.MaskedCrossEntropyLoss(16, 4, label_smoothing=eps)(ld, tgt.to(DEV), mask=mask.to(DEV))
        out.backward()
        got = ld.grad
    else:
        pat = ops.patchify(logits.to(DEV), C, 4, 4, P, P, torch.float32).contiguous()
        h = PatHandle(pat, C, 4, 4, P, P, torch.float32)
        h.token = torch.zeros(1, device=DEV, requires_grad=True)
        out = MaskedCEPatFn.apply(h.token, h, tgt.to(DEV), mask.to(DEV), P, eps)
        out.backward()
        got = ops.unpatchify(h.d_pat[:, :C * P * P].contiguous(), B, C, 4, 4, P, P)
    assert abs(float(out) - float(ref)) < 5e-6
    assert rel_err(got, lr.grad) < 1e-5


@pytest.mark.parametrize('tile', [11, 12, 13, 14])
@pytest.mark.parametrize('shape', [(2504, 520, 256), (1000, 768, 768), (4104, 1032, 128)])
def test_round3_gemm_structures_vs_fp32_reference(tile, shape):
    """The GEMM structures added in round 3 -- 11 / 12: "duo" (two independent 4-wave workgroups per CU, 128 x 256 tiles, and its
    8-wave form), 13 / 14: the ping-pong kernel on 64-wide K tiles (whole-cache-line LDS-DMA pieces; 14 is what the planner picks
    for plain bf16 epilogues) -- with every epilogue flavour they carry, on ragged shapes (M, N not multiples of the tile; rows
    past the edge masked by the out-of-range DMA offset), against an fp32 torch reference of the same bf16 operands."""
    from multimae_amd import _lib, ops
    from multimae_amd._lib import EPI_DGELU, EPI_GELU
    if not hasattr(_lib.load(), 'mmae_gemm_duo_occupancy'):
        pytest.skip('experiment structures are compiled only into -DMMAE_EXPERIMENTS builds of the library (VERDICT r3 hygiene); '
                    'the production library maps tile codes 11-14 to the ping-pong kernel')
    M, N, K = shape
    torch.manual_seed(M + tile)
    x, w = bf(torch.randn(M, K) * 0.5).float(), bf(torch.randn(N, K) * 0.2).float()
    bias, resid = torch.randn(N), torch.randn(M, N)
    xd, wd = x.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16)
    lin = x @ w.t() + bias
    # forward flavours: bias -> bf16, bias + GELU -> 2 x bf16, bias + residual -> f32, bias -> f32
    c16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.linear_fwd(xd, wd, bias.to(DEV), c16, tile=tile)
    assert rel_err(c16.float(), lin) < 4e-3
    aux, act = torch.empty_like(c16), torch.empty_like(c16)
    ops.linear_fwd(xd, wd, bias.to(DEV), act, aux=aux, epi=EPI_GELU, tile=tile)
    assert rel_err(aux.float(), lin) < 4e-3 and rel_err(act.float(), orc.gelu_erf(lin)) < 4e-3
    out = torch.empty(M, N, device=DEV)
    ops.linear_fwd(xd, wd, bias.to(DEV), out, resid=resid.to(DEV), tile=tile)
    assert rel_err(out, lin + resid) < 1e-5
    ops.linear_fwd(xd, wd, bias.to(DEV), out, tile=tile)
    assert rel_err(out, lin) < 1e-5
    # dX flavours (W read through the transposing LDS path): plain bf16, dGELU, dGELU + column sums, f32
    dy = bf(torch.randn(M, N) * 0.3).float()
    dyd = dy.to(DEV, torch.bfloat16)
    dx_ref = dy @ w
    dxo = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
    ops.linear_dx(dyd, wd, dxo, tile=tile)
    assert rel_err(dxo.float(), dx_ref) < 4e-3
    pre = bf(torch.randn(M, K)).float()
    p = pre.clone().requires_grad_(True)
    orc.gelu_erf(p).backward(dx_ref)
    ops.linear_dx(dyd, wd, dxo, aux=pre.to(DEV, torch.bfloat16), epi=EPI_DGELU, tile=tile)
    assert rel_err(dxo.float(), p.grad) < 4e-3
    cs = torch.empty(K, device=DEV)
    ops.linear_dx(dyd, wd, dxo, aux=pre.to(DEV, torch.bfloat16), epi=EPI_DGELU, colsum_out=cs, tile=tile)
    assert rel_err(dxo.float(), p.grad) < 4e-3 and rel_err(cs, p.grad.sum(0)) < 1e-4
    dx32 = torch.empty(M, K, device=DEV)
    ops.gemm(dyd, wd, dx32, M, K, N, lda=N, ldb=K, ldc=K, b_trans=True, tile=tile)
    assert rel_err(dx32, dx_ref) < 1e-5


def test_block_backward_with_mixed_bound_and_unbound_grads_after_composite_forward():
    """a composite bf16 forward leaves GELU'(pre-activation) in `hpre`.  With one .grad view missing the block's
    backward must still multiply by it (fresh tensors for all gradients through the composite call), not run the per-op path that
    would differentiate the derivative again.  Reference: the same block with every gradient handed to autograd."""
    from multimae_amd import functions as F, ops
    g = torch.Generator().manual_seed(11)
    B, N, D, heads, Hd = 2, 50, 128, 2, 512
    shapes = [(D,), (D,), (3 * D, D), (3 * D,), (D, D), (D,), (D,), (D,), (Hd, D), (Hd,), (D, Hd), (D,)]

    def make():
        gg = torch.Generator().manual_seed(12)
        return [torch.nn.Parameter((torch.randn(s, generator=gg) * (0.05 if len(s) == 2 else 0.1) + (1.0 if i in (0, 6) else 0.0)).to(DEV))
                for i, s in enumerate(shapes)]
    x = (torch.randn(B * N, D, generator=g) * 0.7).to(DEV)
    dy = (torch.randn(B * N, D, generator=g) * 0.3).to(DEV)
    act = torch.bfloat16
    wc = lambda w: w.detach().to(act)
    outs = []
    for mixed in (False, True):
        P = make()
        if mixed:                                           # bound gradients for all parameters but one
            for i, p in enumerate(P):
                if i != 8:
                    p.grad = torch.zeros_like(p)
        y, saved = F.block_fwd(x, P, wc, heads, 1e-6, act, B, N, True)
        assert len(saved[5]) > 2 and saved[5][2], 'composite bf16 forward must mark hpre as the stored derivative'
        sink = F.GradSink(direct=mixed)
        dx0, _, _, grads = F.block_bwd(dy, ops.cast(dy, act), False, saved, P, wc, sink, heads, act, B, N)
        torch.cuda.synchronize()
        outs.append((dx0.float().cpu(), [None if gr is None else gr.float().cpu() for gr in grads]))
    (dx_a, g_a), (dx_b, g_b) = outs
    assert all(gr is not None for gr in g_b), 'a mix of bound / unbound gradients hands fresh tensors back for all of them'
    assert rel_err(dx_b, dx_a) < 1e-6
    for ga, gb in zip(g_a, g_b):
        assert rel_err(gb, ga) < 1e-6
    # and the per-op fall-through itself (composites off in backward) multiplies by the stored derivative
    P = make()
    y, saved = F.block_fwd(x, P, wc, heads, 1e-6, act, B, N, True)
    ops.set_composite_blocks(False)
    try:
        dx_c, _, _, g_c = F.block_bwd(dy, ops.cast(dy, act), False, saved, P, wc, F.GradSink(direct=False), heads, act, B, N)
    finally:
        ops.set_composite_blocks(True)
    torch.cuda.synchronize()
    assert rel_err(dx_c.float().cpu(), dx_a) < 2e-2
    assert rel_err(g_c[8].float().cpu(), g_a[8]) < 2e-2     # fc1.weight: wrong by O(1) if gelu' were applied twice


@pytest.mark.parametrize('acc', [False, True])
def test_colsum_batch_one_launch_vs_fp64_and_bit_identical_across_streams(acc):
    """mmae_colsum_batch (round 4): the column sums of a transformer block's backward -- two LayerNorm partial blocks [nblk][3 D], the
    dGELU partials [ceil(R / 32)][Hd], an f32 and a bf16 activation matrix with a ragged width (133 classes) -- in ONE launch with a
    last-workgroup reduction per 256-column group.  Against fp64 sums; repeated launches and launches from two streams give
    bit-identical results (the reduction order does not depend on which workgroup arrives last; tickets reset themselves)."""
    from multimae_amd import ops
    g = torch.Generator().manual_seed(21)
    D, Hd = 768, 3072
    srcs = [torch.randn(1024, 3 * D, generator=g), torch.randn(792, Hd, generator=g), torch.randn(1024, 3 * D, generator=g),
            torch.randn(5000, 133, generator=g), bf(torch.randn(3000, 264, generator=g)), torch.randn(37, 20, generator=g)]
    segs = [D, Hd, D, 133, 264, 8]
    nseg = [3, 1, 3, 1, 1, 3]
    base = [[torch.randn(w, generator=g) for _ in range(n)] for w, n in zip(segs, nseg)]
    drop = (0, 2)                                           # job 0 drops its third segment (a NULL destination)

    def run(stream=None):
        dsts = [[t.clone().to(DEV) for t in row] for row in base]
        jobs = []
        for i, (s, w, d) in enumerate(zip(srcs, segs, dsts)):
            dl = [None if (i, k) == drop else t for k, t in enumerate(d)]
            jobs.append((s.to(DEV), w, dl))
        if stream is None:
            ops.colsum_batch(jobs, acc)
        else:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                ops.colsum_batch(jobs, acc)
            torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        return [[t.cpu() for t in row] for row in dsts]
    got = run()
    for i, (s, w, n) in enumerate(zip(srcs, segs, nseg)):
        ref = s.double().sum(0)
        for k in range(n):
            seg = ref[k * w:(k + 1) * w]
            want = base[i][k].double()[:seg.numel()] * (1.0 if acc else 0.0) + seg
            if (i, k) == drop:
                assert torch.equal(got[i][k], base[i][k])               # untouched
                continue
            assert rel_err(got[i][k][:seg.numel()].double(), want) < 2e-6, (i, k)
    again = run()
    other = run(torch.cuda.Stream())
    for a_, b_, c_ in zip(got, again, other):
        for x_, y_, z_ in zip(a_, b_, c_):
            assert torch.equal(x_, y_) and torch.equal(x_, z_)


@pytest.mark.parametrize('a_trans,b_trans', [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize('shape', [(304, 200, 136), (520, 264, 1000), (256, 2128, 256)])
def test_gemm_f32f16_single_product(shape, a_trans, b_trans):
    """MMAE_F32F16 (round 4): f32 operands rounded to fp16 (11-bit significand = TF32's) for ONE MFMA product, fp32 accumulation.
    Exactly the fp64 product of the fp16-rounded operands (to accumulation order), and TF32-class against the exact product."""
    from multimae_amd import ops
    from multimae_amd._lib import F32F16
    M, N, K = shape
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn((K, M) if a_trans else (M, K), generator=g)
    Bm = torch.randn((K, N) if b_trans else (N, K), generator=g) * 0.05
    C = torch.empty(M, N, device=DEV)
    ops.gemm(A.to(DEV), Bm.to(DEV), C, M, N, K, lda=A.shape[1], ldb=Bm.shape[1], ldc=N, a_trans=a_trans, b_trans=b_trans, f32_as=F32F16)
    a = (A.t() if a_trans else A).double()
    b = (Bm if b_trans else Bm.t()).double()
    a16 = (A.t() if a_trans else A).half().double()
    b16 = (Bm if b_trans else Bm.t()).half().double()
    assert rel_err(C.cpu().double(), a16 @ b16) < 2e-6
    e = rel_err(C.cpu().double(), a @ b)
    assert 1e-5 < e < 6e-4, e                              # 2^-11 per operand, averaged over the contraction


def test_gemm_f32f16_gradient_operand_prescale_and_loss_amax():
    """A gradient operand (~1e-7: 1 / (masked pixels x batch) at the bench geometry) would fall into fp16's subnormals; with
    mmae_gemm_desc.a_amax it is multiplied by 2^-floor(log2 amax) first and the accumulators back afterwards -- TF32-class again.  The
    amax is what the patch-domain loss backward kernels leave behind (atomic max over their d_pat)."""
    from multimae_amd import ops
    from multimae_amd._lib import F32F16
    from multimae_amd.functions import MaskedCEPatFn, PatHandle
    g = torch.Generator().manual_seed(9)
    M, N, K = 520, 264, 392
    dy = torch.randn(M, K, generator=g) * 3e-8
    dy[5, 7] = 4.1e-7                                      # the largest element
    w = torch.randn(K, N, generator=g) * 0.05             # dX product: out[M, N] = dy[M, K] @ w[K, N]
    amax = dy.abs().max().reshape(1).to(DEV)
    ref = dy.double() @ w.double()
    out = torch.empty(M, N, device=DEV)
    ops.gemm(dy.to(DEV), w.to(DEV), out, M, N, K, lda=K, ldb=N, ldc=N, b_trans=True, f32_as=F32F16, a_amax=amax)
    e_scaled = rel_err(out.cpu().double(), ref)
    ops.gemm(dy.to(DEV), w.to(DEV), out, M, N, K, lda=K, ldb=N, ldc=N, b_trans=True, f32_as=F32F16)
    e_raw = rel_err(out.cpu().double(), ref)
    assert e_scaled < 6e-4 and e_raw > 20 * e_scaled, (e_scaled, e_raw)
    # weight-gradient shape through the split-K path: dW[N, K2] = dy^T x
    x = torch.randn(M, 136, generator=g)
    dw = torch.empty(K, 136, device=DEV)
    ops.gemm(dy.to(DEV), x.to(DEV), dw, K, 136, M, lda=K, ldb=136, ldc=136, a_trans=True, b_trans=True, f32_as=F32F16, a_amax=amax)
    assert rel_err(dw.cpu().double(), dy.double().t() @ x.double()) < 6e-4
    # the amax the cross-entropy backward kernel reports == max |d_pat|
    B, C, P, nh = 3, 7, 4, 2
    lr = torch.randn(B, C, nh * P, nh * P, generator=g)
    tgt = torch.randint(0, C, (B, nh * P, nh * P), generator=g)
    mask = torch.tensor([[1, 0, 1, 1], [0, 0, 0, 1], [1, 1, 1, 1]])
    h = PatHandle(ops.patchify(lr.to(DEV), C, nh, nh, P, P, torch.float32).contiguous(), C, nh, nh, P, P, torch.float32)
    h.dy_amax = torch.zeros(1, device=DEV)
    h.token = torch.zeros(1, device=DEV, requires_grad=True)
    MaskedCEPatFn.apply(h.token, h, tgt.to(DEV), mask.to(DEV), P, 0.0).backward()
    torch.cuda.synchronize()
    assert float(h.dy_amax) == float(h.d_pat.abs().max()) > 0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('resid', [True, False])
@pytest.mark.parametrize('M,K', [(1000, 256), (2560, 1024), (50176, 256)])
def test_gemm_layernorm_side_output(dtype, resid, M, K):
    """mmae_gemm_desc.ln_out (round 5): the f32 bias [+ residual] epilogue of an N = 256 product also writes LayerNorm(C) -- and the row
    statistics -- or the plain 16-bit cast of C, in the operands' format (bf16 / fp16 storage).  Against the product followed by the
    stand-alone LayerNorm kernel (the path it replaces) and against the fp64 formula; C itself must not change."""
    from multimae_amd import ops
    g = torch.Generator().manual_seed(7)
    N = 256
    x = (torch.randn(M, K, generator=g) * 0.5).to(dtype)
    w = (torch.randn(N, K, generator=g) * 0.1).to(dtype)
    bias = torch.randn(N, generator=g) * 0.3
    r = torch.randn(M, N, generator=g) + 0.7 if resid else None          # (a residual stream with a mean)
    gam, bet = 1 + 0.2 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    xd, wd, bd = x.to(DEV), w.to(DEV), bias.to(DEV)
    rd = r.to(DEV) if resid else None
    c0 = torch.empty(M, N, device=DEV)
    ops.linear_fwd(xd, wd, bd, c0, resid=rd, tile=9)
    y0, m0, s0 = ops.layernorm_fwd(c0, gam.to(DEV), bet.to(DEV), 1e-6, dtype)
    c1 = torch.empty(M, N, device=DEV)
    y1 = torch.full((M, N), float('nan'), device=DEV, dtype=dtype)
    m1, s1 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.linear_fwd(xd, wd, bd, c1, resid=rd, ln=(gam.to(DEV), bet.to(DEV), y1, m1, s1, 1e-6))
    torch.cuda.synchronize()
    assert torch.equal(c0, c1)                                            # the f32 output is the same kernel's
    ref = torch.nn.functional.layer_norm(c0.double().cpu(), (N,), gam.double(), bet.double(), 1e-6)
    tol = 4e-3 if dtype == torch.bfloat16 else 6e-4                        # output rounding: 2^-9 / 2^-12
    assert not torch.isnan(y1.float()).any()
    assert rel_err(y1.float(), ref) < tol and rel_err(y0.float(), ref) < tol
    assert rel_err(y1.float(), y0.float()) < tol                          # one-pass (E[x^2] - mean^2) against the two-pass kernel
    assert rel_err(m1, m0) < 1e-5 and rel_err(s1, s0) < 1e-4
    # cast-only form (gamma NULL): the 16-bit copy the next Linear reads
    y2 = torch.full((M, N), float('nan'), device=DEV, dtype=dtype)
    c2 = torch.empty(M, N, device=DEV)
    ops.linear_fwd(xd, wd, bd, c2, resid=rd, ln=(None, None, y2, None, None, 0.0))
    torch.cuda.synchronize()
    assert torch.equal(c2, c0) and torch.equal(y2, c0.to(dtype))


def test_gemm_layernorm_side_output_refuses_what_it_cannot_do():
    from multimae_amd import ops
    from multimae_amd._lib import KernelError
    x, w = torch.randn(512, 256, device=DEV).bfloat16(), torch.randn(512, 256, device=DEV).bfloat16()      # N = 512: the row spans two tiles
    out, y = torch.empty(512, 512, device=DEV), torch.empty(512, 512, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(KernelError):
        ops.linear_fwd(x, w, torch.zeros(512, device=DEV), out, ln=(None, None, y, None, None, 0.0))


@pytest.mark.parametrize('xdt,odt', [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.float32, torch.float16),
                                     (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float16, torch.float16)])
def test_dropout_kernel(xdt, odt):
    """mmae_dropout: out = [resid +] keep ? x * scale [* s[i // per]] : 0 -- nn.Dropout forward and backward as one element-wise pass,
    optionally folded with the residual add and the per-sample stochastic-depth scale of a Block branch (multimae_utils.py:229-232)."""
    from multimae_amd import ops
    torch.manual_seed(0)
    B, N, D, p = 3, 7, 64, 0.3
    x = torch.randn(B * N, D, device=DEV).to(xdt)
    keep = ops._dropout_keep((B * N, D), p, DEV)
    inv = 1.0 / (1.0 - p)
    want = x.float() * keep.float() * inv
    got = ops.dropout_apply(x, keep, inv, out_dtype=odt)
    assert got.dtype == odt and torch.equal(got.float(), want.to(odt).float())
    assert torch.equal((got == 0), (keep == 0) | (x == 0))
    if xdt == torch.float32 and odt == torch.float32:
        resid = torch.randn(B * N, D, device=DEV)
        s = torch.tensor([0.0, 1.25, 2.0], device=DEV)
        got = ops.dropout_apply(x, keep, inv, resid=resid, row_scale=s, per=N * D)
        want = resid + x * keep.float() * (inv * s.repeat_interleave(N).view(-1, 1))
        assert torch.allclose(got, want, atol=1e-6, rtol=1e-6)
        # in place (the activation path of the stand-alone modules)
        y = x.clone()
        assert ops.dropout_apply(y, keep, inv, out=y) is y and torch.equal(y, x * keep.float() * inv)
    with pytest.raises(AssertionError):
        ops.dropout_apply(x[:, :63].contiguous(), keep[:, :63].contiguous(), inv)
