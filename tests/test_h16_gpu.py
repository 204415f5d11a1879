"""fp16 STORAGE (mmae.h MMAE_F16, engine.set_fp32_adapter_gemm('h16')): the kernels of the bf16 pipeline instantiated for IEEE-half tensors,
each against an fp64 evaluation of the same fp16-rounded inputs; the gradient scale S = 2^(4 - floor(log2 amax)) and its removal at every
f32 sink; the whole adapter against its f32-activation form."""
import math

import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _S(amax: float) -> float:
    return 2.0 ** (4 - math.floor(math.log2(amax)))


def _gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def _dgelu(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize('shape', [(520, 264, 256), (50176 // 8, 1024, 256), (300, 2128, 256)])
def test_gemm_f16_forward_flavours(shape):
    from multimae_amd import ops
    from multimae_amd._lib import EPI_GELU_G
    M, N, K = shape
    g = torch.Generator().manual_seed(M)
    A = torch.randn(M, K, generator=g).half()
    W = (torch.randn(N, K, generator=g) * 0.05).half()
    bias = torch.randn(N, generator=g) * 0.1
    ref = A.double() @ W.double().t() + bias.double()
    Ad, Wd, bd = A.to(DEV), W.to(DEV), bias.to(DEV)
    # bias -> fp16
    C = torch.empty(M, N, device=DEV, dtype=torch.float16)
    ops.gemm(Ad, Wd, C, M, N, K, lda=K, ldb=K, ldc=N, bias=bd)
    assert rel_err(C.cpu().double(), ref) < 4e-4                      # one fp16 rounding of the output
    # bias + residual -> f32
    R = torch.randn(M, N, generator=g)
    C32 = torch.empty(M, N, device=DEV)
    ops.gemm(Ad, Wd, C32, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, resid=R.to(DEV), ldr=N)
    assert rel_err(C32.cpu().double(), ref + R.double()) < 2e-6
    ops.gemm(Ad, Wd, C32, M, N, K, lda=K, ldb=K, ldc=N, bias=bd)
    assert rel_err(C32.cpu().double(), ref) < 2e-6
    # bias + GELU -> fp16 activations, GELU'(pre-activation) -> fp16 aux
    aux = torch.empty(M, N, device=DEV, dtype=torch.float16)
    ops.gemm(Ad, Wd, C, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, aux=aux, ldaux=N, epi=EPI_GELU_G)
    assert rel_err(C.cpu().double(), _gelu(ref)) < 5e-4
    assert rel_err(aux.cpu().double(), _dgelu(ref)) < 5e-4


def test_gemm_f16_input_gradient_flavours_and_unscale():
    from multimae_amd import ops
    from multimae_amd._lib import EPI_MUL
    M, N, K = 1000, 1024, 256                                     # dx[M, K] = dy[M, N] @ W[N, K]
    g = torch.Generator().manual_seed(3)
    amax = torch.tensor([3.3e-7])
    S = _S(float(amax))
    dy_true = torch.randn(M, N, generator=g) * 4e-8
    dy = (dy_true * S).half()                                     # as the producers store it
    W = (torch.randn(N, K, generator=g) * 0.05).half()
    ref = dy.double() @ W.double()
    dyd, Wd = dy.to(DEV), W.to(DEV)
    out = torch.empty(M, K, device=DEV, dtype=torch.float16)
    ops.gemm(dyd, Wd, out, M, K, N, lda=N, ldb=K, ldc=K, b_trans=True)
    assert rel_err(out.cpu().double(), ref) < 4e-4
    # x GELU' (stored derivative) + column-sum partials of the result
    dg = torch.rand(M, K, generator=g).half()
    part = torch.empty(ops.dx_colsum_part_shape(M, K), device=DEV)
    ops.gemm(dyd, Wd, out, M, K, N, lda=N, ldb=K, ldc=K, b_trans=True, aux=dg.to(DEV), ldaux=K, epi=EPI_MUL, colsum_part=part)
    r2 = ref * dg.double()
    assert rel_err(out.cpu().double(), r2) < 5e-4
    assert rel_err(part.sum(0).cpu().double(), r2.sum(0)) < 2e-4
    # f32 output leaving the fp16-storage domain: times 1/S
    o32 = torch.empty(M, K, device=DEV)
    ops.gemm(dyd, Wd, o32, M, K, N, lda=N, ldb=K, ldc=K, b_trans=True, a_amax=amax.to(DEV))
    assert rel_err(o32.cpu().double(), ref / S) < 2e-6
    assert rel_err(o32.cpu().double(), dy_true.double() @ W.double()) < 6e-4      # TF32-class against the unrounded gradient


def test_gemm_f16_contraction_not_a_multiple_of_32():
    """out_proj's input gradient of the semseg adapter: dh[M, 256] = d_pat[M, 2128] @ W[2128, 256] -- the general address walk"""
    from multimae_amd import ops
    M, N, K = 1000, 2128, 256
    g = torch.Generator().manual_seed(8)
    dy = torch.randn(M, N, generator=g).half()
    W = (torch.randn(N, K, generator=g) * 0.05).half()
    out = torch.empty(M, K, device=DEV, dtype=torch.float16)
    ops.gemm(dy.to(DEV), W.to(DEV), out, M, K, N, lda=N, ldb=K, ldc=K, b_trans=True)
    assert rel_err(out.cpu().double(), dy.double() @ W.double()) < 4e-4


def test_gemm_f16_rejects_what_is_not_compiled():
    from multimae_amd import ops
    A = torch.randn(300, 72, device=DEV).half()                  # forward product with K % 32 != 0
    W = torch.randn(64, 72, device=DEV).half()
    C = torch.empty(300, 64, device=DEV, dtype=torch.float16)
    with pytest.raises(RuntimeError):
        ops.gemm(A, W, C, 300, 64, 72, lda=72, ldb=72, ldc=64)


@pytest.mark.parametrize('rows', [50176 // 4, 1000])
def test_dw_group_f16_unscaled_weight_and_bias_gradients(rows):
    from multimae_amd import ops
    g = torch.Generator().manual_seed(rows)
    amax = torch.tensor([1.7e-6])
    S = _S(float(amax))
    probs, refs = [], []
    for n_out, k_in in [(256, 1024), (768, 256), (2128, 256)]:
        dy = (torch.randn(rows, n_out, generator=g) * 1e-7 * S).half()
        x = torch.randn(rows, k_in, generator=g).half()
        dw = torch.full((n_out, k_in), 2e-5, device=DEV)
        db = torch.full((n_out,), -1e-4, device=DEV)
        probs.append((dy.to(DEV), x.to(DEV), dw, db))
        refs.append((dy.double().t() @ x.double() / S, dy.double().sum(0) / S))
    ops.gemm_dw_group(probs, accumulate=True, unscale=amax.to(DEV))
    for (dy, x, dw, db), (rw, rb) in zip(probs, refs):
        assert rel_err(dw.cpu().double() - 2e-5, rw) < 1e-4         # (f32 cancellation against what was there)
        assert rel_err(db.cpu().double() + 1e-4, rb) < 1e-4
    for p in probs:
        p[2].zero_(); p[3].zero_()
    ops.gemm_dw_group(probs, accumulate=False, unscale=amax.to(DEV))
    for (dy, x, dw, db), (rw, rb) in zip(probs, refs):
        assert rel_err(dw.cpu().double(), rw) < 3e-6
        assert rel_err(db.cpu().double(), rb) < 3e-6


def test_layernorm_colsum_and_casts_f16():
    from multimae_amd import ops
    g = torch.Generator().manual_seed(5)
    R, D = 1001, 256
    x = torch.randn(R, D, generator=g) * 2 + 0.3
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, torch.float16)
    xd = x.double()
    mu, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    xh = (xd - mu) / torch.sqrt(var + 1e-6)
    assert y.dtype == torch.float16 and rel_err(y.cpu().double(), xh * w.double() + b.double()) < 4e-4
    # backward: fp16 dy (scaled units), f32 residual gradient in, f32 + fp16 out; linear in the scale
    dy = (torch.randn(R, D, generator=g) * 20).half()
    dx_in = torch.randn(R, D, generator=g) * 20
    dx, dx_act, part = ops.layernorm_bwd_part(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, dx_in.to(DEV), torch.float16)
    dh = dy.double() * w.double()
    rs = 1 / torch.sqrt(var + 1e-6)
    ref = rs * (dh - dh.mean(1, keepdim=True) - xh * (dh * xh).mean(1, keepdim=True)) + dx_in.double()
    assert rel_err(dx.cpu().double(), ref) < 2e-5
    assert dx_act.dtype == torch.float16 and rel_err(dx_act.cpu().double(), ref) < 4e-4
    # the partial block reduced with the scale removed
    amax = torch.tensor([0.9e-6], device=DEV)
    S = _S(0.9e-6)
    outs = [torch.zeros(D, device=DEV) for _ in range(3)]
    bias_g = torch.zeros(D, device=DEV)
    ops.colsum_batch([(part, D, outs), (dy.to(DEV), D, [bias_g])], False, unscale=amax)
    assert rel_err(outs[0].cpu().double(), (dy.double() * xh).sum(0) / S) < 2e-5
    assert rel_err(outs[1].cpu().double(), dy.double().sum(0) / S) < 2e-5
    assert rel_err(outs[2].cpu().double(), ref.sum(0) / S) < 2e-4
    assert rel_err(bias_g.cpu().double(), dy.double().sum(0) / S) < 2e-5
    # casts: scale in, scale out
    v = torch.randn(3001, generator=g) * 2e-7
    h = ops.cast_f16(v.to(DEV), scale_amax=amax)
    assert h.dtype == torch.float16 and rel_err(h.cpu().double(), v.double() * S) < 4e-4
    back = ops.cast_f16(h, scale_amax=amax)
    assert back.dtype == torch.float32 and rel_err(back.cpu().double(), v.double()) < 4e-4
    # beyond the fp16 range: +-inf, never a silent clamp (round 5) -- the inf reaches the gradient norm and the fused
    # optimiser step skips and counts the update (test_h16_overflow_surfaces_as_a_skipped_counted_step)
    big = ops.cast_f16(torch.tensor([1e6, -1e6, 1.0, 65504.0, float('nan')], device=DEV)).cpu()
    assert big[:4].tolist() == [float('inf'), float('-inf'), 1.0, 65504.0] and bool(torch.isnan(big[4]))


@pytest.mark.parametrize('hd,Nq,Nk', [(32, 196, 98), (64, 98, 98), (32, 196, 196)])
def test_attention_f16_storage_vs_fp64(hd, Nq, Nk):
    from multimae_amd import ops
    from multimae_amd.ops import AttnView
    B, H = 3, 4
    D = H * hd
    g = torch.Generator().manual_seed(hd + Nq)
    q = torch.randn(B * Nq, D, generator=g).half()
    kv = torch.randn(B * Nk, 2 * D, generator=g).half()
    d_o = (torch.randn(B * Nq, D, generator=g) * 8).half()
    qd, kvd, dod = q.to(DEV), kv.to(DEV), d_o.to(DEV)
    o = torch.empty(B * Nq, D, device=DEV, dtype=torch.float16)
    scale = hd ** -0.5
    st = ops.attention_fwd(AttnView(qd, 0, D, Nq), AttnView(kvd, 0, 2 * D, Nk), AttnView(kvd, D, 2 * D, Nk), AttnView(o, 0, D, Nq), B, H, hd, scale)
    assert st[0] == 'fused'
    Q = q.double().view(B, Nq, H, hd).transpose(1, 2).requires_grad_()
    K = kv[:, :D].double().reshape(B, Nk, H, hd).transpose(1, 2).requires_grad_()
    V = kv[:, D:].double().reshape(B, Nk, H, hd).transpose(1, 2).requires_grad_()
    P = torch.softmax(Q @ K.transpose(-1, -2) * scale, -1)
    O = (P @ V).transpose(1, 2).reshape(B * Nq, D)
    assert rel_err(o.cpu().double(), O.detach()) < 1.5e-3
    O.backward(d_o.double())
    dq = torch.empty(B * Nq, D, device=DEV, dtype=torch.float16)
    dkv = torch.empty(B * Nk, 2 * D, device=DEV, dtype=torch.float16)
    ops.attention_bwd(AttnView(qd, 0, D, Nq), AttnView(kvd, 0, 2 * D, Nk), AttnView(kvd, D, 2 * D, Nk), st, AttnView(o, 0, D, Nq),
                      AttnView(dod, 0, D, Nq), AttnView(dq, 0, D, Nq), AttnView(dkv, 0, 2 * D, Nk), AttnView(dkv, D, 2 * D, Nk), B, H, hd, scale)
    f = lambda t, n: t.transpose(1, 2).reshape(B * n, D)
    assert rel_err(dq.cpu().double(), f(Q.grad, Nq)) < 3e-3
    assert rel_err(dkv[:, :D].cpu().double(), f(K.grad, Nk)) < 3e-3
    assert rel_err(dkv[:, D:].cpu().double(), f(V.grad, Nk)) < 3e-3


def test_cross_entropy_rows_in_fp16_units():
    """The cross-entropy backward writes fp16 rows times S(m), m = the largest per-sample weight (written to dy_amax): equal to the f32 rows
    of the same kernel times S, to fp16 rounding; no element exceeds 32."""
    from multimae_amd import ops
    from multimae_amd.functions import MaskedCEPatFn, PatHandle
    g = torch.Generator().manual_seed(11)
    B, C, P, nh = 4, 21, 4, 3
    lr = torch.randn(B, C, nh * P, nh * P, generator=g) * 3
    tgt = torch.randint(0, C, (B, nh * P, nh * P), generator=g)
    mask = (torch.rand(B, nh * nh, generator=g) < 0.6).long()
    mask[1] = 0                                                       # a sample without masked patches
    mask[2, :] = 0; mask[2, 4] = 1                                    # and one with a single patch: the largest weight
    pat = ops.patchify(lr.to(DEV), C, nh, nh, P, P, torch.float32).contiguous()
    rows = {}
    for act in (torch.float32, torch.float16):
        h = PatHandle(pat, C, nh, nh, P, P, act)
        h.dy_amax = torch.zeros(1, device=DEV)
        h.token = torch.zeros(1, device=DEV, requires_grad=True)
        (MaskedCEPatFn.apply(h.token, h, tgt.to(DEV), mask.to(DEV), P, 0.1) * 0.37).backward()
        torch.cuda.synchronize()
        rows[act] = (h.d_pat.float().cpu(), float(h.dy_amax))
    r32, m32 = rows[torch.float32]
    r16, m16 = rows[torch.float16]
    n_valid = int((mask.sum(1) > 0).sum())
    assert abs(m16 - 0.37 / (n_valid * P * P)) < 1e-7 * m16 + 1e-12    # upstream / (samples with a mask x pixels of the smallest mask)
    assert m16 >= m32 > 0
    S = _S(m16)
    assert rel_err(r16.double(), r32.double() * S) < 4e-4
    assert 4.0 < float(r16.abs().max()) < 32.0


def test_adapter_fp16_storage_vs_f32_activations():
    """The whole semseg adapter (cfg3 geometry, B = 4) in 'h16' mode against the same adapter with f32 activations and exact-f32 products:
    prediction, d_enc and every parameter gradient; and against the 'f16' (f32 storage, fp16 operands) mode it replaces."""
    import multimae_amd as M
    from test_parity_geometry_gpu import _seeded_case, _fwd_bwd
    doms = ['rgb', 'depth', 'semseg']
    res = {}
    for mode in ('x3', 'f16', 'h16'):
        model, x, mask_all, ik, ir, ntok = _seeded_case(doms)
        model.to(DEV)
        model.build_arena()
        tm = {d: mask_all[:, i * ntok:(i + 1) * ntok].to(DEV) for i, d in enumerate(doms)}
        xd = {k: v.to(DEV) for k, v in x.items()}
        old = M.engine.fp32_adapter_gemm()
        M.engine.set_fp32_adapter_gemm(mode)
        try:
            preds, losses = _fwd_bwd(model, xd, doms, tm, ik.to(DEV), ir.to(DEV), 98, 'bf16', ('semseg',))
        finally:
            M.engine.set_fp32_adapter_gemm(old)
        res[mode] = (preds['semseg'].float().cpu().double(), float(losses['semseg']),
                     {n: p.grad.detach().cpu().double() for n, p in model.named_parameters() if p.grad is not None and n.startswith('output_adapters.semseg.')},
                     {n: p.grad.detach().cpu().double() for n, p in model.named_parameters() if p.grad is not None and n.startswith('encoder.11.')})
    ref = res['x3']                                                # split-bf16 products on f32 tensors: ~16 operand bits
    for mode in ('f16', 'h16'):
        p, l, ga, ge = res[mode]
        assert rel_err(p, ref[0]) < 3e-3, (mode, rel_err(p, ref[0]))
        assert abs(l - ref[1]) < 2e-3 * abs(ref[1]), (mode, l, ref[1])
        worst = max(rel_err(ga[n], ref[2][n]) for n in ga)
        assert worst < 6e-3, (mode, sorted(((rel_err(ga[n], ref[2][n]), n) for n in ga), reverse=True)[:5])
        worst_e = max(rel_err(ge[n], ref[3][n]) for n in ge)          # what reaches the encoder through d_enc (summed with the bf16 adapters')
        assert worst_e < 6e-3, (mode, worst_e)


def test_standalone_adapter_fp16_storage_from_an_image_gradient_vs_oracle():
    """One SpatialOutputAdapter in 'h16' mode on its own -- parameters NOT in an arena (the fp16 weight copies are cast one by one), the gradient
    arriving on the IMAGE tensor (no loss kernel fixed its units: the adapter's backward takes rows -> their largest element -> one scaled
    cast), twice the same magnitude 1e-6 apart -- against the oracle: prediction, encoder-token gradient, every parameter gradient, TF32 class."""
    import multimae_amd as M
    import multimae_oracle as orc
    torch.manual_seed(5)
    B, Denc, D, P, S, C = 3, 96, 64, 4, 32, 3
    ad = M.SpatialOutputAdapter(num_channels=C, stride_level=1, patch_size_full=P, dim_tokens=D, depth=1, num_heads=2, use_task_queries=True,
                                task='rgb', context_tasks=['rgb', 'depth'], use_xattn=True, image_size=S, dim_tokens_enc=Denc)
    g0 = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for n_, p in ad.named_parameters():
            if p.requires_grad and (n_.endswith('bias') or 'mask_token' in n_):
                p.copy_(0.05 * torch.randn(p.shape, generator=g0))
    n, nkeep, G = (S // P) ** 2, 40, 1
    ids_shuffle = torch.argsort(torch.rand(B, 2 * n), dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :nkeep]
    enc = torch.randn(B, nkeep + G, Denc)
    info = {'tasks': {'rgb': {'num_tokens': n, 'has_2d_posemb': True, 'start_idx': 0, 'end_idx': n},
                      'depth': {'num_tokens': n, 'has_2d_posemb': True, 'start_idx': n, 'end_idx': 2 * n}},
            'image_size': (S, S), 'num_task_tokens': 2 * n, 'num_global_tokens': G}
    cfg = orc.standard_config(['rgb', 'depth'], patch_size=P, image_size=S, dim_tokens=Denc, depth=1, num_heads=2, dec_dim=D, dec_depth=1,
                              dec_heads=2, extra_norm_pix=False)
    sd = {'output_adapters.rgb.' + k: v.detach().clone() for k, v in ad.state_dict().items()}
    for k, p in ad.named_parameters():
        sd['output_adapters.rgb.' + k].requires_grad_(p.requires_grad)
    ad = ad.to(DEV)
    rel = lambda a, b: float((a.cpu().double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    for scale in (1.0, 1e-6):                               # the second: gradients that plain fp16 would flush to zero
        g = torch.randn(B, C, S, S) * scale
        for v in sd.values():
            v.grad = None
        eo = enc.clone().requires_grad_(True)
        po = orc.spatial_adapter(eo, sd, cfg, 'rgb', 'rgb', {'rgb': n, 'depth': n}, ids_keep, ids_restore, (S, S))
        po.backward(g)
        ad.zero_grad()
        eg = enc.to(DEV).requires_grad_(True)
        with M.engine.precision('bf16'):
            pred = ad(eg, info, ids_keep.to(DEV), ids_restore.to(DEV), act_dtype=torch.float32, f32_gemm='h16')
            assert pred._mmae_pat.act == torch.float16, 'the call must have taken the fp16-storage path'
            pred.backward(g.to(DEV))
        torch.cuda.synchronize()
        assert rel(pred.detach(), po.detach()) < 4e-3, rel(pred.detach(), po.detach())
        assert rel(eg.grad, eo.grad) < 6e-3, (scale, rel(eg.grad, eo.grad))
        for k, p in ad.named_parameters():
            if p.requires_grad:
                assert p.grad is not None, k
                assert rel(p.grad, sd['output_adapters.rgb.' + k].grad) < 8e-3, (scale, k, rel(p.grad, sd['output_adapters.rgb.' + k].grad))


def test_fp16_storage_adapter_under_a_pixel_loss_and_two_losses():
    """An fp32 output adapter in 'h16' mode whose gradient does NOT come from the cross-entropy kernel: a masked MSE on its patch rows (f32 rows,
    the adapter's backward finds their largest element and casts them into its units), and two losses on the same prediction (the rows add up
    in f32 first).  Against the same adapter with f32 tensors and split-bf16 products."""
    import multimae_amd as M
    from test_parity_geometry_gpu import _seeded_case
    doms = ['rgb', 'depth', 'semseg']
    res = {}
    for mode in ('x3', 'h16'):
        model, x, mask_all, ik, ir, ntok = _seeded_case(doms)
        model.to(DEV)
        model.build_arena()
        tm = {d: mask_all[:, i * ntok:(i + 1) * ntok].to(DEV) for i, d in enumerate(doms)}
        xd = {k: v.to(DEV) for k, v in x.items()}
        model.generate_random_masks = lambda *a, **k: (tm, ik.to(DEV), ir.to(DEV))
        old = M.engine.fp32_adapter_gemm()
        M.engine.set_fp32_adapter_gemm(mode)
        try:
            with M.engine.precision('bf16'):
                preds, masks = model(xd, num_encoded_tokens=98, alphas=1.0, fp32_output_adapters=['rgb'])
                if mode == 'h16':
                    assert preds['rgb']._mmae_pat.act == torch.float16
                l1 = M.MaskedMSELoss(16, 1)(preds['rgb'].float(), xd['rgb'], mask=masks['rgb'])
                l2 = M.MaskedL1Loss(16, 1)(preds['rgb'].float(), xd['rgb'], mask=masks['rgb'])
                (l1 + 0.5 * l2).backward()
        finally:
            M.engine.set_fp32_adapter_gemm(old)
        torch.cuda.synchronize()
        res[mode] = (float(l1.detach()), float(l2.detach()), {n: p.grad.detach().cpu().double() for n, p in model.named_parameters()
                                           if p.grad is not None and (n.startswith('output_adapters.rgb.') or n.startswith('encoder.11.'))})
    a, b = res['x3'], res['h16']
    assert abs(a[0] - b[0]) < 2e-3 * abs(a[0]) and abs(a[1] - b[1]) < 2e-3 * abs(a[1])
    assert len(b[2]) > 30
    worst = max((rel_err(b[2][n], a[2][n]), n) for n in a[2])
    assert worst[0] < 6e-3, worst


def test_h16_overflow_surfaces_as_a_skipped_counted_step():
    """An fp16-storage adapter whose activations leave the fp16 range must not train on clamped values: the overflow becomes inf,
    travels to the loss / gradient norm, and mmae_opt_step skips the update and counts it -- the GradScaler contract of the
    reference's own fp16 runs.  Forced here by scaling one MLP weight of the semseg adapter by 1e6."""
    import multimae_amd as M
    from multimae_amd.optim import FusedAdamW
    from helpers import MINI, build_mini_engine, make_inputs
    torch.manual_seed(0)
    model = build_mini_engine().to(DEV)
    model.build_arena()
    torch.manual_seed(5)
    x = {k: v.to(DEV) for k, v in make_inputs(MINI['doms'], 4, MINI['S']).items()}
    P = MINI['P']
    fns = {'rgb': M.MaskedMSELoss(P, 1), 'depth': M.MaskedL1Loss(P, 1), 'semseg': M.MaskedCrossEntropyLoss(P, 4), 'norm_rgb': M.MaskedMSELoss(P, 1, norm_pix=True)}
    prev = M.engine.fp32_adapter_gemm()
    M.engine.set_fp32_adapter_gemm('h16')
    M.engine.set_direct_grads(True)
    opt = FusedAdamW(model, lr=1e-3)
    try:
        def step():
            opt.zero_grad()
            preds, masks = model(x, num_encoded_tokens=MINI['nvis'], alphas=1.0, fp32_output_adapters=['semseg'])
            mk = dict(masks, norm_rgb=masks['rgb'])
            tgt = dict(x, norm_rgb=x['rgb'])
            loss = sum(fns[k](preds[k].float(), tgt[k], mask=mk[k]) for k in preds)
            loss.backward()
            opt.step(loss)
            return float(loss)
        l0 = step()
        assert l0 == l0 and opt.counters(detail=True) == dict(steps=1, nonfinite_loss=0, skipped=0, amp_overflow=0, nonfinite_grad=0)
        w = model.output_adapters['semseg'].mlp.fc1.weight
        before = model.encoder[0].attn.qkv.weight.detach().clone()
        with torch.no_grad():
            w.mul_(1e6)
        l1 = step()
        c = opt.counters(detail=True)
        assert c['steps'] == 1 and c['skipped'] == 1, (l1, c)          # the update was NOT applied ...
        assert c['nonfinite_loss'] + c['nonfinite_grad'] >= 1, c        # ... because the overflow was seen, not clamped away
        assert torch.equal(model.encoder[0].attn.qkv.weight.detach(), before)
    finally:
        M.engine.set_direct_grads(False)
        M.engine.set_fp32_adapter_gemm(prev)
