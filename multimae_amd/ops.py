"""Thin Python launchers for the HIP kernels (torch tensors in, raw pointers out).

PyTorch is used here only for device memory (caching allocator) and the current
HIP stream; every arithmetic step is a kernel of libmmae_hip.so.  Nothing in this
module has a CPU or torch-op fallback: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Sequence

import torch

from . import _lib
import contextlib

from ._lib import BF16, F16, F32, F32X3, F32F16, MXFP8, EPI_NONE, EPI_GELU, EPI_DGELU, EPI_GELU_G, EPI_MUL, AdapterDesc, BlockDesc, DwGroupDesc, GemmDesc, OptDesc, PatchSrc, StackDesc, check

Tensor = torch.Tensor


def _require_gpu(t: Tensor, name: str = 'tensor') -> None:
    if not t.is_cuda:
        raise RuntimeError(f'multimae_amd: {name} is on {t.device}; the engine runs on MI355X HIP kernels only '
                           '(no CPU / PyTorch fallback). Move the model and inputs to a GPU.')


def dcode(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float16:           # fp16 storage: an fp32 output adapter in engine.set_fp32_adapter_gemm('h16') mode
        return F16
    raise TypeError(f'unsupported activation dtype {dtype}')


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# --------------------------------------------------------------------------- GEMM --
_F32_GEMM = ['exact']


@contextlib.contextmanager
def f32_gemm_mode(mode: str):
    """How GEMMs with f32 operands are multiplied inside the block: 'exact' (f32-input MFMA, the parity mode), 'x3' (split-bf16 on
    the bf16 MFMA: ~16 operand bits, three MFMAs) or 'f16' (fp16 operands = TF32's significand, one MFMA; the composite adapter
    calls only -- single launches through ops.gemm and the fused attention cores use the split form in both)."""
    assert mode in ('exact', 'x3', 'f16')
    old = _F32_GEMM[0]
    _F32_GEMM[0] = mode
    try:
        yield
    finally:
        _F32_GEMM[0] = old



import os as _os
_FUSED_DB = _os.environ.get('MMAE_FUSED_DB', '1') != '0'


def _f32_split() -> bool:
    """f32 activations on the bf16 / fp16 matrix cores (as opposed to the exact-f32 parity mode)"""
    return _F32_GEMM[0] in ('x3', 'f16')


def _f32_code() -> int:
    return {'x3': F32X3, 'f16': F32F16}.get(_F32_GEMM[0], F32)


def gemm(A: Tensor, B: Tensor, C: Tensor, M: int, N: int, K: int, *, lda: int, ldb: int, ldc: int,
         a_trans: bool = False, b_trans: bool = False, a_off: int = 0, b_off: int = 0, c_off: int = 0,
         batch: int = 1, batch_inner: int = 1, sA=(0, 0), sB=(0, 0), sC=(0, 0),
         bias: Optional[Tensor] = None, resid: Optional[Tensor] = None, ldr: int = 0,
         aux: Optional[Tensor] = None, ldaux: int = 0, epi: int = EPI_NONE, accumulate: bool = False,
         alpha: float = 1.0, tile: int = 0, split_k: int = 0, colsum_part: Optional[Tensor] = None,
         a_colsum: Optional[Tensor] = None, a_colsum_acc: bool = False, f32_as: Optional[int] = None,
         a_amax: Optional[Tensor] = None, ln: Optional[tuple] = None) -> bool:
    """C[M,N] (+)= alpha * A[M,K] . B[N,K]^T with the fused epilogue of mmae_gemm.
    *_off are element offsets into the tensors' storage views (column offsets into packed qkv etc.).
    ln = (gamma, beta, out, mean, rstd, eps): the LayerNorm side output of mmae_gemm_desc.ln_out (gamma None: the plain 16-bit cast of C)."""
    _require_gpu(A, 'gemm A')
    assert A.dtype == B.dtype, (A.dtype, B.dtype)
    d = GemmDesc()
    esz_ab, esz_c = A.element_size(), C.element_size()
    d.A = A.data_ptr() + a_off * esz_ab
    d.B = B.data_ptr() + b_off * esz_ab
    d.C = C.data_ptr() + c_off * esz_c
    d.ab_dtype, d.c_dtype = dcode(A.dtype), dcode(C.dtype)
    if d.ab_dtype == F32 and f32_as is not None:           # explicit: F32X3 / F32F16 (tests, probes); a_amax: device scalar that pre-scales A
        d.ab_dtype = f32_as
        d.a_amax = _p(a_amax)
    elif d.ab_dtype == F32 and _f32_split() and lda % 4 == 0 and ldb % 4 == 0:
        d.ab_dtype = F32X3
    elif d.ab_dtype == F16:                                 # f32 C of a gradient product: multiplied by 1/S(a_amax)
        d.a_amax = _p(a_amax)
        split_k = 1
    d.M, d.N, d.K = M, N, K
    d.a_trans, d.b_trans = int(a_trans), int(b_trans)
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.batch, d.batch_inner = batch, batch_inner
    d.sA_outer, d.sA_inner = sA
    d.sB_outer, d.sB_inner = sB
    d.sC_outer, d.sC_inner = sC
    d.bias = _p(bias)
    d.resid, d.ldr = _p(resid), ldr
    d.aux, d.ldaux = _p(aux), ldaux
    d.aux_dtype = dcode(aux.dtype) if aux is not None else F32
    d.epi, d.accumulate, d.alpha, d.tile = epi, int(accumulate), alpha, tile
    if ln is not None:
        gam, bet, lo, lm, lr, le = ln
        d.ln_gamma, d.ln_beta, d.ln_out, d.ln_mean, d.ln_rstd, d.ln_eps = _p(gam), _p(bet), lo.data_ptr(), _p(lm), _p(lr), float(le)
        split_k = 1
    ws = None
    lib = _lib.load()
    d.colsum_part = _p(colsum_part)
    if split_k == 0:
        t_, s_ = ctypes.c_int(0), ctypes.c_int(1)
        check(lib.mmae_gemm_plan(ctypes.byref(d), ctypes.byref(t_), ctypes.byref(s_)), 'mmae_gemm_plan')
        d.tile, split_k = t_.value, s_.value
    # a_colsum (sum over k of the k-strided A operand = bias gradient of a dW product) rides along only on the kernel that
    # implements it; otherwise the caller gets False back and runs its own column sum
    fused_cs = _FUSED_DB and a_colsum is not None and d.tile == 9 and a_trans and d.ab_dtype == BF16 and batch == 1
    n_ws = (split_k * M * N if split_k > 1 else 0) + (max(split_k, 1) * M if fused_cs else 0)
    if n_ws:
        ws = torch.empty((n_ws,), device=A.device, dtype=torch.float32)
        d.ws, d.ws_elems = ws.data_ptr(), ws.numel()
    d.split_k = max(split_k, 1)
    if fused_cs:
        assert a_colsum.dtype == torch.float32 and a_colsum.numel() >= M
        d.a_colsum, d.a_colsum_acc = a_colsum.data_ptr(), int(a_colsum_acc)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= N
    if resid is not None:
        assert resid.dtype == torch.float32
    check(lib.mmae_gemm(ctypes.byref(d), _stream()), 'mmae_gemm')
    return fused_cs


# ------------------------------------------------------------------------- MX-fp8 --
class MxTensor:
    """An OCP MX-fp8 operand: e4m3 bytes [rows, cols] (blocks of 32 along cols) + the packed E8M0 scales of mmae_mx_quant."""
    __slots__ = ('q', 'scales', 'rows', 'cols')

    def __init__(self, q: Tensor, scales: Tensor, rows: int, cols: int):
        self.q, self.scales, self.rows, self.cols = q, scales, rows, cols


def mx_scale_bytes(rows: int, cols: int) -> int:
    return int(_lib.load().mmae_mx_scale_bytes(rows, cols))


def mx_quant(x: Tensor, out: Optional[MxTensor] = None) -> MxTensor:
    """x [rows, cols] (f32 / bf16, contiguous) -> MX-fp8 with the blocks along cols (mmae_mx_quant)."""
    _require_gpu(x, 'mx_quant input')
    rows, cols = x.shape
    if out is None:
        out = MxTensor(torch.empty((rows, cols), device=x.device, dtype=torch.uint8),
                       torch.empty((mx_scale_bytes(rows, cols),), device=x.device, dtype=torch.uint8), rows, cols)
    check(_lib.load().mmae_mx_quant(x.data_ptr(), dcode(x.dtype), x.stride(0), rows, cols, out.q.data_ptr(), cols, out.scales.data_ptr(),
                                    _stream()), 'mmae_mx_quant')
    return out


def mx_empty(rows: int, cols: int, device) -> MxTensor:
    return MxTensor(torch.empty((rows, cols), device=device, dtype=torch.uint8),
                    torch.zeros((mx_scale_bytes(rows, cols),), device=device, dtype=torch.uint8), rows, cols)


def layernorm_fwd_mx(x: Tensor, w: Tensor, b: Tensor, eps: float):
    """LayerNorm forward with bf16 output that also emits the MX-fp8 quantisation of that output (mmae_layernorm_fwd_mx).
    Returns (y bf16, mean, rstd, MxTensor)."""
    _require_gpu(x, 'layernorm input')
    R, D = x.shape
    y = torch.empty((R, D), device=x.device, dtype=torch.bfloat16)
    mean = torch.empty((R,), device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    q = mx_empty(R, D, x.device)
    check(_lib.load().mmae_layernorm_fwd_mx(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), R, D, eps,
                                            q.q.data_ptr(), q.scales.data_ptr(), _stream()), 'mmae_layernorm_fwd_mx')
    return y, mean, rstd, q


def mx_quant_t(w: Tensor, out: Optional[MxTensor] = None) -> MxTensor:
    """w [n, k] -> MX-fp8 of w^T: bytes [k, n], blocks along n (the weight operand of a dX product; mmae_mx_quant_t)."""
    _require_gpu(w, 'mx_quant_t input')
    n, k = w.shape
    if out is None:
        out = MxTensor(torch.empty((k, n), device=w.device, dtype=torch.uint8),
                       torch.empty((mx_scale_bytes(k, n),), device=w.device, dtype=torch.uint8), k, n)
    check(_lib.load().mmae_mx_quant_t(w.data_ptr(), dcode(w.dtype), w.stride(0), n, k, out.q.data_ptr(), n, out.scales.data_ptr(),
                                      _stream()), 'mmae_mx_quant_t')
    return out


def mx_quant_rows_t(x: Tensor) -> MxTensor:
    """x [M, C] bf16 -> MX-fp8 of x^T: bytes [C, Mp] (Mp = M rounded up to 256, zero rows past M), blocks along M -- an operand of a
    weight-gradient product, whose contraction runs over the rows (mmae_mx_quant_rows_t)."""
    _require_gpu(x, 'mx_quant_rows_t input')
    M, C = x.shape
    Mp = round_up(M, 256)
    out = MxTensor(torch.empty((C, Mp), device=x.device, dtype=torch.uint8),
                   torch.empty((mx_scale_bytes(C, Mp),), device=x.device, dtype=torch.uint8), C, Mp)
    check(_lib.load().mmae_mx_quant_rows_t(x.data_ptr(), dcode(x.dtype), x.stride(0), M, C, out.q.data_ptr(), Mp, out.scales.data_ptr(),
                                           _stream()), 'mmae_mx_quant_rows_t')
    return out


def mx_wgrad(on: Optional[bool] = None) -> bool:
    """Policy of the encoder stack's weight gradients in MX-fp8 mode: True = on the scaled MFMA from row-blocked quantised copies of dy
    and x, False = bf16 (grouped launch).  Sets it when `on` is given; returns the previous value."""
    return bool(_lib.load().mmae_mx_wgrad(-1 if on is None else int(bool(on))))


def gemm_mx(a: MxTensor, b: MxTensor, C: Tensor, *, bias: Optional[Tensor] = None, resid: Optional[Tensor] = None,
            aux: Optional[Tensor] = None, epi: int = EPI_NONE, colsum_part: Optional[Tensor] = None,
            q_out: Optional[MxTensor] = None, split_k: int = 1, accumulate: bool = False) -> Tensor:
    """C[M, N] = a[M, K] . b[N, K]^T on the block-scaled MFMA, with the fused epilogues of mmae_gemm.
    q_out (GELU / dGELU epilogues, bf16 C): also receives the MX-fp8 quantisation of C."""
    assert a.cols == b.cols, (a.cols, b.cols)
    M, N, K = a.rows, b.rows, a.cols
    d = GemmDesc()
    d.A, d.B, d.C = a.q.data_ptr(), b.q.data_ptr(), C.data_ptr()
    d.a_scale, d.b_scale = a.scales.data_ptr(), b.scales.data_ptr()
    d.ab_dtype, d.c_dtype = MXFP8, dcode(C.dtype)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = K, K, N
    d.batch = d.batch_inner = 1
    d.bias = _p(bias)
    d.resid, d.ldr = _p(resid), N
    d.aux, d.ldaux = _p(aux), N
    d.aux_dtype = dcode(aux.dtype) if aux is not None else F32
    d.epi, d.alpha, d.split_k = epi, 1.0, split_k
    d.accumulate = int(accumulate)
    if split_k > 1:                                       # slices of whole 256-element scale groups write f32 slabs, summed in a fixed order
        ws = stream_workspace(_stream(), C.device, max(_WS_ELEMS[0], split_k * M * N))
        d.ws, d.ws_elems = ws.data_ptr(), ws.numel()
    d.colsum_part = _p(colsum_part)
    if q_out is not None:
        assert q_out.rows == M and q_out.cols == N
        d.q_out, d.q_scale, d.ldq = q_out.q.data_ptr(), q_out.scales.data_ptr(), N
    check(_lib.load().mmae_gemm(ctypes.byref(d), _stream()), 'mmae_gemm(mxfp8)')
    return C


def gemm_cu_reserve(k: Optional[int] = None) -> int:
    """Compute units the persistent GEMM grids (ping-pong bf16 / MX-fp8 kernels, grouped weight gradients) leave free: grids
    launched afterwards are at most n_cu - k workgroups wide (mmae_gemm_cu_reserve).  Host-side launch policy, not stream
    ordered: it applies to launches ENQUEUED after the call.  Returns the previous value; k = None only reads it."""
    return int(_lib.load().mmae_gemm_cu_reserve(-1 if k is None else int(k)))


def ln_fuse(on: Optional[bool] = None) -> bool:
    """Policy switch of the composite adapter calls (mmae_ln_fuse): LayerNorm forwards of a D = 256 decoder as side outputs of the
    preceding Linear products' epilogues (default on); returns the previous value."""
    return bool(_lib.load().mmae_ln_fuse(-1 if on is None else int(bool(on))))


def xattn_fuse(on: Optional[bool] = None) -> bool:
    """Policy switch of mmae_adapter_fwd (mmae_xattn_fuse, default off): the cross-attention of a D = 256 bf16 output adapter as ONE launch
    (q-projection + kv-projection + softmax + PV, xattn_fwd_fused) instead of two Linear products + the attention core; returns the previous value."""
    return bool(_lib.load().mmae_xattn_fuse(-1 if on is None else int(bool(on))))


def gemm_cu_share(k: Optional[int] = None) -> int:
    """The second slot of the same launch policy (mmae_gemm_cu_share): grids leave max(reserve, share) CUs free; returns the previous value."""
    return int(_lib.load().mmae_gemm_cu_share(-1 if k is None else int(k)))


def gemm_side_cus(k: Optional[int] = None) -> int:
    """A/B switch: compute units set aside for the side stream's grouped weight gradients (mmae_gemm_side_cus); returns the previous value."""
    return int(_lib.load().mmae_gemm_side_cus(-1 if k is None else int(k)))


def linear_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], out: Tensor, *, resid: Optional[Tensor] = None,
               aux: Optional[Tensor] = None, epi: int = EPI_NONE, tile: int = 0, ln: Optional[tuple] = None) -> Tensor:
    """out[M,N] = x[M,K] @ w[N,K]^T + bias (+ epilogue).  x, w act dtype, contiguous 2-D.  ln: see gemm()."""
    M, K = x.shape
    N = w.shape[0]
    gemm(x, w, out, M, N, K, lda=K, ldb=K, ldc=N, bias=bias, resid=resid, ldr=N, aux=aux, ldaux=N, epi=epi, tile=tile, ln=ln)
    return out


def linear_dx(dy: Tensor, w: Tensor, out: Tensor, *, aux: Optional[Tensor] = None, epi: int = EPI_NONE,
              colsum_out: Optional[Tensor] = None, colsum_part: Optional[Tensor] = None, tile: int = 0) -> Tensor:
    """out[M,K] = dy[M,N] @ w[N,K]   (w read through the transposing LDS path).
    dGELU epilogue only -- the column sums of `out` (= the bias gradient of the Linear whose pre-activation gradient `out`
    is) collected in the GEMM epilogue instead of a separate pass over out:
      colsum_part (f32 [ceil(M/32), K]): receives the per-32-row partial sums (caller reduces them, see GradSink.colsums);
      colsum_out  (f32 [K]): receives the reduced sums."""
    M, N = dy.shape
    K = w.shape[1]
    part = colsum_part
    if colsum_out is not None and part is None:
        part = torch.empty(dx_colsum_part_shape(M, K), device=dy.device, dtype=torch.float32)
    gemm(dy, w, out, M, K, N, lda=dy.stride(0), ldb=K, ldc=K, b_trans=True, aux=aux, ldaux=K, epi=epi, colsum_part=part, tile=tile)
    if colsum_out is not None:
        colsum(part, colsum_out, False)
    return out


def gemm_dw_group(problems: Sequence[tuple], accumulate: bool, split_k: int = 0, unscale: Optional[Tensor] = None) -> None:
    """Up to 8 weight gradients in one launch (mmae_gemm_dw_group).  problems: (dy [rows, n_out] bf16 / fp16, x [rows, k_in] same,
    dw f32 [n_out, k_in] contiguous, db f32 [n_out] or None); all share `rows`.  unscale (fp16 operands): the dy_amax scalar dy is scaled by."""
    d = DwGroupDesc()
    ab = problems[0][0].dtype
    d.n, d.rows, d.ab_dtype, d.accumulate, d.split_k = len(problems), problems[0][0].shape[0], dcode(ab), int(accumulate), split_k
    d.unscale = _p(unscale)
    for i, (dy, x, dw, db) in enumerate(problems):
        assert ab in (torch.bfloat16, torch.float16) and dy.dtype == ab and x.dtype == ab and dw.dtype == torch.float32 and dw.is_contiguous()
        assert dy.shape[0] == d.rows and x.shape[0] == d.rows and dw.shape == (dy.shape[1], x.shape[1])
        q = d.p[i]
        q.dy, q.ldy, q.x, q.ldx, q.dw, q.db, q.n_out, q.k_in = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), _p(db), dy.shape[1], x.shape[1]
    lib = _lib.load()
    n_ws = lib.mmae_gemm_dw_group_ws_elems(ctypes.byref(d))
    ws = torch.empty((max(n_ws, 4),), device=problems[0][0].device, dtype=torch.float32)
    d.ws, d.ws_elems = ws.data_ptr(), ws.numel()
    check(lib.mmae_gemm_dw_group(ctypes.byref(d), _stream()), 'gemm_dw_group')


def dx_colsum_part_shape(M: int, K: int):
    return ((M + 31) // 32, K)


def linear_dw(dy: Tensor, x: Tensor, dw: Tensor, accumulate: bool, *, db: Optional[Tensor] = None, db_accumulate: bool = False,
              x_off: int = 0, ldx: Optional[int] = None, K: Optional[int] = None) -> Tensor:
    """dw[N,K] (+)= dy[M,N]^T @ x[M,K]   (both operands k-strided).
    db (f32 [N]): also the bias gradient  db (+)= column sums of dy  -- inside the GEMM when the kernel supports it (it
    reads dy's tiles anyway), else by a separate column-sum launch."""
    M, N = dy.shape
    K = K if K is not None else x.shape[1]
    fused = gemm(dy, x, dw, N, K, M, lda=dy.stride(0), ldb=ldx if ldx is not None else x.stride(0), ldc=K, a_trans=True, b_trans=True,
                 b_off=x_off, accumulate=accumulate, a_colsum=db, a_colsum_acc=db_accumulate)
    if db is not None and not fused:
        colsum(dy, db, db_accumulate)
    return dw


# ------------------------------------------------------ composite transformer block --
_COMPOSITE = [_os.environ.get('MMAE_COMPOSITE', '1') != '0']
_WS = {}
_WS_ELEMS = [56 << 20]          # f32 elements per stream workspace (224 MB: ViT-L's grouped weight gradients in four row slices need 50.4 M)


def _device_ok(t: Tensor) -> bool:
    return t.is_cuda


def set_composite_blocks(flag: bool) -> None:
    """Transformer blocks as one library call per direction (mmae_block_fwd / mmae_block_bwd) instead of one host launch
    per kernel.  Same kernels and results; default on."""
    _COMPOSITE[0] = bool(flag)


def stream_workspace(stream_handle: int, device, elems: Optional[int] = None) -> Tensor:
    """Persistent f32 scratch private to one HIP stream (split-K slabs, reduction partials of the composite calls).
    Launches on a stream execute in order, so consecutive calls on that stream can share it."""
    elems = elems or _WS_ELEMS[0]
    key = (str(device), stream_handle)
    w = _WS.get(key)
    if w is None or w.numel() < elems:
        w = torch.empty((elems,), device=device, dtype=torch.float32)
        _WS[key] = w
    return w


def block_composite_ok(x: Tensor, act: torch.dtype, heads: int, N: int) -> bool:
    D = x.shape[1]
    hd = D // heads
    if not (_COMPOSITE[0] and _FUSED_ATTN[0] and _device_ok(x) and hd in (32, 64) and N <= 256 and D % 8 == 0):
        return False
    if act == torch.bfloat16:
        return True
    lds_bwd = 4 * 2 * round_up(N, 32) * hd * 2 + 8 * round_up(N, 32)
    return act == torch.float32 and _f32_split() and lds_bwd <= 160 * 1024


def _block_desc(P: Sequence[Tensor], wts: Sequence[Tensor], heads: int, eps: float, act: torch.dtype, B: int, N: int, D: int) -> 'BlockDesc':
    n1w, n1b, qkvw, qkvb, projw, projb, n2w, n2b, fc1w, fc1b, fc2w, fc2b = P
    d = BlockDesc()
    d.B, d.N, d.D, d.heads, d.Hd = B, N, D, heads, fc1w.shape[0]
    d.act_dtype, d.f32_gemm, d.eps = dcode(act), _f32_code(), eps
    d.qkv_w, d.proj_w, d.fc1_w, d.fc2_w = (w.data_ptr() for w in wts)
    d.n1_w, d.n1_b, d.qkv_b, d.proj_b = n1w.data_ptr(), n1b.data_ptr(), qkvb.data_ptr(), projb.data_ptr()
    d.n2_w, d.n2_b, d.fc1_b, d.fc2_b = n2w.data_ptr(), n2b.data_ptr(), fc1b.data_ptr(), fc2b.data_ptr()
    return d


def block_fwd_composite(x: Tensor, P: Sequence[Tensor], wts: Sequence[Tensor], heads: int, eps: float, act: torch.dtype, B: int, N: int):
    """x f32 [B*N, D] -> (x2, saved) with the layout functions.block_fwd produces."""
    R, D = x.shape
    Hd = P[8].shape[0]
    dev, f = x.device, torch.float32
    e = lambda shape, dt: torch.empty(shape, device=dev, dtype=dt)
    ln1, qkv, ao, ln2, hpre, hact = e((R, D), act), e((R, 3 * D), act), e((R, D), act), e((R, D), act), e((R, Hd), act), e((R, Hd), act)
    stats = e((4, R), f)
    lse = e((B, heads, N), f)
    x1, x2 = e((R, D), f), e((R, D), f)
    d = _block_desc(P, wts, heads, eps, act, B, N, D)
    d.x0, d.ln1, d.mean1, d.rstd1, d.qkv, d.lse, d.ao, d.x1 = (x.data_ptr(), ln1.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                                               qkv.data_ptr(), lse.data_ptr(), ao.data_ptr(), x1.data_ptr())
    d.ln2, d.mean2, d.rstd2, d.hpre, d.hact, d.x2 = ln2.data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), hpre.data_ptr(), hact.data_ptr(), x2.data_ptr()
    st = _stream()
    ws = stream_workspace(st, dev)
    d.ws_main, d.ws_main_elems = ws.data_ptr(), ws.numel()
    check(_lib.load().mmae_block_fwd(ctypes.byref(d), st), 'block_fwd')
    # third entry: `hpre` holds GELU'(pre-activation) (bf16 activations, mmae_gelu_grad_aux on) instead of the pre-activation
    hpre_is_grad = act == torch.bfloat16 and bool(_lib.load().mmae_gelu_grad_aux(-1))
    saved = (x, ln1, stats[0], stats[1], qkv, ('fused', lse, hpre_is_grad), ao, x1, ln2, stats[2], stats[3], hpre, hact)
    return x2, saved


def block_bwd_composite(dx: Tensor, dx_act: Tensor, fc2b_done: bool, saved, P: Sequence[Tensor], wts: Sequence[Tensor], grads: Sequence[Optional[Tensor]],
                        g_cs: Optional[Tensor], grad_acc: bool, heads: int, act: torch.dtype, B: int, N: int, side_handle: Optional[int]):
    """One mmae_block_bwd call.  grads: 12 f32 destinations (None = not wanted) in BLOCK_PARAMS order; g_cs: destination of
    colsum(dx0) or None.  Returns (dx0, dx0_act, temporaries to keep alive while the side stream may still read them)."""
    x0, ln1, mean1, rstd1, qkv, Pm, ao, x1, ln2, mean2, rstd2, hpre, hact = saved
    R, D = x0.shape
    Hd = P[8].shape[0]
    dev, f = dx.device, torch.float32
    lib = _lib.load()
    e = lambda shape, dt: torch.empty(shape, device=dev, dtype=dt)
    bf = act != f
    d_hpre, d_ln2, d_ao, d_qkv, d_ln1 = e((R, Hd), act), e((R, D), act), e((R, D), act), e((R, 3 * D), act), e((R, D), act)
    dx1, dx0 = e((R, D), f), e((R, D), f)
    dx1_act = e((R, D), act) if bf else None
    dx0_act = e((R, D), act) if bf else None
    nblk = lib.mmae_layernorm_bwd_nblk(R)
    part = e((2 * nblk * 3 * D + ((R + 31) // 32) * Hd,), f)
    d = _block_desc(P, wts, heads, 0.0, act, B, N, D)
    d.x0, d.ln1, d.mean1, d.rstd1, d.qkv, d.lse, d.ao, d.x1 = (x0.data_ptr(), ln1.data_ptr(), mean1.data_ptr(), rstd1.data_ptr(),
                                                               qkv.data_ptr(), Pm[1].data_ptr(), ao.data_ptr(), x1.data_ptr())
    d.ln2, d.mean2, d.rstd2, d.hpre, d.hact = ln2.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), hpre.data_ptr(), hact.data_ptr()
    d.dx, d.dx_act, d.dx0, d.dx0_act = dx.data_ptr(), dx_act.data_ptr(), dx0.data_ptr(), _p(dx0_act)
    d.d_hpre, d.d_ln2, d.d_ao, d.d_qkv, d.d_ln1, d.dx1, d.dx1_act = (d_hpre.data_ptr(), d_ln2.data_ptr(), d_ao.data_ptr(), d_qkv.data_ptr(),
                                                                      d_ln1.data_ptr(), dx1.data_ptr(), _p(dx1_act))
    es = 4
    d.part1 = part.data_ptr()
    d.part2 = part.data_ptr() + nblk * 3 * D * es
    d.part_h = part.data_ptr() + 2 * nblk * 3 * D * es
    (d.g_n1_w, d.g_n1_b, d.g_qkv_w, d.g_qkv_b, d.g_proj_w, d.g_proj_b, d.g_n2_w, d.g_n2_b, d.g_fc1_w, d.g_fc1_b, d.g_fc2_w,
     d.g_fc2_b) = (_p(g) for g in grads)
    d.g_cs = _p(g_cs)
    d.grad_acc, d.fc2_b_done = int(grad_acc), int(fc2b_done)
    st = _stream()
    wm = stream_workspace(st, dev)
    d.ws_main, d.ws_main_elems = wm.data_ptr(), wm.numel()
    sd = side_handle if side_handle is not None else st
    wsd = stream_workspace(sd, dev) if sd != st else wm
    d.ws_side, d.ws_side_elems = wsd.data_ptr(), wsd.numel()
    check(lib.mmae_block_bwd(ctypes.byref(d), st, sd), 'block_bwd')
    keep = (d_hpre, d_ln2, d_ao, d_qkv, d_ln1, dx1, dx1_act, part, dx, dx_act, saved)
    return dx0, (dx0_act if bf else dx0), keep


# ------------------------------------------------ per-stack / per-adapter composites --
_STACK = [_os.environ.get('MMAE_STACK_COMPOSITE', '1') != '0']
_ADAPTER = [_os.environ.get('MMAE_ADAPTER_COMPOSITE', '1') != '0']


def set_stack_composites(flag: bool) -> None:
    """Encoder stack / output adapters as ONE library call per direction (mmae_stack_fwd/bwd, mmae_adapter_fwd/bwd): same
    kernels and results as the per-block / per-kernel launch sequences, ~40 instead of ~520 host round trips per step."""
    _STACK[0] = bool(flag)


def stack_composites() -> bool:
    return _STACK[0]


def _ptr_arr(ptrs: Sequence[Optional[int]]):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


class _PtrCache:
    """ctypes pointer tables of a parameter list, rebuilt only when a watched address changes (the arena views and their
    bf16 shadows never move; stand-alone modules re-cast their weights every forward and simply miss)."""

    def __init__(self):
        self.tabs = {}

    def get(self, key, probe, build):
        e = self.tabs.get(key)
        if e is not None and e[0] == probe:
            return e[1]
        val = build()
        if len(self.tabs) > 64:
            self.tabs.clear()
        self.tabs[key] = (probe, val)
        return val


_PTRS = _PtrCache()


def _slab(nbytes: int, device) -> Tensor:
    return torch.empty((max(int(nbytes), 256),), device=device, dtype=torch.uint8)


class StackState:
    """What mmae_stack_bwd needs from the forward: the descriptor (geometry + pointer tables) and the activation slab."""
    __slots__ = ('desc', 'act', 'keep', 'x', 'L', 'shape', 'act_dtype', 'dp')


class MxStackWeights:
    """MX-fp8 copies of the four Linear weights of every block of a stack (both orientations, mmae_mx_prepare_weights): one
    byte buffer + the host pointer table mmae_stack_desc.mx_w wants.  refresh() re-quantises from the f32 master weights."""

    def __init__(self, params: Sequence[Tensor]):
        lib = _lib.load()
        L = len(params) // 12
        self.ws = [params[12 * l + i] for l in range(L) for i in (2, 4, 8, 10)]
        dev = self.ws[0].device
        offs, total = [], 0
        for w in self.ws:
            n, k = w.shape
            for nb in (n * k, int(lib.mmae_mx_scale_bytes(n, k)), n * k, int(lib.mmae_mx_scale_bytes(k, n))):
                offs.append(total)
                total += round_up(nb, 256)
        self.buf = torch.empty((total,), device=dev, dtype=torch.uint8)
        base = self.buf.data_ptr()
        self.dst = _ptr_arr([base + o for o in offs])
        self.src = _ptr_arr([w.data_ptr() for w in self.ws])
        self.n_out = (ctypes.c_int32 * len(self.ws))(*[w.shape[0] for w in self.ws])
        self.k_in = (ctypes.c_int32 * len(self.ws))(*[w.shape[1] for w in self.ws])
        self.probe = tuple(w.data_ptr() for w in self.ws[:1] + self.ws[-1:])
        self.shapes = tuple(tuple(w.shape) for w in self.ws)

    def matches(self, params: Sequence[Tensor]) -> bool:
        if self.probe != (params[2].data_ptr(), params[-2].data_ptr()) or len(self.ws) * 3 != len(params):
            return False
        L = len(params) // 12
        return self.shapes == tuple(tuple(params[12 * l + i].shape) for l in range(L) for i in (2, 4, 8, 10))

    def refresh(self) -> None:
        check(_lib.load().mmae_mx_prepare_weights(len(self.ws), ctypes.cast(self.src, ctypes.c_void_p), F32, self.n_out, self.k_in,
                                                  ctypes.cast(self.dst, ctypes.c_void_p), _stream()), 'mmae_mx_prepare_weights')


_MX_STACKS = {}


def mx_stack_weights(params: Sequence[Tensor]) -> MxStackWeights:
    key = (id(params[2]), len(params))
    m = _MX_STACKS.get(key)
    if m is None or not m.matches(params):
        if len(_MX_STACKS) > 8:
            _MX_STACKS.clear()
        m = MxStackWeights(params)
        _MX_STACKS[key] = m
    return m


def mx_stack_ok(D: int, Hd: int, act: torch.dtype) -> bool:
    return act == torch.bfloat16 and D % 256 == 0 and Hd % 256 == 0


def stack_fwd(x: Tensor, params: Sequence[Tensor], wc, heads: int, eps: float, act: torch.dtype, B: int, N: int,
              dp: Optional[Sequence[Optional[Tensor]]] = None, mx: bool = False):
    """L = len(params) // 12 pre-LN blocks on x f32 [B*N, D] in one library call.
    Returns ([output of every block as f32 [B*N, D] views of the slab], StackState).
    mx: forward / dX products on MX-fp8 operands (weights re-quantised here from their f32 masters, once per call); the backward
    call also runs the weight gradients on them (mx_wgrad)."""
    lib = _lib.load()
    R, D = x.shape
    L = len(params) // 12
    Hd = params[8].shape[0]
    d = StackDesc()
    d.L, d.B, d.N, d.D, d.heads, d.Hd = L, B, N, D, heads, Hd
    d.act_dtype, d.f32_gemm, d.eps = dcode(act), _f32_code(), eps
    wts = [wc(params[12 * l + i]) for l in range(L) for i in (2, 4, 8, 10)]
    probe = (wts[0].data_ptr(), wts[-1].data_ptr(), params[0].data_ptr(), params[-1].data_ptr())
    w_arr, p_arr = _PTRS.get(('stack', id(params[0]), L, act), probe, lambda: (
        _ptr_arr([w.data_ptr() for w in wts]),
        _ptr_arr([params[12 * l + i].data_ptr() for l in range(L) for i in (0, 1, 3, 5, 6, 7, 9, 11)])))
    d.w, d.p = ctypes.cast(w_arr, ctypes.c_void_p), ctypes.cast(p_arr, ctypes.c_void_p)
    dp_arr = None
    if dp is not None and any(t is not None for t in dp):
        dp_arr = _ptr_arr([_p(t) for t in dp])
        d.dp = ctypes.cast(dp_arr, ctypes.c_void_p)
    d.x = x.data_ptr()
    mxw = None
    if mx and mx_stack_ok(D, Hd, act):
        assert all(params[12 * l + i].dtype == torch.float32 for l in range(L) for i in (2, 4, 8, 10))
        mxw = mx_stack_weights(params)
        mxw.refresh()
        d.mx_w = ctypes.cast(mxw.dst, ctypes.c_void_p)
    nbytes = lib.mmae_stack_act_bytes(ctypes.byref(d))
    slab = _slab(nbytes, x.device)
    d.act, d.act_bytes = slab.data_ptr(), slab.numel()
    st = _stream()
    ws = stream_workspace(st, x.device)
    d.ws_main, d.ws_main_elems = ws.data_ptr(), ws.numel()
    check(lib.mmae_stack_fwd(ctypes.byref(d), st), 'stack_fwd')
    outs = []
    for l in range(L):
        off = lib.mmae_stack_out_offset(ctypes.byref(d), l)
        outs.append(slab[off:off + R * D * 4].view(torch.float32).view(R, D))
    s = StackState()
    s.desc, s.act, s.keep, s.x, s.L, s.shape, s.act_dtype, s.dp = d, slab, (w_arr, p_arr, dp_arr, wts, dp, mxw), x, L, (B, N, D, Hd), act, dp
    return outs, s


def stack_bwd(s: StackState, d_outs: Sequence[Optional[Tensor]], grads: Sequence[Optional[Tensor]], grad_acc: bool,
              side_handle: Optional[int], chunks: Optional[Sequence] = None, on_chunk=None):
    """Backward of stack_fwd.  d_outs[l]: f32 [R, D] gradient of block l's output or None (the last one is required);
    grads: 12 L destinations (None = not wanted).  chunks: [(lo, hi), ...] from the top down (default one call);
    on_chunk(lo, hi) runs after each chunk is enqueued.  Returns (dx f32 [R, D], tensors to keep alive until the side
    stream has drained)."""
    lib = _lib.load()
    B, N, D, Hd = s.shape
    R = B * N
    d = s.desc
    dev = s.x.device
    d_outs = [None if t is None else t.contiguous() for t in d_outs]
    do_arr = _ptr_arr([_p(t) for t in d_outs])
    g_arr = _ptr_arr([_p(g) for g in grads])
    d.d_out, d.g = ctypes.cast(do_arr, ctypes.c_void_p), ctypes.cast(g_arr, ctypes.c_void_p)
    d.grad_acc = int(grad_acc)
    dx = torch.empty((R, D), device=dev, dtype=torch.float32)
    d.dx = dx.data_ptr()
    tmp = _slab(lib.mmae_stack_tmp_bytes(ctypes.byref(d)), dev)
    d.tmp, d.tmp_bytes = tmp.data_ptr(), tmp.numel()
    st = _stream()
    wm = stream_workspace(st, dev)
    d.ws_main, d.ws_main_elems = wm.data_ptr(), wm.numel()
    sd = side_handle if side_handle is not None else st
    wsd = stream_workspace(sd, dev) if sd != st else wm
    d.ws_side, d.ws_side_elems = wsd.data_ptr(), wsd.numel()
    for lo, hi in (chunks or [(0, s.L)]):
        d.l_begin, d.l_end = lo, hi
        check(lib.mmae_stack_bwd(ctypes.byref(d), st, sd), 'stack_bwd')
        if on_chunk is not None:
            on_chunk(lo, hi)
    return dx, (tmp, s.act, s.keep, d_outs, do_arr, g_arr, grads)


class AdapterState:
    __slots__ = ('desc', 'act', 'keep', 'pat', 'ld_pat')


def adapter_composite_ok(enc: Tensor, act: torch.dtype, heads: int, D: int, n_q: int, NC: int, depth: int, T: int, KP: int) -> bool:
    hd = D // heads
    if not (_STACK[0] and _ADAPTER[0] and _COMPOSITE[0] and _FUSED_ATTN[0] and enc.is_cuda and hd in (32, 64) and n_q <= 256 and NC <= 256 and depth <= 8
            and T <= 7 and D % 8 == 0 and enc.shape[-1] % 8 == 0 and KP % 4 == 0):
        return False
    if act == torch.bfloat16:
        return True
    lds_bwd = 4 * (round_up(n_q, 32) + round_up(NC, 32)) * hd * 2 + 8 * round_up(n_q, 32)
    return act == torch.float32 and _f32_split() and lds_bwd <= 160 * 1024


class X3Weights:
    """Pre-split bf16 copies of the f32 weights of an fp32 output adapter (mmae_x3_prepare_weights): [n][3k] = [hi | lo | hi] for the
    forward product, [3n][k] = [hi ; lo ; hi] for dX, and the host table of {weight, copy, copy} triples mmae_adapter_desc.x3_w wants."""

    def __init__(self, w_list: Sequence[Tensor]):
        self.ws = list(w_list)
        dev = self.ws[0].device
        total, offs = 0, []
        for w in self.ws:
            nb = round_up(w.numel() * 3 * 2, 256)
            offs += [total, total + nb]
            total += 2 * nb
        self.buf = torch.empty((total,), device=dev, dtype=torch.uint8)
        base = self.buf.data_ptr()
        self.dst = _ptr_arr([base + o for o in offs])
        self.src = _ptr_arr([w.data_ptr() for w in self.ws])
        self.n_out = (ctypes.c_int32 * len(self.ws))(*[w.shape[0] for w in self.ws])
        self.k_in = (ctypes.c_int32 * len(self.ws))(*[w.numel() // w.shape[0] for w in self.ws])
        self.triples = _ptr_arr([v for i, w in enumerate(self.ws) for v in (w.data_ptr(), base + offs[2 * i], base + offs[2 * i + 1])])
        self.probe = tuple(w.data_ptr() for w in self.ws)

    def matches(self, w_list: Sequence[Tensor]) -> bool:
        return self.probe == tuple(w.data_ptr() for w in w_list)

    def refresh(self) -> None:
        check(_lib.load().mmae_x3_prepare_weights(len(self.ws), ctypes.cast(self.src, ctypes.c_void_p), self.n_out, self.k_in,
                                                  ctypes.cast(self.dst, ctypes.c_void_p), _stream()), 'mmae_x3_prepare_weights')


_X3_SETS = {}
# Off by default: measured +1.2 ms per cfg3 step (36.2 against 35.0 ms, profiles/r02_x3_presplit_ab.txt).  The fp32 adapter's products
# have K = 256: as ONE bf16 product over 3 K on the persistent 256 x 256 ping-pong kernel they are prologue / f32-epilogue bound and
# no faster than on the 128 x 128 kernel that splits in registers (three workgroups per CU hide each other's epilogues), and the
# split passes come on top.  MMAE_X3_PRESPLIT=1 turns it on for experiments.
_X3_PRESPLIT = _os.environ.get('MMAE_X3_PRESPLIT', '0') == '1'


def x3_weights(w_list: Sequence[Tensor]) -> X3Weights:
    key = (id(w_list[0]), len(w_list))
    m = _X3_SETS.get(key)
    if m is None or not m.matches(w_list):
        if len(_X3_SETS) > 16:
            _X3_SETS.clear()
        m = X3Weights(w_list)
        _X3_SETS[key] = m
    return m


def adapter_fwd(enc: Tensor, enc_act: Optional[Tensor], ids_keep: Tensor, ids_restore: Tensor, cfg, w_list: Sequence[Tensor],
                p_list: Sequence[Tensor], mask_token: Tensor, temb: Sequence[Optional[Tensor]], want_img: bool = True,
                store: Optional[torch.dtype] = None):
    """SpatialOutputAdapter.forward in one library call.  enc f32 [B, NC, Denc]; w_list / p_list in mmae_adapter_desc order.
    store: the activation dtype in memory when it is not cfg.act (torch.float16: an fp32 adapter in 'h16' mode, w_list fp16 copies).
    Returns (img or None, AdapterState)."""
    act = store if store is not None else cfg.act
    lib = _lib.load()
    B, NC, Denc = enc.shape
    T = len(cfg.task_offsets) - 1
    n_q = cfg.nh * cfg.nw if cfg.q_task < 0 else cfg.task_offsets[cfg.q_task + 1] - cfg.task_offsets[cfg.q_task]
    d = AdapterDesc()
    d.B, d.NC, d.Denc, d.D, d.heads, d.Hd, d.depth, d.T, d.q_task, d.G, d.n_q = (B, NC, Denc, cfg.D, cfg.heads, w_list[3].shape[0], cfg.depth, T,
                                                                                  cfg.q_task, cfg.G, n_q)
    d.C, d.nh, d.nw, d.ph, d.pw = cfg.C, cfg.nh, cfg.nw, cfg.ph, cfg.pw
    d.act_dtype, d.f32_gemm, d.eps = dcode(act), _f32_code(), cfg.eps
    offs = _i32_array(cfg.task_offsets)
    d.task_offsets_host = ctypes.cast(offs, ctypes.c_void_p)
    probe = (w_list[0].data_ptr(), w_list[-1].data_ptr(), p_list[0].data_ptr(), p_list[-1].data_ptr(), mask_token.data_ptr())
    w_arr, p_arr, te_arr = _PTRS.get(('adapter', id(p_list[0]), len(w_list), act), probe, lambda: (
        _ptr_arr([w.data_ptr() for w in w_list]), _ptr_arr([t.data_ptr() for t in p_list]), _ptr_arr([_p(t) for t in temb])))
    d.w, d.p, d.task_emb = ctypes.cast(w_arr, ctypes.c_void_p), ctypes.cast(p_arr, ctypes.c_void_p), ctypes.cast(te_arr, ctypes.c_void_p)
    d.mask_token, d.pos = mask_token.data_ptr(), cfg.pos.data_ptr()
    d.enc = enc.data_ptr()
    d.enc_act = enc.data_ptr() if act == torch.float32 else _p(enc_act)
    d.ids_keep, d.ids_restore = ids_keep.data_ptr(), ids_restore.data_ptr()
    x3 = None
    if _X3_PRESPLIT and act == torch.float32 and _F32_GEMM[0] == 'x3' and all(w.dtype == torch.float32 and w.is_contiguous() and w.numel() % (8 * w.shape[0]) == 0 for w in w_list):
        x3 = x3_weights(w_list)                 # forward / dX products of this fp32 adapter as ONE bf16 product over 3 K each
        x3.refresh()
        d.x3_w, d.x3_n = ctypes.cast(x3.triples, ctypes.c_void_p), len(w_list)
    # the prediction rows in their own allocation (mmae_adapter_desc.pat): a lazily written image / a patch-domain loss keep THEM
    # alive, not the whole activation slab
    KP = cfg.C * cfg.ph * cfg.pw
    pat = torch.empty((B * n_q, KP), device=enc.device, dtype=torch.float32)
    d.pat = pat.data_ptr()
    slab = _slab(lib.mmae_adapter_act_bytes(ctypes.byref(d)), enc.device)
    d.act, d.act_bytes = slab.data_ptr(), slab.numel()
    img = torch.empty((B, cfg.C, cfg.nh * cfg.ph, cfg.nw * cfg.pw), device=enc.device, dtype=torch.float32) if want_img else None
    d.img = _p(img)
    st = _stream()
    ws = stream_workspace(st, enc.device)
    d.ws_main, d.ws_main_elems = ws.data_ptr(), ws.numel()
    check(lib.mmae_adapter_fwd(ctypes.byref(d), st), 'adapter_fwd')
    s = AdapterState()
    s.pat = pat
    s.desc, s.act, s.ld_pat = d, slab, KP
    s.keep = (offs, w_arr, p_arr, te_arr, w_list, p_list, temb, mask_token, enc, enc_act, ids_keep, ids_restore, cfg.pos, x3)
    return img, s


def adapter_bwd(s: AdapterState, d_img: Optional[Tensor], d_pat: Optional[Tensor], grads: Sequence[Optional[Tensor]], grad_acc: bool,
                side_handle: Optional[int], dy_amax: Optional[Tensor] = None):
    """Backward of adapter_fwd from the image-domain gradient d_img (f32 [B,C,H,W]) or the patch-domain gradient d_pat (act
    dtype [B*n_q, ld]).  grads: destinations in mmae_adapter_desc.g order.  Returns (d_enc f32 [B, NC, Denc], keep-alives)."""
    lib = _lib.load()
    d = s.desc
    dev = s.act.device
    if d_img is not None:
        d_img = d_img.contiguous()
    d.d_img, d.d_pat = _p(d_img), _p(d_pat)
    d.ld_pat = d_pat.stride(0) if d_pat is not None else 0
    d.dy_amax = _p(dy_amax)            # f32 adapter with fp16-operand products: scales the gradient operands (None: they run split-bf16)
    g_arr = _ptr_arr([_p(g) for g in grads])
    d.g, d.grad_acc = ctypes.cast(g_arr, ctypes.c_void_p), int(grad_acc)
    d_enc = torch.empty((d.B, d.NC, d.Denc), device=dev, dtype=torch.float32)
    d.d_enc = d_enc.data_ptr()
    tmp = _slab(lib.mmae_adapter_tmp_bytes(ctypes.byref(d)), dev)
    d.tmp, d.tmp_bytes = tmp.data_ptr(), tmp.numel()
    st = _stream()
    wm = stream_workspace(st, dev)
    d.ws_main, d.ws_main_elems = wm.data_ptr(), wm.numel()
    sd = side_handle if side_handle is not None else st
    wsd = stream_workspace(sd, dev) if sd != st else wm
    d.ws_side, d.ws_side_elems = wsd.data_ptr(), wsd.numel()
    check(lib.mmae_adapter_bwd(ctypes.byref(d), st, sd), 'adapter_bwd')
    return d_enc, (tmp, s.act, s.keep, d_img, d_pat, g_arr, grads, dy_amax)


# ------------------------------------------------------------------- row kernels --
def layernorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, out_dtype: torch.dtype):
    _require_gpu(x, 'layernorm input')
    R, D = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty((R, D), device=x.device, dtype=out_dtype)
    mean = torch.empty((R,), device=x.device, dtype=torch.float32)
    rstd = torch.empty((R,), device=x.device, dtype=torch.float32)
    check(_lib.load().mmae_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), dcode(out_dtype),
                                         mean.data_ptr(), rstd.data_ptr(), R, D, eps, _stream()), 'layernorm_fwd')
    return y, mean, rstd


def layernorm_bwd_part(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dx_in: Optional[Tensor],
                       act_dtype: Optional[torch.dtype]):
    """returns dx (f32, = dx_in + LN'(dy)), dx_act (act copy or None) and the UNREDUCED per-workgroup partial sums
    part f32 [nblk, 3*D] = [dgamma | dbeta | colsum(dx)] (reduce with colsum / colsum_scatter, off the critical path)."""
    R, D = x.shape
    lib = _lib.load()
    nblk = lib.mmae_layernorm_bwd_nblk(R)
    part = torch.empty((nblk, 3 * D), device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x)
    dx_act = torch.empty((R, D), device=x.device, dtype=act_dtype) if act_dtype is not None else None
    check(lib.mmae_layernorm_bwd(dy.data_ptr(), dcode(dy.dtype), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                 rstd.data_ptr(), _p(dx_in), dx.data_ptr(), _p(dx_act),
                                 dcode(act_dtype) if act_dtype is not None else F32, part.data_ptr(), R, D, _stream()),
          'layernorm_bwd')
    return dx, dx_act, part


def layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dx_in: Optional[Tensor],
                  act_dtype: Optional[torch.dtype]):
    """returns dx (f32, = dx_in + LN'(dy)), dx_act (act copy or None), dgamma, dbeta, colsum(dx) (f32 [D] each)."""
    D = x.shape[1]
    dx, dx_act, part = layernorm_bwd_part(dy, x, gamma, mean, rstd, dx_in, act_dtype)
    dgb = torch.empty((3 * D,), device=x.device, dtype=torch.float32)
    colsum(part, dgb, False)
    return dx, dx_act, dgb[:D], dgb[D:2 * D], dgb[2 * D:]


def colsum(dy: Tensor, out: Tensor, accumulate: bool) -> Tensor:
    """out[n] (+)= sum_m dy[m][n]"""
    M, N = dy.shape
    lib = _lib.load()
    ws = torch.empty((lib.mmae_colsum_ws_elems(M, N),), device=dy.device, dtype=torch.float32)
    check(lib.mmae_colsum(dy.data_ptr(), dcode(dy.dtype), M, N, dy.stride(0), out.data_ptr(), int(accumulate),
                          ws.data_ptr(), _stream()), 'colsum')
    return out


def colsum_scatter(dy: Tensor, seg_w: int, dsts: Sequence[Optional[Tensor]], accumulate: bool) -> None:
    """column sums of dy [M, len(dsts)*seg_w]; segment i (columns i*seg_w .. (i+1)*seg_w) (+)= into dsts[i] (f32, seg_w
    elements, contiguous); None drops the segment."""
    M, N = dy.shape
    assert 1 <= len(dsts) <= 8 and seg_w * len(dsts) >= N, (N, seg_w, len(dsts))
    for d in dsts:
        assert d is None or (d.dtype == torch.float32 and d.numel() >= min(seg_w, N) and d.is_contiguous())
    lib = _lib.load()
    ws = torch.empty((lib.mmae_colsum_ws_elems(M, N),), device=dy.device, dtype=torch.float32)
    arr = (ctypes.c_void_p * len(dsts))(*[_p(d) for d in dsts])
    check(lib.mmae_colsum_scatter(dy.data_ptr(), dcode(dy.dtype), M, N, dy.stride(0), seg_w, ctypes.cast(arr, ctypes.c_void_p),
                                  len(dsts), int(accumulate), ws.data_ptr(), _stream()), 'colsum_scatter')


def colsum_batch(jobs: Sequence[tuple], accumulate: bool, unscale: Optional[Tensor] = None) -> None:
    """Up to 8 colsum_scatter reductions in ONE launch (mmae_colsum_batch).  jobs: (src [rows, cols] f32 / bf16 / fp16 (row stride = stride(0)),
    seg_w, [dst tensors f32 or None]).  unscale: the dy_amax scalar of an fp16-storage adapter whose (scaled) gradients the sources hold."""
    from ._lib import ColsumJob
    assert 1 <= len(jobs) <= 8
    arr = (ColsumJob * len(jobs))()
    for q, (src, seg_w, dsts) in zip(arr, jobs):
        rows, cols = src.shape
        assert 1 <= len(dsts) <= 8 and seg_w * len(dsts) >= cols
        q.src, q.dtype, q.cols, q.rows, q.ld, q.seg_w, q.nseg = src.data_ptr(), dcode(src.dtype), cols, rows, src.stride(0), seg_w, len(dsts)
        for i, d in enumerate(dsts):
            assert d is None or (d.dtype == torch.float32 and d.is_contiguous())
            q.dst[i] = _p(d)
        q.unscale = _p(unscale)
    lib = _lib.load()
    n_ws = int(lib.mmae_colsum_batch_ws_elems(ctypes.cast(arr, ctypes.c_void_p), len(jobs)))
    ws = torch.empty((max(n_ws, 4),), device=jobs[0][0].device, dtype=torch.float32)
    check(lib.mmae_colsum_batch(ctypes.cast(arr, ctypes.c_void_p), len(jobs), int(accumulate), ws.data_ptr(), ws.numel(), _stream()), 'colsum_batch')


def reduce_partials(part: Tensor, out: Tensor, accumulate: bool) -> Tensor:
    nrows = part.shape[0]
    ncols = part.numel() // nrows
    check(_lib.load().mmae_colsum_partials(part.data_ptr(), out.data_ptr(), nrows, ncols, int(accumulate), _stream()),
          'colsum_partials')
    return out


def softmax_fwd(S: Tensor, P: Tensor, rows: int, n: int, scale: float) -> None:
    check(_lib.load().mmae_softmax_fwd(S.data_ptr(), S.shape[-1], P.data_ptr(), dcode(P.dtype), P.shape[-1], rows, n, scale,
                                       _stream()), 'softmax_fwd')


def softmax_bwd(P: Tensor, dP: Tensor, dS: Tensor, rows: int, n: int, scale: float) -> None:
    check(_lib.load().mmae_softmax_bwd(P.data_ptr(), dcode(P.dtype), P.shape[-1], dP.data_ptr(), dP.shape[-1], dS.data_ptr(),
                                       dS.shape[-1], rows, n, scale, _stream()), 'softmax_bwd')


def cast(x: Tensor, dtype: torch.dtype) -> Tensor:
    """f32 <-> bf16 cast kernel (identity if already that dtype)."""
    if x.dtype == dtype:
        return x
    _require_gpu(x, 'cast input')
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=dtype)
    lib = _lib.load()
    if dtype == torch.bfloat16:
        check(lib.mmae_cast_f32_to_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), 'cast_f32_to_bf16')
    else:
        check(lib.mmae_cast_bf16_to_f32(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), 'cast_bf16_to_f32')
    return y


def cast_f16(x: Tensor, scale_amax: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """fp16 storage (mmae.h MMAE_F16): f32 -> fp16 (times S(scale_amax): a loss gradient entering an 'h16' adapter's backward) or
    fp16 -> f32 (times 1/S).  scale_amax: the adapter's dy_amax device scalar, None = values as they are."""
    _require_gpu(x, 'cast input')
    x = x.contiguous()
    lib = _lib.load()
    if x.dtype == torch.float32:
        y = out if out is not None else torch.empty(x.shape, device=x.device, dtype=torch.float16)
        assert y.dtype == torch.float16 and y.numel() == x.numel() and y.is_contiguous()
        check(lib.mmae_cast_f32_to_f16(x.data_ptr(), y.data_ptr(), x.numel(), _p(scale_amax), _stream()), 'cast_f32_to_f16')
        return y
    assert x.dtype == torch.float16
    y = out if out is not None else torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(lib.mmae_cast_f16_to_f32(x.data_ptr(), y.data_ptr(), x.numel(), _p(scale_amax), _stream()), 'cast_f16_to_f32')
    return y


class H16Weights:
    """fp16 copies of the f32 weights of an fp32 output adapter in 'h16' mode, refreshed once per forward.  The weights of one adapter
    sit close together in the parameter arena: one cast launch over the range that spans them (biases and LayerNorm vectors in
    between are cast along, unused) instead of one per weight; scattered weights fall back to a launch each."""

    def __init__(self, w_list: Sequence[Tensor]):
        self.ws = list(w_list)
        dev = self.ws[0].device
        self.probe = tuple(w.data_ptr() for w in self.ws)
        lo = min(self.probe)
        hi = max(w.data_ptr() + w.numel() * 4 for w in self.ws)
        total = sum(w.numel() for w in self.ws)
        same = all(w.dtype == torch.float32 and w.is_contiguous() and w.untyped_storage().data_ptr() == self.ws[0].untyped_storage().data_ptr()
                   for w in self.ws)
        self.span = same and lo % 16 == 0 and (hi - lo) // 4 <= 2 * total + 4096
        if self.span:
            n = (hi - lo) // 4
            self.src = torch.empty(0, device=dev, dtype=torch.float32).set_(self.ws[0].untyped_storage(), (lo - self.ws[0].untyped_storage().data_ptr()) // 4, (n,))
            self.buf = torch.empty((n,), device=dev, dtype=torch.float16)
            self.copies = [self.buf[(w.data_ptr() - lo) // 4:(w.data_ptr() - lo) // 4 + w.numel()].view(w.shape) for w in self.ws]
        else:
            self.copies = [torch.empty(w.shape, device=dev, dtype=torch.float16) for w in self.ws]

    def matches(self, w_list: Sequence[Tensor]) -> bool:
        return self.probe == tuple(w.data_ptr() for w in w_list)

    def refresh(self) -> Sequence[Tensor]:
        if self.span:
            cast_f16(self.src, out=self.buf)
        else:
            for w, c in zip(self.ws, self.copies):
                cast_f16(w.detach(), out=c)
        return self.copies


_H16_SETS = {}


def h16_weights(w_list: Sequence[Tensor]) -> H16Weights:
    # keyed by where the first weight LIVES (the callers hand in fresh detach() views every forward: their id() is not stable)
    key = (w_list[0].data_ptr(), len(w_list))
    s = _H16_SETS.get(key)
    if s is None or not s.matches(w_list):
        if len(_H16_SETS) > 16:
            _H16_SETS.clear()
        s = H16Weights(w_list)
        _H16_SETS[key] = s
    return s


def adapter_h16_ok(enc: Tensor, heads: int, D: int, Hd: int, n_q: int, NC: int, depth: int, T: int, KP: int) -> bool:
    """fp16 storage for an fp32 output adapter: the composite call with every width on the fp16 ping-pong kernels' grid"""
    return (adapter_composite_ok(enc, torch.bfloat16, heads, D, n_q, NC, depth, T, KP) and D % 32 == 0 and Hd % 32 == 0
            and enc.shape[-1] % 32 == 0 and KP % 8 == 0)


def cast_into(src: Tensor, dst: Tensor) -> None:
    lib = _lib.load()
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
    check(lib.mmae_cast_f32_to_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), 'cast_f32_to_bf16')


def rowscale_add(resid: Tensor, y: Tensor, s: Tensor, N: int) -> Tensor:
    """resid + s[row // N] * y on f32 [R, D] activations (stochastic depth: per-sample scale of a residual branch)."""
    R, D = y.shape
    out = torch.empty((R, D), device=y.device, dtype=torch.float32)
    check(_lib.load().mmae_rowscale_add(resid.data_ptr(), y.data_ptr(), s.data_ptr(), out.data_ptr(), R, N, D, _stream()), 'rowscale_add')
    return out


def rowscale_cast(x: Tensor, s: Tensor, N: int, dtype: torch.dtype) -> Tensor:
    """cast(s[row // N] * x) for f32 x [R, D]."""
    R, D = x.shape
    out = torch.empty((R, D), device=x.device, dtype=dtype)
    check(_lib.load().mmae_rowscale_cast(x.data_ptr(), s.data_ptr(), out.data_ptr(), dcode(dtype), R, N, D, _stream()), 'rowscale_cast')
    return out


def add_n(xs: Sequence[Tensor]) -> Tensor:
    """Sum of 1 <= n <= 8 f32 tensors of one shape in ONE pass (mmae_add_n_f32), in index order."""
    xs = [x.contiguous() for x in xs]
    assert all(x.dtype == torch.float32 and x.shape == xs[0].shape for x in xs) and xs[0].numel() % 4 == 0
    out = torch.empty_like(xs[0])
    ptrs = (ctypes.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    check(_lib.load().mmae_add_n_f32(out.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p), len(xs), out.numel(), _stream()), 'add_n')
    return out


def axpy_(y: Tensor, x: Tensor, a: float = 1.0) -> Tensor:
    assert y.dtype == torch.float32 and x.dtype == torch.float32 and y.numel() == x.numel()
    check(_lib.load().mmae_axpy_f32(y.data_ptr(), x.data_ptr(), a, y.numel(), _stream()), 'axpy')
    return y


# --------------------------------------------------------------------------- dropout --
def _dropout_keep(shape, p: float, device) -> Tensor:
    """The keep mask of an nn.Dropout(p) call (uint8, 1 = kept): torch.rand(shape) >= p on the device generator.  Tests replace this
    hook to replay the masks the reference drew (tests/golden/make_golden_dropout.py)."""
    return (torch.rand(shape, device=device) >= p).to(torch.uint8)


def dropout_apply(x: Tensor, keep: Tensor, scale: float, out_dtype: Optional[torch.dtype] = None, resid: Optional[Tensor] = None,
                  row_scale: Optional[Tensor] = None, per: int = 0, out: Optional[Tensor] = None) -> Tensor:
    """mmae_dropout: out = [resid +] keep ? x * scale [* row_scale[i // per]] : 0 -- forward and backward of nn.Dropout alike
    (multimae_utils.py:138-155, 158-214).  keep: uint8 of x's shape; row_scale: a stochastic-depth scale per sample (per elements each)."""
    _require_gpu(x, 'dropout input')
    x = x.contiguous()
    keep = keep.contiguous()
    assert keep.dtype == torch.uint8 and keep.numel() == x.numel() and x.numel() % 4 == 0, (keep.dtype, keep.shape, x.shape)
    odt = out_dtype or (torch.float32 if resid is not None else x.dtype)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=odt)
    check(_lib.load().mmae_dropout(x.data_ptr(), dcode(x.dtype), keep.data_ptr(), float(scale), _p(row_scale), int(per), _p(resid), out.data_ptr(),
                                   dcode(out.dtype), x.numel(), _stream()), 'dropout')
    return out


# ------------------------------------------------------------- attention (unfused) --
class AttnView:
    """A (B, N, heads, hd) operand living inside a packed 2-D activation [B*N, ld] at column `col`."""
    __slots__ = ('t', 'col', 'ld', 'N')

    def __init__(self, t: Tensor, col: int, ld: int, N: int):
        self.t, self.col, self.ld, self.N = t, col, ld, N


_FUSED_ATTN = [True]


def set_fused_attention(flag: bool) -> None:
    """bf16 attention: fused single-kernel path (default) or the batched-GEMM + softmax path."""
    _FUSED_ATTN[0] = bool(flag)


def _ptr(v: AttnView) -> int:
    return v.t.data_ptr() + v.col * v.t.element_size()


def _fusable(q: AttnView, k: AttnView, hd: int) -> bool:
    if not (_FUSED_ATTN[0] and hd in (32, 64) and q.N <= 256 and k.N <= 256):
        return False
    if q.t.dtype in (torch.bfloat16, torch.float16):
        return True
    # f32 activations: only where the surrounding GEMMs are split-bf16 too (fp32 adapters in speed mode); the backward's
    # eight hi/lo tiles must fit the 160 KB LDS
    lds_bwd = 4 * (round_up(q.N, 32) + round_up(k.N, 32)) * hd * 2 + 8 * round_up(q.N, 32)
    return q.t.dtype == torch.float32 and _f32_split() and lds_bwd <= 160 * 1024


def attention_fwd(q: AttnView, k: AttnView, v: AttnView, out: AttnView, B: int, H: int, hd: int, scale: float, f16: bool = False,
                  drop_p: float = 0.0):
    """softmax(q k^T * scale) v.  Returns the state backward needs: ('fused', lse) or ('gemm', P).
    multimae_utils.py:175-179 / 206-210.  f16 (f32 tensors only): fp16-operand products instead of the split-bf16 ones.
    drop_p > 0: attn_drop (multimae_utils.py:177 / 208) on the materialised probabilities -- the batched-GEMM path, state
    ('gemm', P, P_dropped, keep, 1 / (1 - p))."""
    Nq, Nk = q.N, k.N
    dev, act = q.t.device, q.t.dtype
    if drop_p <= 0.0 and _fusable(q, k, hd):
        lse = torch.empty((B, H, Nq), device=dev, dtype=torch.float32)
        fwd = _lib.load().mmae_attn_fwd if act == torch.bfloat16 else (_lib.load().mmae_attn_fwd_f16 if act == torch.float16 else (
            _lib.load().mmae_attn_fwd_f32f16 if f16 else _lib.load().mmae_attn_fwd_f32x3))
        check(fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), lse.data_ptr(), B, H, Nq, Nk, hd,
                                        Nq * q.ld, q.ld, Nk * k.ld, k.ld, Nk * v.ld, v.ld, Nq * out.ld, out.ld, scale, _stream()),
              'attn_fwd')
        return ('fused', lse)
    Np = round_up(Nk, 8)
    S = torch.empty((B, H, Nq, Np), device=dev, dtype=torch.float32)
    gemm(q.t, k.t, S, Nq, Nk, hd, lda=q.ld, ldb=k.ld, ldc=Np, a_off=q.col, b_off=k.col,
         batch=B * H, batch_inner=H, sA=(Nq * q.ld, hd), sB=(Nk * k.ld, hd), sC=(H * Nq * Np, Nq * Np))
    P = torch.empty((B, H, Nq, Np), device=dev, dtype=act)
    softmax_fwd(S, P, B * H * Nq, Nk, scale)
    Pv, state = P, ('gemm', P)
    if drop_p > 0.0:
        keep = _dropout_keep((B, H, Nq, Nk), drop_p, dev)
        if Np != Nk:                                       # the padded key columns hold zeros either way
            kp = torch.zeros((B, H, Nq, Np), device=dev, dtype=torch.uint8)
            kp[..., :Nk] = keep
            keep = kp
        Pv = dropout_apply(P, keep, 1.0 / (1.0 - drop_p))
        state = ('gemm', P, Pv, keep, 1.0 / (1.0 - drop_p))
    gemm(Pv, v.t, out.t, Nq, hd, Nk, lda=Np, ldb=v.ld, ldc=out.ld, b_trans=True, b_off=v.col, c_off=out.col,
         batch=B * H, batch_inner=H, sA=(H * Nq * Np, Nq * Np), sB=(Nk * v.ld, hd), sC=(Nq * out.ld, hd))
    return state


def xattn_fwd_fused(qn: Tensor, cn: Tensor, wq: Tensor, bq: Tensor, wkv: Tensor, bkv: Tensor, B: int, Nq: int, Nk: int, heads: int = 8):
    """CrossAttention.forward up to (not including) its output projection, multimae_utils.py:199-214, as ONE launch (mmae_xattn_fwd_fused): qn [B * Nq, D]
    and cn [B * Nk, D] bf16 (the normalised queries / context), nn.Linear weights wq [D, D], wkv [2 D, D] bf16, biases f32; D = heads * 32 = 256, Nk <= 128.
    Returns (q, kv, out, lse) -- q and kv are what the backward differentiates."""
    D = qn.shape[1]
    dev = qn.device
    q = torch.empty((B * Nq, D), device=dev, dtype=torch.bfloat16)
    kv = torch.empty((B * Nk, 2 * D), device=dev, dtype=torch.bfloat16)
    out = torch.empty((B * Nq, D), device=dev, dtype=torch.bfloat16)
    lse = torch.empty((B, heads, Nq), device=dev, dtype=torch.float32)
    for t in (qn, cn, wq, wkv):
        assert t.dtype == torch.bfloat16 and t.is_contiguous()
    check(_lib.load().mmae_xattn_fwd_fused(qn.data_ptr(), cn.data_ptr(), wq.data_ptr(), bq.data_ptr(), wkv.data_ptr(), bkv.data_ptr(), q.data_ptr(), kv.data_ptr(),
                                           out.data_ptr(), lse.data_ptr(), B, heads, Nq, Nk, D, 1.0 / math.sqrt(D // heads), _stream()), 'xattn_fwd_fused')
    return q, kv, out, lse


def attention_bwd(q: AttnView, k: AttnView, v: AttnView, state, out: AttnView, d_out: AttnView, dq: AttnView, dk: AttnView,
                  dv: AttnView, B: int, H: int, hd: int, scale: float, f16: bool = False, dy_amax: Optional[Tensor] = None) -> None:
    Nq, Nk = q.N, k.N
    if state[0] == 'fused' and f16 and q.t.dtype == torch.float32:
        check(_lib.load().mmae_attn_bwd_f32f16(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(d_out), state[1].data_ptr(), _ptr(dq), _ptr(dk), _ptr(dv),
                                               B, H, Nq, Nk, hd, Nq * q.ld, q.ld, Nk * k.ld, k.ld, Nk * v.ld, v.ld, Nq * out.ld, out.ld, Nq * dq.ld,
                                               dq.ld, Nk * dk.ld, dk.ld, Nk * dv.ld, dv.ld, scale, _p(dy_amax), _stream()), 'attn_bwd_f32f16')
        return
    if state[0] == 'fused':
        assert d_out.ld == out.ld
        bwd = _lib.load().mmae_attn_bwd if q.t.dtype == torch.bfloat16 else (_lib.load().mmae_attn_bwd_f16 if q.t.dtype == torch.float16 else _lib.load().mmae_attn_bwd_f32x3)
        check(bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(d_out), state[1].data_ptr(), _ptr(dq), _ptr(dk),
                                        _ptr(dv), B, H, Nq, Nk, hd, Nq * q.ld, q.ld, Nk * k.ld, k.ld, Nk * v.ld, v.ld, Nq * out.ld,
                                        out.ld, Nq * dq.ld, dq.ld, Nk * dk.ld, dk.ld, Nk * dv.ld, dv.ld, scale, _stream()), 'attn_bwd')
        return
    P = state[1]
    Np = P.shape[-1]
    dev, act = q.t.device, q.t.dtype
    bh = dict(batch=B * H, batch_inner=H)
    sP = (H * Nq * Np, Nq * Np)
    # dP = dO V^T
    dP = torch.empty((B, H, Nq, Np), device=dev, dtype=torch.float32)
    gemm(d_out.t, v.t, dP, Nq, Nk, hd, lda=d_out.ld, ldb=v.ld, ldc=Np, a_off=d_out.col, b_off=v.col,
         sA=(Nq * d_out.ld, hd), sB=(Nk * v.ld, hd), sC=sP, **bh)
    Pv = P
    if len(state) > 2:                                     # attn_drop: O = (keep * P / (1 - p)) V
        Pv = state[2]
        dP = dropout_apply(dP, state[3], state[4])
    dS = torch.empty((B, H, Nq, Np), device=dev, dtype=act)
    softmax_bwd(P, dP, dS, B * H * Nq, Nk, scale)
    # dV = P^T dO
    gemm(Pv, d_out.t, dv.t, Nk, hd, Nq, lda=Np, ldb=d_out.ld, ldc=dv.ld, a_trans=True, b_trans=True, b_off=d_out.col,
         c_off=dv.col, sA=sP, sB=(Nq * d_out.ld, hd), sC=(Nk * dv.ld, hd), **bh)
    # dQ = dS K
    gemm(dS, k.t, dq.t, Nq, hd, Nk, lda=Np, ldb=k.ld, ldc=dq.ld, b_trans=True, b_off=k.col, c_off=dq.col,
         sA=sP, sB=(Nk * k.ld, hd), sC=(Nq * dq.ld, hd), **bh)
    # dK = dS^T Q
    gemm(dS, q.t, dk.t, Nk, hd, Nq, lda=Np, ldb=q.ld, ldc=dk.ld, a_trans=True, b_trans=True, b_off=q.col,
         c_off=dk.col, sA=sP, sB=(Nq * q.ld, hd), sC=(Nk * dk.ld, hd), **bh)


# ----------------------------------------------------------------- token kernels --
def _i32_array(vals: Sequence[int]):
    return (ctypes.c_int32 * len(vals))(*vals)


def mask_sample(samples_per_task: Tensor, task_noise: Tensor, all_noise: Tensor, task_offsets: Sequence[int], n_keep: int):
    """Deterministic core of generate_random_masks (multimae.py:191-216) on the GPU."""
    _require_gpu(task_noise, 'mask noise')
    B, Ntot = all_noise.shape
    T = len(task_offsets) - 1
    dev = all_noise.device
    mask_all = torch.empty((B, Ntot), device=dev, dtype=torch.int64)
    ids_keep = torch.empty((B, n_keep), device=dev, dtype=torch.int64)
    ids_restore = torch.empty((B, Ntot), device=dev, dtype=torch.int64)
    # pinned + non_blocking: a pageable H2D copy blocks the host until the stream has drained, i.e. once per step the host
    # would lose its whole launch lead (measured: ~1.2 ms of GPU idle at every step start behind the CPU Dirichlet draw)
    spt = samples_per_task.to(dtype=torch.int64).contiguous()
    if spt.device.type == 'cpu' and dev.type == 'cuda':
        spt = spt.pin_memory().to(dev, non_blocking=True)
    elif spt.device != dev:
        spt = spt.to(dev)
    check(_lib.load().mmae_mask_sample(spt.data_ptr(), task_noise.contiguous().data_ptr(), all_noise.contiguous().data_ptr(),
                                       ctypes.cast(_i32_array(task_offsets), ctypes.c_void_p), T, B, Ntot, n_keep,
                                       mask_all.data_ptr(), ids_keep.data_ptr(), ids_restore.data_ptr(), _stream()),
          'mask_sample')
    return mask_all, ids_keep, ids_restore


def patch_rows(srcs: Sequence[dict], task_offsets: Sequence[int], sel: Tensor, B: int, n_sel: int, Ktot: int,
               dtype: torch.dtype) -> Tensor:
    """srcs: one dict per task: data (tensor), emb (tensor|None), kind, C, H, W, ph, pw, k_off."""
    T = len(srcs)
    arr = (PatchSrc * T)()
    for i, s in enumerate(srcs):
        _require_gpu(s['data'], 'input image')
        arr[i].data = s['data'].data_ptr()
        arr[i].emb = _p(s.get('emb'))
        arr[i].kind, arr[i].C, arr[i].H, arr[i].W = s['kind'], s['C'], s['H'], s['W']
        arr[i].ph, arr[i].pw, arr[i].k_off = s['ph'], s['pw'], s['k_off']
        arr[i].n_cls = s['emb'].shape[0] if s.get('emb') is not None else 0
    rows = torch.empty((B * n_sel, Ktot), device=sel.device, dtype=dtype)
    check(_lib.load().mmae_patch_rows(ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(_i32_array(task_offsets), ctypes.c_void_p),
                                      T, sel.data_ptr(), rows.data_ptr(), dcode(dtype), B, n_sel, Ktot, _stream()), 'patch_rows')
    return rows


def _patch_src_array(srcs: Sequence[dict]):
    arr = (PatchSrc * len(srcs))()
    for i, s in enumerate(srcs):
        _require_gpu(s['data'], 'input image')
        arr[i].data = s['data'].data_ptr()
        arr[i].emb = _p(s.get('emb'))
        arr[i].kind, arr[i].C, arr[i].H, arr[i].W = s['kind'], s['C'], s['H'], s['W']
        arr[i].ph, arr[i].pw, arr[i].k_off = s['ph'], s['pw'], s['k_off']
        arr[i].n_cls = s['emb'].shape[0] if s.get('emb') is not None else 0
    return arr


_FUSED_EMBED = [_os.environ.get('MMAE_FUSED_EMBED', '1') != '0']


def patch_embed_supported(srcs: Sequence[dict], n_sel: int, D: int) -> bool:
    """True if mmae_patch_embed_fwd (ONE kernel: gather + bf16 MFMA + bias + pos-emb, csrc/embed.hip) takes this geometry."""
    if not _FUSED_EMBED[0]:
        return False
    arr = _patch_src_array(srcs)
    return bool(_lib.load().mmae_patch_embed_supported(ctypes.cast(arr, ctypes.c_void_p), len(srcs), n_sel, D))


def patch_embed_fwd(srcs: Sequence[dict], weights_bf16: Sequence[Tensor], biases: Sequence[Tensor], poss: Sequence[Tensor],
                    task_offsets: Sequence[int], sel: Tensor, global_tok: Optional[Tensor], B: int, n_sel: int, G: int, D: int, Ktot: int,
                    want_rows: bool = True):
    """Fused patch embedding: returns (tok f32 [B, n_sel+G, D], rows bf16 [B*n_sel, Ktot] or None)."""
    arr = _patch_src_array(srcs)
    for w, s in zip(weights_bf16, srcs):
        if w.dtype != torch.bfloat16 or not w.is_contiguous() or w.numel() != D * s['C'] * s['ph'] * s['pw']:
            raise ValueError('patch_embed_fwd: weights must be contiguous bf16 [D, C*ph*pw]')
    tok = torch.empty((B, n_sel + G, D), device=sel.device, dtype=torch.float32)
    rows = torch.empty((B * n_sel, Ktot), device=sel.device, dtype=torch.bfloat16) if want_rows else None
    check(_lib.load().mmae_patch_embed_fwd(ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(_ptr_array(weights_bf16), ctypes.c_void_p),
                                           ctypes.cast(_ptr_array(biases), ctypes.c_void_p), ctypes.cast(_ptr_array(poss), ctypes.c_void_p),
                                           ctypes.cast(_i32_array(task_offsets), ctypes.c_void_p), len(srcs), sel.data_ptr(), _p(global_tok),
                                           tok.data_ptr(), _p(rows), B, n_sel, G, D, Ktot, _stream()), 'patch_embed_fwd')
    return tok, rows


def semseg_emb_bwd(d_rows: Tensor, cls: Tensor, sel: Tensor, d_emb: Tensor, *, B, H, W, E, ph, pw, n_sel, k_off, tok_off,
                   n_patches, n_cls, accumulate: bool = True, deterministic: bool = True) -> None:
    """Gradient of the class-embedding table.  deterministic (default, round 6): fixed summation order, no atomics (mmae_semseg_emb_bwd_det) for
    bf16 / f32 rows whose table fits the LDS; accumulate=False stores the gradient, so d_emb needs no zero-fill.  Otherwise (fp16 rows, huge tables)
    the float-atomic form, which ADDS into d_emb (accumulate=False then zero-fills it first)."""
    lib = _lib.load()
    ws_elems = int(lib.mmae_semseg_emb_bwd_ws_elems(B, n_sel, E, n_cls)) if (deterministic and d_rows.dtype in (torch.bfloat16, torch.float32)) else -1
    if ws_elems > 0:
        ws = torch.empty(ws_elems, device=d_rows.device, dtype=torch.float32)
        check(lib.mmae_semseg_emb_bwd_det(d_rows.data_ptr(), dcode(d_rows.dtype), d_rows.stride(0), cls.data_ptr(), sel.data_ptr(), d_emb.data_ptr(),
                                          B, H, W, E, ph, pw, n_sel, k_off, tok_off, n_patches, n_cls, ws.data_ptr(), ws_elems, int(bool(accumulate)),
                                          _stream()), 'semseg_emb_bwd_det')
        return
    if not accumulate:
        d_emb.zero_()
    check(lib.mmae_semseg_emb_bwd(d_rows.data_ptr(), dcode(d_rows.dtype), d_rows.stride(0), cls.data_ptr(), sel.data_ptr(),
                                  d_emb.data_ptr(), B, H, W, E, ph, pw, n_sel, k_off, tok_off, n_patches, n_cls,
                                  _stream()), 'semseg_emb_bwd')


def _ptr_array(ts: Sequence[Tensor]):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def semseg_avg_emb_fwd(x: Tensor, class_emb: Tensor, ph: int, pw: int) -> Tensor:
    """[B, H, W] class ids -> f32 [B, E, H/ph, W/pw]: the class-embedding image resized by 1 / patch (interpolate_class_emb)."""
    _require_gpu(x, 'semseg input')
    B, H, W = x.shape
    n_cls, E = class_emb.shape
    out = torch.empty((B, E, H // ph, W // pw), device=x.device, dtype=torch.float32)
    check(_lib.load().mmae_semseg_avg_emb_fwd(x.data_ptr(), class_emb.data_ptr(), out.data_ptr(), B, H, W, E, ph, pw, n_cls, _stream()), 'semseg_avg_emb_fwd')
    return out


def semseg_avg_emb_bwd(d_img: Tensor, x: Tensor, n_cls: int, ph: int, pw: int, pad_idx: int = -1) -> Tensor:
    B, H, W = x.shape
    E = d_img.shape[1]
    d_emb = torch.zeros((n_cls, E), device=x.device, dtype=torch.float32)
    check(_lib.load().mmae_semseg_avg_emb_bwd(d_img.data_ptr(), x.data_ptr(), d_emb.data_ptr(), B, H, W, E, ph, pw, n_cls, pad_idx, _stream()), 'semseg_avg_emb_bwd')
    return d_emb


def rows_to_image(d_rows: Tensor, k_off: int, sel: Tensor, B: int, n_sel: int, tok_off: int, n_patches: int, C: int, H: int, W: int, ph: int,
                  pw: int) -> Tensor:
    """data gradient of the patch embedding for one image-like input: f32 [B, C, H, W] (zeros where no token was selected)."""
    d_img = torch.zeros((B, C, H, W), device=d_rows.device, dtype=torch.float32)
    check(_lib.load().mmae_rows_to_image(d_rows.data_ptr(), d_rows.stride(0), k_off, sel.data_ptr(), d_img.data_ptr(), B, n_sel, tok_off, n_patches,
                                         C, H, W, ph, pw, _stream()), 'rows_to_image')
    return d_img


def pos_emb_bwd(d_tok: Tensor, sel: Tensor, n_pos: int, B: int, n_sel: int, G: int, D: int) -> Tensor:
    """d_pos f32 [n_pos, D]: the selected tokens' gradients summed per position (learnable positional embeddings)."""
    d_pos = torch.zeros((n_pos, D), device=d_tok.device, dtype=torch.float32)
    check(_lib.load().mmae_pos_emb_bwd(d_tok.data_ptr(), sel.data_ptr(), d_pos.data_ptr(), B, n_sel, G, D, n_pos, _stream()), 'mmae_pos_emb_bwd')
    return d_pos


def tokens_assemble(proj: Tensor, biases: Sequence[Tensor], poss: Sequence[Tensor], task_offsets: Sequence[int], sel: Tensor,
                    global_tok: Optional[Tensor], B: int, n_sel: int, G: int, D: int) -> Tensor:
    tok = torch.empty((B, n_sel + G, D), device=proj.device, dtype=torch.float32)
    check(_lib.load().mmae_tokens_assemble(tok.data_ptr(), proj.data_ptr(), ctypes.cast(_ptr_array(biases), ctypes.c_void_p),
                                           ctypes.cast(_ptr_array(poss), ctypes.c_void_p),
                                           ctypes.cast(_i32_array(task_offsets), ctypes.c_void_p), len(biases), sel.data_ptr(),
                                           _p(global_tok), B, n_sel, G, D, _stream()), 'tokens_assemble')
    return tok


def tokens_assemble_bwd(d_tok: Tensor, task_offsets: Sequence[int], sel: Tensor, B: int, n_sel: int, G: int, D: int,
                        act: torch.dtype, raw: bool = False):
    """returns d_proj (act) [B*n_sel, D] and sums f32 [T+G, D] (bias grads per task, then global-token grads);
    raw=True: the unreduced partials [nblk, (T+G)*D] instead of the sums."""
    lib = _lib.load()
    T = len(task_offsets) - 1
    nblk = lib.mmae_tokens_assemble_bwd_nblk(B)
    part = torch.empty((nblk, T + G, D), device=d_tok.device, dtype=torch.float32)
    d_proj = torch.empty((B * n_sel, D), device=d_tok.device, dtype=act)
    check(lib.mmae_tokens_assemble_bwd(d_tok.data_ptr(), d_proj.data_ptr(), dcode(act),
                                       ctypes.cast(_i32_array(task_offsets), ctypes.c_void_p), T, sel.data_ptr(), part.data_ptr(),
                                       B, n_sel, G, D, _stream()), 'tokens_assemble_bwd')
    if raw:
        return d_proj, part.view(nblk, (T + G) * D)
    sums = torch.empty((T + G, D), device=d_tok.device, dtype=torch.float32)
    reduce_partials(part, sums, False)
    return d_proj, sums


def decoder_build(ctx: Tensor, ids_keep: Tensor, ids_restore: Tensor, mask_token: Tensor, task_emb: Tensor, pos: Tensor,
                  task_offsets: Sequence[int], q_task: int, B: int, n_keep: int, G: int, D: int, n_q: int):
    queries = torch.empty((B * n_q, D), device=ctx.device, dtype=torch.float32)
    context = torch.empty((B * (n_keep + G), D), device=ctx.device, dtype=torch.float32)
    check(_lib.load().mmae_decoder_build(ctx.data_ptr(), ids_keep.data_ptr(), ids_restore.data_ptr(), mask_token.data_ptr(),
                                         task_emb.data_ptr(), pos.data_ptr(), ctypes.cast(_i32_array(task_offsets), ctypes.c_void_p),
                                         len(task_offsets) - 1, q_task, B, n_keep, G, D, n_q, queries.data_ptr(),
                                         context.data_ptr(), _stream()), 'decoder_build')
    return queries, context


def decoder_build_bwd(d_queries: Tensor, d_context: Tensor, ids_keep: Tensor, ids_restore: Tensor, task_offsets: Sequence[int],
                      q_task: int, B: int, n_keep: int, G: int, D: int, n_q: int, raw: bool = False):
    """returns d_ctx f32 [B*(n_keep+G), D] and sums f32 [T+1, D] (task embeddings, then mask token);
    raw=True: the unreduced partials [nblk, (T+1)*D] instead of the sums."""
    lib = _lib.load()
    T = len(task_offsets) - 1
    nblk = lib.mmae_decoder_build_bwd_nblk(B)
    part = torch.empty((nblk, T + 1, D), device=d_queries.device, dtype=torch.float32)
    d_ctx = torch.empty((B * (n_keep + G), D), device=d_queries.device, dtype=torch.float32)
    check(lib.mmae_decoder_build_bwd(d_queries.data_ptr(), d_context.data_ptr(), ids_keep.data_ptr(), ids_restore.data_ptr(),
                                     ctypes.cast(_i32_array(task_offsets), ctypes.c_void_p), T, q_task, B, n_keep, G, D, n_q,
                                     d_ctx.data_ptr(), part.data_ptr(), _stream()), 'decoder_build_bwd')
    if raw:
        return d_ctx, part.view(nblk, (T + 1) * D)
    sums = torch.empty((T + 1, D), device=d_queries.device, dtype=torch.float32)
    reduce_partials(part, sums, False)
    return d_ctx, sums


def unpatchify(pat: Tensor, B: int, C: int, nh: int, nw: int, ph: int, pw: int) -> Tensor:
    img = torch.empty((B, C, nh * ph, nw * pw), device=pat.device, dtype=torch.float32)
    check(_lib.load().mmae_unpatchify(pat.data_ptr(), img.data_ptr(), B, C, nh, nw, ph, pw, _stream()), 'unpatchify')
    return img


def unpatchify_into(pat: Tensor, img: Tensor, B: int, C: int, nh: int, nw: int, ph: int, pw: int) -> None:
    """the same rearrangement into an existing (B, C, nh*ph, nw*pw) f32 tensor (lazy.LazyPrediction's deferred write)"""
    assert img.dtype == torch.float32 and img.is_contiguous() and img.numel() == B * C * nh * ph * nw * pw
    check(_lib.load().mmae_unpatchify(pat.data_ptr(), img.data_ptr(), B, C, nh, nw, ph, pw, _stream()), 'unpatchify')


def patchify(img: Tensor, C: int, nh: int, nw: int, ph: int, pw: int, dtype: torch.dtype) -> Tensor:
    """(B,C,H,W) f32 -> [B*nh*nw, C*ph*pw] act dtype.  The row stride is padded to a multiple of 8 elements
    (zero-filled) so the result is a legal MFMA GEMM operand for any patch dimension; a [:, :C*ph*pw] view is
    returned."""
    B = img.shape[0]
    KP = C * ph * pw
    ld = round_up(KP, 8)
    img = img.contiguous()
    buf = (torch.empty if ld == KP else torch.zeros)((B * nh * nw, ld), device=img.device, dtype=dtype)
    check(_lib.load().mmae_patchify(img.data_ptr(), buf.data_ptr(), dcode(dtype), ld, B, C, nh, nw, ph, pw, _stream()), 'patchify')
    return buf if ld == KP else buf[:, :KP]


# ----------------------------------------------------------------------- optimiser --
def sumsq(x: Tensor, out: Tensor, ws: Tensor) -> None:
    check(_lib.load().mmae_sumsq(x.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), _stream()), 'sumsq')


def adamw_dev(p: Tensor, g: Tensor, m: Tensor, v: Tensor, hyper: Tensor, *, beta1: float, beta2: float, eps: float,
              grad_scale: Optional[Tensor] = None, skip_flag: Optional[Tensor] = None, shadow: Optional[Tensor] = None) -> None:
    """AdamW with {lr, weight_decay, 1 - beta1^t, sqrt(1 - beta2^t)} read from the device tensor `hyper` (f32 [4])."""
    assert hyper.dtype == torch.float32 and hyper.numel() >= 4 and hyper.is_cuda
    check(_lib.load().mmae_adamw_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), hyper.data_ptr(), beta1, beta2,
                                     eps, _p(grad_scale), _p(skip_flag), _p(shadow),
                                     dcode(shadow.dtype) if shadow is not None else F32, _stream()), 'adamw_dev')


def adamw(p: Tensor, g: Tensor, m: Tensor, v: Tensor, *, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float,
          step: int, grad_scale: Optional[Tensor] = None, skip_flag: Optional[Tensor] = None, shadow: Optional[Tensor] = None) -> None:
    check(_lib.load().mmae_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                                 weight_decay, step, _p(grad_scale), _p(skip_flag), _p(shadow),
                                 dcode(shadow.dtype) if shadow is not None else F32, _stream()), 'adamw')


def opt_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, state: Tensor, istate: Tensor, ws: Tensor, *, lr: float, weight_decay: float,
             beta1: float, beta2: float, eps: float, clip_grad: Optional[float], skip_grad: Optional[float], grad_prescale: float = 1.0,
             lrwd_dev: Optional[Tensor] = None, loss_dev: Optional[Tensor] = None, shadow: Optional[Tensor] = None,
             found_inf_dev: Optional[Tensor] = None, grad_scale_dev: Optional[Tensor] = None) -> None:
    """mmae_opt_step: grad-norm, clip / skip / non-finite decisions, step counter and AdamW -- all on the device.
    ``found_inf_dev`` / ``grad_scale_dev``: torch.amp.GradScaler's device scalars (an overflow verdict that skips the update and is
    counted on its own, istate[4]; the loss scale still on the gradients, folded into the step's multiply)."""
    d = OptDesc()
    d.p, d.g, d.m, d.v, d.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
    d.shadow, d.shadow_dtype = _p(shadow), (dcode(shadow.dtype) if shadow is not None else F32)
    d.lr, d.weight_decay, d.beta1, d.beta2, d.eps = lr, weight_decay, beta1, beta2, eps
    d.lrwd_dev, d.loss_dev = _p(lrwd_dev), _p(loss_dev)
    d.found_inf_dev, d.grad_scale_dev = _p(found_inf_dev), _p(grad_scale_dev)
    d.clip_grad, d.skip_grad, d.grad_prescale = (clip_grad or 0.0), (skip_grad or 0.0), grad_prescale
    assert state.dtype == torch.float32 and state.numel() >= 8 and istate.dtype == torch.int32 and istate.numel() >= 8
    d.state, d.istate, d.ws = state.data_ptr(), istate.data_ptr(), ws.data_ptr()
    check(_lib.load().mmae_opt_step(ctypes.byref(d), _stream()), 'opt_step')
