"""CPU restatement of the OCP Microscaling (MX) v1.0 fp8 format used by the engine's cfg5 path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module (same rule as multimae_oracle.py).

The reference has no fp8 path (it trains under fp16 autocast, run_pretraining_multimae.py:514-516; BASELINE.json configs[4]
asks for the "fp8 MFMA path" on MI355X), so there is nothing in /root/reference to pin this against.  It follows the published
specification instead -- "OCP Microscaling Formats (MX) Specification v1.0", section 5.3 (MXFP8, element e4m3, block 32, scale
E8M0) and section 6.3 (conversion: shared exponent = floor(log2(max |x|)) - emax_elem, elements = round-to-nearest-even of
x / 2^shared, saturating), with ONE deliberate difference the engine makes: the shared exponent is rounded UP when the floor
rule would saturate the block's largest element (mantissa of max |x| above 1.75; `round_up`, default on; the rule of NVIDIA's
MX-fp8 pre-training recipe, arXiv 2506.08027) -- and is pinned by tests/test_oracle_golden.py::test_mx_* against PyTorch's own float8_e4m3fn cast
(in-range values, round-to-nearest-even) and against hand-computed vectors from the specification's tables.
"""
from __future__ import annotations

import numpy as np

E4M3_MAX = 448.0
E4M3_EMAX = 8          # largest power of two of the element format (2^8 = 256 <= 448)
BLOCK = 32


def e4m3_decode_table() -> np.ndarray:
    """value of every e4m3fn byte (OCP: bias 7, no infinities, 0x7f / 0xff = NaN)"""
    t = np.zeros(256, dtype=np.float32)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 15, b & 7
        if e == 15 and m == 7:
            v = np.nan
        elif e == 0:
            v = m * 2.0 ** -9
        else:
            v = (8 + m) * 2.0 ** (e - 10)
        t[b] = -v if s else v
    return t


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """float32 -> e4m3fn bytes, round-to-nearest-even, saturating at +-448 (MX spec 6.3: clamp)"""
    x = np.asarray(x, dtype=np.float32)
    a = np.minimum(np.abs(x).astype(np.float64), E4M3_MAX)
    with np.errstate(divide='ignore'):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, -6.0)                       # subnormals share the exponent of the smallest normal
    quantum = 2.0 ** (e - 3)
    r = np.rint(a / quantum)                      # numpy rint: ties to even
    v = r * quantum                               # may carry into the next binade (r == 16): still exactly representable
    v = np.minimum(v, E4M3_MAX)
    with np.errstate(divide='ignore'):
        e2 = np.floor(np.log2(np.where(v > 0, v, 1.0)))
    normal = v >= 2.0 ** -6
    exp_field = np.where(normal, e2 + 7, 0).astype(np.int64)
    mant = np.where(normal, v / 2.0 ** (e2 - 3) - 8, v / 2.0 ** -9).astype(np.int64)
    byte = (exp_field << 3) | mant
    byte = np.where(np.signbit(x), byte | 0x80, byte)
    return byte.astype(np.uint8)


def shared_exponent(amax: np.ndarray, round_up: bool = True) -> np.ndarray:
    """biased E8M0 exponent of a block, clamped at 0 (amax = 0 or tiny -> 2^-127).
    round_up=False: the specification's rule, floor(log2(amax)) - 8 + 127 -- elements in (448, 512) x scale saturate.
    round_up=True (what the engine does): one more when amax / 2^floor(log2 amax) > 1.75, so that no element saturates."""
    bits = np.asarray(amax, dtype=np.float32).view(np.uint32)
    e = ((bits >> 23) & 0xff).astype(np.int64) - E4M3_EMAX
    if round_up:
        e = e + ((bits & 0x7fffff) > 0x600000)
    return np.maximum(e, 0).astype(np.uint8)


def mx_quantize(x: np.ndarray, round_up: bool = True):
    """x [rows, cols] (cols % 32 == 0) -> (bytes [rows, cols] uint8, exps [rows, cols // 32] uint8), blocks along cols"""
    x = np.asarray(x, dtype=np.float32)
    rows, cols = x.shape
    assert cols % BLOCK == 0
    xb = x.reshape(rows, cols // BLOCK, BLOCK)
    exps = shared_exponent(np.abs(xb).max(axis=2), round_up)
    inv = np.ldexp(np.float32(1.0), 127 - exps.astype(np.int32)).astype(np.float32)       # 2^(127 - e): exact in f32
    q = e4m3_encode(xb * inv[:, :, None])
    return q.reshape(rows, cols), exps


def mx_dequantize(q: np.ndarray, exps: np.ndarray) -> np.ndarray:
    rows, cols = q.shape
    v = e4m3_decode_table()[q].reshape(rows, cols // BLOCK, BLOCK).astype(np.float64)
    return (v * np.ldexp(1.0, exps.astype(np.int32) - 127)[:, :, None]).reshape(rows, cols)


def pack_scales(exps: np.ndarray) -> np.ndarray:
    """[rows, nblk] exponents -> the engine's packed layout uint8 [ceil(nblk / 8)][rows][2][4] (include/mmae.h, mmae_mx_quant):
    byte j of dword (g, r, h) = exponent of block 8 g + 2 j + h"""
    rows, nblk = exps.shape
    groups = (nblk + 7) // 8
    full = np.zeros((rows, groups * 8), dtype=np.uint8)
    full[:, :nblk] = exps
    return np.ascontiguousarray(full.reshape(rows, groups, 4, 2).transpose(1, 0, 3, 2)).reshape(-1)


def mx_matmul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """(a [M, K]) . (b [N, K])^T with both operands MX-quantised along K; products and sums in float64"""
    qa, ea = mx_quantize(a)
    qb, eb = mx_quantize(b)
    return mx_dequantize(qa, ea) @ mx_dequantize(qb, eb).T


# ---------------------------------------------------------------------------------------------------------------------
# a transformer block whose Linear layers see MX-quantised operands the way the engine's 'mxfp8' mode feeds them
# (forward: x and W quantised along the input features; dX: dY and W quantised along the output features; dW unquantised)
# ---------------------------------------------------------------------------------------------------------------------
def fake_quant(t):
    """torch [rows, cols] -> dequantised MX-fp8 of it (blocks along cols), float32"""
    import torch
    q, e = mx_quantize(t.detach().float().numpy())
    return torch.from_numpy(mx_dequantize(q, e).astype(np.float32))


def _bf16(t):
    import torch
    return t.to(torch.bfloat16).float()


def mx_linear(x, w, b):
    """x [..., K] @ w [N, K]^T + b with the engine's operand treatment (activations pass through bf16 before quantisation)."""
    import torch

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x2, w, b):
            xb = _bf16(x2)
            ctx.save_for_backward(xb, w)
            return fake_quant(xb) @ fake_quant(w).t() + b

        @staticmethod
        def backward(ctx, dy):
            xb, w = ctx.saved_tensors
            dyb = _bf16(dy)
            dx = fake_quant(dyb) @ fake_quant(w.t().contiguous()).t()          # w^T [K, N] quantised along N
            return dx, dyb.t() @ xb, dy.sum(0)

    shp = x.shape
    return Fn.apply(x.reshape(-1, shp[-1]), w, b).reshape(*shp[:-1], w.shape[0])


def mx_block(x, sd, prefix: str, heads: int, eps: float):
    """multimae_oracle.block (multimae_utils.py:229-232) with mx_linear in place of the four nn.Linear products"""
    from oracle import multimae_oracle as orc
    B, N, C = x.shape
    d = C // heads
    h = orc.layer_norm(x, sd[prefix + 'norm1.weight'], sd[prefix + 'norm1.bias'], eps)
    qkv = mx_linear(h, sd[prefix + 'attn.qkv.weight'], sd[prefix + 'attn.qkv.bias']).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    a = ((qkv[0] @ qkv[1].transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
    o = (a @ qkv[2]).transpose(1, 2).reshape(B, N, C)
    x = x + mx_linear(o, sd[prefix + 'attn.proj.weight'], sd[prefix + 'attn.proj.bias'])
    h = orc.layer_norm(x, sd[prefix + 'norm2.weight'], sd[prefix + 'norm2.bias'], eps)
    h = orc.gelu_erf(mx_linear(h, sd[prefix + 'mlp.fc1.weight'], sd[prefix + 'mlp.fc1.bias']))
    return x + mx_linear(h, sd[prefix + 'mlp.fc2.weight'], sd[prefix + 'mlp.fc2.bias'])
