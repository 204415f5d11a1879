// Fused attention for the ragged visible-token sets of MultiMAE (N <= 256 tokens, head_dim 32 / 64):
// one workgroup per (batch, head); the whole K and V of the head live in LDS, scores never touch HBM.
//
//   forward : S^T = K Q^T on MFMA (32x32x16 bf16) so that a lane owns ONE query row -> the softmax
//             max / sum are in-register plus one cross-half shuffle; P^T feeds the P.V MFMA straight from
//             the accumulator registers (the MFMA k-index permutation is absorbed by fetching the V^T
//             fragment with the transposing LDS read ds_read_b64_tr_b16); saves only the row LSE.
//   backward: recomputes P from (Q, K, LSE); pass 1 (lane = query) produces dQ, pass 2 (lane = key)
//             produces dK and dV; delta = rowsum(dO . O).  7 small MFMA products instead of 5, no atomics,
//             no HBM traffic beyond Q, K, V, O, dO in and dQ, dK, dV out.
//
// X3 = true: the same kernels for f32 activations (fp32_output_adapters in speed mode).  Operands are split on the fly
// into bf16 hi + lo parts (tiles: when they are written to LDS; register operands: right before the MFMA) and every
// product runs as  a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi  with fp32 accumulation -- the precision of the split-bf16
// ("x3") GEMMs those adapters use, ~16 operand mantissa bits.  Replaces, for them, 2-5 batched x3 GEMM launches plus a
// softmax pass over materialised (B, H, Nq, Nk) f32 scores per attention.
//
// MODE 2 ("f32f16", round 4): f32 activations with fp16 operands -- TF32's 11-bit significand, one MFMA per product instead of
// three -- for adapters that run engine.set_fp32_adapter_gemm('f16').  Tiles are rounded to fp16 when they are written to LDS, P
// and dS right before their MFMA; the backward scales dO by the power of two of the loss gradient's amax (AttnArgs.dy_amax, as
// the GEMMs' mmae_gemm_desc.a_amax) and unscales dQ / dK / dV at their store.
//
// Replaces Attention.forward / CrossAttention.forward cores (multimae_utils.py:175-179, 206-210) + autograd.
#include "common.h"
#include <mutex>

#define LDS_AS __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(8))) short s16x8;

namespace {

constexpr unsigned OOB = 0x80000000u;

struct AttnArgs {
    const void *q, *k, *v, *o, *d_o;          // bf16 (uint16_t) or, in the X3 kernels, f32
    void *out, *dq, *dk, *dv;
    float* lse;
    int B, H, Nq, Nk, nqp, nkp;
    long long q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, dq_sb, dq_sr, dk_sb, dk_sr, dv_sb, dv_sr;
    float scale;
    // optional MX-fp8 mirror of the bf16 outputs (mmae_attn_*_mx): e4m3 bytes [mx_rows][mx_ld] + packed scales; the forward
    // output's / dq, dk, dv's head columns start at mx_col[0..2]
    unsigned char *mx_q, *mx_s;
    long long mx_rows;
    int mx_ld, mx_col[3];
    const float* dy_amax;                      // MODE 2 backward: device scalar whose power of two pre-scales dO (NULL: no scaling)
};

// swizzled byte offset of 16-byte chunk c of row `row` in a row-major bf16 tile with HD columns
template <int HD> __device__ __forceinline__ int tile_off(int row, int c);
template <> __device__ __forceinline__ int tile_off<64>(int row, int c) {
    return (row >> 1) * 256 + (((((row & 1) << 3) | c) ^ ((row >> 1) & 15)) << 4);
}
template <> __device__ __forceinline__ int tile_off<32>(int row, int c) {
    return (row >> 2) * 256 + (((((row & 3) << 2) | c) ^ ((row >> 2) & 15)) << 4);
}

// rows [0, nrows) of a [.., HD] slice (row stride `sr` elements) -> LDS tile of nrows_pad rows (zero padded).
// Two-phase so that a kernel can put ALL its tile loads in flight before the first LDS write: with load -> ds_write per
// chunk (and one loop per tile) every chunk exposed a full HBM round trip -- 8 in a row in the backward prologue.
template <int HD, int NTHR>
struct TileLoader {
    static constexpr int CPR = HD / 8;
    static constexpr int MAXIT = (256 * CPR + NTHR - 1) / NTHR;       // nrows_pad <= 256
    i32x4 r[MAXIT];
    __device__ __forceinline__ void issue(const uint16_t* base, long long sr, int nrows, int nrows_pad, int tid) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000, 0x00020000);
#pragma unroll
        for (int i = 0; i < MAXIT; ++i) {
            const int c = tid + i * NTHR;
            const int row = c / CPR, ch = c % CPR;
            const unsigned off = (c < nrows_pad * CPR && row < nrows) ? (unsigned)((row * sr + ch * 8) * 2) : OOB;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
    }
    __device__ __forceinline__ void commit(char* lds, int nrows_pad, int tid) {
#pragma unroll
        for (int i = 0; i < MAXIT; ++i) {
            const int c = tid + i * NTHR;
            if (c < nrows_pad * CPR) *reinterpret_cast<i32x4*>(lds + tile_off<HD>(c / CPR, c % CPR)) = r[i];
        }
    }
};

__device__ __forceinline__ void split8(const i32x4& a, const i32x4& b, bf16x8& hi, bf16x8& lo) {
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = (__bf16)fa[j]; lo[j] = (__bf16)(fa[j] - (float)hi[j]);
        hi[4 + j] = (__bf16)fb[j]; lo[4 + j] = (__bf16)(fb[j] - (float)hi[4 + j]);
    }
}
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
// 8 f32 (times a power of two) -> fp16 bit patterns, carried as bf16x8 (the LDS tiles / fragment fetches are type-agnostic)
__device__ __forceinline__ bf16x8 cvt8_f16(const i32x4& a, const i32x4& b, float s) {
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = f16_cvt(fa[j] * s);
        h[4 + j] = f16_cvt(fb[j] * s);
    }
    return __builtin_bit_cast(bf16x8, h);
}
// f32 rows -> a hi tile and a lo tile (same swizzled layout, `lo_off` bytes apart)
template <int HD, int NTHR>
struct TileLoaderF32 {
    static constexpr int CPR = HD / 8;
    static constexpr int MAXIT = (256 * CPR + NTHR - 1) / NTHR;
    i32x4 r0[MAXIT], r1[MAXIT];
    __device__ __forceinline__ void issue(const float* base, long long sr, int nrows, int nrows_pad, int tid) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x80000000, 0x00020000);
#pragma unroll
        for (int i = 0; i < MAXIT; ++i) {
            const int c = tid + i * NTHR;
            const int row = c / CPR, ch = c % CPR;
            const bool ok = c < nrows_pad * CPR && row < nrows;
            const unsigned off = (unsigned)((row * sr + ch * 8) * 4);
            r0[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0);
            r1[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : OOB, 0, 0);
        }
    }
    __device__ __forceinline__ void commit(char* lds, int lo_off, int nrows_pad, int tid) {
#pragma unroll
        for (int i = 0; i < MAXIT; ++i) {
            const int c = tid + i * NTHR;
            if (c < nrows_pad * CPR) {
                bf16x8 hi, lo;
                split8(r0[i], r1[i], hi, lo);
                char* p = lds + tile_off<HD>(c / CPR, c % CPR);
                *reinterpret_cast<bf16x8*>(p) = hi;
                *reinterpret_cast<bf16x8*>(p + lo_off) = lo;
            }
        }
    }
    // one fp16 tile (values times the power of two `s`, saturating)
    __device__ __forceinline__ void commit16(char* lds, int nrows_pad, int tid, float s) {
#pragma unroll
        for (int i = 0; i < MAXIT; ++i) {
            const int c = tid + i * NTHR;
            if (c < nrows_pad * CPR) *reinterpret_cast<bf16x8*>(lds + tile_off<HD>(c / CPR, c % CPR)) = cvt8_f16(r0[i], r1[i], s);
        }
    }
};

// MFMA operand "row fragment": lane supplies row (row0 + lane&31), k = 16*ks + 8*(lane>>5) + 0..7
template <int HD>
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int row0, int ks, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + tile_off<HD>(row0 + (lane & 31), ks * 2 + (lane >> 5)));
}
// MFMA operand "column fragment" of the TRANSPOSED tile: lane supplies column (col0 + lane&31) and the 8
// rows { krow0 + 4*kh + 0..3, krow0 + 8 + 4*kh + 0..3 }, kh = lane>>5 -- exactly the key set a lane of the
// other operand holds in accumulator registers 8s..8s+7 of a 32x32 tile (see pack8 callers).
template <int HD>
__device__ __forceinline__ bf16x8 frag_cols(const char* tile, int col0, int krow0, int lane) {
    const int p = lane & 15, g4 = lane >> 4;
    const int col = col0 + (g4 & 1) * 16 + (p & 3) * 4;
    const int kr = krow0 + 4 * (g4 >> 1) + (p >> 2);
    const char* p0 = tile + tile_off<HD>(kr, col >> 3) + (col & 7) * 2;
    const char* p1 = tile + tile_off<HD>(kr + 8, col >> 3) + (col & 7) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)p1);
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[8 * s + j];
    return r;
}
__device__ __forceinline__ bf16x8 pack8_lo(const f32x16& v, int s) {     // residual of pack8: v - float(bf16(v))
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)(v[8 * s + j] - (float)(__bf16)v[8 * s + j]);
    return r;
}
__device__ __forceinline__ bf16x8 pack8_f16(const f32x16& v, int s) {     // fp16 bit patterns of accumulator registers 8s .. 8s+7
    f16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f16_cvt(v[8 * s + j]);
    return __builtin_bit_cast(bf16x8, r);
}
template <int MODE> __device__ __forceinline__ bf16x8 pack8m(const f32x16& v, int s) { return MODE >= 2 ? pack8_f16(v, s) : pack8(v, s); }
__device__ __forceinline__ bf16x8 load_frag_global(const __amdgpu_buffer_rsrc_t rs, bool ok, long long row, long long sr, int ks, int hi) {
    const unsigned off = ok ? (unsigned)((row * sr + ks * 16 + hi * 8) * 2) : OOB;
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
__device__ __forceinline__ void load_frag_global_f32(const __amdgpu_buffer_rsrc_t rs, bool ok, long long row, long long sr, int ks, int hi,
                                                     bf16x8& h, bf16x8& l) {
    const unsigned off = (unsigned)((row * sr + ks * 16 + hi * 8) * 4);
    const i32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0);
    const i32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : OOB, 0, 0);
    split8(a, b, h, l);
}
__device__ __forceinline__ void load_frag_global_f16(const __amdgpu_buffer_rsrc_t rs, bool ok, long long row, long long sr, int ks, int hi, bf16x8& h) {
    const unsigned off = (unsigned)((row * sr + ks * 16 + hi * 8) * 4);
    const i32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0);
    const i32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off + 16u : OOB, 0, 0);
    h = cvt8_f16(a, b, 1.0f);
}
// c += a . b.  MODE 0: bf16 operands; 1: operands given as (hi, lo) bf16 pairs, three products; 2, 3: fp16 bit patterns, one f16 product
template <int MODE>
__device__ __forceinline__ f32x16 mma(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x16 c) {
    if (MODE >= 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), c, 0, 0, 0);
    if (MODE == 1) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}
// store 4 consecutive head-dim values of one row
__device__ __forceinline__ void store4(uint16_t* p, float a, float b, float c, float d) {
    f32x4 t = {a, b, c, d};
    st4(p, t);
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    f32x4 t = {a, b, c, d};
    st4(p, t);
}
// One row's 16 values of a 32-column MFMA tile (a lane holds columns 8g + 4*hi .. +3 for g = 0..3; lanes l and l + 32 hold the
// two halves of the same row) -> TWO 16-byte stores per lane instead of four 8-byte ones: v_permlane32_swap exchanges the
// half-waves' column groups pairwise so that a lane ends up with 8 contiguous columns (the store tail of a row-per-lane
// epilogue is issue-bound: half the store instructions, same bytes -- MI355X guide, technique T21).  Every lane must execute
// the swaps; `ok` only gates the stores.
__device__ __forceinline__ i32x2 pack4_bf16_rne(float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
    bf16x4_t v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    return __builtin_bit_cast(i32x2, v);
}
__device__ __forceinline__ i32x2 pack4_f16_rne(float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    f16x4_t v = {f16_cvt(a), f16_cvt(b),
                 f16_cvt(c), f16_cvt(d)};
    return __builtin_bit_cast(i32x2, v);
}
template <bool H16 = false>
__device__ __forceinline__ void store_row32(uint16_t* p, const f32x16& v, float s, int hi, bool ok) {    // p: column 0 of this tile's row
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
        const i32x2 a = H16 ? pack4_f16_rne(v[g * 4] * s, v[g * 4 + 1] * s, v[g * 4 + 2] * s, v[g * 4 + 3] * s)
                            : pack4_bf16_rne(v[g * 4] * s, v[g * 4 + 1] * s, v[g * 4 + 2] * s, v[g * 4 + 3] * s);
        const i32x2 b = H16 ? pack4_f16_rne(v[g * 4 + 4] * s, v[g * 4 + 5] * s, v[g * 4 + 6] * s, v[g * 4 + 7] * s)
                            : pack4_bf16_rne(v[g * 4 + 4] * s, v[g * 4 + 5] * s, v[g * 4 + 6] * s, v[g * 4 + 7] * s);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
        i32x4 o; o[0] = r0[0]; o[1] = r1[0]; o[2] = r0[1]; o[3] = r1[1];
        if (ok) *reinterpret_cast<i32x4*>(p + (g + hi) * 8) = o;
    }
}
// the MX-fp8 copy of the same 32 values (one block: lanes l and l + 32 hold the two halves of the row), quantised from the
// bf16-rounded values so that it equals mmae_mx_quant of the stored tensor.  Every lane must execute the shuffle.
__device__ __forceinline__ void store_row32_mx(const AttnArgs& a, long long row, int col, const f32x16& v, float s, int hi, bool ok) {
    float w[16];
    float am = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { w[j] = bf16_bits_to_f32(f32_to_bf16_bits(v[j] * s)); am = fmaxf(am, fabsf(w[j])); }
    am = fmaxf(am, __shfl_xor(am, 32, 64));
    const int e = mx_shared_exp(am);
    const float inv = mx_inv_scale(e);
    if (!ok) return;
    unsigned char* dst = a.mx_q + row * a.mx_ld + col;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<int*>(dst + 8 * g + 4 * hi) = mx_cvt4_e4m3(w[4 * g] * inv, w[4 * g + 1] * inv, w[4 * g + 2] * inv, w[4 * g + 3] * inv);
    if (hi == 0) a.mx_s[mx_scale_addr(a.mx_rows, row, col >> 5)] = (unsigned char)e;
}
template <bool H16 = false>
__device__ __forceinline__ void store_row32(float* p, const f32x16& v, float s, int hi, bool ok) {       // f32 rows: already 16-byte stores
    if (!ok) return;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) store4(p + rg * 8 + 4 * hi, v[rg * 4] * s, v[rg * 4 + 1] * s, v[rg * 4 + 2] * s, v[rg * 4 + 3] * s);
}
template <bool X3> struct ActOf { typedef uint16_t T; };
template <> struct ActOf<true> { typedef float T; };

// -------------------------------------------------------------------------------------------------
// GRP (with NT = 4): more than 128 keys are walked in groups of four key tiles with the running (max, sum) rescale of flash attention -- 64
// score registers instead of 112 / 128, so three waves per SIMD fit where the single-group form of the 196-key decoder grids (NT = 7) holds two.
template <int HD, int NT, int MODE, bool GRP = false>
__global__ void __launch_bounds__(256, (NT == 4 && MODE != 1) ? 3 : (NT <= 7 ? 2 : 1)) attn_fwd_kernel(const AttnArgs a) {
    static_assert(!GRP || NT == 4, "grouped form: four key tiles per group");
    constexpr bool X3 = MODE == 1, F32IO = MODE == 1 || MODE == 2;      // MODE 3: fp16 tensors in memory (MMAE_F16)
    typedef typename ActOf<F32IO>::T AT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int tile_b = a.nkp * HD * 2;
    const int lo = 2 * tile_b;                          // X3: [K hi | V hi | K lo | V lo]
    char* Ks = smem;
    char* Vs = smem + tile_b;
    const AT* kg = (const AT*)a.k + b * a.k_sb + h * HD;
    const AT* vg = (const AT*)a.v + b * a.v_sb + h * HD;
    if constexpr (F32IO) {
        TileLoaderF32<HD, 256> lk, lv;
        lk.issue(kg, a.k_sr, a.Nk, a.nkp, tid);
        lv.issue(vg, a.v_sr, a.Nk, a.nkp, tid);
        if constexpr (X3) { lk.commit(Ks, lo, a.nkp, tid); lv.commit(Vs, lo, a.nkp, tid); }
        else { lk.commit16(Ks, a.nkp, tid, 1.0f); lv.commit16(Vs, a.nkp, tid, 1.0f); }
    } else {
        TileLoader<HD, 256> lk, lv;
        lk.issue(kg, a.k_sr, a.Nk, a.nkp, tid);
        lv.issue(vg, a.v_sr, a.Nk, a.nkp, tid);
        lk.commit(Ks, a.nkp, tid);
        lv.commit(Vs, a.nkp, tid);
    }
    __syncthreads();
    const int nt = a.nkp >> 5, nqb = (a.Nq + 31) >> 5;
    const float sc2 = a.scale * 1.44269504089f;            // v_exp_f32 is 2^x: fold log2(e) into the score scale (one VALU op per score less)
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)((const AT*)a.q + b * a.q_sb + h * HD), 0, 0x80000000, 0x00020000);
    AT* ob = (AT*)a.out + b * a.o_sb + h * HD;
    for (int qblk = wave; qblk < nqb; qblk += 4) {
        const int q = qblk * 32 + (lane & 31);
        const bool qok = q < a.Nq;
        bf16x8 qf[HD / 16], ql[HD / 16];
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
            if constexpr (X3) load_frag_global_f32(rsQ, qok, q, a.q_sr, ks, hi, qf[ks], ql[ks]);
            else if constexpr (MODE == 2) { load_frag_global_f16(rsQ, qok, q, a.q_sr, ks, hi, qf[ks]); ql[ks] = qf[ks]; }
            else { qf[ks] = load_frag_global(rsQ, qok, q, a.q_sr, ks, hi); ql[ks] = qf[ks]; }
        }
        f32x16 s[NT];
        float m = -INFINITY, l = 0.f;
        f32x16 o[HD / 32];
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        for (int t0 = 0; t0 < (GRP ? nt : 1); t0 += NT) {                  // one pass unless GRP
            float mg = -INFINITY;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t0 + t < nt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < HD / 16; ++ks) {
                        const bf16x8 kh = frag_rows<HD>(Ks, (t0 + t) * 32, ks, lane);
                        const bf16x8 kl = X3 ? frag_rows<HD>(Ks + lo, (t0 + t) * 32, ks, lane) : kh;
                        s[t] = mma<MODE>(kh, kl, qf[ks], ql[ks], s[t]);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = (t0 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const float v = key < a.Nk ? s[t][r] * sc2 : -INFINITY;          // scores in the base-2 domain: exp(x) = 2^(x log2 e)
                        s[t][r] = v;
                        mg = fmaxf(mg, v);
                    }
                }
            }
            mg = fmaxf(mg, __shfl_xor(mg, 32, 64));
            const float mn = fmaxf(m, mg);                                 // (every group holds at least one real key: mn is finite)
            float lg = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t0 + t < nt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float p = __builtin_amdgcn_exp2f(s[t][r] - mn); s[t][r] = p; lg += p; }
                }
            }
            lg += __shfl_xor(lg, 32, 64);
            if (GRP) {                                                     // running rescale (first group: m = -inf -> alpha = 0 on zeros)
                const float alpha = __builtin_amdgcn_exp2f(m - mn);
                l = l * alpha + lg;
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            } else {
                l = lg;
            }
            m = mn;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t0 + t < nt) {
#pragma unroll
                    for (int sI = 0; sI < 2; ++sI) {
                        const bf16x8 pf = pack8m<MODE>(s[t], sI);
                        const bf16x8 pl = X3 ? pack8_lo(s[t], sI) : pf;
#pragma unroll
                        for (int dt = 0; dt < HD / 32; ++dt) {
                            const bf16x8 vh = frag_cols<HD>(Vs, dt * 32, (t0 + t) * 32 + 16 * sI, lane);
                            const bf16x8 vl = X3 ? frag_cols<HD>(Vs + lo, dt * 32, (t0 + t) * 32 + 16 * sI, lane) : vh;
                            o[dt] = mma<MODE>(vh, vl, pf, pl, o[dt]);
                        }
                    }
                }
            }
        }
        {
            const float inv = 1.0f / l;
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt) store_row32<MODE == 3>(ob + q * a.o_sr + dt * 32, o[dt], inv, hi, qok);
            if (MODE == 0 && a.mx_q) {
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt) store_row32_mx(a, (long long)b * a.Nq + q, a.mx_col[0] + h * HD + dt * 32, o[dt], inv, hi, qok);
            }
            if (qok && hi == 0) a.lse[((long long)b * a.H + h) * a.Nq + q] = m * 0.69314718056f + __logf(l);     // natural-log lse, as before
        }
    }
}

// -------------------------------------------------------------------------------------------------
// NW = 8 (512 threads): after the shared prologue (tiles -> LDS, delta), waves 0-3 run pass 1 and waves 4-7 run pass 2
// concurrently (the passes only read LDS and write disjoint outputs).  NW = 4 (256 threads): the same four waves run pass 1,
// then pass 2.  At head_dim 64 the passes need ~209 VGPRs, so a CU holds 8 waves either way -- as ONE 8-wave workgroup whose
// prologue (tile loads, LDS commit, delta, barrier) nothing overlaps, or as TWO 4-wave workgroups that overlap each other's
// prologue and compute: 111 vs 132 us on the encoder geometry.  At head_dim 32 (123 VGPRs, two 8-wave workgroups per CU
// already) the 8-wave form is the faster one (107 vs 120 us).
// RS (NW = 4, at most 4 key tiles = 128 keys: the encoder's 98-token self-attention): pass 1 keeps the S and dP tiles of its query block
// in registers (8 x 16 accumulators), forms delta from them and goes straight on to dS and dQ -- the separate delta sweep (a second
// evaluation of the same S / dP products and exponentials: 32 of a wave's 144 MFMAs, 64 of its 192 v_exp) disappears.
template <int HD, int MODE, int NW, bool RS = false>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) attn_bwd_kernel(const AttnArgs a) {
    static_assert(!RS || NW == 4, "register-resident pass 1: the passes run one after the other");
    constexpr int NTH = NW * 64;
    constexpr bool X3 = MODE == 1, F32IO = MODE == 1 || MODE == 2;      // MODE 3: fp16 tensors in memory (MMAE_F16)
    typedef typename ActOf<F32IO>::T AT;
    // MODE 2: dO is scaled by 2^-floor(log2 amax) on its way into LDS; dQ, dK, dV are scaled back at their store
    float do_s = 1.0f, do_inv = 1.0f;
    if (MODE == 2 && a.dy_amax) {
        const unsigned e = (__float_as_uint(*a.dy_amax) >> 23) & 0xffu;
        if (e >= 1 && e <= 253) { do_s = __uint_as_float((254u - e) << 23); do_inv = __uint_as_float(e << 23); }
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    char* Qs = smem;
    char* dOs = Qs + a.nqp * HD * 2;
    char* Ks = dOs + a.nqp * HD * 2;
    char* Vs = Ks + a.nkp * HD * 2;
    const int lo = 2 * (a.nqp + a.nkp) * HD * 2;        // X3: the four lo tiles follow the four hi tiles
    float* lse_s = (float*)(Vs + a.nkp * HD * 2 + (X3 ? lo : 0));
    float* delta_s = lse_s + a.nqp;
    const AT* qg = (const AT*)a.q + b * a.q_sb + h * HD;
    const AT* dog = (const AT*)a.d_o + b * a.o_sb + h * HD;
    const AT* kg = (const AT*)a.k + b * a.k_sb + h * HD;
    const AT* vg = (const AT*)a.v + b * a.v_sb + h * HD;
    if constexpr (F32IO) {
        {
            TileLoaderF32<HD, NTH> lq, ld;
            lq.issue(qg, a.q_sr, a.Nq, a.nqp, tid);
            ld.issue(dog, a.o_sr, a.Nq, a.nqp, tid);
            for (int q = tid; q < a.nqp; q += NTH) lse_s[q] = q < a.Nq ? a.lse[((long long)b * a.H + h) * a.Nq + q] * 1.44269504089f : 0.f;
            if constexpr (X3) { lq.commit(Qs, lo, a.nqp, tid); ld.commit(dOs, lo, a.nqp, tid); }
            else { lq.commit16(Qs, a.nqp, tid, 1.0f); ld.commit16(dOs, a.nqp, tid, do_s); }
        }
        {
            TileLoaderF32<HD, NTH> lk, lv;
            lk.issue(kg, a.k_sr, a.Nk, a.nkp, tid);
            lv.issue(vg, a.v_sr, a.Nk, a.nkp, tid);
            if constexpr (X3) { lk.commit(Ks, lo, a.nkp, tid); lv.commit(Vs, lo, a.nkp, tid); }
            else { lk.commit16(Ks, a.nkp, tid, 1.0f); lv.commit16(Vs, a.nkp, tid, 1.0f); }
        }
    } else {
        TileLoader<HD, NTH> lq, ld, lk, lv;
        lq.issue(qg, a.q_sr, a.Nq, a.nqp, tid);
        ld.issue(dog, a.o_sr, a.Nq, a.nqp, tid);
        lk.issue(kg, a.k_sr, a.Nk, a.nkp, tid);
        lv.issue(vg, a.v_sr, a.Nk, a.nkp, tid);
        for (int q = tid; q < a.nqp; q += NTH) lse_s[q] = q < a.Nq ? a.lse[((long long)b * a.H + h) * a.Nq + q] * 1.44269504089f : 0.f;
        lq.commit(Qs, a.nqp, tid);
        ld.commit(dOs, a.nqp, tid);
        lk.commit(Ks, a.nkp, tid);
        lv.commit(Vs, a.nkp, tid);
    }
    const int nt = a.nkp >> 5, nqb = a.nqp >> 5;
    // base-2 softmax recomputation: lse_s holds lse * log2(e), P = 2^(s * sc2 - lse_s) (fma + v_exp_f32); delta_s holds delta * scale, so that
    // dS = P * fma(dP, scale, -delta_s): five VALU operations per score where the natural-base form took seven
    const float sc2 = a.scale * 1.44269504089f;
    __syncthreads();
    // delta[q] = sum_j P[q][j] dP[q][j], from the SAME fp32 P and dP the two passes form dS = P (dP - delta) with -- exactly what
    // autograd's softmax backward computes.  (Rounds 1-3 used the flash-attention shortcut delta = rowsum(dO . O) with the STORED
    // bf16 O.  O = sum_j P_j V_j carries the keys' common component V_mean; its 2^-9 rounding error, dotted with dO, is a common-mode
    // error of every dS_j of the row that the dQ / dK products multiply by the keys' / queries' common component K_mean -- while
    // the true signal only sees V_j - V_mean and K_j - K_mean.  On the 24-layer ViT-L step of cfg5, where |mean| / |spread| is ~4
    // for both, that was 22 % on dQ of the decoders' cross-attention and 4.8 % on decoder.q.weight, against 0.7 % / 0.4 % with
    // this delta: tools/xattn_delta_probe.py)  One extra S / dP sweep per query block, spread over all waves;
    // O is no longer read.
    for (int qblk = wave; !RS && qblk < nqb; qblk += NW) {
        const int q = qblk * 32 + (lane & 31);
        bf16x8 qf[HD / 16], dof[HD / 16], ql[HD / 16], dol[HD / 16];
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
            qf[ks] = frag_rows<HD>(Qs, qblk * 32, ks, lane); dof[ks] = frag_rows<HD>(dOs, qblk * 32, ks, lane);
            ql[ks] = X3 ? frag_rows<HD>(Qs + lo, qblk * 32, ks, lane) : qf[ks];
            dol[ks] = X3 ? frag_rows<HD>(dOs + lo, qblk * 32, ks, lane) : dof[ks];
        }
        const float lse_q = lse_s[q];
        float d0 = 0.f, d1 = 0.f;
        for (int t = 0; t < nt; ++t) {
            f32x16 st, dpt;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) {
                const bf16x8 kh = frag_rows<HD>(Ks, t * 32, ks, lane), vh = frag_rows<HD>(Vs, t * 32, ks, lane);
                const bf16x8 kl = X3 ? frag_rows<HD>(Ks + lo, t * 32, ks, lane) : kh, vl = X3 ? frag_rows<HD>(Vs + lo, t * 32, ks, lane) : vh;
                st = mma<MODE>(kh, kl, qf[ks], ql[ks], st);
                dpt = mma<MODE>(vh, vl, dof[ks], dol[ks], dpt);
            }
            // (zero-padded keys: dP is exactly 0 there, whatever exp(-lse) their P is)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                d0 += __builtin_amdgcn_exp2f(st[r] * sc2 - lse_q) * dpt[r];
                d1 += __builtin_amdgcn_exp2f(st[r + 1] * sc2 - lse_q) * dpt[r + 1];
            }
        }
        float d = d0 + d1;
        d += __shfl_xor(d, 32, 64);
        if (hi == 0) delta_s[q] = d * a.scale;
    }
    if (!RS) __syncthreads();

    // ---- pass 1, register-resident form
    for (int qblk = wave; RS && qblk < nqb; qblk += 4) {
        const int q = qblk * 32 + (lane & 31);
        const bool qok = q < a.Nq;
        f32x16 st[4], dpt[4];
        {
            bf16x8 qf[HD / 16], dof[HD / 16];
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) { qf[ks] = frag_rows<HD>(Qs, qblk * 32, ks, lane); dof[ks] = frag_rows<HD>(dOs, qblk * 32, ks, lane); }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < nt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { st[t][r] = 0.f; dpt[t][r] = 0.f; }
#pragma unroll
                    for (int ks = 0; ks < HD / 16; ++ks) {
                        const bf16x8 kh = frag_rows<HD>(Ks, t * 32, ks, lane), vh = frag_rows<HD>(Vs, t * 32, ks, lane);
                        st[t] = mma<MODE>(kh, kh, qf[ks], qf[ks], st[t]);
                        dpt[t] = mma<MODE>(vh, vh, dof[ks], dof[ks], dpt[t]);
                    }
                }
            }
        }
        const float lse_q = lse_s[q];
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(st[t][r] * sc2 - lse_q), p1 = __builtin_amdgcn_exp2f(st[t][r + 1] * sc2 - lse_q);
                    st[t][r] = p0; st[t][r + 1] = p1;
                    d0 += p0 * dpt[t][r]; d1 += p1 * dpt[t][r + 1];
                }
            }
        }
        float delta_q = d0 + d1;
        delta_q += __shfl_xor(delta_q, 32, 64);
        delta_q *= a.scale;
        if (hi == 0) delta_s[q] = delta_q;                               // for pass 2, behind the barrier below
        f32x16 dq[HD / 32];
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[t][r] = st[t][r] * (dpt[t][r] * a.scale - delta_q);      // dS^T
#pragma unroll
                for (int sI = 0; sI < 2; ++sI) {
                    const bf16x8 dsf = pack8m<MODE>(st[t], sI);
#pragma unroll
                    for (int dt = 0; dt < HD / 32; ++dt) {
                        const bf16x8 kh = frag_cols<HD>(Ks, dt * 32, t * 32 + 16 * sI, lane);
                        dq[dt] = mma<MODE>(kh, kh, dsf, dsf, dq[dt]);
                    }
                }
            }
        }
        AT* dst = (AT*)a.dq + b * a.dq_sb + h * HD + q * a.dq_sr;
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt) store_row32<MODE == 3>(dst + dt * 32, dq[dt], do_inv, hi, qok);
        if (MODE == 0 && a.mx_q) {
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt) store_row32_mx(a, (long long)b * a.Nq + q, a.mx_col[0] + h * HD + dt * 32, dq[dt], 1.0f, hi, qok);
        }
    }
    if (RS) __syncthreads();                                             // delta of every query block

    // ---- pass 1 (waves 0-3): lane = query row -> dQ
    for (int qblk = wave; !RS && (NW == 4 || wave < 4) && qblk < nqb; qblk += 4) {
        const int q = qblk * 32 + (lane & 31);
        const bool qok = q < a.Nq;
        bf16x8 qf[HD / 16], dof[HD / 16], ql[HD / 16], dol[HD / 16];
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
            qf[ks] = frag_rows<HD>(Qs, qblk * 32, ks, lane); dof[ks] = frag_rows<HD>(dOs, qblk * 32, ks, lane);
            ql[ks] = X3 ? frag_rows<HD>(Qs + lo, qblk * 32, ks, lane) : qf[ks];
            dol[ks] = X3 ? frag_rows<HD>(dOs + lo, qblk * 32, ks, lane) : dof[ks];
        }
        const float lse_q = lse_s[q], delta_q = delta_s[q];
        f32x16 dq[HD / 32];
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
        for (int t = 0; t < nt; ++t) {
            f32x16 st, dpt;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) {
                const bf16x8 kh = frag_rows<HD>(Ks, t * 32, ks, lane), vh = frag_rows<HD>(Vs, t * 32, ks, lane);
                const bf16x8 kl = X3 ? frag_rows<HD>(Ks + lo, t * 32, ks, lane) : kh, vl = X3 ? frag_rows<HD>(Vs + lo, t * 32, ks, lane) : vh;
                st = mma<MODE>(kh, kl, qf[ks], ql[ks], st);
                dpt = mma<MODE>(vh, vl, dof[ks], dol[ks], dpt);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[r] * sc2 - lse_q);
                st[r] = p * (dpt[r] * a.scale - delta_q);            // dS^T
            }
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) {
                const bf16x8 dsf = pack8m<MODE>(st, sI);
                const bf16x8 dsl = X3 ? pack8_lo(st, sI) : dsf;
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt) {
                    const bf16x8 kh = frag_cols<HD>(Ks, dt * 32, t * 32 + 16 * sI, lane);
                    const bf16x8 kl = X3 ? frag_cols<HD>(Ks + lo, dt * 32, t * 32 + 16 * sI, lane) : kh;
                    dq[dt] = mma<MODE>(kh, kl, dsf, dsl, dq[dt]);
                }
            }
        }
        {
            AT* dst = (AT*)a.dq + b * a.dq_sb + h * HD + q * a.dq_sr;
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt) store_row32<MODE == 3>(dst + dt * 32, dq[dt], do_inv, hi, qok);
            if (MODE == 0 && a.mx_q) {
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt) store_row32_mx(a, (long long)b * a.Nq + q, a.mx_col[0] + h * HD + dt * 32, dq[dt], 1.0f, hi, qok);
            }
        }
    }

    // ---- pass 2 (NW = 8: waves 4-7; NW = 4: the same waves again): lane = key row -> dK, dV
    for (int kblk = (NW == 8 ? wave - 4 : wave); (NW == 4 || wave >= 4) && kblk < nt; kblk += 4) {
        const int key = kblk * 32 + (lane & 31);
        bf16x8 kf[HD / 16], vf[HD / 16], kl[HD / 16], vl[HD / 16];
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
            kf[ks] = frag_rows<HD>(Ks, kblk * 32, ks, lane); vf[ks] = frag_rows<HD>(Vs, kblk * 32, ks, lane);
            kl[ks] = X3 ? frag_rows<HD>(Ks + lo, kblk * 32, ks, lane) : kf[ks];
            vl[ks] = X3 ? frag_rows<HD>(Vs + lo, kblk * 32, ks, lane) : vf[ks];
        }
        f32x16 dk[HD / 32], dv[HD / 32];
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
        for (int qt = 0; qt < nqb; ++qt) {
            f32x16 sm, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sm[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < HD / 16; ++ks) {
                const bf16x8 qh = frag_rows<HD>(Qs, qt * 32, ks, lane), doh = frag_rows<HD>(dOs, qt * 32, ks, lane);
                const bf16x8 qlo = X3 ? frag_rows<HD>(Qs + lo, qt * 32, ks, lane) : qh, dolo = X3 ? frag_rows<HD>(dOs + lo, qt * 32, ks, lane) : doh;
                sm = mma<MODE>(qh, qlo, kf[ks], kl[ks], sm);
                dp = mma<MODE>(doh, dolo, vf[ks], vl[ks], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float p = __builtin_amdgcn_exp2f(sm[r] * sc2 - lse_s[qr]);
                sm[r] = p;                                            // P
                dp[r] = p * (dp[r] * a.scale - delta_s[qr]);          // dS
            }
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) {
                const bf16x8 pf = pack8m<MODE>(sm, sI), dsf = pack8m<MODE>(dp, sI);
                const bf16x8 pl = X3 ? pack8_lo(sm, sI) : pf, dsl = X3 ? pack8_lo(dp, sI) : dsf;
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt) {
                    const bf16x8 doh = frag_cols<HD>(dOs, dt * 32, qt * 32 + 16 * sI, lane), qh = frag_cols<HD>(Qs, dt * 32, qt * 32 + 16 * sI, lane);
                    const bf16x8 dolo = X3 ? frag_cols<HD>(dOs + lo, dt * 32, qt * 32 + 16 * sI, lane) : doh;
                    const bf16x8 qlo = X3 ? frag_cols<HD>(Qs + lo, dt * 32, qt * 32 + 16 * sI, lane) : qh;
                    dv[dt] = mma<MODE>(doh, dolo, pf, pl, dv[dt]);
                    dk[dt] = mma<MODE>(qh, qlo, dsf, dsl, dk[dt]);
                }
            }
        }
        {
            const bool kok = key < a.Nk;
            AT* dkd = (AT*)a.dk + b * a.dk_sb + h * HD + key * a.dk_sr;
            AT* dvd = (AT*)a.dv + b * a.dv_sb + h * HD + key * a.dv_sr;
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt) {
                store_row32<MODE == 3>(dkd + dt * 32, dk[dt], do_inv, hi, kok);
                store_row32<MODE == 3>(dvd + dt * 32, dv[dt], do_inv, hi, kok);
            }
            if (MODE == 0 && a.mx_q) {
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt) {
                    store_row32_mx(a, (long long)b * a.Nk + key, a.mx_col[1] + h * HD + dt * 32, dk[dt], 1.0f, hi, kok);
                    store_row32_mx(a, (long long)b * a.Nk + key, a.mx_col[2] + h * HD + dt * 32, dv[dt], 1.0f, hi, kok);
                }
            }
        }
    }
}


// -------------------------------------------------------------------------------------------------
// Round 6: the fusion the north star names, for the output adapters' cross-attention (D = 256: 8 heads x 32, <= 128 context rows, bf16) --
// q-projection + kv-projection + softmax + PV in ONE kernel, one workgroup per (image, head):
//   1. the image's normalised context rows (<= 128 x 256 bf16 = 64 KiB) are staged in LDS once and projected by the head's 64 rows of the
//      kv weight ([K_h ; V_h] = cn . Wkv_h^T + b, one 32-key block per wave, 2 x 16 MFMAs on 256-wide contractions); the results go to the
//      K / V tiles the attention core reads (over the staged context, behind a barrier) AND to the kv activation the backward differentiates;
//   2. a wave projects each of its 32-query blocks straight from global memory (qn rows and the head's 32 rows of Wq as MFMA operands, no
//      staging: every fragment is used once), writes q for the backward and keeps the bf16 fragments for
//   3. the attention core of attn_fwd_kernel<32, 4, 0>, unchanged.
// The weight rows are fetched in the order i -> d(i) = i with bits 2 and 3 swapped, so that accumulator registers 8 s .. 8 s + 7 of a lane hold
// the eight CONSECUTIVE head-dim columns 16 s + 8 (lane >> 5) .. + 7 of its row: the projected values are MFMA operand fragments (and 16-byte
// store units) as they leave the accumulator.  Same arithmetic as the three-launch form: fp32 accumulation over k = 0 .. 255 in steps of 16,
// bias in fp32, one rounding to bf16 -- q / kv / out equal the GEMM + attention path's (tests/test_kernels_gpu.py::test_fused_cross_attention_forward).
// Replaces CrossAttention.forward, multimae_utils.py:199-214 (q, kv, softmax(q k^T) v; the output projection stays the next GEMM).
struct XAttnArgs {
    const uint16_t *qn, *cn, *wq, *wkv;
    const float *bq, *bkv;
    uint16_t *q, *kv, *out;
    float* lse;
    int B, H, Nq, Nk;
    float scale;
};
__device__ __forceinline__ int xa_coff(int row, int ch) { return row * 512 + ((ch ^ (row & 31)) << 4); }       // 256-wide bf16 rows, 16-byte chunks
__device__ __forceinline__ int xa_dperm(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

__global__ void __launch_bounds__(256, 2) xattn_fwd_fused_kernel(const XAttnArgs a) {
    constexpr int D = 256, HD = 32, NT = 4, KS = D / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    char* Cs = smem;                                       // [128][256] bf16 context (phase 1-2), then K | V tiles of the head (phase 3)
    char* Ks = smem;
    char* Vs = smem + 128 * HD * 2;
    // ---- 1. stage the context rows (zero padded to 128)
    {
        const auto rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(a.cn + (long long)b * a.Nk * D), 0, 0x80000000, 0x00020000);
        i32x4 r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = tid + i * 256, row = c >> 5, ch = c & 31;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsC, row < a.Nk ? (unsigned)((row * D + ch * 8) * 2) : OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = tid + i * 256;
            *reinterpret_cast<i32x4*>(Cs + xa_coff(c >> 5, c & 31)) = r[i];
        }
    }
    __syncthreads();
    // ---- 2. K_h, V_h of key block `wave`: D[i -> d(i)][j = key] = sum_k Wkv[s D + h 32 + d][k] cn[key][k]
    const int wrow = h * HD + xa_dperm(lane & 31);
    f32x16 kv2[2];
    {
        const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wkv, 0, 0x80000000, 0x00020000);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) kv2[s][r] = 0.f;
#pragma unroll 4
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 cf = *reinterpret_cast<const bf16x8*>(Cs + xa_coff(wave * 32 + (lane & 31), ks * 2 + hi));
            const bf16x8 wk = load_frag_global(rsW, true, wrow, D, ks, hi);
            const bf16x8 wv = load_frag_global(rsW, true, D + wrow, D, ks, hi);
            kv2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk, cf, kv2[0], 0, 0, 0);
            kv2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, cf, kv2[1], 0, 0, 0);
        }
    }
    __syncthreads();                                       // every wave is done with the staged context: its first 16 KiB become the K / V tiles
    {
        const int key = wave * 32 + (lane & 31);
        const bool kok = key < a.Nk;
        uint16_t* kvg = a.kv + ((long long)b * a.Nk + key) * (2 * D) + h * HD;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            char* T = s ? Vs : Ks;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f32x4 b0 = ld4(a.bkv + s * D + h * HD + 16 * ks + 8 * hi), b1 = ld4(a.bkv + s * D + h * HD + 16 * ks + 8 * hi + 4);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = (__bf16)(kv2[s][8 * ks + j] + b0[j]); o[4 + j] = (__bf16)(kv2[s][8 * ks + 4 + j] + b1[j]); }
                *reinterpret_cast<bf16x8*>(T + tile_off<HD>(key, 2 * ks + hi)) = o;
                if (kok) *reinterpret_cast<bf16x8*>(kvg + s * D + 16 * ks + 8 * hi) = o;
            }
        }
    }
    __syncthreads();
    // ---- 3. per 32-query block: project, then attend
    const int nqb = (a.Nq + 31) >> 5;
    const float sc2 = a.scale * 1.44269504089f;
    const auto rsQn = __builtin_amdgcn_make_buffer_rsrc((void*)(a.qn + (long long)b * a.Nq * D), 0, 0x80000000, 0x00020000);
    const auto rsWq = __builtin_amdgcn_make_buffer_rsrc((void*)a.wq, 0, 0x80000000, 0x00020000);
    for (int qblk = wave; qblk < nqb; qblk += 4) {
        const int q = qblk * 32 + (lane & 31);
        const bool qok = q < a.Nq;
        bf16x8 qf[2];
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 wf = load_frag_global(rsWq, true, wrow, D, ks, hi);
                const bf16x8 xf = load_frag_global(rsQn, qok, q, D, ks, hi);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
            }
            uint16_t* qg = a.q + ((long long)b * a.Nq + q) * D + h * HD;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f32x4 b0 = ld4(a.bq + h * HD + 16 * ks + 8 * hi), b1 = ld4(a.bq + h * HD + 16 * ks + 8 * hi + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { qf[ks][j] = (__bf16)(acc[8 * ks + j] + b0[j]); qf[ks][4 + j] = (__bf16)(acc[8 * ks + 4 + j] + b1[j]); }
                if (qok) *reinterpret_cast<bf16x8*>(qg + 16 * ks + 8 * hi) = qf[ks];
            }
        }
        f32x16 s[NT];
        float m = -INFINITY, l = 0.f;
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Ks, t * 32, ks, lane), qf[ks], s[t], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float v = key < a.Nk ? s[t][r] * sc2 : -INFINITY;
                s[t][r] = v;
                m = fmaxf(m, v);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float p = __builtin_amdgcn_exp2f(s[t][r] - m); s[t][r] = p; l += p; }
        l += __shfl_xor(l, 32, 64);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<HD>(Vs, 0, t * 32 + 16 * sI, lane), pack8(s[t], sI), o, 0, 0, 0);
        store_row32<false>(a.out + ((long long)b * a.Nq + q) * D + h * HD, o, 1.0f / l, hi, qok);
        if (qok && hi == 0) a.lse[((long long)b * a.H + h) * a.Nq + q] = m * 0.69314718056f + __logf(l);
    }
}

int check_common(int B, int H, int Nq, int Nk, int hd, const long long* strides, int n) {
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || Nq > 256 || Nk > 256) return -1;
    if (hd != 32 && hd != 64) return -2;
    for (int i = 0; i < n; ++i) if (strides[i] % 8) return -3;
    return 0;
}

}  // namespace

extern "C" {

static int attn_fwd_impl(int mode, const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                         int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr,
                         float scale, void* stream, void* mx_q = nullptr, void* mx_scale = nullptr) {
    const bool x3 = mode == 1;                            // mode: 0 bf16, 1 f32 activations / split-bf16 products, 2 f32 activations / fp16 products, 3 fp16 activations
    MMAE_REQUIRE(q && k && v && o && lse, "attn_fwd: null pointer");
    const long long st[] = {q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr};
    const int rc = check_common(B, H, Nq, Nk, hd, st, 8);
    MMAE_REQUIRE(rc == 0, rc == -2 ? "attn: head_dim must be 32 or 64" : (rc == -3 ? "attn: strides must be multiples of 8" : "attn: need 1 <= Nq, Nk <= 256"));
    MMAE_REQUIRE(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)o % 16 == 0), "attn_fwd: unaligned pointer");
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.out = o; a.lse = lse;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.nqp = (Nq + 31) / 32 * 32; a.nkp = (Nk + 31) / 32 * 32;
    a.q_sb = q_sb; a.q_sr = q_sr; a.k_sb = k_sb; a.k_sr = k_sr; a.v_sb = v_sb; a.v_sr = v_sr; a.o_sb = o_sb; a.o_sr = o_sr;
    a.scale = scale;
    if (mx_q) {
        MMAE_REQUIRE(!x3 && mx_scale && o_sr == (int64_t)H * hd && o_sb == (int64_t)Nq * o_sr, "attn_fwd_mx: bf16 output, dense [B * Nq][H * head_dim]");
        a.mx_q = (unsigned char*)mx_q; a.mx_s = (unsigned char*)mx_scale; a.mx_rows = (long long)B * Nq; a.mx_ld = H * hd;
    }
    const size_t lds = (size_t)2 * a.nkp * hd * 2 * (x3 ? 2 : 1);
    hipStream_t st_ = (hipStream_t)stream;
    dim3 grid(B * H), block(256);
#define LAUNCH_FWD(HD, NT, X3, ...)                                                                                                       \
    do {                                                                                                                                  \
        hipFuncSetAttribute((const void*)attn_fwd_kernel<HD, NT, X3, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((attn_fwd_kernel<HD, NT, X3, ##__VA_ARGS__>), grid, block, lds, st_, a);                                         \
    } while (0)
    static const int env_grp = mmae_env_int("MMAE_ATTN_FWD_GRP", 1);    // 0: all key tiles of a query block in registers at once (A/B)
    // NT = key tiles held in registers per query block: 4 (<= 128 keys), 7 (<= 224: the 196-token decoder grids; one 16-register
    // score tile less than NT = 8 is what lets two waves per SIMD fit) or 8
    const bool small = a.nkp <= 128, mid = a.nkp <= 224;
    if (mode == 1) {
        if (hd == 64) { if (small) LAUNCH_FWD(64, 4, 1); else if (mid) LAUNCH_FWD(64, 7, 1); else LAUNCH_FWD(64, 8, 1); }
        else { if (small) LAUNCH_FWD(32, 4, 1); else if (mid) LAUNCH_FWD(32, 7, 1); else LAUNCH_FWD(32, 8, 1); }
    } else if (mode == 2) {
        if (hd == 64) { if (small) LAUNCH_FWD(64, 4, 2); else if (mid) LAUNCH_FWD(64, 7, 2); else LAUNCH_FWD(64, 8, 2); }
        else { if (small) LAUNCH_FWD(32, 4, 2); else if (mid) LAUNCH_FWD(32, 7, 2); else LAUNCH_FWD(32, 8, 2); }
    } else if (mode == 3) {
        if (hd == 64) { if (small) LAUNCH_FWD(64, 4, 3); else if (mid) LAUNCH_FWD(64, 7, 3); else LAUNCH_FWD(64, 8, 3); }      // (grouped form at head_dim 64: 168 VGPRs + scratch)
        else { if (small) LAUNCH_FWD(32, 4, 3); else if (env_grp) LAUNCH_FWD(32, 4, 3, true); else if (mid) LAUNCH_FWD(32, 7, 3); else LAUNCH_FWD(32, 8, 3); }
    } else {
        if (hd == 64) { if (small) LAUNCH_FWD(64, 4, 0); else if (mid) LAUNCH_FWD(64, 7, 0); else LAUNCH_FWD(64, 8, 0); }      // (grouped form at head_dim 64: 168 VGPRs + scratch)
        else { if (small) LAUNCH_FWD(32, 4, 0); else if (env_grp) LAUNCH_FWD(32, 4, 0, true); else if (mid) LAUNCH_FWD(32, 7, 0); else LAUNCH_FWD(32, 8, 0); }
    }
#undef LAUNCH_FWD
    return mmae_check_launch("attn_fwd");
}

static int attn_bwd_impl(int mode, const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq,
                         void* dk, void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr,
                         int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr, int64_t dk_sb, int64_t dk_sr,
                         int64_t dv_sb, int64_t dv_sr, float scale, void* stream, void* mx_q = nullptr, void* mx_scale = nullptr,
                         const float* dy_amax = nullptr) {
    const bool x3 = mode == 1;
    MMAE_REQUIRE(q && k && v && o && d_o && lse && dq && dk && dv, "attn_bwd: null pointer");
    MMAE_REQUIRE(((uintptr_t)dq % 16 == 0) && ((uintptr_t)dk % 16 == 0) && ((uintptr_t)dv % 16 == 0), "attn_bwd: unaligned output pointer");
    const long long st[] = {q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, dq_sb, dq_sr, dk_sb, dk_sr, dv_sb, dv_sr};
    const int rc = check_common(B, H, Nq, Nk, hd, st, 14);
    MMAE_REQUIRE(rc == 0, rc == -2 ? "attn: head_dim must be 32 or 64" : (rc == -3 ? "attn: strides must be multiples of 8" : "attn: need 1 <= Nq, Nk <= 256"));
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = o; a.d_o = d_o;
    a.dq = dq; a.dk = dk; a.dv = dv; a.lse = (float*)lse;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.nqp = (Nq + 31) / 32 * 32; a.nkp = (Nk + 31) / 32 * 32;
    a.q_sb = q_sb; a.q_sr = q_sr; a.k_sb = k_sb; a.k_sr = k_sr; a.v_sb = v_sb; a.v_sr = v_sr; a.o_sb = o_sb; a.o_sr = o_sr;
    a.dq_sb = dq_sb; a.dq_sr = dq_sr; a.dk_sb = dk_sb; a.dk_sr = dk_sr; a.dv_sb = dv_sb; a.dv_sr = dv_sr;
    a.scale = scale;
    a.dy_amax = dy_amax;
    if (mx_q) {                                           // self-attention with dq | dk | dv packed in one [B * N][3 * H * head_dim] tensor
        const int64_t Dm = (int64_t)H * hd;
        MMAE_REQUIRE(!x3 && mx_scale && Nq == Nk && dq_sr == 3 * Dm && dk_sr == 3 * Dm && dv_sr == 3 * Dm && dq_sb == Nq * 3 * Dm && dk_sb == dq_sb && dv_sb == dq_sb &&
                     (const uint16_t*)dk == (const uint16_t*)dq + Dm && (const uint16_t*)dv == (const uint16_t*)dq + 2 * Dm,
                     "attn_bwd_mx: dq, dk, dv must be the column slices of one packed bf16 [B * N][3 D] tensor");
        a.mx_q = (unsigned char*)mx_q; a.mx_s = (unsigned char*)mx_scale; a.mx_rows = (long long)B * Nq; a.mx_ld = (int)(3 * Dm);
        a.mx_col[0] = 0; a.mx_col[1] = (int)Dm; a.mx_col[2] = (int)(2 * Dm);
    }
    const size_t lds = (size_t)2 * (a.nqp + a.nkp) * hd * 2 * (x3 ? 2 : 1) + (size_t)2 * a.nqp * 4;
    if (lds > 160 * 1024) { mmae_set_error("attn_bwd: tiles exceed the 160 KB LDS (f32 split path: (Nq + Nk) * head_dim too large)"); return MMAE_ESUPPORT; }
    hipStream_t st_ = (hipStream_t)stream;
    dim3 grid(B * H);
#define LAUNCH_BWD(HD, X3, NW, ...)                                                                                                       \
    do {                                                                                                                                  \
        hipFuncSetAttribute((const void*)attn_bwd_kernel<HD, X3, NW, ##__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((attn_bwd_kernel<HD, X3, NW, ##__VA_ARGS__>), grid, dim3(NW * 64), lds, st_, a);                                 \
    } while (0)
    static const int env_rs = mmae_env_int("MMAE_ATTN_BWD_RS", 1);      // 0: the separate delta sweep everywhere (A/B)
    const bool rs = env_rs && a.nkp <= 128 && lds <= 80 * 1024;
    if (mode == 1) { if (hd == 64) LAUNCH_BWD(64, 1, 8); else LAUNCH_BWD(32, 1, 8); }
    else if (mode == 2) { if (hd == 64) { if (lds <= 80 * 1024) LAUNCH_BWD(64, 2, 4); else LAUNCH_BWD(64, 2, 8); } else LAUNCH_BWD(32, 2, 8); }
    else if (mode == 3) { if (hd == 64) { if (lds <= 80 * 1024) LAUNCH_BWD(64, 3, 4); else LAUNCH_BWD(64, 3, 8); } else if (rs) LAUNCH_BWD(32, 3, 4, true); else LAUNCH_BWD(32, 3, 8); }
    else if (hd == 64) { if (rs) LAUNCH_BWD(64, 0, 4, true); else if (lds <= 80 * 1024) LAUNCH_BWD(64, 0, 4); else LAUNCH_BWD(64, 0, 8); }
    else {
        // head_dim 32 (the output adapters' cross-attention: 196 queries x 99 keys): the register-resident pass 1 too where the keys fit four tiles
        // (round 6: one evaluation of S / dP and the exponentials less -- the separate delta sweep was a third of the kernel's v_exp work)
        static const int env_rs32 = mmae_env_int("MMAE_ATTN_BWD_RS32", 1);
        if (rs && env_rs32) LAUNCH_BWD(32, 0, 4, true); else LAUNCH_BWD(32, 0, 8);
    }
#undef LAUNCH_BWD
    return mmae_check_launch("attn_bwd");
}


/* q-projection + kv-projection + cross-attention forward in one launch (xattn_fwd_fused_kernel): D = H * 32 = 256, Nk <= 128, bf16, dense rows.
 * qn [B * Nq][D], cn [B * Nk][D]; wq [D][D], wkv [2 D][D] (nn.Linear layout), bq [D], bkv [2 D] f32.  Writes q [B * Nq][D] and kv [B * Nk][2 D]
 * (what the backward differentiates), out [B * Nq][D] and lse [B][H][Nq].  MMAE_ESUPPORT for any other geometry. */
int mmae_xattn_fwd_fused(const void* qn, const void* cn, const void* wq, const float* bq, const void* wkv, const float* bkv, void* q, void* kv,
                         void* out, float* lse, int B, int H, int Nq, int Nk, int D, float scale, void* stream) {
    MMAE_REQUIRE(qn && cn && wq && bq && wkv && bkv && q && kv && out && lse, "xattn_fwd_fused: null pointer");
    if (D != 256 || H != 8 || Nk < 1 || Nk > 128 || Nq < 1 || B < 1) { mmae_set_error("xattn_fwd_fused: D = 8 x 32 = 256 and 1 <= Nk <= 128 only"); return MMAE_ESUPPORT; }
    MMAE_REQUIRE(((uintptr_t)qn | (uintptr_t)cn | (uintptr_t)wq | (uintptr_t)wkv | (uintptr_t)q | (uintptr_t)kv | (uintptr_t)out | (uintptr_t)bq | (uintptr_t)bkv) % 16 == 0,
                 "xattn_fwd_fused: unaligned pointer");
    XAttnArgs a = {(const uint16_t*)qn, (const uint16_t*)cn, (const uint16_t*)wq, (const uint16_t*)wkv, bq, bkv, (uint16_t*)q, (uint16_t*)kv, (uint16_t*)out, lse,
                   B, H, Nq, Nk, scale};
    const size_t lds = (size_t)128 * 256 * 2;
    static std::once_flag once;
    std::call_once(once, [&] { (void)hipFuncSetAttribute((const void*)xattn_fwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL(xattn_fwd_fused_kernel, dim3(B * H), dim3(256), lds, (hipStream_t)stream, a);
    return mmae_check_launch("xattn_fwd_fused");
}

int mmae_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                  int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr,
                  float scale, void* stream) {
    return attn_fwd_impl(0, q, k, v, o, lse, B, H, Nq, Nk, hd, q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, scale, stream);
}
int mmae_attn_fwd_mx(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                     int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr,
                     float scale, void* mx_q, void* mx_scale, void* stream) {
    MMAE_REQUIRE(mx_q && mx_scale, "attn_fwd_mx: null MX destination");
    return attn_fwd_impl(0, q, k, v, o, lse, B, H, Nq, Nk, hd, q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, scale, stream, mx_q, mx_scale);
}
int mmae_attn_fwd_f32x3(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                        int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr,
                        float scale, void* stream) {
    return attn_fwd_impl(1, q, k, v, o, lse, B, H, Nq, Nk, hd, q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, scale, stream);
}

int mmae_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq, void* dk,
                  void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb,
                  int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr, int64_t dk_sb, int64_t dk_sr, int64_t dv_sb,
                  int64_t dv_sr, float scale, void* stream) {
    return attn_bwd_impl(0, q, k, v, o, d_o, lse, dq, dk, dv, B, H, Nq, Nk, hd, q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, dq_sb,
                         dq_sr, dk_sb, dk_sr, dv_sb, dv_sr, scale, stream);
}
int mmae_attn_bwd_mx(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq, void* dk,
                     void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb,
                     int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr, int64_t dk_sb, int64_t dk_sr, int64_t dv_sb,
                     int64_t dv_sr, float scale, void* mx_q, void* mx_scale, void* stream) {
    MMAE_REQUIRE(mx_q && mx_scale, "attn_bwd_mx: null MX destination");
    return attn_bwd_impl(0, q, k, v, o, d_o, lse, dq, dk, dv, B, H, Nq, Nk, hd, q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, dq_sb,
                         dq_sr, dk_sb, dk_sr, dv_sb, dv_sr, scale, stream, mx_q, mx_scale);
}
int mmae_attn_bwd_f32x3(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq, void* dk,
                        void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb,
                        int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr, int64_t dk_sb, int64_t dk_sr, int64_t dv_sb,
                        int64_t dv_sr, float scale, void* stream) {
    return attn_bwd_impl(1, q, k, v, o, d_o, lse, dq, dk, dv, B, H, Nq, Nk, hd, q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, dq_sb,
                         dq_sr, dk_sb, dk_sr, dv_sb, dv_sr, scale, stream);
}

/* f32 activations, fp16 operands (engine.set_fp32_adapter_gemm('f16')): TF32-class products, one MFMA each */
int mmae_attn_fwd_f32f16(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                         int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr,
                         float scale, void* stream) {
    return attn_fwd_impl(2, q, k, v, o, lse, B, H, Nq, Nk, hd, q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, scale, stream);
}
int mmae_attn_bwd_f32f16(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq, void* dk,
                         void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb,
                         int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr, int64_t dk_sb, int64_t dk_sr, int64_t dv_sb,
                         int64_t dv_sr, float scale, const float* dy_amax, void* stream) {
    return attn_bwd_impl(2, q, k, v, o, d_o, lse, dq, dk, dv, B, H, Nq, Nk, hd, q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, dq_sb,
                         dq_sr, dk_sb, dk_sr, dv_sb, dv_sr, scale, stream, nullptr, nullptr, dy_amax);
}

/* fp16 tensors in memory (MMAE_F16: an fp32 output adapter in 'h16' mode): the bf16 kernels' data movement, fp16 MFMA products.
 * Backward: d_o arrives in the adapter's scaled gradient units and dq / dk / dv leave in them -- no scaling here. */
int mmae_attn_fwd_f16(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                      int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr,
                      float scale, void* stream) {
    return attn_fwd_impl(3, q, k, v, o, lse, B, H, Nq, Nk, hd, q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, scale, stream);
}
int mmae_attn_bwd_f16(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq, void* dk,
                      void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb,
                      int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr, int64_t dk_sb, int64_t dk_sr, int64_t dv_sb,
                      int64_t dv_sr, float scale, void* stream) {
    return attn_bwd_impl(3, q, k, v, o, d_o, lse, dq, dk, dv, B, H, Nq, Nk, hd, q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, dq_sb,
                         dq_sr, dk_sb, dk_sr, dv_sb, dv_sr, scale, stream);
}

}  // extern "C"
