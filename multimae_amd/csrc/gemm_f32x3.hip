// f32-operand GEMM on the bf16 matrix cores by operand splitting ("bf16x3"):
//     a = a_hi + a_lo  (a_hi = bf16(a), a_lo = bf16(a - a_hi)),   a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi
// with fp32 accumulation.  Each f32 operand keeps ~16 mantissa bits, i.e. the products are 25-50x more
// accurate than TF32 (10 bits) -- the precision the reference's "fp32" output adapters actually ran at on
// A100 under torch 1.10 defaults (allow_tf32 = True) -- at 3 bf16 MFMAs per tile step instead of the
// 16x slower exact-f32 MFMA.  Used for fp32_output_adapters in bf16 speed mode; the exact-f32 parity mode
// keeps gemm_f32.hip.
//
// F16 = true ("f32f16", MMAE_F32F16, round 4): the same kernel with ONE product per tile step -- both operands rounded to fp16
// (11-bit significand = TF32's; fp32 accumulation) on v_mfma_f32_32x32x16_f16.  That is exactly the operand precision the
// reference's fp32 adapters ran at on A100 (TF32), at a third of the MFMA work and half the LDS traffic of the split form.  fp16
// has TF32's significand but not its exponent range: forward operands (normalised activations, weights) fit as they are (values
// beyond +-65504 become inf, which the optimiser step's non-finite test turns into a skipped, counted update: common.h f16_cvt); a GRADIENT operand (dy of a dX / dW product, ~1e-6 at the bench batch) is multiplied by a power of two
// read from device memory first -- 2^-floor(log2(amax)), amax = the largest |element| of the loss gradient the backward pass
// started from, left there by the loss kernel -- and the accumulators are multiplied back before the epilogue, so nothing in
// memory changes scale.  Without an amax the gradient products stay on the split form.
//
// Structure = gemm_bf16.hip's VGPR-staged kernel with BK = 32: operands are loaded as f32 (2 float4 per
// 8-element chunk), split in registers, and written to separate hi / lo LDS tiles (same swizzled layouts,
// same ds_read_b128 / ds_read_b64_tr_b16 fragment fetch as the bf16 kernels).
#include "gemm_common.h"
#include <mutex>

#define LDS_AS __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(8))) short s16x8;

namespace {

constexpr int BK = 32;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ int kc_off(int row, int c) {       // rows of 64 B (4 chunks), 4 rows per bank row
    return (row >> 2) * 256 + (((((row & 3) << 2) | c) ^ ((row >> 2) & 15)) << 4);
}
template <int COLS>
__device__ __forceinline__ int ks_off(int krow, int chunk) {
    return krow * (COLS * 2) + ((chunk ^ ((krow & 3) << 2)) << 4);
}

// 8 consecutive f32 -> hi / lo bf16x8 (as i32x4)
__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, i32x4& hi, i32x4& lo) {
    float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h[j] = (__bf16)x[j];
        l[j] = (__bf16)(x[j] - (float)h[j]);
    }
    hi = __builtin_bit_cast(i32x4, h);
    lo = __builtin_bit_cast(i32x4, l);
}

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
// 8 consecutive f32 (times a power of two) -> fp16x8, round to nearest even, saturating
__device__ __forceinline__ i32x4 cvt8_f16(const f32x4 x0, const f32x4 x1, float s) {
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = f16_cvt(x0[j] * s);
        h[4 + j] = f16_cvt(x1[j] * s);
    }
    return __builtin_bit_cast(i32x4, h);
}

template <bool AKS, bool BKS, bool F16 = false>
__global__ void __launch_bounds__(256) gemm_f32x3_kernel(const GemmArgs g) {
    constexpr int BM = 128, BN = 128, NT = 256;
    constexpr int LCH = BM * 4 / NT;                       // 8-element chunks per thread per operand tile (2)
    constexpr int T_BYTES = BM * 64;                       // one 16-bit tile (hi or lo / fp16) of one operand: 8 KiB
    constexpr int STAGE = (F16 ? 2 : 4) * T_BYTES;         // A_hi, A_lo, B_hi, B_lo  |  A16, B16
    // F16: power-of-two pre-scale of the A operand (a gradient): amax in [2^e, 2^(e+1)) -> 2^-e, undone on the accumulators
    float a_s = 1.0f, a_inv = 1.0f;
    if (F16 && g.a_amax) {
        const unsigned e = (__float_as_uint(*g.a_amax) >> 23) & 0xffu;
        if (e >= 1 && e <= 253) { a_s = __uint_as_float((254u - e) << 23); a_inv = __uint_as_float(e << 23); }
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // Split-K products (the weight gradients: a handful of output tiles, the batch as contraction): workgroups reach the XCDs round-robin in
    // flattened (x, z) order, so with the plain mapping an XCD kept the same two output tiles through all slices and every XCD streamed
    // all of the narrower operand -- 482 MB fetched per launch against 256 MB of operands.  Remapped: an XCD takes whole K slices, all
    // output tiles of a slice back to back, so both operand slabs of a slice are fetched by one L2 only: 243 MB per launch, 3 % faster
    // (the re-reads had been MALL hits; profiles/r03_x3_splitk_remap.txt).
    // (slices beyond the last multiple of 8 keep the plain order.)
    int tile, zsplit = blockIdx.z;
    if (gridDim.z >= 8 && gridDim.y == 1) {
        const int T = gridDim.x, S8 = (int)(gridDim.z & ~7u), L = blockIdx.x + T * blockIdx.z;
        if (L < T * S8) { zsplit = (L & 7) + 8 * (L / (8 * T)); tile = (L >> 3) % T; }
        else { const int Lr = L - T * S8; zsplit = S8 + Lr / T; tile = Lr % T; }
    } else {
        tile = g.xcd_swizzle ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
    }
    const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;
    const int z = blockIdx.y, zo = z / g.nb_inner, zi = z % g.nb_inner;
    const float* Az = (const float*)g.A + zo * g.sAo + zi * g.sAi;
    const float* Bz = (const float*)g.B + zo * g.sBo + zi * g.sBi;
    char* Cz = (char*)g.C + (zo * g.sCo + zi * g.sCi) * (g.c_f32 ? 4 : 2);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, 0x80000000, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bz, 0, 0x80000000, 0x00020000);

    unsigned a_off[LCH], b_off[LCH];      // byte offset (f32) of the chunk at K tile 0
    int a_kq[LCH], b_kq[LCH], a_lds[LCH], b_lds[LCH];
#pragma unroll
    for (int i = 0; i < LCH; ++i) {
        const int c = tid + i * NT;
        if (!AKS) {
            const int row = c >> 2, kc = c & 3;
            a_kq[i] = kc * 8;
            a_off[i] = (m0 + row < g.M) ? (unsigned)((((long long)(m0 + row)) * g.lda + kc * 8) * 4) : OOB;
            a_lds[i] = kc_off(row, kc);
        } else {
            constexpr int CPR = BM / 8;
            const int krow = c / CPR, ch = c % CPR;
            a_kq[i] = krow;
            a_off[i] = (m0 + ch * 8 < g.M) ? (unsigned)((((long long)krow) * g.lda + m0 + ch * 8) * 4) : OOB;
            a_lds[i] = ks_off<BM>(krow, ch);
        }
        if (!BKS) {
            const int row = c >> 2, kc = c & 3;
            b_kq[i] = kc * 8;
            b_off[i] = (n0 + row < g.N) ? (unsigned)((((long long)(n0 + row)) * g.ldb + kc * 8) * 4) : OOB;
            b_lds[i] = kc_off(row, kc);
        } else {
            constexpr int CPR = BN / 8;
            const int krow = c / CPR, ch = c % CPR;
            b_kq[i] = krow;
            b_off[i] = (n0 + ch * 8 < g.N) ? (unsigned)((((long long)krow) * g.ldb + n0 + ch * 8) * 4) : OOB;
            b_lds[i] = ks_off<BN>(krow, ch);
        }
    }
    const unsigned a_step = AKS ? (unsigned)(g.lda * BK * 4) : (unsigned)(BK * 4);
    const unsigned b_step = BKS ? (unsigned)(g.ldb * BK * 4) : (unsigned)(BK * 4);

    f32x4 ra[LCH][2], rb[LCH][2];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < LCH; ++i) {
            const bool oka = (a_off[i] != OOB) && (k0 + a_kq[i] < g.K);
            const unsigned oa = oka ? a_off[i] + (unsigned)kt * a_step : OOB;
            ra[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, oa, 0, 0));
            ra[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, oka ? oa + 16 : OOB, 0, 0));
            const bool okb = (b_off[i] != OOB) && (k0 + b_kq[i] < g.K);
            const unsigned ob = okb ? b_off[i] + (unsigned)kt * b_step : OOB;
            rb[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, ob, 0, 0));
            rb[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, okb ? ob + 16 : OOB, 0, 0));
        }
    };
    auto lstore = [&](int stage) {
        char* s = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < LCH; ++i) {
            if (F16) {
                *reinterpret_cast<i32x4*>(s + a_lds[i]) = cvt8_f16(ra[i][0], ra[i][1], a_s);
                *reinterpret_cast<i32x4*>(s + T_BYTES + b_lds[i]) = cvt8_f16(rb[i][0], rb[i][1], 1.0f);
                continue;
            }
            i32x4 hi, lo;
            split8(ra[i][0], ra[i][1], hi, lo);
            *reinterpret_cast<i32x4*>(s + a_lds[i]) = hi;
            *reinterpret_cast<i32x4*>(s + T_BYTES + a_lds[i]) = lo;
            split8(rb[i][0], rb[i][1], hi, lo);
            *reinterpret_cast<i32x4*>(s + 2 * T_BYTES + b_lds[i]) = hi;
            *reinterpret_cast<i32x4*>(s + 3 * T_BYTES + b_lds[i]) = lo;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int fr = lane & 31, fk = lane >> 5;
    const int tg = lane >> 4, tp = lane & 15;
    const int t_i0 = (tg & 1) * 16, t_kh = (tg >> 1) * 8;
    auto frag_kc = [&](const char* base, int row0, int kk) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(base + kc_off(row0 + fr, kk * 2 + fk));
    };
    auto frag_ks = [&](const char* base, int col0, int kk) -> bf16x8 {
        const int col = col0 + t_i0 + (tp & 3) * 4, k_lo = kk * 16 + t_kh + (tp >> 2);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(base + ks_off<128>(k_lo, col >> 3) + (col & 7) * 2));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(base + ks_off<128>(k_lo + 4, col >> 3) + (col & 7) * 2));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    const int nkt_all = (g.K + BK - 1) / BK;
    const int kt_begin = zsplit * g.kt_per_split;
    const int nkt = (kt_begin + g.kt_per_split < nkt_all) ? kt_begin + g.kt_per_split : nkt_all;
    gload(kt_begin);
    lstore(0);
    __syncthreads();
    for (int kt = kt_begin; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload(kt + 1);
        const char* s = smem + ((kt - kt_begin) & 1) * STAGE;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            if (F16) {
                bf16x8 a16[2], b16[2];                      // fp16 bit patterns (the fragment fetch is type-agnostic)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a16[t] = AKS ? frag_ks(s, wm * 64 + t * 32, kk) : frag_kc(s, wm * 64 + t * 32, kk);
                    b16[t] = BKS ? frag_ks(s + T_BYTES, wn * 64 + t * 32, kk) : frag_kc(s + T_BYTES, wn * 64 + t * 32, kk);
                }
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b16[tn]), __builtin_bit_cast(f16x8, a16[tm]),
                                                                             acc[tn][tm], 0, 0, 0);
                continue;
            }
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = AKS ? frag_ks(s, wm * 64 + t * 32, kk) : frag_kc(s, wm * 64 + t * 32, kk);
                al[t] = AKS ? frag_ks(s + T_BYTES, wm * 64 + t * 32, kk) : frag_kc(s + T_BYTES, wm * 64 + t * 32, kk);
                bh[t] = BKS ? frag_ks(s + 2 * T_BYTES, wn * 64 + t * 32, kk) : frag_kc(s + 2 * T_BYTES, wn * 64 + t * 32, kk);
                bl[t] = BKS ? frag_ks(s + 3 * T_BYTES, wn * 64 + t * 32, kk) : frag_kc(s + 3 * T_BYTES, wn * 64 + t * 32, kk);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[tn], ah[tm], acc[tn][tm], 0, 0, 0);   // small terms first
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[tn], al[tm], acc[tn][tm], 0, 0, 0);
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[tn], ah[tm], acc[tn][tm], 0, 0, 0);
                }
        }
        if (kt + 1 < nkt) lstore((kt + 1 - kt_begin) & 1);
        __syncthreads();
    }
    if (F16 && a_inv != 1.0f) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] *= a_inv;
    }
    GemmArgs gs = g;                                       // the shared epilogue addresses the slab of slice blockIdx.z: point it at slice zsplit's
    if (g.splitk > 1) gs.ws = g.ws + ((long long)zsplit - (long long)blockIdx.z) * g.M * g.N;
    gemm_store_tile64(gs, Cz, smem + wave * 8192, lane, acc, m0 + wm * 64, n0 + wn * 64);   // (loop ended on a barrier)
}

template <bool AKS, bool BKS, bool F16 = false>
int launch(const GemmArgs& g, int batch, hipStream_t st) {
    GemmArgs a = g;
    a.tiles_n = (g.N + 127) / 128;
    a.kt_per_split = g.kt_per_split * 2;           // runtime.hip counts 64-wide K tiles for 16-bit operands
    dim3 grid(((g.M + 127) / 128) * a.tiles_n, batch, a.splitk), block(256);
    const size_t lds = 2 * (F16 ? 2 : 4) * 128 * 64;
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute((const void*)gemm_f32x3_kernel<AKS, BKS, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    hipLaunchKernelGGL((gemm_f32x3_kernel<AKS, BKS, F16>), grid, block, lds, st, a);
    return mmae_check_launch(F16 ? "gemm_f32f16" : "gemm_f32x3");
}


// Pre-split operands ("x3p"): out bf16 [rows][out_ld] with three segments of `cols` elements each, segment s at element offset
// s * seg_stride of the row; segment lo_seg holds the low parts bf16(x - hi), the other two the high parts bf16(x).  With
// A' = [hi | hi | lo] and B' = [hi | lo | hi] along the contraction, the three-term split product is ONE bf16 product over 3 K:
// it runs on the ping-pong kernel (LDS-DMA cannot split on the fly) instead of this file's VGPR-staged 128 x 128 kernel.
__global__ void __launch_bounds__(256) x3_split_kernel(const float* __restrict__ x, long long ldx, long long rows, int cols, uint16_t* __restrict__ out,
                                                       long long out_ld, long long seg_stride, int lo_seg) {
    const int cpr = cols >> 3;
    const long long total = rows * cpr;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long r = idx / cpr;
        const int c = (int)(idx - r * cpr) * 8;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + r * ldx + c), x1 = *reinterpret_cast<const f32x4*>(x + r * ldx + c + 4);
        i32x4 hi, lo;
        split8(x0, x1, hi, lo);
        uint16_t* o = out + r * out_ld + c;
#pragma unroll
        for (int sgm = 0; sgm < 3; ++sgm) *reinterpret_cast<i32x4*>(o + sgm * seg_stride) = (sgm == lo_seg) ? lo : hi;
    }
}

}  // namespace

int mmae_gemm_f32x3_impl(const mmae_gemm_desc* d, const GemmArgs& g, hipStream_t st) {
    MMAE_REQUIRE(d->lda % 4 == 0 && d->ldb % 4 == 0, "gemm f32x3: lda/ldb must be multiples of 4");
    MMAE_REQUIRE(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0, "gemm f32x3: A/B must be 16-byte aligned");
    MMAE_REQUIRE(d->sA_outer % 4 == 0 && d->sA_inner % 4 == 0 && d->sB_outer % 4 == 0 && d->sB_inner % 4 == 0,
                 "gemm f32x3: batch strides must be multiples of 4");
    MMAE_REQUIRE((d->a_trans ? d->M : d->K) % 8 == 0 || (d->lda >= (((d->a_trans ? d->M : d->K) + 7) / 8) * 8),
                 "gemm f32x3: A contiguous extent must be readable up to a multiple of 8");
    MMAE_REQUIRE((d->b_trans ? d->N : d->K) % 8 == 0 || (d->ldb >= (((d->b_trans ? d->N : d->K) + 7) / 8) * 8),
                 "gemm f32x3: B contiguous extent must be readable up to a multiple of 8");
    const long long a_rows = d->a_trans ? d->K : d->M, b_rows = d->b_trans ? d->K : d->N;
    MMAE_REQUIRE(a_rows * d->lda * 4 < 0x7fffffffLL && b_rows * d->ldb * 4 < 0x7fffffffLL, "gemm f32x3: operand >= 2 GiB");
    const bool aks = d->a_trans != 0, bks = d->b_trans != 0;
    if (d->ab_dtype == MMAE_F32F16) {
        if (!aks && !bks) return launch<false, false, true>(g, d->batch, st);
        if (!aks && bks) return launch<false, true, true>(g, d->batch, st);
        if (aks && !bks) return launch<true, false, true>(g, d->batch, st);
        return launch<true, true, true>(g, d->batch, st);
    }
    if (!aks && !bks) return launch<false, false>(g, d->batch, st);
    if (!aks && bks) return launch<false, true>(g, d->batch, st);
    if (aks && !bks) return launch<true, false>(g, d->batch, st);
    return launch<true, true>(g, d->batch, st);
}

extern "C" int mmae_x3_split(const float* x, int64_t ldx, int64_t rows, int cols, void* out, int64_t out_ld, int64_t seg_stride, int lo_seg, void* stream) {
    MMAE_REQUIRE(x && out && rows > 0 && cols > 0, "x3_split: bad argument");
    MMAE_REQUIRE(cols % 8 == 0 && ldx % 4 == 0 && out_ld % 8 == 0 && seg_stride % 8 == 0 && lo_seg >= 0 && lo_seg < 3 &&
                 (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0, "x3_split: widths multiples of 8, 16-byte aligned rows");
    const long long total = rows * (cols / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(x3_split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, (long long)rows, cols, (uint16_t*)out,
                       (long long)out_ld, (long long)seg_stride, lo_seg);
    return mmae_check_launch("x3_split");
}

// weights w[i] = [n_out[i]][k_in[i]] f32 -> dst[2 i]: bf16 [n_out][3 k_in] = [hi | lo | hi] (forward product),
//                                           dst[2 i + 1]: bf16 [3 n_out][k_in] = rows [hi ; lo ; hi] (dX product)
extern "C" int mmae_x3_prepare_weights(int n, const void* const* w, const int32_t* n_out, const int32_t* k_in, void* const* dst, void* stream) {
    MMAE_REQUIRE(n >= 0 && (n == 0 || (w && n_out && k_in && dst)), "x3_prepare_weights: null argument");
    for (int i = 0; i < n; ++i) {
        MMAE_REQUIRE(w[i] && dst[2 * i] && dst[2 * i + 1], "x3_prepare_weights: null pointer");
        int rc = mmae_x3_split((const float*)w[i], k_in[i], n_out[i], k_in[i], dst[2 * i], 3LL * k_in[i], k_in[i], 1, stream);
        if (rc) return rc;
        rc = mmae_x3_split((const float*)w[i], k_in[i], n_out[i], k_in[i], dst[2 * i + 1], k_in[i], (int64_t)n_out[i] * k_in[i], 1, stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int64_t mmae_x3_tmp_bytes(int64_t rows, int cols) { return (rows * 3 * cols * 2 + 255) / 256 * 256; }
