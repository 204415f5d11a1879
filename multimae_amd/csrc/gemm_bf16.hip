// bf16 MFMA GEMM for gfx950:  C[M,N] (+)= alpha * A[M,K] . B[N,K]^T  (+ fused epilogue)
//
//  * v_mfma_f32_32x32x16_bf16, fp32 accumulate; each wave owns a 64x64 output tile
//    (2x2 MFMA tiles), a workgroup is WM x WN waves (128x128 or 256x128).
//  * operands staged HBM -> VGPR (16-byte bounds-checked buffer loads, OOB reads 0) ->
//    LDS (XOR-swizzled, conflict-free) -> MFMA fragments; LDS is double buffered and the
//    next K tile's loads are in flight while the current tile is multiplied.
//  * either operand may be stored with its reduction index as the row index
//    ("k-strided": dW = dY^T X, P.V, ...): its fragments are then fetched with the
//    gfx950 transposing LDS read ds_read_b64_tr_b16.
//  * the MFMA is issued as D[n][m] so that a lane owns 4 consecutive n of one row m.
#include <stdlib.h>
#include <mutex>
#include "gemm_common.h"

#define LDS_AS __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(8))) short s16x8;

namespace {

constexpr int BK = 64;   // bf16 elements per K tile (128 bytes per k-contiguous row)

// k-contiguous tile: rows of 128 B; two rows share one 256-B bank row, 16-byte chunks are
// XORed with (row>>1)&15 -> a ds_read_b128 lane group (16 distinct rows) hits 16 distinct slots.
__device__ __forceinline__ int lds_kc_off(int row, int c) {
    return (row >> 1) * 256 + (((((row & 1) << 3) | c) ^ ((row >> 1) & 15)) << 4);
}
// k-strided tile: [64 k-rows][COLS] bf16; 16-byte chunk index XORed with (krow&3)<<2 so the 4
// k-rows touched by one ds_read_b64_tr_b16 16-lane group fall on distinct 32-byte spans.
template <int COLS>
__device__ __forceinline__ int lds_ks_off(int krow, int chunk) {
    return krow * (COLS * 2) + ((chunk ^ ((krow & 3) << 2)) << 4);
}

// DMA = true: operand tiles go HBM -> LDS directly (buffer_load ... lds, 1 KiB per wave instruction,
// no VGPR staging and none of the slow ds_write_b128 traffic); the XOR swizzle is applied on the
// SOURCE side: lane p of a wave instruction owns LDS slot p of its 1-KiB segment and fetches the
// global 16-byte chunk that belongs there.  DMA = false: VGPR-staged loads + ds_write_b128.
template <int WM, int WN, bool AKS, bool BKS, bool DMA>
__global__ void __launch_bounds__(WM * WN * 64) gemm_bf16_kernel(const GemmArgs g) {
    constexpr int BM = WM * 64, BN = WN * 64, NT = WM * WN * 64, NW = WM * WN;
    constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT;      // 16-byte chunks per thread per tile
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int tile = g.xcd_swizzle ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
    const int m0 = (tile / g.tiles_n) * BM;
    const int n0 = (tile % g.tiles_n) * BN;
    const int z = blockIdx.y;
    const int zo = z / g.nb_inner, zi = z % g.nb_inner;

    const uint16_t* Az = (const uint16_t*)g.A + zo * g.sAo + zi * g.sAi;
    const uint16_t* Bz = (const uint16_t*)g.B + zo * g.sBo + zi * g.sBi;
    char* Cz = (char*)g.C + (zo * g.sCo + zi * g.sCi) * (g.c_f32 ? 4 : 2);

    // buffer descriptors: num_records = 2^31 so that 0x80000000 is an always-OOB offset (-> 0)
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, 0x80000000, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bz, 0, 0x80000000, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // per-thread chunk coordinates (fixed over the K loop)
    unsigned a_off[LA], b_off[LB];        // byte offset at k-tile 0 (OOB if the row is outside)
    int a_kq[LA], b_kq[LB];               // k index (elements) this chunk starts at, within the tile
    int a_lds[LA], b_lds[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int c = tid + i * NT;
        if (DMA) {   // slot (segment i*NW + wave, lane) -> logical chunk index whose swizzled home is that slot
            const int seg = i * NW + wave;
            if (!AKS) { const int b_abs = 4 * seg + (lane >> 4), j = (lane & 15) ^ (b_abs & 15); c = (2 * b_abs + (j >> 3)) * 8 + (j & 7); }
            else { constexpr int CPR = BM / 8; const int krow = seg * (64 / CPR) + lane / CPR; c = krow * CPR + ((lane % CPR) ^ ((krow & 3) << 2)); }
        }
        if (!AKS) {
            const int row = c >> 3, kc = c & 7;
            a_kq[i] = kc * 8;
            a_off[i] = (m0 + row < g.M) ? (unsigned)((((long long)(m0 + row)) * g.lda + kc * 8) * 2) : OOB;
            a_lds[i] = lds_kc_off(row, kc);
        } else {
            constexpr int CPR = BM / 8;
            const int krow = c / CPR, ch = c % CPR;
            a_kq[i] = krow;
            a_off[i] = (m0 + ch * 8 < g.M) ? (unsigned)((((long long)krow) * g.lda + m0 + ch * 8) * 2) : OOB;
            a_lds[i] = lds_ks_off<BM>(krow, ch);
        }
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        int c = tid + i * NT;
        if (DMA) {
            const int seg = i * NW + wave;
            if (!BKS) { const int b_abs = 4 * seg + (lane >> 4), j = (lane & 15) ^ (b_abs & 15); c = (2 * b_abs + (j >> 3)) * 8 + (j & 7); }
            else { constexpr int CPR = BN / 8; const int krow = seg * (64 / CPR) + lane / CPR; c = krow * CPR + ((lane % CPR) ^ ((krow & 3) << 2)); }
        }
        if (!BKS) {
            const int row = c >> 3, kc = c & 7;
            b_kq[i] = kc * 8;
            b_off[i] = (n0 + row < g.N) ? (unsigned)((((long long)(n0 + row)) * g.ldb + kc * 8) * 2) : OOB;
            b_lds[i] = lds_kc_off(row, kc);
        } else {
            constexpr int CPR = BN / 8;
            const int krow = c / CPR, ch = c % CPR;
            b_kq[i] = krow;
            b_off[i] = (n0 + ch * 8 < g.N) ? (unsigned)((((long long)krow) * g.ldb + n0 + ch * 8) * 2) : OOB;
            b_lds[i] = lds_ks_off<BN>(krow, ch);
        }
    }
    const unsigned a_step = AKS ? (unsigned)(g.lda * BK * 2) : (unsigned)(BK * 2);
    const unsigned b_step = BKS ? (unsigned)(g.ldb * BK * 2) : (unsigned)(BK * 2);

    i32x4 ra[LA], rb[LB];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const bool ok = (a_off[i] != OOB) && (k0 + a_kq[i] < g.K);
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? a_off[i] + (unsigned)kt * a_step : OOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const bool ok = (b_off[i] != OOB) && (k0 + b_kq[i] < g.K);
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? b_off[i] + (unsigned)kt * b_step : OOB, 0, 0);
        }
    };
    auto lstore = [&](int stage) {
        char* sa = smem + stage * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<i32x4*>(sa + a_lds[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *reinterpret_cast<i32x4*>(sb + b_lds[i]) = rb[i];
    };

    auto dma = [&](int kt, int stage) {
        const int k0 = kt * BK;
        char* sa = smem + stage * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const bool ok = (a_off[i] != OOB) && (k0 + a_kq[i] < g.K);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(sa + (i * NW + wave) * 1024), 16,
                                                     (int)(ok ? a_off[i] + (unsigned)kt * a_step : OOB), 0, 0, 0);   // explicit int: an implicit unsigned->int here makes the host pass drop the kernel stub
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const bool ok = (b_off[i] != OOB) && (k0 + b_kq[i] < g.K);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(sb + (i * NW + wave) * 1024), 16,
                                                     (int)(ok ? b_off[i] + (unsigned)kt * b_step : OOB), 0, 0, 0);
        }
    };

    f32x16 acc[2][2];   // [tn][tm]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment coordinates
    const int fr = lane & 31, fk = lane >> 5;                 // k-contiguous: row, 8-element k group
    const int tg = lane >> 4, tp = lane & 15;                 // k-strided: 16-lane group / lane in group
    const int t_i0 = (tg & 1) * 16, t_kh = (tg >> 1) * 8;

    auto frag_kc = [&](const char* base, int row0, int kk) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(base + lds_kc_off(row0 + fr, kk * 2 + fk));
    };
    auto frag_ks_a = [&](const char* base, int col0, int kk) -> bf16x8 {
        const int col = col0 + t_i0 + (tp & 3) * 4;
        const int k_lo = kk * 16 + t_kh + (tp >> 2);
        const char* p0 = base + lds_ks_off<BM>(k_lo, col >> 3) + (col & 7) * 2;
        const char* p1 = base + lds_ks_off<BM>(k_lo + 4, col >> 3) + (col & 7) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)p0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)p1);
        s16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, r);
    };
    auto frag_ks_b = [&](const char* base, int col0, int kk) -> bf16x8 {
        const int col = col0 + t_i0 + (tp & 3) * 4;
        const int k_lo = kk * 16 + t_kh + (tp >> 2);
        const char* p0 = base + lds_ks_off<BN>(k_lo, col >> 3) + (col & 7) * 2;
        const char* p1 = base + lds_ks_off<BN>(k_lo + 4, col >> 3) + (col & 7) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)p0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)p1);
        s16x8 r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, r);
    };

    const int nkt_all = (g.K + BK - 1) / BK;
    const int kt_begin = blockIdx.z * g.kt_per_split;
    const int nkt = (kt_begin + g.kt_per_split < nkt_all) ? kt_begin + g.kt_per_split : nkt_all;
    if (DMA) { dma(kt_begin, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else { gload(kt_begin); lstore(0); }
    __syncthreads();
    for (int kt = kt_begin; kt < nkt; ++kt) {
        if (kt + 1 < nkt) { if (DMA) dma(kt + 1, (kt + 1 - kt_begin) & 1); else gload(kt + 1); }
        const char* sa = smem + ((kt - kt_begin) & 1) * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = AKS ? frag_ks_a(sa, wm * 64 + t * 32, kk) : frag_kc(sa, wm * 64 + t * 32, kk);
                bf[t] = BKS ? frag_ks_b(sb, wn * 64 + t * 32, kk) : frag_kc(sb, wn * 64 + t * 32, kk);
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[tn], af[tm], acc[tn][tm], 0, 0, 0);
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (kt + 1 < nkt) lstore((kt + 1 - kt_begin) & 1);
        __syncthreads();
    }

    // epilogue: D[i = n][j = m]; lane: m = .. + (lane&31); n = .. + 8*(r>>2) + 4*(lane>>5) + (r&3)
    __syncthreads();                       // the operand ring is dead: reuse it as per-wave staging
    gemm_store_tile64(g, Cz, smem + wave * 8192, lane, acc, m0 + wm * 64, n0 + wn * 64);
}

template <int WM, int WN, bool AKS, bool BKS, bool DMA>
int launch(const GemmArgs& g, int batch, hipStream_t st) {
    constexpr int BM = WM * 64, BN = WN * 64;
    const int tiles_m = (g.M + BM - 1) / BM;
    GemmArgs a = g;
    a.tiles_n = (g.N + BN - 1) / BN;
    dim3 grid(tiles_m * a.tiles_n, batch, a.splitk), block(WM * WN * 64);
    const size_t lds = 2 * (BM + BN) * 128;
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        hipFuncSetAttribute((const void*)gemm_bf16_kernel<WM, WN, AKS, BKS, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, AKS, BKS, DMA>), grid, block, lds, st, a);
    return mmae_check_launch("gemm_bf16");
}

template <int WM, int WN, bool DMA>
int dispatch_layout(const GemmArgs& g, int batch, bool aks, bool bks, hipStream_t st) {
    if (!aks && !bks) return launch<WM, WN, false, false, DMA>(g, batch, st);
    if (!aks && bks) return launch<WM, WN, false, true, DMA>(g, batch, st);
    if (aks && !bks) return launch<WM, WN, true, false, DMA>(g, batch, st);
    return launch<WM, WN, true, true, DMA>(g, batch, st);
}

}  // namespace

int mmae_gemm_bf16_pipe_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st);
int mmae_gemm_bf16_pp_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st);
int mmae_gemm_bf16_ext_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st);

int mmae_gemm_bf16_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st) {
    MMAE_REQUIRE(d->lda % 8 == 0 && d->ldb % 8 == 0, "gemm bf16: lda/ldb must be multiples of 8");
    MMAE_REQUIRE(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0, "gemm bf16: A/B must be 16-byte aligned");
    MMAE_REQUIRE(d->sA_outer % 8 == 0 && d->sA_inner % 8 == 0 && d->sB_outer % 8 == 0 && d->sB_inner % 8 == 0,
                 "gemm bf16: batch strides must be multiples of 8");
    const long long a_rows = d->a_trans ? d->K : d->M, b_rows = d->b_trans ? d->K : d->N;
    MMAE_REQUIRE(a_rows * d->lda * 2 < 0x7fffffffLL && b_rows * d->ldb * 2 < 0x7fffffffLL, "gemm bf16: operand >= 2 GiB");
    if (d->a_trans) MMAE_REQUIRE(d->M % 8 == 0 || d->lda >= ((d->M + 7) / 8) * 8, "gemm bf16: transposed A row too short");
    if (d->b_trans) MMAE_REQUIRE(d->N % 8 == 0 || d->ldb >= ((d->N + 7) / 8) * 8, "gemm bf16: transposed B row too short");
    // tile codes (chosen by runtime.hip's gemm_plan): 1 = 128x128 LDS-DMA, 2 = 256x128 LDS-DMA, 3 = 128x128 VGPR-staged, 4 = 256x128 VGPR-staged
#ifdef MMAE_EXPERIMENTS
    // Round-3 structures that lost to the ping-pong kernel (profiles/r03_duo_*, r03_pp64_*) and their dissection builds: tile codes
    // 11 / 12 ("duo", gemm_duo_body.h), 13 / 14 (64-wide K tiles, gemm_pp64_body.h), > 14 (dissections).  They exist only in
    // experiment builds (make EXTRA=-DMMAE_EXPERIMENTS links gemm_bf16_duo.hip / gemm_bf16_pp64.hip); the production library
    // maps those codes to the ping-pong kernel below.
    if (code > 10) {
        const int rc = mmae_gemm_bf16_ext_impl(d, g, code, st);
        if (rc != MMAE_ESUPPORT) return rc;
    }
#endif
    if (code > 10) code = code == 14 ? 10 : 9;
    switch (code) {
        case 5: case 6: case 7: case 8: return mmae_gemm_bf16_pipe_impl(d, g, code, st);     // LDS-DMA ring, BK = 32
        case 9: case 10: return mmae_gemm_bf16_pp_impl(d, g, code, st);                     // 8-wave ping-pong, 256/320 x 256
        case 2: return dispatch_layout<4, 2, true>(g, d->batch, d->a_trans != 0, d->b_trans != 0, st);
        case 3: return dispatch_layout<2, 2, false>(g, d->batch, d->a_trans != 0, d->b_trans != 0, st);
        case 4: return dispatch_layout<4, 2, false>(g, d->batch, d->a_trans != 0, d->b_trans != 0, st);
        default: return dispatch_layout<2, 2, true>(g, d->batch, d->a_trans != 0, d->b_trans != 0, st);
    }
}
