// bf16 MFMA GEMM, deep-pipelined variant for the short-K / huge-M products of the ViT step.
//
// Same math, layouts and epilogue as gemm_bf16.hip, different memory pipeline: K tiles of 32
// elements are streamed HBM -> LDS by the LDS-DMA path (buffer_load ... lds, 1 KiB per wave
// instruction, source-side XOR swizzle) into a 4-deep ring; up to 3 tiles are in flight while one is
// multiplied, tracked with COUNTED s_waitcnt vmcnt(N) and one raw s_barrier per K tile.  The 2-stage
// kernel waits a full HBM/L2 round trip (~1 us) per 64-wide K tile -- with K = 768 that is 12 exposed
// round trips per output tile (measured 21 % MFMA utilisation); here the round trips overlap.
#include "gemm_common.h"
#include <mutex>

#define LDS_AS __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(8))) short s16x8;

namespace {

constexpr int BK = 32;
constexpr unsigned OOB = 0x80000000u;

// k-contiguous tile, rows of 64 B (4 chunks); 4 rows share a 256-B bank row
__device__ __forceinline__ int kc_off(int row, int c) {
    return (row >> 2) * 256 + (((((row & 3) << 2) | c) ^ ((row >> 2) & 15)) << 4);
}
template <int COLS>
__device__ __forceinline__ int ks_off(int krow, int chunk) {
    return krow * (COLS * 2) + ((chunk ^ ((krow & 3) << 2)) << 4);
}

template <int N> __device__ __forceinline__ void wait_vm();
template <> __device__ __forceinline__ void wait_vm<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<2>() { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<3>() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<4>() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }

template <int WM, int WN, bool AKS, bool BKS, int NSTAGE>
__global__ void __launch_bounds__(WM * WN * 64) gemm_bf16_pipe_kernel(const GemmArgs g) {
    constexpr int BM = WM * 64, BN = WN * 64, NW = WM * WN;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int LA = A_BYTES / 1024 / NW, LB = B_BYTES / 1024 / NW;     // DMA instructions per wave per K tile
    constexpr int IPT = LA + LB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tile = g.xcd_swizzle ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
    const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;
    const int z = blockIdx.y, zo = z / g.nb_inner, zi = z % g.nb_inner;
    const uint16_t* Az = (const uint16_t*)g.A + zo * g.sAo + zi * g.sAi;
    const uint16_t* Bz = (const uint16_t*)g.B + zo * g.sBo + zi * g.sBi;
    char* Cz = (char*)g.C + (zo * g.sCo + zi * g.sCi) * (g.c_f32 ? 4 : 2);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, 0x80000000, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bz, 0, 0x80000000, 0x00020000);

    // lane p of DMA instruction (segment s) fills LDS slot p of that 1-KiB segment: find the chunk living there
    unsigned a_off[LA], b_off[LB];
    int a_kq[LA], b_kq[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int seg = i * NW + wave;
        if (!AKS) {
            const int b_abs = 4 * seg + (lane >> 4), j = (lane & 15) ^ (b_abs & 15);
            const int row = 4 * b_abs + (j >> 2), c = j & 3;
            a_kq[i] = c * 8;
            a_off[i] = (m0 + row < g.M) ? (unsigned)((((long long)(m0 + row)) * g.lda + c * 8) * 2) : OOB;
        } else {
            constexpr int CPR = BM / 8;
            const int krow = seg * (64 / CPR) + lane / CPR, ch = (lane % CPR) ^ ((krow & 3) << 2);
            a_kq[i] = krow;
            a_off[i] = (m0 + ch * 8 < g.M) ? (unsigned)((((long long)krow) * g.lda + m0 + ch * 8) * 2) : OOB;
        }
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int seg = i * NW + wave;
        if (!BKS) {
            const int b_abs = 4 * seg + (lane >> 4), j = (lane & 15) ^ (b_abs & 15);
            const int row = 4 * b_abs + (j >> 2), c = j & 3;
            b_kq[i] = c * 8;
            b_off[i] = (n0 + row < g.N) ? (unsigned)((((long long)(n0 + row)) * g.ldb + c * 8) * 2) : OOB;
        } else {
            constexpr int CPR = BN / 8;
            const int krow = seg * (64 / CPR) + lane / CPR, ch = (lane % CPR) ^ ((krow & 3) << 2);
            b_kq[i] = krow;
            b_off[i] = (n0 + ch * 8 < g.N) ? (unsigned)((((long long)krow) * g.ldb + n0 + ch * 8) * 2) : OOB;
        }
    }
    const unsigned a_step = AKS ? (unsigned)(g.lda * BK * 2) : (unsigned)(BK * 2);
    const unsigned b_step = BKS ? (unsigned)(g.ldb * BK * 2) : (unsigned)(BK * 2);

    auto dma = [&](int kt, int stage) {
        const int k0 = kt * BK;
        char* sa = smem + stage * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const bool ok = (a_off[i] != OOB) && (k0 + a_kq[i] < g.K);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(sa + (i * NW + wave) * 1024), 16,
                                                     (int)(ok ? a_off[i] + (unsigned)kt * a_step : OOB), 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const bool ok = (b_off[i] != OOB) && (k0 + b_kq[i] < g.K);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(sb + (i * NW + wave) * 1024), 16,
                                                     (int)(ok ? b_off[i] + (unsigned)kt * b_step : OOB), 0, 0, 0);
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
    auto frag_ks_a = [&](const char* base, int col0, int kk) -> bf16x8 {
        const int col = col0 + t_i0 + (tp & 3) * 4, k_lo = kk * 16 + t_kh + (tp >> 2);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(base + ks_off<BM>(k_lo, col >> 3) + (col & 7) * 2));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(base + ks_off<BM>(k_lo + 4, col >> 3) + (col & 7) * 2));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto frag_ks_b = [&](const char* base, int col0, int kk) -> bf16x8 {
        const int col = col0 + t_i0 + (tp & 3) * 4, k_lo = kk * 16 + t_kh + (tp >> 2);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(base + ks_off<BN>(k_lo, col >> 3) + (col & 7) * 2));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(base + ks_off<BN>(k_lo + 4, col >> 3) + (col & 7) * 2));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    const int nkt_all = (g.K + BK - 1) / BK;
    const int kt_begin = blockIdx.z * g.kt_per_split;
    const int nkt = (kt_begin + g.kt_per_split < nkt_all) ? kt_begin + g.kt_per_split : nkt_all;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (kt_begin + s < nkt) dma(kt_begin + s, s);
    for (int kt = kt_begin; kt < nkt; ++kt) {
        // tile kt has landed once at most the tiles issued after it (<= 2) are still outstanding
        const int rem = nkt - 1 - kt;
        if (NSTAGE >= 4 && rem >= 2) wait_vm<2 * IPT>(); else if (NSTAGE >= 3 && rem >= 1) wait_vm<IPT>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();          // everyone's share of tile kt landed; everyone left tile kt-1
        asm volatile("" ::: "memory");
        if (kt + NSTAGE - 1 < nkt) dma(kt + NSTAGE - 1, (kt + NSTAGE - 1 - kt_begin) % NSTAGE);   // refills the slot of tile kt-1
        const char* sa = smem + ((kt - kt_begin) % NSTAGE) * STAGE;
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
    }

    __syncthreads();                       // the operand ring is dead: reuse it as per-wave staging
    gemm_store_tile64(g, Cz, smem + wave * 8192, lane, acc, m0 + wm * 64, n0 + wn * 64);
}

template <int WM, int WN, bool AKS, bool BKS, int NSTAGE>
int launch(const GemmArgs& g, int batch, hipStream_t st) {
    constexpr int BM = WM * 64, BN = WN * 64;
    const int tiles_m = (g.M + BM - 1) / BM;
    GemmArgs a = g;
    a.tiles_n = (g.N + BN - 1) / BN;
    // kt_per_split arrives in units of 64-wide K tiles (runtime.hip); this kernel steps K by 32
    a.kt_per_split = g.kt_per_split * 2;
    dim3 grid(tiles_m * a.tiles_n, batch, a.splitk), block(WM * WN * 64);
    const size_t lds = (size_t)NSTAGE * (BM + BN) * 64;
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pipe_kernel<WM, WN, AKS, BKS, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    hipLaunchKernelGGL((gemm_bf16_pipe_kernel<WM, WN, AKS, BKS, NSTAGE>), grid, block, lds, st, a);
    return mmae_check_launch("gemm_bf16_pipe");
}

template <int WM, int WN, int NSTAGE>
int dispatch(const mmae_gemm_desc* d, const GemmArgs& g, hipStream_t st) {
    const bool aks = d->a_trans != 0, bks = d->b_trans != 0;
    if (!aks && !bks) return launch<WM, WN, false, false, NSTAGE>(g, d->batch, st);
    if (!aks && bks) return launch<WM, WN, false, true, NSTAGE>(g, d->batch, st);
    if (aks && !bks) return launch<WM, WN, true, false, NSTAGE>(g, d->batch, st);
    return launch<WM, WN, true, true, NSTAGE>(g, d->batch, st);
}

}  // namespace

// tile codes: 5 = 128x128 4-stage, 6 = 256x128 4-stage, 7 = 128x128 2-stage (32 KiB LDS, 3+ workgroups / CU),
//             8 = 128x128 3-stage
int mmae_gemm_bf16_pipe_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st) {
    switch (code) {
        case 6: return dispatch<4, 2, 4>(d, g, st);
        case 7: return dispatch<2, 2, 2>(d, g, st);
        case 8: return dispatch<2, 2, 3>(d, g, st);
        default: return dispatch<2, 2, 4>(d, g, st);
    }
}
