// bf16 MFMA GEMM, "ping-pong" variant for the large products of the ViT step (M = batch x tokens >= 1k rows).
//
// Why another kernel: with 64x64 wave tiles (gemm_bf16.hip / gemm_bf16_pipe.hip) one K step of 16 moves 4 KiB of
// fragments LDS -> VGPR per 4 MFMAs, i.e. half of the LDS read peak at full MFMA rate, and a wave's own loads, LDS
// reads and MFMAs serialise behind each other: every variant of that structure measured ~30 % of the MFMA peak
// (profiles/r01_gemm_ksweep.txt).  Here
//   * a workgroup is 8 waves (2 per SIMD) on a (2 x TM x 32) x 256 output tile; a wave owns (TM x 32) x 64
//     (TM = 4 or 5 -> 8 or 10 MFMAs per 6 or 7 fragment reads),
//   * the two waves of a SIMD run half a phase apart: while one issues its 8-10 back-to-back MFMAs for a
//     16-wide K slice, the other fetches its fragments for the next slice and issues its share of the LDS-DMA
//     prefetch; workgroup barriers between the half-phases keep the alternation exact, so the MFMA pipe always
//     has a wave with nothing else to do and the memory instructions never sit between two MFMAs,
//   * operand K tiles (32 wide) stream HBM -> LDS by LDS-DMA into a 4-deep ring; about 2.5 tiles are in flight,
//     tracked with counted s_waitcnt vmcnt(N) -- never 0 inside the loop.
// TM = 5 (320-row tiles) exists because M = 256 x 99 rows = 79.2 tiles of 320: N = 768 / 2304 / 3072 give 240 /
// 720 / 960 workgroups = 0.94 / 2.81 / 3.75 rounds of the 256 CUs, where 256-row tiles give 1.16 / 3.48 / 4.64.
//
// Barrier numbering (b_j = j-th workgroup barrier; phase q = K tile q/2, 16-wide slice q%2):
//   group 0 (waves 0-3):  MEM(q) b_2q MFMA(q) b_2q+1          group 1 (waves 4-7):  b_2q MEM(q) b_2q+1 MFMA(q)
// Read-after-DMA: every wave waits (counted) for its pieces of tile t before b_4t-1; the first read of tile t is
// after b_4t-1.  Write-after-read: the ring slot of tile t is re-targeted (tile t+4) in MEM(2t+3) / MEM(2t+4),
// which every wave reaches after b_4t+4, when the last reader (group 1, MEM(2t+1)) has drained its lgkmcnt.
#include <stdlib.h>
#include "gemm_common.h"

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
template <> __device__ __forceinline__ void wait_vm<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<7>() { asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }

template <typename T>
__device__ __forceinline__ T* sgpr_ptr(T* p) {            // wave-uniform pointer -> SGPR pair
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void wg_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// v0 / vstep: first output tile of this workgroup and the stride of its walk over the tile list (the plain kernel: its block
// index and grid size; the grouped weight-gradient kernel: one tile per workgroup).  blockIdx.y = batch slice, blockIdx.z = K slice.
template <int TM, bool AKS, bool BKS>
__device__ __forceinline__ void pp_body(const GemmArgs& g, const int v0, const int vstep) {
    constexpr int WMR = TM * 32;                         // output rows per wave
    constexpr int BM = 2 * WMR, BN = 256, NW = 8;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int PA = A_BYTES / 1024, PB = B_BYTES / 1024;     // 1-KiB DMA pieces per K tile
    constexpr int LA = (PA + NW - 1) / NW, LB = PB / NW;        // pieces per wave (the last A round may be padding)
    constexpr int NST = 4;
    constexpr int DUMP = NST * STAGE;                    // 1 KiB that swallows the padding pieces
    constexpr int W = LA + LB + 2;                       // pieces issued after tile t at the point tile t must have landed
    static_assert(!AKS || TM == 4, "k-strided A needs a 256-column tile image");
    static_assert(LB == 2 && LA >= 2 && LA <= 3, "piece schedule below assumes 2 + (2|3) pieces per wave and tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    // `lane` is laundered through an empty asm at the top of every output tile: everything derived from it (fragment and
    // DMA addresses) is then recomputed per tile instead of being hoisted out of the persistent loop, where ~70 address
    // VGPRs would stay live across the epilogue (on top of the 128-160 accumulators: 100-190 spilled registers)
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int z = blockIdx.y, zo = __builtin_amdgcn_readfirstlane(z / g.nb_inner), zi = z - zo * g.nb_inner;
    const uint16_t* Az = sgpr_ptr((const uint16_t*)g.A + zo * g.sAo + zi * g.sAi);
    const uint16_t* Bz = sgpr_ptr((const uint16_t*)g.B + zo * g.sBo + zi * g.sBi);
    char* Cz = (char*)g.C + (zo * g.sCo + zi * g.sCi) * (g.c_f32 ? 4 : 2);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, 0x80000000, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bz, 0, 0x80000000, 0x00020000);

    // lane p of DMA piece s fills LDS slot p of that 1-KiB segment: find the chunk living there.
    // Persistent workgroups: v walks the tile list from v0 in steps of vstep; set_tile() re-targets the DMA offsets.
    unsigned a_off[LA], b_off[LB];
    int a_kq[LA], b_kq[LB];
    int m0 = 0, n0 = 0;
    auto set_tile = [&](int v, int& tm0, int& tn0) {
        const int tile = g.xcd_swizzle ? xcd_tile(v, g.tiles_total) : v;
        // integer division runs on the VALU: pin the (wave-uniform) result back into an SGPR
        const int tile_m = __builtin_amdgcn_readfirstlane(tile / g.tiles_n);
        tm0 = tile_m * BM; tn0 = (tile - tile_m * g.tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int seg = i * NW + wave;
            if (!AKS) {
                const int b_abs = 4 * seg + (lane >> 4), j = (lane & 15) ^ (b_abs & 15);
                const int row = 4 * b_abs + (j >> 2), c = j & 3;
                a_kq[i] = c * 8;
                a_off[i] = (seg < PA && tm0 + row < g.M) ? (unsigned)((((long long)(tm0 + row)) * g.lda + c * 8) * 2) : OOB;
            } else {
                constexpr int CPR = BM / 8;
                const int krow = seg * (64 / CPR) + lane / CPR, ch = (lane % CPR) ^ ((krow & 3) << 2);
                a_kq[i] = krow;
                a_off[i] = (seg < PA && tm0 + ch * 8 < g.M) ? (unsigned)((((long long)krow) * g.lda + tm0 + ch * 8) * 2) : OOB;
            }
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int seg = i * NW + wave;
            if (!BKS) {
                const int b_abs = 4 * seg + (lane >> 4), j = (lane & 15) ^ (b_abs & 15);
                const int row = 4 * b_abs + (j >> 2), c = j & 3;
                b_kq[i] = c * 8;
                b_off[i] = (tn0 + row < g.N) ? (unsigned)((((long long)(tn0 + row)) * g.ldb + c * 8) * 2) : OOB;
            } else {
                constexpr int CPR = BN / 8;
                const int krow = seg * (64 / CPR) + lane / CPR, ch = (lane % CPR) ^ ((krow & 3) << 2);
                b_kq[i] = krow;
                b_off[i] = (tn0 + ch * 8 < g.N) ? (unsigned)((((long long)krow) * g.ldb + tn0 + ch * 8) * 2) : OOB;
            }
        }
    };
    set_tile(v0, m0, n0);
    const unsigned a_step = AKS ? (unsigned)(g.lda * BK * 2) : (unsigned)(BK * 2);
    const unsigned b_step = BKS ? (unsigned)(g.ldb * BK * 2) : (unsigned)(BK * 2);

    const int nkt_all = (g.K + BK - 1) / BK;
    const int kt_begin = blockIdx.z * g.kt_per_split;
    const int kt_end = (kt_begin + g.kt_per_split < nkt_all) ? kt_begin + g.kt_per_split : nkt_all;
    const int T = kt_end - kt_begin;

    // Every call issues the same number of DMA instructions (the vmcnt arithmetic depends on it): tiles past the end of
    // this K slice and the padding pieces load from the out-of-range sentinel (zeros, no memory traffic).
    auto dma_a = [&](int u, int i) {                     // u = K tile relative to kt_begin
        const int kt = kt_begin + u;
        const bool ok = (a_off[i] != OOB) & (kt < kt_end) & (kt * BK + a_kq[i] < g.K);
        char* dst = (i * NW + wave < PA) ? smem + (u & (NST - 1)) * STAGE + (i * NW + wave) * 1024 : smem + DUMP;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)dst, 16, (int)(ok ? a_off[i] + (unsigned)kt * a_step : OOB), 0, 0, 0);
    };
    auto dma_b = [&](int u, int i) {
        const int kt = kt_begin + u;
        const bool ok = (b_off[i] != OOB) & (kt < kt_end) & (kt * BK + b_kq[i] < g.K);
        char* dst = smem + (u & (NST - 1)) * STAGE + A_BYTES + (i * NW + wave) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)dst, 16, (int)(ok ? b_off[i] + (unsigned)kt * b_step : OOB), 0, 0, 0);
    };
    auto dma_first = [&](int u) { dma_a(u, 0); dma_a(u, 1); };                                  // 2 pieces
    auto dma_second = [&](int u) { dma_b(u, 0); dma_b(u, 1); if (LA == 3) dma_a(u, 2); };       // LA + LB - 2 pieces

    f32x16 acc[2][TM];

    int fr = 0, fk = 0, tp = 0, t_i0 = 0, t_kh = 0;
    auto derive = [&]() {
        fr = lane & 31; fk = lane >> 5;
        const int tg = lane >> 4;
        tp = lane & 15; t_i0 = (tg & 1) * 16; t_kh = (tg >> 1) * 8;
    };
    derive();
    auto frag_kc = [&](const char* base, int row0, int kk) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(base + kc_off(row0 + fr, kk * 2 + fk));
    };
    auto frag_ks = [&](const char* base, int col0, int kk) -> bf16x8 {            // both operands' images are 256 columns wide
        const int col = col0 + t_i0 + (tp & 3) * 4, k_lo = kk * 16 + t_kh + (tp >> 2);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(base + ks_off<256>(k_lo, col >> 3) + (col & 7) * 2));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(base + ks_off<256>(k_lo + 4, col >> 3) + (col & 7) * 2));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    bf16x8 af[TM], bf[2];
    // optional column sums of the k-strided A operand (bias gradient of a dW product): the wn = 0 waves of the n-tile-0
    // workgroups add up the A fragments they hold anyway (v_dot2c with a vector of ones, in the shadow of the MFMAs)
    bool do_acs = false;
    float acs[TM];
    auto mem_phase = [&](int u, int kk) {
        const char* sa = smem + (u & (NST - 1)) * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int t = 0; t < 2; ++t) bf[t] = BKS ? frag_ks(sb, wn * 64 + t * 32, kk) : frag_kc(sb, wn * 64 + t * 32, kk);
#pragma unroll
        for (int t = 0; t < TM; ++t) af[t] = AKS ? frag_ks(sa, wm * WMR + t * 32, kk) : frag_kc(sa, wm * WMR + t * 32, kk);
        if (kk == 0) dma_second(u + 2); else dma_first(u + 3);
    };
    // (Issuing all, or one, of the phase's DMA pieces between the MFMAs instead -- an LDS-DMA issue stalls its wave 60-180
    // cycles -- measured 0-10 % slower than keeping them in the memory half-phase.)
    auto mfma_phase = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[tn], af[tm], acc[tn][tm], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (AKS && do_acs) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const i32x4 w = __builtin_bit_cast(i32x4, af[tm]);
#pragma unroll
                for (int j = 0; j < 4; ++j) asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acs[tm]) : "v"(w[j]), "v"(0x3f803f80));
            }
        }
    };

    // prologue of the first tile: K tiles 0, 1 and the first half of K tile 2
    dma_first(0); dma_second(0);
    dma_first(1); dma_second(1);
    dma_first(2);
    wait_vm<W>();                                        // K tile 0 has landed (this wave's share)
    wg_barrier();

    // epilogue staging lives in ring slots 2-3 so that slots 0-1 can already receive the NEXT output tile's first two K
    // tiles while this tile's results are written out (the DMA latency and most of the prologue hide under the epilogue)
    char* stage = smem + 2 * STAGE + wave * 8192;
    static_assert(2 * STAGE >= 8 * 8192, "staging must fit in ring slots 2-3");
    for (int v = v0; v < g.tiles_total; v += vstep) {
        asm volatile("" : "+v"(lane));
        derive();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        do_acs = AKS && g.acs != nullptr && n0 == 0 && wn == 0;
#pragma unroll
        for (int t = 0; t < TM; ++t) acs[t] = 0.f;

        if (wm == 0) {
            for (int u = 0; u < T; ++u) {
                mem_phase(u, 0);
                wg_barrier();
                mfma_phase();
                wg_barrier();
                mem_phase(u, 1);
                wg_barrier();
                mfma_phase();
                wait_vm<W>();                                // K tile u + 1
                wg_barrier();
            }
        } else {
            for (int u = 0; u < T; ++u) {
                wg_barrier();
                mem_phase(u, 0);
                wg_barrier();
                mfma_phase();
                wg_barrier();
                mem_phase(u, 1);
                wait_vm<W>();                                // K tile u + 1
                wg_barrier();
                mfma_phase();
            }
        }
        if (AKS && do_acs) {                                 // lanes l and l + 32 hold the two k-halves of row l
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const float sv = acs[tm] + __shfl_xor(acs[tm], 32, 64);
                const int m = m0 + wm * WMR + tm * 32 + (lane & 31);
                if (lane < 32 && m < g.M) g.acs[(long long)blockIdx.z * g.M + m] = sv;
            }
        }
        wait_vm<0>();                                        // the zero-fill tail pieces must not land on live data
        __syncthreads();                                     // every wave is out of the ring

        const int mw = m0 + wm * WMR, nw = n0 + wn * 64;
        const bool has_next = v + vstep < g.tiles_total;
        if (has_next) {                                      // next tile: K tiles 0, 1 -> slots 0, 1 (in flight during the epilogue)
            asm volatile("" : "+v"(lane));
            set_tile(v + vstep, m0, n0);
            dma_first(0); dma_second(0);
            dma_first(1); dma_second(1);
        }
        {
            f32x16 sub[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
            gemm_store_tile64(g, Cz, stage, lane, sub, mw, nw);
        }
        {
            f32x16 sub[2][2] = {{acc[0][2], acc[0][3]}, {acc[1][2], acc[1][3]}};
            gemm_store_tile64(g, Cz, stage, lane, sub, mw + 64, nw);
        }
        if (TM & 1) {
            f32x16 sub[2][2] = {{acc[0][TM - 1], acc[0][TM - 1]}, {acc[1][TM - 1], acc[1][TM - 1]}};
            gemm_store_tile64(g, Cz, stage, lane, sub, mw + (TM - 1) * 32, nw, 1);
        }
        if (has_next) {
            // loads and stores share vmcnt and may retire out of order with respect to each other: drain everything (the
            // prefetched K tiles landed long ago; this waits for the last store acknowledgements only), then hand ring
            // slots 2-3 back to the DMA ring
            wait_vm<0>();
            __syncthreads();
            dma_first(2);
        }
    }
}

template <int TM, bool AKS, bool BKS>
__global__ void __launch_bounds__(512) gemm_bf16_pp_kernel(const GemmArgs g) {
    pp_body<TM, AKS, BKS>(g, blockIdx.x, gridDim.x);
}

// Grouped weight-gradient launch: up to 8 dW[N_out][K_in] = dY^T X products that share the reduction length (the rows of the
// batch) and the number of K slices, in ONE grid -- blockIdx.x walks the concatenated tile lists, blockIdx.z is the K slice.
// Every workgroup computes one 256 x 256 tile over its slice of the rows: the four weight gradients of a transformer block fill
// the chip with 2 slices (216 workgroups at ViT-B) instead of 7-28 slices per product, i.e. 1/4-1/14 of the partial-slab
// traffic, K loops 4-14x longer, and one reduction launch per block instead of four (+ the bias reductions).
struct DwProblem {
    const void* A; const void* B; float* ws; float* acs;
    int M, N;                    // M = N_out (columns of dY), N = K_in (columns of X)
    long long lda, ldb;
    int tiles_n, tile_begin;
};
struct DwGroupArgs {
    int n, K, splitk, kt_per_split;
    DwProblem p[8];
};

__global__ void __launch_bounds__(512) gemm_bf16_pp_dwgroup_kernel(const DwGroupArgs ga) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) pi += (i < ga.n && (int)blockIdx.x >= ga.p[i].tile_begin) ? 1 : 0;
    pi = __builtin_amdgcn_readfirstlane(pi);
    const DwProblem& pr = ga.p[pi];
    GemmArgs g;
    g.A = pr.A; g.B = pr.B; g.C = nullptr;
    g.M = pr.M; g.N = pr.N; g.K = ga.K;
    g.lda = pr.lda; g.ldb = pr.ldb; g.ldc = pr.N;
    g.nb_inner = 1;
    g.sAo = g.sAi = g.sBo = g.sBi = g.sCo = g.sCi = 0;
    g.bias = nullptr; g.resid = nullptr; g.ldr = 0; g.aux = nullptr; g.ldaux = 0;
    g.c_f32 = 1; g.aux_f32 = 0; g.epi = MMAE_EPI_NONE; g.accumulate = 0; g.vec = 1;
    g.alpha = 1.0f;
    g.tiles_n = pr.tiles_n;
    const int local = (int)blockIdx.x - pr.tile_begin;
    g.tiles_total = local + 1;                            // exactly one tile for this workgroup
    g.splitk = 2;                                         // always through the partial slabs (slice z -> ws[z][M][N]); > 1 only selects that epilogue
    g.kt_per_split = ga.kt_per_split;
    g.ws = pr.ws;
    g.xcd_swizzle = 0;
    g.acs = pr.acs;
    g.colpart = nullptr;
    g.wide_st = 0;
    g.dbg = 0;
    pp_body<4, true, true>(g, local, 1 << 30);
}

template <int TM, bool AKS, bool BKS>
int launch(const GemmArgs& g, int batch, hipStream_t st) {
    constexpr int BM = TM * 64, BN = 256;
    const int tiles_m = (g.M + BM - 1) / BM;
    GemmArgs a = g;
    a.tiles_n = (g.N + BN - 1) / BN;
    a.kt_per_split = g.kt_per_split * 2;                 // runtime.hip counts 64-wide K tiles; this kernel steps by 32
    a.tiles_total = tiles_m * a.tiles_n;
    static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    static const int env_persist = getenv("MMAE_PP_PERSIST") ? atoi(getenv("MMAE_PP_PERSIST")) : 1;
    const int gx = (env_persist && a.tiles_total > n_cu) ? n_cu : a.tiles_total;      // one resident workgroup per CU walks the tile list
    dim3 grid(gx, batch, a.splitk), block(512);
    const size_t lds = (size_t)4 * (BM + BN) * 64 + 1024;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp_kernel<TM, AKS, BKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_bf16_pp_kernel<TM, AKS, BKS>), grid, block, lds, st, a);
    return mmae_check_launch("gemm_bf16_pp");
}

}  // namespace

// tile codes: 9 = 256 x 256 (TM = 4), 10 = 320 x 256 (TM = 5; k-contiguous A only, no column-sum epilogue)
int mmae_gemm_bf16_pp_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st) {
    const bool aks = d->a_trans != 0, bks = d->b_trans != 0;
    if (code == 10 && !aks && !d->colsum_part) {
        return bks ? launch<5, false, true>(g, d->batch, st) : launch<5, false, false>(g, d->batch, st);
    }
    if (!aks && !bks) return launch<4, false, false>(g, d->batch, st);
    if (!aks && bks) return launch<4, false, true>(g, d->batch, st);
    if (aks && !bks) return launch<4, true, false>(g, d->batch, st);
    return launch<4, true, true>(g, d->batch, st);
}

// ------------------------------------------------------------------------------------------------------------------
// grouped weight gradients (mmae_gemm_dw_group)
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct DwReduceProblem { const float* ws; float* C; const float* acs; float* bias; long long mn; int M; long long begin4; };
struct DwReduceArgs { int n, splits, accumulate; long long total4; DwReduceProblem p[8]; };

// C_p (+)= sum_z ws_p[z] (fixed order: deterministic); bias_p (+)= sum_z acs_p[z]
__global__ void __launch_bounds__(256) dw_group_reduce_kernel(const DwReduceArgs ra) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ra.total4; i += (long long)gridDim.x * 256) {
        int pi = 0;
#pragma unroll
        for (int k = 1; k < 8; ++k) pi += (k < ra.n && i >= ra.p[k].begin4) ? 1 : 0;
        const DwReduceProblem& pr = ra.p[pi];
        const long long e = (i - pr.begin4) * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (ra.accumulate) a = ld4(pr.C + e);
        for (int z = 0; z < ra.splits; ++z) {
            const f32x4 v = ld4(pr.ws + z * pr.mn + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += v[j];
        }
        st4(pr.C + e, a);
    }
    if (blockIdx.x == 0) {
        for (int pi = 0; pi < ra.n; ++pi) {
            const DwReduceProblem& pr = ra.p[pi];
            if (!pr.bias) continue;
            for (int m = threadIdx.x; m < pr.M; m += 256) {
                float a = ra.accumulate ? pr.bias[m] : 0.f;
                for (int z = 0; z < ra.splits; ++z) a += pr.acs[(long long)z * pr.M + m];
                pr.bias[m] = a;
            }
        }
    }
}

int dw_group_splits(const mmae_dw_group_desc* d, long long* tiles_out) {
    static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    long long tiles = 0;
    for (int i = 0; i < d->n; ++i) tiles += (long long)((d->p[i].n_out + 255) / 256) * ((d->p[i].k_in + 255) / 256);
    int s = d->split_k > 0 ? d->split_k : (int)(n_cu / (tiles > 0 ? tiles : 1));
    const int nkt = (d->rows + 31) / 32;                      // 32-wide K tiles
    if (s > nkt / 16) s = nkt / 16;                            // >= 16 K tiles per slice
    if (s < 1) s = 1;
    const int kt_per = (nkt + s - 1) / s;
    s = (nkt + kt_per - 1) / kt_per;
    if (tiles_out) *tiles_out = tiles;
    return s;
}

}  // namespace

extern "C" int64_t mmae_gemm_dw_group_ws_elems(const mmae_dw_group_desc* d) {
    if (!d || d->n < 1 || d->n > 8 || d->rows <= 0) return -1;
    const int s = dw_group_splits(d, nullptr);
    long long elems = 0;
    for (int i = 0; i < d->n; ++i) elems += (long long)s * ((long long)d->p[i].n_out * d->p[i].k_in + d->p[i].n_out);
    return elems;
}

extern "C" int mmae_gemm_dw_group(const mmae_dw_group_desc* d, void* stream) {
    MMAE_REQUIRE(d && d->n >= 1 && d->n <= 8 && d->rows > 0, "dw_group: bad descriptor");
    if (d->ab_dtype != MMAE_BF16) { mmae_set_error("dw_group: bf16 operands only"); return MMAE_ESUPPORT; }
    for (int i = 0; i < d->n; ++i) {
        const mmae_dw_problem& q = d->p[i];
        MMAE_REQUIRE(q.dy && q.x && q.dw && q.n_out > 0 && q.k_in > 0, "dw_group: null / empty problem");
        if ((q.n_out % 8) || (q.k_in % 8) || (q.ldy % 8) || (q.ldx % 8) || ((uintptr_t)q.dy % 16) || ((uintptr_t)q.x % 16) || ((uintptr_t)q.dw % 16) ||
            (q.db && ((uintptr_t)q.db % 4))) { mmae_set_error("dw_group: widths / leading dimensions must be multiples of 8, bases 16-byte aligned"); return MMAE_ESUPPORT; }
    }
    long long tiles = 0;
    const int s = dw_group_splits(d, &tiles);
    const int nkt = (d->rows + 31) / 32;
    const int kt_per = (nkt + s - 1) / s;
    MMAE_REQUIRE(d->ws && ((uintptr_t)d->ws % 16) == 0, "dw_group: workspace missing / unaligned");
    MMAE_REQUIRE(d->ws_elems >= mmae_gemm_dw_group_ws_elems(d), "dw_group: workspace too small (mmae_gemm_dw_group_ws_elems)");
    DwGroupArgs ga = {};
    DwReduceArgs ra = {};
    ga.n = d->n; ga.K = d->rows; ga.splitk = s; ga.kt_per_split = kt_per;
    ra.n = d->n; ra.splits = s; ra.accumulate = d->accumulate;
    float* w = d->ws;
    int tb = 0;
    long long b4 = 0;
    for (int i = 0; i < d->n; ++i) {
        const mmae_dw_problem& q = d->p[i];
        const long long mn = (long long)q.n_out * q.k_in;
        DwProblem& p = ga.p[i];
        p.A = q.dy; p.B = q.x; p.ws = w; p.acs = q.db ? w + (long long)s * mn : nullptr;
        p.M = q.n_out; p.N = q.k_in; p.lda = q.ldy; p.ldb = q.ldx;
        p.tiles_n = (q.k_in + 255) / 256; p.tile_begin = tb;
        tb += ((q.n_out + 255) / 256) * p.tiles_n;
        DwReduceProblem& r = ra.p[i];
        r.ws = w; r.C = q.dw; r.acs = p.acs; r.bias = q.db; r.mn = mn; r.M = q.n_out; r.begin4 = b4;
        b4 += mn / 4;
        w += (long long)s * (mn + q.n_out);      // n_out, k_in multiples of 8: every slab base stays 16-byte aligned
    }
    ra.total4 = b4;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)4 * (256 + 256) * 64 + 1024;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp_dwgroup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm_bf16_pp_dwgroup_kernel, dim3(tb, 1, s), dim3(512), lds, st, ga);
    int rc = mmae_check_launch("gemm_bf16_pp_dwgroup");
    if (rc) return rc;
    long long nb = (b4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(dw_group_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, ra);
    return mmae_check_launch("dw_group_reduce");
}
