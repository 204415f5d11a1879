// (gemm_pp_body.h: kernel body shared by gemm_bf16_pp.hip and gemm_bf16_pp_fl.hip)
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
#pragma once
#include <stdlib.h>
#include <mutex>
#include "gemm_common.h"

#define LDS_AS __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(8))) short s16x8;

namespace {

// Phase stamps of workgroup 0 (waves 0 and 4: one of each half-phase group) -- builds with -DMMAE_PP_TRACE only (tools/pp_trace.py):
// per output tile  A tile start | B main loop done | C ring drained + barrier | D next tile's first DMA issued | E, F, G store calls done |
// H stores acknowledged + barrier.
#ifdef MMAE_PP_TRACE
__device__ long long g_pp_trace[2][128];
__device__ long long g_pp_wg[1024][4];                   // per workgroup: start / end (100 MHz), XCC id, tiles done
// each stamp: s_memrealtime (100 MHz, wall time) and s_memtime (shader-clock cycles) -- their ratio is the clock the phase ran at
#define PP_STAMP() do { if (tr_on && tr_n < 126) { g_pp_trace[tr_w][1 + tr_n++] = (long long)__builtin_amdgcn_s_memrealtime(); \
                                                    g_pp_trace[tr_w][1 + tr_n++] = (long long)__builtin_readcyclecounter(); } } while (0)
#define PP_TRACE_END() do { if (tr_on) g_pp_trace[tr_w][0] = tr_n; } while (0)
#else
#define PP_STAMP() do {} while (0)
#define PP_TRACE_END() do {} while (0)
#endif

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
template <> __device__ __forceinline__ void wait_vm<5>() { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<7>() { asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }

template <int N> __device__ __forceinline__ void duo_like_wait() {
    static_assert(N >= 3 && N <= 5, "add the immediate");
    if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
}

template <typename T>
__device__ __forceinline__ T* sgpr_ptr(T* p) {            // wave-uniform pointer -> SGPR pair
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

// sum over the 16 lanes of a DPP row (all of them end up with it): two quad permutes, a half-row and a row mirror -- VALU only
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
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
// KF: K is a multiple of the 32-wide K tile (every ViT product): the DMA pieces then walk running byte offsets -- one v_or
// (out-of-range mask of the prefetch overrun past the K slice) and one add per piece and K tile instead of the compare / select
// chains of the general path.  The memory half-phase, not the MFMA half-phase, paces the loop (MFMA pipe 46 % busy at 2.08 GHz on
// the best product, profiles/r02_pmc_mfma.txt), so every instruction taken out of it counts.
// UNR: the K loop unrolled over the NST ring slots (slot-dependent LDS addresses and M0 values become immediates: 29 fewer
// instructions per K tile in the memory half-phases); off where the extra address registers would spill (generic flavours,
// 320-row k-strided B).
// WIDE: one memory half-phase and one MFMA half-phase per 32-wide K TILE instead of per 16-wide slice (2 x TM x 2 MFMAs per phase,
// both slices' fragments in registers): half the workgroup barriers per K tile.  Schedule: G0: MEM(u) b_2u MFMA(u) b_2u+1, G1: b_2u
// MEM(u) b_2u+1 MFMA(u); MEM(u) reads tile u and issues the pieces of tile u+2 (its slot held tile u-2, whose last reads were
// drained before b_2u-1); every wave waits for its pieces of tile u+2 -- those of tile u+3 may still fly -- before b_2u+3.
// H16: the 16-bit operands are fp16 (v_mfma_f32_32x32x16_f16, fp16 conversions in the epilogue) instead of bf16: the activations of an fp32
// output adapter in engine.set_fp32_adapter_gemm('h16') mode (MMAE_F16).  LDS images, DMA pieces and fragment reads are type-agnostic.
template <bool H16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 b, bf16x8 a, f32x16 c) {
    if constexpr (H16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0);
}
template <bool H16>
__device__ __forceinline__ void dot2_ones(float& acc, int w) {          // acc += the two 16-bit halves of w
    if constexpr (H16) asm("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(w), "v"(0x3c003c00));
    else asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(w), "v"(0x3f803f80));
}

// LNE (f32 bias [+ residual] flavours, TM = 4, N == 256 -- the tile spans the row): besides C, write the LayerNorm (gamma / beta given) or
// the plain 16-bit cast (gamma NULL) of the rows this tile completed -- the nn.LayerNorm / autocast cast that follows the Linear in a
// D = 256 decoder block (multimae_utils.py:229-232, output_adapters.py:265-266) without its own pass over the residual stream.  The final
// values stay in registers (where the accumulators were); per-row (sum, sum of squares) of a wave's 64 columns are reduced over the 16
// lanes that share a row, exchanged through 8 KiB of LDS behind ONE workgroup barrier, and every wave normalises its own columns.
template <int TM, bool AKS, bool BKS, int FL = 0, bool KF = false, bool UNR = false, bool WIDE = false, bool H16 = false, bool LNE = false>
__device__ __forceinline__ void pp_body(const GemmArgs& g, const int v0, const int vstep) {
    constexpr int WMR = TM * 32;                         // output rows per wave
    constexpr int BM = 2 * WMR, BN = 256, NW = 8;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int PA = A_BYTES / 1024, PB = B_BYTES / 1024;     // 1-KiB DMA pieces per K tile
    constexpr int LA = (PA + NW - 1) / NW, LB = PB / NW;        // pieces per wave (the last A round may be padding)
    constexpr int NST = 4;
    constexpr int DUMP = NST * STAGE;                    // 1 KiB that swallows the padding pieces
    constexpr int W = LA + LB + 2;                       // pieces issued after tile t at the point tile t must have landed
    // TAILD (round 6; flavoured kernels on K % 32 == 0): what the tile boundary no longer waits for.  A wave's vector-memory operations
    // retire IN ORDER on gfx950 -- loads, LDS-DMA pieces and stores share one in-order vmcnt (hipcc itself waits vmcnt(N) across mixed loads
    // and stores: tools/vmcnt_order.hip) -- so (a) the zero-fill pieces of the prefetch overrun past the K slice target the 1-KiB dump
    // instead of ring slots: nothing in flight at the end of the loop can land on the epilogue's staging, and the full drain in front of
    // it becomes a bare barrier; (b) behind the last store call a wave waits only until its EIGHT youngest operations are outstanding
    // (everything older -- the next tile's first two K tiles, issued before the stores -- has then landed) instead of for every store
    // acknowledgement: the tail of the store stream retires under the next tile's first K tiles, and the loop's counted waits, which now
    // also cover those stores, stay exact because nothing can overtake them.
    constexpr bool TAILD = KF && FL != 0;
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
    const int nkt_all = (g.K + BK - 1) / BK;
    const int kt_begin = blockIdx.z * g.kt_per_split;
    const int kt_end = (kt_begin + g.kt_per_split < nkt_all) ? kt_begin + g.kt_per_split : nkt_all;
    const int T = kt_end - kt_begin;
    const unsigned a_step = AKS ? (unsigned)(g.lda * BK * 2) : (unsigned)(BK * 2);
    const unsigned b_step = BKS ? (unsigned)(g.ldb * BK * 2) : (unsigned)(BK * 2);
    unsigned a_off[LA], b_off[LB];
    unsigned a_cur[LA], b_cur[LB];                       // KF: byte offset each piece fetches next (invalid rows keep the out-of-range bit)
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
            if (KF) a_cur[i] = a_off[i] == OOB ? OOB : a_off[i] + (unsigned)kt_begin * a_step;
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
            if (KF) b_cur[i] = b_off[i] == OOB ? OOB : b_off[i] + (unsigned)kt_begin * b_step;
        }
    };
    set_tile(v0, m0, n0);


    // Every call issues the same number of DMA instructions (the vmcnt arithmetic depends on it): tiles past the end of
    // this K slice and the padding pieces load from the out-of-range sentinel (zeros, no memory traffic).
    // slot: ring slot of K tile u (= u mod NST) -- passed as a compile-time constant by the 4x-unrolled K loop so that LDS
    // addresses and M0 values are immediates instead of per-phase scalar arithmetic
    auto dma_a = [&](int u, int i, int slot) {           // u = K tile relative to kt_begin
        const int kt = kt_begin + u;
        char* dst = (i * NW + wave < PA) ? smem + slot * STAGE + (i * NW + wave) * 1024 : smem + DUMP;
        if (KF) {                                        // pieces are issued once per K tile, in tile order: a running offset suffices
            const unsigned tail = kt < kt_end ? 0u : OOB;                    // wave-uniform
            if (TAILD && kt >= kt_end) dst = smem + DUMP;                    // the zero-fill overrun never touches a ring slot (see TAILD)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)dst, 16, (int)(a_cur[i] | tail), 0, 0, 0);
            a_cur[i] += a_step;
            return;
        }
        const bool ok = (a_off[i] != OOB) & (kt < kt_end) & (kt * BK + a_kq[i] < g.K);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)dst, 16, (int)(ok ? a_off[i] + (unsigned)kt * a_step : OOB), 0, 0, 0);
    };
    auto dma_b = [&](int u, int i, int slot) {
        const int kt = kt_begin + u;
        char* dst = smem + slot * STAGE + A_BYTES + (i * NW + wave) * 1024;
        if (KF) {
            const unsigned tail = kt < kt_end ? 0u : OOB;
            if (TAILD && kt >= kt_end) dst = smem + DUMP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)dst, 16, (int)(b_cur[i] | tail), 0, 0, 0);
            b_cur[i] += b_step;
            return;
        }
        const bool ok = (b_off[i] != OOB) & (kt < kt_end) & (kt * BK + b_kq[i] < g.K);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)dst, 16, (int)(ok ? b_off[i] + (unsigned)kt * b_step : OOB), 0, 0, 0);
    };
    auto dma_first = [&](int u, int slot) { dma_a(u, 0, slot); dma_a(u, 1, slot); };                                  // 2 pieces
    // third A piece (320-row tiles: 20 pieces for 8 waves): only the waves that own a real one issue it -- a padding piece costs
    // its wave a full LDS-DMA issue slot (60-180 cycles) per K tile for nothing; their counted waits are one lower
    const bool has3 = LA == 3 && 2 * NW + wave < PA;
    auto dma_second = [&](int u, int slot) { dma_b(u, 0, slot); dma_b(u, 1, slot); if (LA == 3 && has3) dma_a(u, 2, slot); };  // LA + LB - 2 pieces
    auto wait_tile = [&]() { if (LA == 3 && !has3) wait_vm<W - 1>(); else wait_vm<W>(); };

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
    // Transposing fragment read of a k-strided operand image.  Issued as INLINE ASM: for the builtin (a DS read with an LDS memory
    // operand) hipcc's wait-count pass assumes it may alias the LDS-DMA writes in flight and puts s_waitcnt vmcnt(0) in front of
    // every group of transposing reads -- the whole prefetch ring drained once per 16-wide K slice in every dX and dW product
    // (found in the round-3 ISA audit; the plain ds_read_b128 fragment loads do not get that wait).  The compiler neither counts
    // nor waits for an asm load: mfma_phase() opens with s_waitcnt lgkmcnt(0) + sched_barrier (cdna_hip_programming.md 5.7).
    // The two 64-bit halves (k rows k_lo .. k_lo+3 and +4: same swizzle term, 4 x 512 B further) share one address register.
    auto frag_ks = [&](const char* base, int col0, int kk) -> bf16x8 {            // both operands' images are 256 columns wide
        const int col = col0 + t_i0 + (tp & 3) * 4, k_lo = kk * 16 + t_kh + (tp >> 2);
        const unsigned addr = (unsigned)(size_t)(LDS_AS const char*)(base + ks_off<256>(k_lo, col >> 3) + (col & 7) * 2);
        s16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(lo), "=&v"(hi) : "v"(addr));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    bf16x8 af[TM], bf[2];
    bf16x8 af2[WIDE ? TM : 1], bf2[2];                      // WIDE: the second 16-wide slice of the K tile
    // optional column sums of the k-strided A operand (bias gradient of a dW product): the wn = 0 waves of the n-tile-0
    // workgroups add up the A fragments they hold anyway (v_dot2c with a vector of ones, in the shadow of the MFMAs)
    bool do_acs = false;
    float acs[TM];
    auto mem_phase = [&](int u, int kk, int slot) {
        const char* sa = smem + slot * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int t = 0; t < 2; ++t) bf[t] = BKS ? frag_ks(sb, wn * 64 + t * 32, kk) : frag_kc(sb, wn * 64 + t * 32, kk);
#pragma unroll
        for (int t = 0; t < TM; ++t) af[t] = AKS ? frag_ks(sa, wm * WMR + t * 32, kk) : frag_kc(sa, wm * WMR + t * 32, kk);
        if (kk == 0) dma_second(u + 2, (slot + 2) & (NST - 1)); else dma_first(u + 3, (slot + 3) & (NST - 1));
    };
    // (Issuing all, or one, of the phase's DMA pieces between the MFMAs instead -- an LDS-DMA issue stalls its wave 60-180
    // cycles -- measured 0-10 % slower than keeping them in the memory half-phase.)
    // The asm fragment reads (frag_ks) are invisible to the compiler's wait-count pass: their results are valid only behind this
    // s_waitcnt.  Every fragment register is passed THROUGH an asm statement that follows the wait ("+v": redefined there), so
    // no copy, live-range split or spill of a fragment can be scheduled between its read and the wait -- the
    // dependency no longer rests on sched_barrier and the allocator's behaviour.  (volatile asms keep their order.)
    auto frag_fence = [&](bf16x8& f) {
        i32x4 t = __builtin_bit_cast(i32x4, f);
        asm volatile("" : "+v"(t));
        f = __builtin_bit_cast(bf16x8, t);
    };
    auto mfma_phase = [&]() {
        if (AKS || BKS) {                                    // the asm fragment reads of this phase's operands (see frag_ks)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 2; ++t) if (BKS) frag_fence(bf[t]);
#pragma unroll
            for (int t = 0; t < TM; ++t) if (AKS) frag_fence(af[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                acc[tn][tm] = mfma16<H16>(bf[tn], af[tm], acc[tn][tm]);
        __builtin_amdgcn_s_setprio(0);
        if (AKS && do_acs) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const i32x4 w = __builtin_bit_cast(i32x4, af[tm]);
#pragma unroll
                for (int j = 0; j < 4; ++j) dot2_ones<H16>(acs[tm], w[j]);
            }
        }
    };

    auto mem_phase_w = [&](int u, int slot) {
        const char* sa = smem + slot * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int t = 0; t < 2; ++t) bf[t] = BKS ? frag_ks(sb, wn * 64 + t * 32, 0) : frag_kc(sb, wn * 64 + t * 32, 0);
#pragma unroll
        for (int t = 0; t < TM; ++t) af[t] = AKS ? frag_ks(sa, wm * WMR + t * 32, 0) : frag_kc(sa, wm * WMR + t * 32, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) bf2[t] = BKS ? frag_ks(sb, wn * 64 + t * 32, 1) : frag_kc(sb, wn * 64 + t * 32, 1);
#pragma unroll
        for (int t = 0; t < (WIDE ? TM : 1); ++t) af2[t] = AKS ? frag_ks(sa, wm * WMR + t * 32, 1) : frag_kc(sa, wm * WMR + t * 32, 1);
        dma_first(u + 2, (slot + 2) & (NST - 1));
        dma_second(u + 2, (slot + 2) & (NST - 1));
    };
    auto mfma_phase_w = [&]() {
        if (AKS || BKS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < 2; ++t) if (BKS) { frag_fence(bf[t]); frag_fence(bf2[t]); }
#pragma unroll
            for (int t = 0; t < TM; ++t) if (AKS) frag_fence(af[t]);
#pragma unroll
            for (int t = 0; t < (WIDE ? TM : 1); ++t) if (AKS) frag_fence(af2[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                acc[tn][tm] = mfma16<H16>(bf[tn], af[tm], acc[tn][tm]);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < (WIDE ? TM : 1); ++tm)
                acc[tn][tm] = mfma16<H16>(bf2[tn], af2[tm], acc[tn][tm]);
        __builtin_amdgcn_s_setprio(0);
        if (AKS && do_acs) {
#pragma unroll
            for (int tm = 0; tm < (WIDE ? TM : 1); ++tm) {
                const i32x4 w = __builtin_bit_cast(i32x4, af[tm]), w2 = __builtin_bit_cast(i32x4, af2[tm]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dot2_ones<H16>(acs[tm], w[j]);
                    dot2_ones<H16>(acs[tm], w2[j]);
                }
            }
        }
    };
    // pieces of ONE K tile may still be in flight at the point the tile before it must have landed
    auto wait_tile_w = [&]() {
        constexpr int NPW = LA + LB;
        if (LA == 3 && !has3) duo_like_wait<NPW - 1>(); else duo_like_wait<NPW>();
    };

    if (g.dephase) {                                     // experiment (env MMAE_PP_DEPHASE = n + 256 * mode): some CUs start n x ~4 us late
        const int n_sleep = g.dephase & 0xff, mode = g.dephase >> 8;
        const int n_mine = (g.tiles_total - v0 + vstep - 1) / vstep, n_max = (g.tiles_total + vstep - 1) / vstep;
        // mode 0: the odd workgroups; 1: the workgroups with a tile less than the busiest (their delay is free); 2: the odd ones among those
        const bool late = mode == 0 ? (blockIdx.x & 1) : (mode == 1 ? n_mine < n_max : (n_mine < n_max && (blockIdx.x & 1)));
        if (late) for (int i = 0; i < n_sleep; ++i) __builtin_amdgcn_s_sleep(127);
    }
    // prologue of the first tile: K tiles 0, 1 and the first half of K tile 2
    dma_first(0, 0); dma_second(0, 0);
    dma_first(1, 1); dma_second(1, 1);
    if (WIDE) {
        wait_tile_w();                                  // K tile 0 has landed, tile 1 may still fly
    } else {
        dma_first(2, 2);
        wait_tile();                                    // K tile 0 has landed (this wave's share)
    }
    wg_barrier();

    // epilogue staging lives in ring slots 2-3 so that slots 0-1 can already receive the NEXT output tile's first two K
    // tiles while this tile's results are written out (the DMA latency and most of the prologue hide under the epilogue)
    char* stage = smem + 2 * STAGE + wave * 8192;
    static_assert(2 * STAGE >= 8 * 8192, "staging must fit in ring slots 2-3");
#ifdef MMAE_PP_TRACE
    const bool tr_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x == 0 || threadIdx.x == 256);
    const int tr_w = threadIdx.x >> 8;
    int tr_n = 0;
    const bool wg_on = threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 1024;
    int wg_tiles = 0;
    if (wg_on) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_pp_wg[blockIdx.x][0] = (long long)__builtin_amdgcn_s_memrealtime();
        g_pp_wg[blockIdx.x][2] = (long long)(xcc & 0xf);
    }
#endif
    for (int v = v0; v < g.tiles_total; v += vstep) {
        PP_STAMP();                                          // A
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

        if (WIDE) {
            if (wm == 0) {
                for (int u0 = 0; u0 < T; u0 += NST) {
#pragma unroll
                    for (int j = 0; j < NST; ++j) {
                        const int u = u0 + j;
                        if (u < T) {
                            mem_phase_w(u, j);
                            wg_barrier();
                            mfma_phase_w();
                            wait_tile_w();                  // K tile u + 1 (tile u + 2 may still fly)
                            wg_barrier();
                        }
                    }
                }
            } else {
                for (int u0 = 0; u0 < T; u0 += NST) {
#pragma unroll
                    for (int j = 0; j < NST; ++j) {
                        const int u = u0 + j;
                        if (u < T) {
                            wg_barrier();
                            mem_phase_w(u, j);
                            wait_tile_w();
                            wg_barrier();
                            mfma_phase_w();
                        }
                    }
                }
            }
        } else if (UNR) {
            // the K loop unrolled over the NST ring slots: slot = u mod NST is a constant in every copy of the body
            if (wm == 0) {
                for (int u0 = 0; u0 < T; u0 += NST) {
#pragma unroll
                    for (int j = 0; j < NST; ++j) {
                        const int u = u0 + j;
                        if (u < T) {
                            mem_phase(u, 0, j);
                            wg_barrier();
                            mfma_phase();
                            wg_barrier();
                            mem_phase(u, 1, j);
                            wg_barrier();
                            mfma_phase();
                            wait_tile();                    // K tile u + 1
                            wg_barrier();
                        }
                    }
                }
            } else {
                for (int u0 = 0; u0 < T; u0 += NST) {
#pragma unroll
                    for (int j = 0; j < NST; ++j) {
                        const int u = u0 + j;
                        if (u < T) {
                            wg_barrier();
                            mem_phase(u, 0, j);
                            wg_barrier();
                            mfma_phase();
                            wg_barrier();
                            mem_phase(u, 1, j);
                            wait_tile();                    // K tile u + 1
                            wg_barrier();
                            mfma_phase();
                        }
                    }
                }
            }
        } else if (wm == 0) {
            for (int u = 0; u < T; ++u) {
                mem_phase(u, 0, u & (NST - 1));
                wg_barrier();
                mfma_phase();
                wg_barrier();
                mem_phase(u, 1, u & (NST - 1));
                wg_barrier();
                mfma_phase();
                wait_tile();                                // K tile u + 1
                wg_barrier();
            }
        } else {
            for (int u = 0; u < T; ++u) {
                wg_barrier();
                mem_phase(u, 0, u & (NST - 1));
                wg_barrier();
                mfma_phase();
                wg_barrier();
                mem_phase(u, 1, u & (NST - 1));
                wait_tile();                                // K tile u + 1
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
        PP_STAMP();                                          // B
        if constexpr (TAILD) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wg_barrier();                                    // every wave is out of the ring; the overrun pieces still in flight target the dump
        } else {
            wait_vm<0>();                                    // the zero-fill tail pieces must not land on live data
            __syncthreads();                                 // every wave is out of the ring
        }
        PP_STAMP();                                          // C

        if constexpr (H16) {                                 // a gradient leaving the fp16-storage domain (f32 C): 1/S, mmae.h MMAE_F16
            if (g.a_amax) {
                const float us = h16_grad_unscale(g.a_amax);
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[tn][tm][r] *= us;
            }
        }
        const int mw = m0 + wm * WMR, nw = n0 + wn * 64;
        const bool has_next = v + vstep < g.tiles_total;
        if (has_next) {                                      // next tile: K tiles 0, 1 -> slots 0, 1 (in flight during the epilogue)
            asm volatile("" : "+v"(lane));
            set_tile(v + vstep, m0, n0);
            dma_first(0, 0); dma_second(0, 0);
            dma_first(1, 1); dma_second(1, 1);
        }
        PP_STAMP();                                          // D
        if constexpr (LNE) {
            static_assert(TM == 4 && !AKS && (FL == FL_F32_BIAS_RESID || FL == FL_F32_BIAS), "LayerNorm side output: f32 bias [+ residual] flavours on 256-row tiles");
            // (N == 256 = four 64-column quarters of a row, one per wn: runtime.hip checks)
            constexpr bool RES = FL == FL_F32_BIAS_RESID;
            f32x4 keep[2][16];
            {
                f32x16 sub[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
                store_tile64_fast<true, 0, RES, true, false, false, false, true>(g, Cz, g.ldc, stage, lane, sub, mw, nw, 2, keep[0]);
            }
            {
                f32x16 sub[2][2] = {{acc[0][2], acc[0][3]}, {acc[1][2], acc[1][3]}};
                store_tile64_fast<true, 0, RES, true, false, false, false, true>(g, Cz, g.ldc, stage, lane, sub, mw + 64, nw, 2, keep[1]);
            }
            float* part = reinterpret_cast<float*>(smem + NST * STAGE + 1024);         // [256 rows][4 column quarters][sum, sum of squares]
            const int c16 = lane & 15, rsub = lane >> 4;
            const bool do_ln = g.ln_g != nullptr;
            if (do_ln) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        // per 64-column quarter of the row: its mean and the sum of squared deviations FROM THAT MEAN (two DPP row sums: the values
                        // are still in registers) -- not (sum, sum of squares), whose difference cancels for rows with |mean| >> std
                        const f32x4 v = keep[s][k];
                        const float mq = row16_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);   // the 16 lanes that share the row: DPP, no LDS traffic
                        const float d0 = v[0] - mq, d1 = v[1] - mq, d2 = v[2] - mq, d3 = v[3] - mq;
                        const float m2 = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
                        const int row_l = wm * WMR + s * 64 + (k >> 3) * 32 + (k & 7) * 4 + rsub;
                        if (c16 == 0) { part[(row_l * 4 + wn) * 2] = mq; part[(row_l * 4 + wn) * 2 + 1] = m2; }
                    }
                __syncthreads();
            }
            const int ncol = nw + c16 * 4;
            f32x4 g4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
            if (do_ln) { g4 = ld4(g.ln_g + ncol); b4 = ld4(g.ln_b + ncol); }
            const float inv_n = 1.0f / (float)g.N;
            const auto rsL = row_rsrc((uint16_t*)g.ln_out, mw, (long long)g.N);
            const int rows_left = g.M - mw;
            // two rows per store: lanes c16 and c16 ^ 1 swap halves (one DPP quad_perm each way), so that the even lane writes eight
            // columns of row k and the odd lane eight columns of row k + 1 -- 16 dwordx4 stores per lane instead of 32 dwordx2
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int k2 = 0; k2 < 16; k2 += 2) {
                    i32x2 pk[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int k = k2 + u;
                        const int r = s * 64 + (k >> 3) * 32 + (k & 7) * 4 + rsub;        // row of this wave's 128
                        f32x4 o = keep[s][k];
                        if (do_ln) {
                            const f32x4 p0 = *reinterpret_cast<const f32x4*>(part + (wm * WMR + r) * 8);
                            const f32x4 p1 = *reinterpret_cast<const f32x4*>(part + (wm * WMR + r) * 8 + 4);
                            // the four quarters (64 columns each) combined as in Chan et al.: M2 = sum M2_q + 64 sum (mean_q - mu)^2
                            const float mu = ((p0[0] + p0[2]) + (p1[0] + p1[2])) * 0.25f;
                            const float e0 = p0[0] - mu, e1 = p0[2] - mu, e2 = p1[0] - mu, e3 = p1[2] - mu;
                            const float var = (((p0[1] + p0[3]) + (p1[1] + p1[3])) + 64.0f * ((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3))) * inv_n;
                            const float rs = 1.0f / sqrtf(var + g.ln_eps);
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = (o[j] - mu) * rs * g4[j] + b4[j];
                            if (wn == 0 && c16 == 0 && r < rows_left) { g.ln_mean[mw + r] = mu; g.ln_rstd[mw + r] = rs; }
                        }
                        pk[u] = pack4_16<H16>(o);
                    }
                    // even lane keeps row k (its own 4 columns + the neighbour's), odd lane row k + 1
                    const bool odd = (c16 & 1) != 0;
                    const i32x2 give = odd ? pk[0] : pk[1], mine = odd ? pk[1] : pk[0];
                    i32x2 got;
                    got[0] = __builtin_amdgcn_update_dpp(0, give[0], 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]: lane ^ 1
                    got[1] = __builtin_amdgcn_update_dpp(0, give[1], 0xB1, 0xF, 0xF, true);
                    i32x4 w4;
                    if (odd) { w4[0] = got[0]; w4[1] = got[1]; w4[2] = mine[0]; w4[3] = mine[1]; }
                    else { w4[0] = mine[0]; w4[1] = mine[1]; w4[2] = got[0]; w4[3] = got[1]; }
                    const int k = k2 + (odd ? 1 : 0);
                    const int r = s * 64 + (k >> 3) * 32 + (k & 7) * 4 + rsub;
                    const int off = (r < rows_left) ? (int)(((long long)r * g.N + nw + (c16 & ~1) * 4) * 2) : (int)OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(w4, rsL, off, 0, 0);
                }
        } else {
        {
            f32x16 sub[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
            gemm_store_tile64_fl<FL, H16>(g, Cz, stage, lane, sub, mw, nw);
        }
        PP_STAMP();                                          // E
        {
            f32x16 sub[2][2] = {{acc[0][2], acc[0][3]}, {acc[1][2], acc[1][3]}};
            gemm_store_tile64_fl<FL, H16>(g, Cz, stage, lane, sub, mw + 64, nw);
        }
        PP_STAMP();                                          // F
        }
        if (!LNE && (TM & 1)) {
            f32x16 sub[2][2] = {{acc[0][TM - 1], acc[0][TM - 1]}, {acc[1][TM - 1], acc[1][TM - 1]}};
            gemm_store_tile64_fl<FL, H16>(g, Cz, stage, lane, sub, mw + (TM - 1) * 32, nw, 1);
        }
        PP_STAMP();                                          // G
        if (has_next) {
            if constexpr (TAILD) {
                // in-order vmcnt: at most the eight youngest operations (stores of the last call) stay outstanding -- the next tile's K tiles
                // 0 and 1 were issued before every store of this epilogue (>= 16 per wave in every compiled flavour) and have landed
                wait_vm<8>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wg_barrier();                                // every wave is out of the staging: slots 2-3 go back to the DMA ring
            } else {
                // generic flavours (masked stores may be skipped: no lower bound on the operations behind the prefetch): drain
                wait_vm<0>();
                __syncthreads();
            }
            if (!WIDE) dma_first(2, 2);
        }
        PP_STAMP();                                          // H
#ifdef MMAE_PP_TRACE
        ++wg_tiles;
#endif
    }
    PP_TRACE_END();
#ifdef MMAE_PP_TRACE
    if (wg_on) { g_pp_wg[blockIdx.x][1] = (long long)__builtin_amdgcn_s_memrealtime(); g_pp_wg[blockIdx.x][3] = wg_tiles; }
#endif
}

template <int TM, bool AKS, bool BKS, int FL = 0, bool KF = false, bool H16 = false, bool LNE = false>
__global__ void __launch_bounds__(512) gemm_bf16_pp_kernel(const GemmArgs g) {
    // unrolled wherever the flavoured instantiation has the registers for it (measured: no scratch)
    pp_body<TM, AKS, BKS, FL, KF, (FL != 0 && KF && !(TM == 5 && BKS)), false, H16, LNE>(g, blockIdx.x, gridDim.x);
}

template <int TM, bool AKS, bool BKS, int FL = 0, bool KF = false, bool H16 = false, bool LNE = false>
int launch(const GemmArgs& g, int batch, hipStream_t st) {
    constexpr int BM = TM * 64, BN = 256;
    const int tiles_m = (g.M + BM - 1) / BM;
    GemmArgs a = g;
    a.tiles_n = (g.N + BN - 1) / BN;
    a.kt_per_split = g.kt_per_split * 2;                 // runtime.hip counts 64-wide K tiles; this kernel steps by 32
    a.tiles_total = tiles_m * a.tiles_n;
    const int n_cu = mmae_cu_avail();
    static const int env_persist = mmae_env_int("MMAE_PP_PERSIST", 1);
    const int gx = (env_persist && a.tiles_total > n_cu) ? n_cu : a.tiles_total;      // one resident workgroup per CU walks the tile list
    dim3 grid(gx, batch, a.splitk), block(512);
    const size_t lds = (size_t)4 * (BM + BN) * 64 + 1024 + (LNE ? (size_t)BM * 4 * 2 * 4 : 0);      // + the row-statistics exchange of the LayerNorm side output
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp_kernel<TM, AKS, BKS, FL, KF, H16, LNE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    hipLaunchKernelGGL((gemm_bf16_pp_kernel<TM, AKS, BKS, FL, KF, H16, LNE>), grid, block, lds, st, a);
    return mmae_check_launch("gemm_bf16_pp");
}

}  // namespace
