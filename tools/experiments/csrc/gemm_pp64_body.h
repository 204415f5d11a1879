// (gemm_pp64_body.h: the 8-wave ping-pong GEMM of gemm_pp_body.h on 64-wide K tiles)
//
// Why: the LDS-DMA path moves whole 128-byte cache lines L2 -> L1 whatever a piece asks of them.  With 32-wide K tiles a
// k-contiguous operand contributes 64 B per row and piece -- HALF a line -- and the other half comes back one K tile later, after
// 32+ KiB of other pieces have gone through the 32-KiB L1: every line is fetched twice.  Measured on an L2-resident panel
// (tools/dma_probe.hip, profiles/r03_dma_probe_lines.txt): 33 cycles per 1-KiB piece and CU for 16 rows x 64 B against 17 for
// 8 rows x 128 B.  At 32 (256-row tiles) to 36 pieces per 32-wide K tile that is 1 060-1 200 cycles of DMA per 1 024-1 280
// cycles of MFMA work: the loop was DMA-bound (MFMA pipe 31-50 % busy, profiles/r02_pmc_mfma.txt), which is also why the
// products whose B operand is k-strided -- 512-byte rows, whole lines -- were the fast ones.  Here a K tile is 64 wide: a
// k-contiguous piece is 8 rows x 128 B = 8 whole lines, half the pieces per FLOP, each at the full-line rate.
//
// LDS: two slots of (BM + 256) x 128 B (144 KiB at BM = 320).  Only one K tile can be in flight behind the one being
// multiplied, and its slot is free only when the previous tile has been read out, so the pieces of tile u+1 are issued in the
// FIRST sub-steps of tile u: waves 4-7 (group 1, whose memory half-phase follows the barrier at which tile u-1 is drained)
// stream the A operand -- the activations, which miss L2 once per row panel -- in sub-steps 0-2; waves 0-3 stream B (weights:
// L2 / MALL hits) in sub-steps 1-2.  Nothing else is in flight inside the loop, so the wait before tile u+1's first read is a
// plain vmcnt(0) of the wave's own 8-10 pieces.
//
// Barrier numbering (b_j = j-th workgroup barrier; sub-step q = 4u + s covers the 16-wide slice s of K tile u):
//   group 0 (waves 0-3):  MEM(q) b_2q MFMA(q) b_2q+1          group 1 (waves 4-7):  b_2q MEM(q) b_2q+1 MFMA(q)
// Write-after-read: slot (u+1)%2 held tile u-1, whose last reads (group 1, MEM(4u-1)) are drained before b_8u; group 1 issues
// from MEM(4u) (after b_8u), group 0 from MEM(4u+1) (after b_8u+1).  Read-after-DMA: every wave waits for its pieces before
// b_8u+7; the first read of tile u+1 (group 0, MEM(4u+4)) comes after b_8u+7.
#pragma once
#include "gemm_pp_body.h"

namespace {

// k-contiguous operand image of a 64-wide K tile: rows of 128 B (8 chunks of 16 B = one cache line), two rows per 256-B bank row;
// chunk c of row r lives in slot c ^ ((r >> 1) & 7): a 16-lane ds_read_b128 group (rows {0-3,12-15,20-27} + 32 j of one chunk
// column) touches 16 distinct 16-byte bank groups
__device__ __forceinline__ int kc64_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }

// VAR: 0 = production; dissection builds (wrong results): 1 = no DMA in the K loop, 2 = no fragment reads, 3 = neither, 4 = neither and no barriers,
// 5 = the loop never waits for its DMA pieces, 6 = every K tile re-fetches K tile 0 (L2-hot lines)
template <int TM, bool AKS, bool BKS, int FL, int VAR = 0, bool PF = false>
__device__ __forceinline__ void pp64_body(const GemmArgs& g, const int v0, const int vstep) {
    constexpr int WMR = TM * 32;
    constexpr int BM = 2 * WMR, BN = 256;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int PA = A_BYTES / 1024, PB = B_BYTES / 1024;
    constexpr int N1 = PA / 4, N0 = PB / 4;              // pieces per wave and K tile: group 1 streams A, group 0 streams B
    constexpr int NMAX = N1 > N0 ? N1 : N0;
    constexpr int G1_S0 = (N1 + 2) / 3, G1_S1 = (N1 - G1_S0 + 1) / 2;     // group 1: sub-steps 0 / 1 / 2 (10 -> 4 3 3, 8 -> 3 3 2)
    constexpr int G0_S1 = (N0 + 1) / 2;                                   // group 0: sub-steps 1 / 2
    static_assert(PA % 4 == 0 && PB % 4 == 0, "whole pieces per wave");
    static_assert(!AKS || TM == 4, "k-strided A needs a 256-column tile image");
    static_assert(STAGE >= 8 * 8192, "epilogue staging lives in slot 1");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    int lane = tid & 63;                                 // laundered per output tile (gemm_pp_body.h)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int z = blockIdx.y, zo = __builtin_amdgcn_readfirstlane(z / g.nb_inner), zi = z - zo * g.nb_inner;
    const uint16_t* Az = sgpr_ptr((const uint16_t*)g.A + zo * g.sAo + zi * g.sAi);
    const uint16_t* Bz = sgpr_ptr((const uint16_t*)g.B + zo * g.sBo + zi * g.sBi);
    char* Cz = (char*)g.C + (zo * g.sCo + zi * g.sCi) * (g.c_f32 ? 4 : 2);
    // this wave's operand: group 1 (wm = 1) streams A, group 0 streams B
    const auto rsMine = __builtin_amdgcn_make_buffer_rsrc((void*)(wm ? Az : Bz), 0, 0x80000000, 0x00020000);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, 0x80000000, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bz, 0, 0x80000000, 0x00020000);
    const int n_mine = wm ? N1 : N0;

    const int nkt_all = g.K / 64;                        // K % 64 == 0 (checked by the launcher)
    const int kt_begin = blockIdx.z * g.kt_per_split;
    const int kt_end = (kt_begin + g.kt_per_split < nkt_all) ? kt_begin + g.kt_per_split : nkt_all;
    const int T = kt_end - kt_begin;
    const unsigned a_step = AKS ? (unsigned)(g.lda * 64 * 2) : 128u;
    const unsigned b_step = BKS ? (unsigned)(g.ldb * 64 * 2) : 128u;
    const unsigned my_step = wm ? a_step : b_step;
    // DMA addressing: piece i of this wave = piece 0 shifted by i x (32 rows | 8 k-rows): ONE per-lane offset (voffset) + a scalar
    // i * rowstep + K offset (soffset) instead of a running per-piece offset register.  Rows past the operand's edge: M, N are
    // multiples of 8 (checked by the launcher), so a k-contiguous piece (8 rows) is inside or outside as a whole -- bit i of vmask.
    unsigned base = OOB, vmask = 0;
    // L2 prefetch (PF, experiment -- measured 4 % SLOWER, profiles/r03_pp64_dissection.txt): one lane per 128-byte line of a K
    // tile, issued three K tiles ahead as plain sc1 loads into a dead register, on the idea that the two-slot ring's half K-tile
    // period of cover is too short for an HBM miss of the activation operand.  The prefetch loads travel through the same per-CU
    // L1 miss machinery as the DMA pieces, so they take from the resource they were meant to relieve.
    unsigned pfa = OOB, pfb = OOB;                       // byte offset of this lane's line in K tile 0
    const unsigned my_ld2 = (unsigned)((wm ? g.lda : g.ldb) * 2);
    const bool my_ks = wm ? AKS : BKS;
    const unsigned rowstep = my_ks ? 8u * my_ld2 : 32u * my_ld2;
    int m0 = 0, n0 = 0;
    auto set_tile = [&](int v, int& tm0, int& tn0) {
        const int tile = g.xcd_swizzle ? xcd_tile(v, g.tiles_total) : v;
        const int tile_m = __builtin_amdgcn_readfirstlane(tile / g.tiles_n);
        tm0 = tile_m * BM; tn0 = (tile - tile_m * g.tiles_n) * BN;
        const int e0 = wm ? tm0 : tn0, lim = wm ? g.M : g.N;          // this wave's operand: first row / column of the tile, extent
        if (!my_ks) {
            const int row = 8 * wn + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);     // (row >> 1) & 7 is the same for every piece (rows + 32 i)
            base = (unsigned)(((long long)(e0 + row)) * my_ld2 + c * 16);
            vmask = 0;
#pragma unroll
            for (int i = 0; i < NMAX; ++i) vmask |= (e0 + 32 * i + 8 * wn + 8 <= lim) ? (1u << i) : 0u;
        } else {
            const int krow = wn * 2 + (lane >> 5), ch = (lane & 31) ^ ((krow & 3) << 2);   // 256-column image: 2 k-rows per piece; (krow + 8 i) & 3 is the same for every piece
            base = (e0 + ch * 8 < lim) ? (unsigned)((((long long)krow) * my_ld2) + (e0 + ch * 8) * 2) : OOB;
            vmask = 0xffffffffu;
        }
        if (PF) {
            if (!AKS) { const int row = wave * (BM / 8) + lane; pfa = (lane < BM / 8 && tm0 + row < g.M) ? (unsigned)(((long long)(tm0 + row)) * g.lda * 2) : OOB; }
            else { const int krow = wave * 8 + (lane >> 2), col = tm0 + (lane & 3) * 64; pfa = (lane < 32 && col < g.M) ? (unsigned)((((long long)krow) * g.lda + col) * 2) : OOB; }
            if (!BKS) { const int row = wave * 32 + lane; pfb = (lane < 32 && tn0 + row < g.N) ? (unsigned)(((long long)(tn0 + row)) * g.ldb * 2) : OOB; }
            else { const int krow = wave * 8 + (lane >> 2), col = tn0 + (lane & 3) * 64; pfb = (lane < 32 && col < g.N) ? (unsigned)((((long long)krow) * g.ldb + col) * 2) : OOB; }
        }
    };
    set_tile(v0, m0, n0);
    int pf0 = 0, pf1 = 0;
    // touch the lines of K tile t (relative to kt_begin): results are never used, the registers stay reserved until pf_retire()
    auto pf_issue = [&](int t) {
        if (PF && t < T) {
            pf0 = __builtin_amdgcn_raw_buffer_load_b32(rsA, (int)pfa, (int)((unsigned)(kt_begin + t) * a_step), 16);     // aux 16 = sc1: served by L2, not allocated in L1
            pf1 = __builtin_amdgcn_raw_buffer_load_b32(rsB, (int)pfb, (int)((unsigned)(kt_begin + t) * b_step), 16);
        }
    };
    auto pf_retire = [&]() { if (PF) asm volatile("" :: "v"(pf0), "v"(pf1)); };

    // piece i of this wave of K tile t (relative to kt_begin) into ring slot `slot`
    const int my_base = (wm ? 0 : A_BYTES) + wn * 1024;
    auto dma_mine = [&](int i, int slot, int t) {
        char* dst = smem + slot * STAGE + my_base + i * 4096;
        const unsigned voff = ((vmask >> i) & 1u) ? base : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsMine, (LDS_AS void*)dst, 16, (int)voff, (int)((unsigned)i * rowstep + (unsigned)(kt_begin + t) * my_step), 0, 0);
    };
    auto dma_all = [&](int slot, int t) {                // a whole K tile (prologue of an output tile)
#pragma unroll
        for (int i = 0; i < NMAX; ++i) if (i < n_mine) dma_mine(i, slot, t);
    };

    f32x16 acc[2][TM];
    int fr = 0, fk = 0, tp = 0, t_i0 = 0, t_kh = 0;
    auto derive = [&]() {
        fr = lane & 31; fk = lane >> 5;
        const int tg = lane >> 4;
        tp = lane & 15; t_i0 = (tg & 1) * 16; t_kh = (tg >> 1) * 8;
    };
    derive();
    auto frag_kc = [&](const char* base, int row0, int s) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(base + kc64_off(row0 + fr, s * 2 + fk));
    };
    auto frag_ks = [&](const char* base, int col0, int s) -> bf16x8 {            // both operands' k-strided images are 256 columns wide
        // inline asm, not the builtin: see frag_ks in gemm_pp_body.h (hipcc drains vmcnt before every builtin transposing read)
        const int col = col0 + t_i0 + (tp & 3) * 4, k_lo = s * 16 + t_kh + (tp >> 2);
        const unsigned addr = (unsigned)(size_t)(LDS_AS const char*)(base + ks_off<256>(k_lo, col >> 3) + (col & 7) * 2);
        s16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(lo), "=&v"(hi) : "v"(addr));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    bf16x8 af[TM], bf[2];
    bool do_acs = false;
    float acs[TM];
    auto read_frags = [&](int s, int slot) {
        const char* sa = smem + slot * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int t = 0; t < 2; ++t) bf[t] = BKS ? frag_ks(sb, wn * 64 + t * 32, s) : frag_kc(sb, wn * 64 + t * 32, s);
#pragma unroll
        for (int t = 0; t < TM; ++t) af[t] = AKS ? frag_ks(sa, wm * WMR + t * 32, s) : frag_kc(sa, wm * WMR + t * 32, s);
    };
    // memory half-phase of sub-step s of a K tile in `slot`; `more`: a next K tile exists and streams into the other slot
    auto mem_phase = [&](int s, int slot, bool more, int tn) {
        if (VAR == 0 || VAR == 1 || VAR >= 5) read_frags(s, slot);
        if (VAR == 6) tn = 0;                               // dissection: every K tile re-fetches tile 0 (L2-hot lines)
        if (more && (VAR == 0 || VAR == 2 || VAR >= 5)) {
            if (wm) {
                if (s == 0) {
#pragma unroll
                    for (int i = 0; i < G1_S0; ++i) dma_mine(i, slot ^ 1, tn);
                } else if (s == 1) {
#pragma unroll
                    for (int i = G1_S0; i < G1_S0 + G1_S1; ++i) dma_mine(i, slot ^ 1, tn);
                } else if (s == 2) {
#pragma unroll
                    for (int i = G1_S0 + G1_S1; i < N1; ++i) dma_mine(i, slot ^ 1, tn);
                }
            } else {
                if (s == 1) {
#pragma unroll
                    for (int i = 0; i < G0_S1; ++i) dma_mine(i, slot ^ 1, tn);
                } else if (s == 2) {
#pragma unroll
                    for (int i = G0_S1; i < N0; ++i) dma_mine(i, slot ^ 1, tn);
                }
            }
        }
    };
    auto mfma_phase = [&]() {
        if (AKS || BKS) {                                    // the asm fragment reads of this phase's operands
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
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

    auto loop_barrier = [&]() { if (VAR != 4) wg_barrier(); else __builtin_amdgcn_sched_barrier(0); };
    // prologue of the first output tile: K tile 0
    dma_all(0, 0);
    pf_issue(1);
    wait_vm<0>();
    pf_retire();
    pf_issue(2);
    wg_barrier();

    char* stage = smem + STAGE + wave * 8192;            // epilogue staging in slot 1; slot 0 receives the next output tile's K tile 0 meanwhile
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
        if (VAR >= 2 && VAR <= 4) read_frags(0, 0);                     // dissection builds without fragment reads: defined operands

        if (wm == 0) {
            for (int u0 = 0; u0 < T; u0 += 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int u = u0 + j;
                    if (u < T) {
                        const bool more = u + 1 < T;
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            mem_phase(s, j, more, u + 1);
                            loop_barrier();
                            mfma_phase();
                            if (s == 3) { if (VAR != 5) wait_vm<0>(); pf_retire(); pf_issue(u + 3); }       // this wave's pieces of K tile u + 1 have landed
                            loop_barrier();
                        }
                    }
                }
            }
        } else {
            for (int u0 = 0; u0 < T; u0 += 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int u = u0 + j;
                    if (u < T) {
                        const bool more = u + 1 < T;
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            loop_barrier();
                            mem_phase(s, j, more, u + 1);
                            if (s == 3) { if (VAR != 5) wait_vm<0>(); pf_retire(); pf_issue(u + 3); }
                            loop_barrier();
                            mfma_phase();
                        }
                    }
                }
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
        __syncthreads();                                     // every wave is out of the ring (no DMA is outstanding here)

        const int mw = m0 + wm * WMR, nw = n0 + wn * 64;
        const bool has_next = v + vstep < g.tiles_total;
        if (has_next) {                                      // next output tile: its K tile 0 streams into slot 0 during the epilogue
            asm volatile("" : "+v"(lane));
            set_tile(v + vstep, m0, n0);
            dma_all(0, 0);
            pf_issue(1);
        }
        {
            f32x16 sub[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
            gemm_store_tile64_fl<FL>(g, Cz, stage, lane, sub, mw, nw);
        }
        {
            f32x16 sub[2][2] = {{acc[0][2], acc[0][3]}, {acc[1][2], acc[1][3]}};
            gemm_store_tile64_fl<FL>(g, Cz, stage, lane, sub, mw + 64, nw);
        }
        if (TM & 1) {
            f32x16 sub[2][2] = {{acc[0][TM - 1], acc[0][TM - 1]}, {acc[1][TM - 1], acc[1][TM - 1]}};
            gemm_store_tile64_fl<FL>(g, Cz, stage, lane, sub, mw + (TM - 1) * 32, nw, 1);
        }
        if (has_next) {
            wait_vm<0>();                                    // K tile 0 landed, stores acknowledged
            pf_retire();
            pf_issue(2);
            __syncthreads();
        }
    }
}

}  // namespace
