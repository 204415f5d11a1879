// (gemm_duo_body.h: kernel body of the "duo" bf16 MFMA GEMM -- gemm_bf16_duo.hip)
// Two INDEPENDENT 4-wave workgroups per CU on 128 x 256 output tiles (a wave owns 128 x 64), instead of the one 8-wave
// ping-pong workgroup per CU of gemm_pp_body.h.
//
// Why (profiles/r02_pmc_mfma.txt, r02_epilogue_dissection.txt): in the ping-pong kernel the two waves of a SIMD alternate in
// lock step behind four workgroup barriers per 32-wide K tile -- the memory half-phase (fragment reads + LDS-DMA issue), not the
// MFMA half-phase, paces the loop (MFMA pipe 31-50 % busy) -- and all 256 CUs reach their store-heavy epilogues together, with
// nothing left on the CU to compute under the store tail (fc1 + GELU: 121 us of loop, 50-75 us of epilogue).  Here
//   * the two waves that share a SIMD belong to DIFFERENT workgroups: no barrier couples them, so one wave's LDS-DMA issue
//     stalls, fragment reads, barrier waits and -- above all -- its whole epilogue run under the other wave's MFMAs;
//   * inside a workgroup every wave software-pipelines itself: the fragments of the next 16-wide K slice are read into a
//     second register set while the 8 MFMAs of the current slice issue (128 accumulators + 2 x 24 fragment registers), and
//     there is ONE workgroup barrier per K tile (16 MFMAs per wave) instead of four;
//   * K tiles stream HBM -> LDS by LDS-DMA into a 3-slot ring (24 KiB per slot: 72 + 8 KiB per workgroup, two per CU); the
//     slot of tile u is re-targeted right after the barrier at which every wave has drained its reads of tile u, so two
//     tiles are in flight behind the one being multiplied, tracked with counted s_waitcnt vmcnt(N).
// The LDS images, DMA piece mapping and epilogue routines are the ping-pong kernel's (kc_off / ks_off, gemm_store_tile64_fl).
//
// Iteration u (K tile u in ring slot u % NST; F0 / F1 = fragment register sets of its two 16-wide slices):
//     read F1 <- slot u          | 8 MFMAs on F0
//     lgkmcnt(0)  (this wave is out of slot u)   vmcnt((NST-2) NP)  (its pieces of tile u+1 have landed)   barrier B_u
//     DMA tile u+NST -> slot u   | read F0 <- slot u+1      | 8 MFMAs on F1
// Read-after-DMA: tile u+1 is first read after B_u, which every wave passes after its counted wait.  Write-after-read: slot u
// is re-targeted after B_u, which every wave passes with its reads of tile u drained.
#pragma once
#include "gemm_pp_body.h"
#ifndef DUO_TRACE
#define DUO_TRACE(tile_no, what)
#endif

namespace {

template <int N> __device__ __forceinline__ void duo_wait_vm() {
    static_assert(N == 0 || N == 4 || N == 6 || N == 8 || N == 12, "add the immediate");
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// NW = 4: 128 x 256 tile, 3-slot ring, two workgroups per CU.  NW = 8: 256 x 256 tile, 4-slot ring, one workgroup per CU (the
// same free-running schedule on the ping-pong kernel's geometry; A/B reference).
// VAR: schedule variant (0 = production; others: experiments / dissection, see half_b)
template <int NW, bool AKS, bool BKS, int FL, bool KF, int VAR = 0>
__device__ __forceinline__ void duo_body(const GemmArgs& g, const int v0, const int vstep) {
    constexpr int TM = 4;
    constexpr int BM = (NW / 4) * 128, BN = 256;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int PA = A_BYTES / 1024, PB = B_BYTES / 1024;
    constexpr int LA = PA / NW, LB = PB / NW, NP = LA + LB;     // DMA pieces per wave and K tile
    constexpr int NST = NW == 4 ? 3 : 4;
    constexpr int PRE = NW == 4 ? 1 : 2;                         // K tiles of the NEXT output tile prefetched under the epilogue (NW = 4: one, so that the
                                                                 // staging fits in 72 KiB -- two 80-KiB workgroups were NOT co-resident although the occupancy query said so)
    constexpr int STG = PRE * STAGE;                             // epilogue staging: NW x 8 KiB above the first PRE slots
    static_assert(PA % NW == 0 && PB % NW == 0, "whole pieces per wave");
    static_assert(STG + NW * 8192 <= NST * STAGE, "staging must fit behind the prefetch slots");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    int lane = tid & 63;                                         // laundered per output tile (see gemm_pp_body.h)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int z = blockIdx.y, zo = __builtin_amdgcn_readfirstlane(z / g.nb_inner), zi = z - zo * g.nb_inner;
    const uint16_t* Az = sgpr_ptr((const uint16_t*)g.A + zo * g.sAo + zi * g.sAi);
    const uint16_t* Bz = sgpr_ptr((const uint16_t*)g.B + zo * g.sBo + zi * g.sBi);
    char* Cz = (char*)g.C + (zo * g.sCo + zi * g.sCi) * (g.c_f32 ? 4 : 2);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, 0x80000000, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bz, 0, 0x80000000, 0x00020000);

    const int nkt_all = (g.K + BK - 1) / BK;
    const int kt_begin = blockIdx.z * g.kt_per_split;
    const int kt_end = (kt_begin + g.kt_per_split < nkt_all) ? kt_begin + g.kt_per_split : nkt_all;
    const int T = kt_end - kt_begin;
    const unsigned a_step = AKS ? (unsigned)(g.lda * BK * 2) : (unsigned)(BK * 2);
    const unsigned b_step = BKS ? (unsigned)(g.ldb * BK * 2) : (unsigned)(BK * 2);
    unsigned a_off[LA], b_off[LB];
    unsigned a_cur[LA], b_cur[LB];
    int a_kq[LA], b_kq[LB];
    int m0 = 0, n0 = 0;
    auto set_tile = [&](int v, int& tm0, int& tn0) {
        const int tile = g.xcd_swizzle ? xcd_tile(v, g.tiles_total) : v;
        const int tile_m = __builtin_amdgcn_readfirstlane(tile / g.tiles_n);
        tm0 = tile_m * BM; tn0 = (tile - tile_m * g.tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int seg = i * NW + wave;
            if (!AKS) {
                const int b_abs = 4 * seg + (lane >> 4), j = (lane & 15) ^ (b_abs & 15);
                const int row = 4 * b_abs + (j >> 2), c = j & 3;
                a_kq[i] = c * 8;
                a_off[i] = (tm0 + row < g.M) ? (unsigned)((((long long)(tm0 + row)) * g.lda + c * 8) * 2) : OOB;
            } else {
                constexpr int CPR = BM / 8;
                const int krow = seg * (64 / CPR) + lane / CPR, ch = (lane % CPR) ^ ((krow & 3) << 2);
                a_kq[i] = krow;
                a_off[i] = (tm0 + ch * 8 < g.M) ? (unsigned)((((long long)krow) * g.lda + tm0 + ch * 8) * 2) : OOB;
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

    // piece i of K tile u (relative to kt_begin) -> ring slot `slot`; i < LA: A pieces, then B.  Every call issues exactly
    // one DMA instruction (the counted waits depend on it): past the K slice it fetches the out-of-range sentinel (zeros).
    auto dma_piece = [&](int u, int i, int slot) {
        const int kt = kt_begin + u;
        if (i < LA) {
            char* dst = smem + slot * STAGE + (i * NW + wave) * 1024;
            if (KF) {
                const unsigned tail = kt < kt_end ? 0u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)dst, 16, (int)(a_cur[i] | tail), 0, 0, 0);
                a_cur[i] += a_step;
            } else {
                const bool ok = (a_off[i] != OOB) & (kt < kt_end) & (kt * BK + a_kq[i] < g.K);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)dst, 16, (int)(ok ? a_off[i] + (unsigned)kt * a_step : OOB), 0, 0, 0);
            }
        } else {
            const int ib = i - LA;
            char* dst = smem + slot * STAGE + A_BYTES + (ib * NW + wave) * 1024;
            if (KF) {
                const unsigned tail = kt < kt_end ? 0u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)dst, 16, (int)(b_cur[ib] | tail), 0, 0, 0);
                b_cur[ib] += b_step;
            } else {
                const bool ok = (b_off[ib] != OOB) & (kt < kt_end) & (kt * BK + b_kq[ib] < g.K);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)dst, 16, (int)(ok ? b_off[ib] + (unsigned)kt * b_step : OOB), 0, 0, 0);
            }
        }
    };
    auto dma_tile = [&](int u, int slot) {
#pragma unroll
        for (int i = 0; i < NP; ++i) dma_piece(u, i, slot);
    };

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
    auto read_b = [&](int slot, int kk, int t) -> bf16x8 {
        const char* sb = smem + slot * STAGE + A_BYTES;
        return BKS ? frag_ks_b(sb, wn * 64 + t * 32, kk) : frag_kc(sb, wn * 64 + t * 32, kk);
    };
    auto read_a = [&](int slot, int kk, int t) -> bf16x8 {
        const char* sa = smem + slot * STAGE;
        return AKS ? frag_ks_a(sa, wm * 128 + t * 32, kk) : frag_kc(sa, wm * 128 + t * 32, kk);
    };

    bf16x8 af0[TM], bf0[2], af1[TM], bf1[2];
    constexpr int CA = AKS ? 2 : 1, CB = BKS ? 2 : 1;      // DS instructions per fragment (transposing reads come in pairs)
    // optional column sums of the k-strided A operand (bias gradient of a dW product), as in the ping-pong kernel
    bool do_acs = false;
    float acs[TM];
    auto acs_add = [&](const bf16x8 (&af)[TM]) {
        if (AKS && do_acs) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const i32x4 w = __builtin_bit_cast(i32x4, af[tm]);
#pragma unroll
                for (int j = 0; j < 4; ++j) asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acs[tm]) : "v"(w[j]), "v"(0x3f803f80));
            }
        }
    };
    // MFMA i of a 16-wide slice: i = tn * TM + tm (the two B fragments are needed first, then one A fragment per pair)
    auto mfma1 = [&](const bf16x8 (&af)[TM], const bf16x8 (&bf)[2], int i) {
        const int tm = i >> 1, tn = i & 1;
        acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[tn], af[tm], acc[tn][tm], 0, 0, 0);
    };

    // first half of iteration u: F1 <- (slot, kk = 1) beside the MFMAs on F0
    auto half_a = [&](int slot) {
        if (VAR != 3 && VAR != 4) {
            bf1[0] = read_b(slot, 1, 0);
            bf1[1] = read_b(slot, 1, 1);
#pragma unroll
            for (int t = 0; t < TM; ++t) af1[t] = read_a(slot, 1, t);
        }
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) mfma1(af0, bf0, i);
        acs_add(af0);
        // interleave: one fragment read behind each of the first six MFMAs
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, CB, 0);     // DS read(s) of one B fragment
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, CA, 0);
        }
    };
    // second half: re-target slot (tile u + NST), F0 <- (next slot, kk = 0), MFMAs on F1.  Source order = issue order of the
    // LDS traffic (the compiler keeps LDS-DMA and ds_read in program order): one DMA piece and one fragment read per MFMA
    auto half_b = [&](int u, int slot, int nslot) {
        if (VAR == 1) {
            // DMA pieces among bare MFMAs (an LDS-DMA issue costs ~60 cycles there, 100-185 when mixed with fragment reads --
            // MI355X_MICROARCH.md), the fragment reads of the next slice behind the last MFMA
#pragma unroll
            for (int i = 0; i < NP; ++i) dma_piece(u + NST, i, slot);
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) mfma1(af1, bf1, i);
            acs_add(af1);
#pragma unroll
            for (int i = 0; i < 2; ++i) bf0[i] = read_b(nslot, 0, i);
#pragma unroll
            for (int t = 0; t < TM; ++t) af0[t] = read_a(nslot, 0, t);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM - NP, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * CB + TM * CA, 0);
            return;
        }
        if (VAR >= 2) {                                   // dissection builds (results are wrong): 2 = no DMA, 3 = no fragment reads, 4 = neither
#pragma unroll
            for (int i = 0; i < 2 + TM; ++i) {
                if (VAR == 3 && i < NP) dma_piece(u + NST, i, slot);
                if (VAR == 2) { if (i < 2) bf0[i] = read_b(nslot, 0, i); else af0[i - 2] = read_a(nslot, 0, i - 2); }
            }
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) mfma1(af1, bf1, i);
            if (VAR == 3) {
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            if (VAR == 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, CB, 0);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, CA, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2 + TM; ++i) {
            if (i < NP) dma_piece(u + NST, i, slot);
            if (i < 2) bf0[i] = read_b(nslot, 0, i); else af0[i - 2] = read_a(nslot, 0, i - 2);
        }
#pragma unroll
        for (int i = 2 + TM; i < NP; ++i) dma_piece(u + NST, i, slot);
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) mfma1(af1, bf1, i);
        acs_add(af1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one LDS-DMA piece (VMEM read)
            __builtin_amdgcn_sched_group_barrier(0x100, CB, 0);     // DS read(s) of one B fragment
        }
#pragma unroll
        for (int i = 2; i < (NP < 2 + TM ? NP : 2 + TM); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, CA, 0);
        }
#pragma unroll
        for (int i = NP; i < 2 + TM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, CA, 0);
        }
    };

    // Two workgroups that share a CU must run OUT OF PHASE for one's epilogue to sit under the other's K loop: started
    // together on identical work they stay in lock step (both in the loop at half speed, then both in the epilogue).  The
    // second half of the grid (the workgroups dispatched onto already occupied CUs) starts g.dephase x ~4 us late.
    if (NW == 4 && g.dephase && blockIdx.x >= (gridDim.x >> 1)) {
        for (int i = 0; i < g.dephase; ++i) __builtin_amdgcn_s_sleep(127);
    }
    // prologue of the first tile: K tiles 0 .. NST-1
#pragma unroll
    for (int s = 0; s < NST; ++s) dma_tile(s, s);

    char* stage = smem + STG + wave * 8192;
    int tile_no = 0;
    for (int v = v0; v < g.tiles_total; v += vstep, ++tile_no) {
        DUO_TRACE(tile_no, 0);
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

        duo_wait_vm<(NST - 1) * NP>();                     // K tile 0 has landed (this wave's pieces)
        wg_barrier();
        bf0[0] = read_b(0, 0, 0);
        bf0[1] = read_b(0, 0, 1);
#pragma unroll
        for (int t = 0; t < TM; ++t) af0[t] = read_a(0, 0, t);
        if (VAR == 3 || VAR == 4) {                          // dissection builds without fragment reads: defined operands
            bf1[0] = bf0[0]; bf1[1] = bf0[1];
#pragma unroll
            for (int t = 0; t < TM; ++t) af1[t] = af0[t];
        }

        for (int u0 = 0; u0 < T; u0 += NST) {
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                const int u = u0 + j;
                if (u < T) {
                    half_a(j);
                    __builtin_amdgcn_sched_barrier(0);      // every MFMA of the slice is issued before the wave parks on the waits
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    duo_wait_vm<(NST - 2) * NP>();          // K tile u + 1
                    wg_barrier();
                    half_b(u, j, (j + 1) % NST);
                }
            }
        }
        if (AKS && do_acs) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const float sv = acs[tm] + __shfl_xor(acs[tm], 32, 64);
                const int m = m0 + wm * 128 + tm * 32 + (lane & 31);
                if (lane < 32 && m < g.M) g.acs[(long long)blockIdx.z * g.M + m] = sv;
            }
        }
        duo_wait_vm<0>();                                    // the zero-fill tail pieces must not land on live data
        __syncthreads();                                     // every wave is out of the ring
        DUO_TRACE(tile_no, 1);

        const int mw = m0 + wm * 128, nw = n0 + wn * 64;
        const bool has_next = v + vstep < g.tiles_total;
        if (has_next) {                                      // next tile: K tiles 0 .. PRE-1 stream in during the epilogue
            asm volatile("" : "+v"(lane));
            set_tile(v + vstep, m0, n0);
#pragma unroll
            for (int s = 0; s < PRE; ++s) dma_tile(s, s);
        }
        {
            f32x16 sub[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
            gemm_store_tile64_fl<FL>(g, Cz, stage, lane, sub, mw, nw);
        }
        {
            f32x16 sub[2][2] = {{acc[0][2], acc[0][3]}, {acc[1][2], acc[1][3]}};
            gemm_store_tile64_fl<FL>(g, Cz, stage, lane, sub, mw + 64, nw);
        }
        if (has_next) {
            // loads and stores share vmcnt and retire out of order with respect to each other: drain, then hand the staging
            // region back to the ring
            duo_wait_vm<0>();
            __syncthreads();
#pragma unroll
            for (int s = PRE; s < NST; ++s) dma_tile(s, s);
        }
        DUO_TRACE(tile_no, 2);
    }
}

}  // namespace
