// Kernel-argument block and fused epilogue shared by the bf16 and f32 MFMA GEMM kernels.
#pragma once
#include "common.h"

struct GemmArgs {
    const void* A; const void* B; void* C;
    int M, N, K;
    long long lda, ldb, ldc;
    int nb_inner;
    long long sAo, sAi, sBo, sBi, sCo, sCi;
    const float* bias;
    const float* resid; long long ldr;
    void* aux; long long ldaux;
    int c_f32, aux_f32, epi, accumulate, vec;
    float alpha;
    int tiles_n, tiles_total;
    int splitk, kt_per_split;     // splitk > 1: z-slice s writes its partial product to ws[s][M][N] (f32)
    float* ws;
    int xcd_swizzle;
    float* acs;                   // optional [splitk][M] partial column sums of the k-strided A operand (ping-pong kernel)
    float* colpart;               // optional [ceil(M/32)][N] column sums of the epilogue output per 32-row block (dGELU flavour)
    int wide_st;                  // bf16 epilogues with 8-column (16-byte) lanes (store_tile64_bf16x8); env MMAE_EPI_WIDE=0 turns it off
    int aux_grad;                 // MMAE_EPI_GELU_G / MMAE_EPI_MUL: aux keeps GELU'(pre-activation) instead of the pre-activation (mmae.h)
    int dephase;                  // experiment (env MMAE_PP_DEPHASE = n + 256 * mode): some workgroups of the ping-pong kernel start n x ~4 us late
    const void* scA; const void* scB;   // MX-fp8 products: packed E8M0 scales of the two operands (mxfp8.hip)
    unsigned char* qout; unsigned char* qsc; long long ldq;   // ..._Q flavours: also emit the MX-fp8 quantisation of the bf16 output C ([M][ldq] bytes + packed scales)
    int h16;                      // the 16-bit operands / outputs / aux of this product are fp16 (MMAE_F16), not bf16: flavoured ping-pong kernels only
    const float* a_amax;          // MMAE_F32F16 products: device scalar whose power of two pre-scales the A operand (gemm_f32x3.hip), or NULL
    // LayerNorm (or plain 16-bit cast) of the f32 rows this product completes, written beside C (mmae_gemm_desc.ln_out; N == 256: the
    // 256-column tile spans the row): gamma / beta NULL = cast only
    const float* ln_g; const float* ln_b; void* ln_out; float* ln_mean; float* ln_rstd; float ln_eps;
    int dbg;                      // epilogue dissection for profiling (env MMAE_EPI_DBG, GELU flavour only): 1 = no GELU arithmetic, 2 = no pre-activation store,
                                  // 3 = arithmetic but no stores, 4 = nothing.  0 in production.
};

// XCD-aware tile order (MI355X: workgroup b runs on XCD b % 8, each XCD has its own 4 MiB L2): hand every XCD a
// contiguous range of the row-major tile list, so the workgroups resident on one XCD share A row-panels (and walk B
// in step) instead of each XCD streaming every A panel through its own L2.  Bijective for any tile count.
__device__ __forceinline__ int xcd_tile(int bid, int ntiles) {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// The kernels compute D[i = n][j = m] (B fragment as the MFMA "A" operand) so that one lane
// owns 4 consecutive n for a fixed m: a row-per-lane epilogue with 8/16-byte accesses.
// acc holds columns n .. n+3 of row m.
__device__ __forceinline__ void gemm_epilogue4(const GemmArgs& g, char* Cz, int m, int n, f32x4 acc) {
    if (m >= g.M || n >= g.N) return;
    const int cnt = (g.N - n) < 4 ? (g.N - n) : 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = acc[j] * g.alpha;
    if (g.splitk > 1) {   // split-K partial slab (dense [M][N] f32); combined by splitk_reduce_kernel
        float* c = g.ws + ((long long)blockIdx.z * g.M + m) * g.N + n;
        if (cnt == 4 && (g.N & 3) == 0) { f32x4 t = {v[0], v[1], v[2], v[3]}; st4(c, t); } else for (int j = 0; j < cnt; ++j) c[j] = v[j];
        return;
    }
    const bool full = g.vec && (cnt == 4);
    if (g.bias) {
        if (full) { f32x4 b = ld4(g.bias + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += b[j];
        } else { for (int j = 0; j < cnt; ++j) v[j] += g.bias[n + j]; }
    }
    if (g.epi == MMAE_EPI_GELU) {
        const long long ao = (long long)m * g.ldaux + n;
        float s4[4];                                     // what aux keeps: the pre-activation, or (aux_grad) GELU' of it
        if (!g.c_f32 && !g.aux_f32) {                    // bf16 outputs: the same polynomial pair as the fast routines (one result per dtype, whatever the tile)
            f32x4 y4, d4;
            gelu_both_fast4((f32x4){v[0], v[1], v[2], v[3]}, y4, d4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s4[j] = g.aux_grad ? d4[j] : v[j]; v[j] = y4[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float y, dy; gelu_both(v[j], y, dy); s4[j] = g.aux_grad ? dy : v[j]; v[j] = y; }
        }
        if (g.aux_f32) {
            float* a = (float*)g.aux + ao;
            if (full) { f32x4 t = {s4[0], s4[1], s4[2], s4[3]}; st4(a, t); } else for (int j = 0; j < cnt; ++j) a[j] = s4[j];
        } else {
            uint16_t* a = (uint16_t*)g.aux + ao;
            if (full) { f32x4 t = {s4[0], s4[1], s4[2], s4[3]}; st4(a, t); } else for (int j = 0; j < cnt; ++j) a[j] = f32_to_bf16_bits(s4[j]);
        }
    } else if (g.epi == MMAE_EPI_DGELU) {
        const long long ao = (long long)m * g.ldaux + n;
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.aux_f32) {
            const float* a = (const float*)g.aux + ao;
            if (full) { f32x4 t = ld4(a); p[0] = t[0]; p[1] = t[1]; p[2] = t[2]; p[3] = t[3]; } else for (int j = 0; j < cnt; ++j) p[j] = a[j];
        } else {
            const uint16_t* a = (const uint16_t*)g.aux + ao;
            if (full) { f32x4 t = ld4(a); p[0] = t[0]; p[1] = t[1]; p[2] = t[2]; p[3] = t[3]; } else for (int j = 0; j < cnt; ++j) p[j] = bf16_bits_to_f32(a[j]);
        }
        if (g.aux_grad) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= p[j];
        } else if (!g.c_f32 && !g.aux_f32) {
            const f32x4 g4 = gelu_grad_fast4((f32x4){p[0], p[1], p[2], p[3]});
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= g4[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_grad(p[j]);
        }
    }
    if (g.resid) {
        const float* r = g.resid + (long long)m * g.ldr + n;
        if (full) { f32x4 t = ld4(r);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += t[j];
        } else { for (int j = 0; j < cnt; ++j) v[j] += r[j]; }
    }
    const long long co = (long long)m * g.ldc + n;
    if (g.c_f32) {
        float* c = (float*)Cz + co;
        if (g.accumulate) {
            if (full) { f32x4 t = ld4(c);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += t[j];
            } else { for (int j = 0; j < cnt; ++j) v[j] += c[j]; }
        }
        if (full) { f32x4 t = {v[0], v[1], v[2], v[3]}; st4(c, t); } else for (int j = 0; j < cnt; ++j) c[j] = v[j];
    } else {
        uint16_t* c = (uint16_t*)Cz + co;
        if (full) { f32x4 t = {v[0], v[1], v[2], v[3]}; st4(c, t); } else for (int j = 0; j < cnt; ++j) c[j] = f32_to_bf16_bits(v[j]);
    }
}

// ------------------------------------------------------------------------------------------------
// Epilogue of one wave's 64x64 output tile (acc[tn][tm], D[i = n][j = m] MFMA layout), staged through a
// wave-private 8-KiB LDS region so that every global access (C, residual, aux) is a full 128/256-byte
// row segment: 16 lanes x 4 consecutive columns per row, 4 rows per instruction.  The epilogue flavour
// is a compile-time parameter of the inner loop (chosen once per wave): with run-time flag tests per
// 4-element group the epilogue was instruction-bound -- ~13 us per output tile regardless of K
// (rocprof K-sweep, profiles/r01_gemm_ksweep.txt).
// Caller must have passed a workgroup barrier after the last operand read of the LDS ring.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_acc_tile(char* wave_lds, int lane, const f32x16 (&acc)[2][2], int tm) {
    const int hi = lane >> 5, ml = lane & 31;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int c16 = tn * 8 + rg * 2 + hi;
            f32x4 v = {acc[tn][tm][rg * 4 + 0], acc[tn][tm][rg * 4 + 1], acc[tn][tm][rg * 4 + 2], acc[tn][tm][rg * 4 + 3]};
            *reinterpret_cast<f32x4*>(wave_lds + ml * 256 + ((c16 ^ (ml & 15)) << 4)) = v;
        }
}

__device__ __forceinline__ void st4_bf16_hw(uint16_t* p, f32x4 v) {   // v_cvt_pk_bf16_f32 x2 + one 8-byte store
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
    bf16x4_t b;
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x4_t*>(p) = b;
}

// fast paths: alpha == 1, N % 4 == 0, every pointer/ld 4-element aligned (g.vec), aux is bf16 (or f32 with AUX_F32).
// All global traffic goes through raw buffer instructions on descriptors based at the tile's first row: masked lanes use
// the out-of-range offset (loads return 0, stores are dropped), so the loop has no branches, and the epilogue's INPUT
// streams (dGELU pre-activations, residual) are prefetched one 32-row round ahead -- with the loads inside a bounds
// branch every 4-row step of the dGELU / residual epilogues exposed a full HBM round trip (~30 us per 256 x 256 tile).
__device__ __forceinline__ i32x2 pack4_bf16(f32x4 v) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
    bf16x4_t b;
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = (__bf16)v[j];
    return __builtin_bit_cast(i32x2, b);
}
__device__ __forceinline__ f32x4 unpack4_bf16(i32x2 r) {
    f32x4 o;
    o[0] = __uint_as_float(((uint32_t)r[0]) << 16); o[1] = __uint_as_float(((uint32_t)r[0]) & 0xffff0000u);
    o[2] = __uint_as_float(((uint32_t)r[1]) << 16); o[3] = __uint_as_float(((uint32_t)r[1]) & 0xffff0000u);
    return o;
}
// the same two for the 16-bit format of the instantiation: bf16 (H16 = false) or fp16
template <bool H16> __device__ __forceinline__ i32x2 pack4_16(f32x4 v) {
    if constexpr (!H16) return pack4_bf16(v);
    i32x2 r;
    r[0] = (int)((uint32_t)f32_to_f16_bits(v[0]) | ((uint32_t)f32_to_f16_bits(v[1]) << 16));
    r[1] = (int)((uint32_t)f32_to_f16_bits(v[2]) | ((uint32_t)f32_to_f16_bits(v[3]) << 16));
    return r;
}
template <bool H16> __device__ __forceinline__ f32x4 unpack4_16(i32x2 r) {
    if constexpr (!H16) return unpack4_bf16(r);
    f32x4 o;
    o[0] = f16_bits_to_f32((uint16_t)((uint32_t)r[0] & 0xffffu)); o[1] = f16_bits_to_f32((uint16_t)((uint32_t)r[0] >> 16));
    o[2] = f16_bits_to_f32((uint16_t)((uint32_t)r[1] & 0xffffu)); o[3] = f16_bits_to_f32((uint16_t)((uint32_t)r[1] >> 16));
    return o;
}
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(T* base, long long row, long long ld) {
    const unsigned long long v = (unsigned long long)(base + row * ld);       // wave-uniform: keep it in SGPRs
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x80000000, 0x00020000);
}

// KEEP: also hand the final values back (keep[tm * 8 + it] = the four columns of row tm * 32 + it * 4 + (lane >> 4)): the LayerNorm side
// output of the ping-pong kernel normalises them once the row statistics of all four column quarters are known (gemm_pp_body.h)
template <bool BIAS, int EPI, bool RESID, bool C_F32, bool ACC, bool AUX_F32 = false, bool COLSUM = false, bool KEEP = false>
__device__ __forceinline__ void store_tile64_fast(const GemmArgs& g, char* Cbase, long long ldc, char* wave_lds, int lane,
                                                  const f32x16 (&acc)[2][2], int m_base, int n_base, int ntm = 2, f32x4* keep = nullptr) {
    constexpr unsigned OOB_OFF = 0x80000000u;
    const int c16 = lane & 15, rsub = lane >> 4;
    const int n = n_base + c16 * 4;
    const bool n_ok = n < g.N;
    const int rows_left = g.M - m_base;                  // rows r (tile-relative) < rows_left exist
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    if (BIAS && n_ok) b4 = ld4(g.bias + n);
    constexpr int C_ESZ = C_F32 ? 4 : 2, AUX_ESZ = AUX_F32 ? 4 : 2;
    const auto rsC = C_F32 ? row_rsrc((float*)Cbase, m_base, ldc) : row_rsrc((uint16_t*)Cbase, m_base, ldc);
    const auto rsAux = (EPI != MMAE_EPI_NONE) ? (AUX_F32 ? row_rsrc((float*)g.aux, m_base, g.ldaux) : row_rsrc((uint16_t*)g.aux, m_base, g.ldaux)) : rsC;
    const auto rsRes = RESID ? row_rsrc((float*)g.resid, m_base, g.ldr) : rsC;
    auto voff = [&](int tm, int it, long long ld, int esz) -> int {
        const int r = tm * 32 + it * 4 + rsub;
        return (n_ok && r < rows_left) ? (int)((r * ld + n) * esz) : (int)OOB_OFF;
    };
    // input prefetch ring: slot gi % PD holds the inputs of global 4-row step gi = tm * 8 + it; refilled for step gi + PD
    // right after use (f32 streams: 4 steps = 16 VGPRs; the bf16 aux stream affords 8)
    constexpr int PD = (RESID || AUX_F32) ? 4 : 8;
    i32x4 pre_res[PD];
    i32x4 pre_aux[PD];                                   // bf16 aux uses the low two dwords
    auto prefetch = [&](int gi) {
        const int tm = gi >> 3, it = gi & 7, sl = gi % PD;
        if (tm >= ntm) return;
        if (RESID) pre_res[sl] = __builtin_amdgcn_raw_buffer_load_b128(rsRes, voff(tm, it, g.ldr, 4), 0, 0);
        if (EPI == MMAE_EPI_DGELU) {
            if (AUX_F32) pre_aux[sl] = __builtin_amdgcn_raw_buffer_load_b128(rsAux, voff(tm, it, g.ldaux, 4), 0, 0);
            else {
                const i32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsAux, voff(tm, it, g.ldaux, 2), 0, 0);
                pre_aux[sl][0] = t[0]; pre_aux[sl][1] = t[1];
            }
        }
    };
    if (RESID || EPI == MMAE_EPI_DGELU) {
#pragma unroll
        for (int gi = 0; gi < PD; ++gi) prefetch(gi);
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        if (tm < ntm) {                  // ntm = 1: only the first 32-row half of the tile exists (odd MFMA-tile counts)
            stage_acc_tile(wave_lds, lane, acc, tm);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 4 + rsub;
                f32x4 v = *reinterpret_cast<const f32x4*>(wave_lds + r * 256 + ((c16 ^ (r & 15)) << 4));
                const bool ok = n_ok && (tm * 32 + r) < rows_left;
                if (BIAS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += b4[j];
                }
                if (EPI == MMAE_EPI_GELU) {
                    f32x4 sv;
                    if (!C_F32 && !AUX_F32) {              // bf16 outputs: the packed polynomial pair (common.h), as the 8-column routine
                        f32x4 y4, d4;
                        gelu_both_fast4(v, y4, d4);
                        sv = g.aux_grad ? d4 : v; v = y4;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { float y, dy; gelu_both(v[j], y, dy); sv[j] = g.aux_grad ? dy : v[j]; v[j] = y; }
                    }
                    if (AUX_F32) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, sv), rsAux, voff(tm, it, g.ldaux, 4), 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(pack4_bf16(sv), rsAux, voff(tm, it, g.ldaux, 2), 0, 0);
                } else if (EPI == MMAE_EPI_DGELU) {
                    i32x2 lo2; lo2[0] = pre_aux[(tm * 8 + it) % PD][0]; lo2[1] = pre_aux[(tm * 8 + it) % PD][1];
                    const f32x4 p = AUX_F32 ? __builtin_bit_cast(f32x4, pre_aux[(tm * 8 + it) % PD]) : unpack4_bf16(lo2);
                    if (g.aux_grad) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= p[j];
                    } else if (!C_F32 && !AUX_F32) {
                        const f32x4 g4 = gelu_grad_fast4(p);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= g4[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_grad(p[j]);
                    }
                }
                if (RESID) {
                    const f32x4 t = __builtin_bit_cast(f32x4, pre_res[(tm * 8 + it) % PD]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += t[j];
                }
                if ((RESID || EPI == MMAE_EPI_DGELU) && tm * 8 + it + PD < 16) prefetch(tm * 8 + it + PD);   // refill this slot
                if (KEEP) keep[tm * 8 + it] = v;
                if (COLSUM) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) cs[j] += ok ? v[j] : 0.f;
                }
                if (C_F32) {
                    const int o = voff(tm, it, ldc, 4);
                    if (ACC) {
                        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsC, o, 0, 0));
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += t[j];
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rsC, o, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b64(pack4_bf16(v), rsC, voff(tm, it, ldc, C_ESZ), 0, 0);
                }
            }
            if (COLSUM) {       // the 4 lanes that share a column group (rsub = 0..3) -> one partial per column and 32-row block
#pragma unroll
                for (int j = 0; j < 4; ++j) { cs[j] += __shfl_xor(cs[j], 16, 64); cs[j] += __shfl_xor(cs[j], 32, 64); }
                if (rsub == 0 && n_ok && m_base + tm * 32 < g.M) st4(g.colpart + (long long)((m_base + tm * 32) >> 5) * g.N + n, cs);
                cs = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// bf16 C (and bf16 aux) with 16-byte global accesses: a lane owns EIGHT consecutive columns of a row (8 lanes x 8 columns
// per 64-column row, 8 rows per step, 4 steps per 32-row MFMA tile), so every store / aux access is a dwordx4 instead of the
// dwordx2 of the 4-column layout above -- half the vector-memory instructions.  The epilogue's tail is store-ISSUE-bound
// (MI355X_MICROARCH.md: 16 x dwordx2 per lane ~ 2x the cycles of 8 x dwordx4), and the GELU / dGELU products issue two bf16
// streams per element.  Needs N, ldc (and ldaux) multiples of 8 and 16-byte aligned bases; flavours: bf16 C, no residual,
// no accumulate, bf16 aux.
__device__ __forceinline__ i32x4 pack8_bf16(const f32x4 a, const f32x4 b) {
    const i32x2 lo = pack4_bf16(a), hi = pack4_bf16(b);
    i32x4 r; r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
    return r;
}
// MXQ: additionally write the MX-fp8 quantisation of the (bf16-rounded) output -- a 32-column block is four adjacent lanes
template <bool H16> __device__ __forceinline__ i32x4 pack8_16(const f32x4 a, const f32x4 b) {
    const i32x2 lo = pack4_16<H16>(a), hi = pack4_16<H16>(b);
    i32x4 r; r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
    return r;
}
// AUXG: what aux holds (GemmArgs::aux_grad) fixed at compile time (1 / 0; -1 = read the argument).  As a run-time test of a uniform
// argument hipcc if-converts it: BOTH forms computed per element and selected -- the dX product then evaluated the whole GELU' polynomial
// (9 packed FMAs + clamps per pair) next to the multiply that replaces it, and fc1's epilogue paid a v_cndmask per element (round 5,
// read off the ISA: 32 -> 15 and 34 -> 28 VALU operations per element pair).  The flavoured callers branch once per tile instead.
template <bool BIAS, int EPI, bool COLSUM, int DBG = 0, int PDEPTH = 4, bool MXQ = false, bool H16 = false, int AUXG = -1>
__device__ __forceinline__ void store_tile64_bf16x8(const GemmArgs& g, char* Cbase, char* wave_lds, int lane, const f32x16 (&acc)[2][2],
                                                    int m_base, int n_base, int ntm) {
    constexpr unsigned OOB_OFF = 0x80000000u;
    const int c8 = lane & 7, r8 = lane >> 3;
    const int n = n_base + c8 * 8;
    const bool n_ok = n < g.N;
    const int rows_left = g.M - m_base;
    const bool aux_grad = AUXG < 0 ? (g.aux_grad != 0) : (AUXG != 0);
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (BIAS && n_ok) { b0 = ld4(g.bias + n); b1 = ld4(g.bias + n + 4); }
    f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = cs0;
    const auto rsC = row_rsrc((uint16_t*)Cbase, m_base, g.ldc);
    const auto rsAux = (EPI != MMAE_EPI_NONE) ? row_rsrc((uint16_t*)g.aux, m_base, g.ldaux) : rsC;
    auto voff = [&](int gi, long long ld) -> int {       // gi = global 8-row step: tm * 4 + it
        const int r = gi * 8 + r8;
        return (n_ok && r < rows_left) ? (int)((r * ld + n) * 2) : (int)OOB_OFF;
    };
    const int nsteps = ntm * 4;
    constexpr int PD = PDEPTH;                           // dGELU: pre-activation rows prefetched PD steps (8 rows each) ahead; 8 = the whole 64-row tile up front
    i32x4 pre_aux[PD];
    if (EPI == MMAE_EPI_DGELU) {
#pragma unroll
        for (int gi = 0; gi < PD; ++gi) if (gi < nsteps) pre_aux[gi] = __builtin_amdgcn_raw_buffer_load_b128(rsAux, voff(gi, g.ldaux), 0, 0);
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        if (tm < ntm) {
            stage_acc_tile(wave_lds, lane, acc, tm);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int gi = tm * 4 + it, r = it * 8 + r8;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(wave_lds + r * 256 + (((2 * c8) ^ (r & 15)) << 4));
                f32x4 v1 = *reinterpret_cast<const f32x4*>(wave_lds + r * 256 + (((2 * c8 + 1) ^ (r & 15)) << 4));
                const bool ok = n_ok && (tm * 32 + r) < rows_left;
                if (BIAS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v0[j] += b0[j]; v1[j] += b1[j]; }
                }
                if (EPI == MMAE_EPI_GELU) {
                    f32x4 s0 = v0, s1 = v1;                  // what aux keeps: the pre-activation, or (aux_grad) GELU' of it
                    if (DBG != 1) {                          // bf16 outputs: the packed polynomial pair (common.h); fp16-storage adapters (the
                        f32x4 y0, d0, y1, d1;                // reference's fp32 adapters): the exact erf form, as every f32 output
                        if constexpr (H16) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float ya, da, yb, db;
                                gelu_both(v0[j], ya, da); gelu_both(v1[j], yb, db);
                                y0[j] = ya; d0[j] = da; y1[j] = yb; d1[j] = db;
                            }
                        } else {
                            gelu_both_fast4(v0, y0, d0);
                            gelu_both_fast4(v1, y1, d1);
                        }
                        if (aux_grad) { s0 = d0; s1 = d1; }
                        v0 = y0; v1 = y1;
                    }
                    if (DBG != 2 && DBG != 3) __builtin_amdgcn_raw_buffer_store_b128(pack8_16<H16>(s0, s1), rsAux, voff(gi, g.ldaux), 0, 0);
                } else if (EPI == MMAE_EPI_DGELU) {
                    const i32x4 pa = pre_aux[gi % PD];
                    i32x2 lo2, hi2; lo2[0] = pa[0]; lo2[1] = pa[1]; hi2[0] = pa[2]; hi2[1] = pa[3];
                    const f32x4 p0 = unpack4_16<H16>(lo2), p1 = unpack4_16<H16>(hi2);
                    if (aux_grad) {                          // the forward stored GELU' itself: no transcendental work here
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v0[j] *= p0[j]; v1[j] *= p1[j]; }
                    } else if constexpr (H16) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v0[j] *= gelu_erf_grad(p0[j]); v1[j] *= gelu_erf_grad(p1[j]); }
                    } else {
                        const f32x4 g0 = gelu_grad_fast4(p0), g1 = gelu_grad_fast4(p1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v0[j] *= g0[j]; v1[j] *= g1[j]; }
                    }
                    if (gi + PD < nsteps) pre_aux[gi % PD] = __builtin_amdgcn_raw_buffer_load_b128(rsAux, voff(gi + PD, g.ldaux), 0, 0);
                }
                if (COLSUM) {                                // (rows / columns outside the matrix hold exact zeros: their operands were loaded as zeros)
                    const float okf = ok ? 1.f : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { cs0[j] = __builtin_fmaf(okf, v0[j], cs0[j]); cs1[j] = __builtin_fmaf(okf, v1[j], cs1[j]); }
                }
                if (MXQ) {
                    const i32x4 pk = pack8_16<H16>(v0, v1);
                    __builtin_amdgcn_raw_buffer_store_b128(pk, rsC, voff(gi, g.ldc), 0, 0);
                    float w[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { w[2 * j] = __uint_as_float(((unsigned)pk[j]) << 16); w[2 * j + 1] = __uint_as_float(((unsigned)pk[j]) & 0xffff0000u); }
                    float am = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(w[j]));
                    am = fmaxf(am, __shfl_xor(am, 1, 64));
                    am = fmaxf(am, __shfl_xor(am, 2, 64));
                    const int e = mx_shared_exp(am);
                    const float inv = mx_inv_scale(e);
                    i32x2 o;
                    o[0] = mx_cvt4_e4m3(w[0] * inv, w[1] * inv, w[2] * inv, w[3] * inv);
                    o[1] = mx_cvt4_e4m3(w[4] * inv, w[5] * inv, w[6] * inv, w[7] * inv);
                    if (ok) {
                        const long long grow = m_base + tm * 32 + r;
                        *reinterpret_cast<i32x2*>(g.qout + grow * g.ldq + n) = o;
                        if ((c8 & 3) == 0) g.qsc[mx_scale_addr(g.M, grow, n >> 5)] = (unsigned char)e;
                    }
                } else
                if (DBG != 3) __builtin_amdgcn_raw_buffer_store_b128(pack8_16<H16>(v0, v1), rsC, voff(gi, g.ldc), 0, 0);
                else if (v0[0] == 1234.5678f) __builtin_amdgcn_raw_buffer_store_b128(pack8_16<H16>(v0, v1), rsC, voff(gi, g.ldc), 0, 0);   // keeps the arithmetic alive
            }
            if (COLSUM) {       // the 8 lanes that share a column group (r8 = 0..7) -> one partial per column and 32-row block
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs0[j] += __shfl_xor(cs0[j], 8, 64); cs0[j] += __shfl_xor(cs0[j], 16, 64); cs0[j] += __shfl_xor(cs0[j], 32, 64);
                    cs1[j] += __shfl_xor(cs1[j], 8, 64); cs1[j] += __shfl_xor(cs1[j], 16, 64); cs1[j] += __shfl_xor(cs1[j], 32, 64);
                }
                if (r8 == 0 && n_ok && m_base + tm * 32 < g.M) {
                    float* cp = g.colpart + (long long)((m_base + tm * 32) >> 5) * g.N + n;
                    st4(cp, cs0); st4(cp + 4, cs1);
                }
                cs0 = f32x4{0.f, 0.f, 0.f, 0.f}; cs1 = cs0;
            }
        }
    }
}

__device__ __forceinline__ void gemm_store_tile64(const GemmArgs& g, char* Cz, char* wave_lds, int lane, f32x16 (&acc)[2][2],
                                                  int m_base, int n_base, int ntm = 2) {
    const bool aligned = g.vec && ((g.N & 3) == 0) && g.alpha == 1.0f;
    if (g.splitk > 1) {                      // dense f32 partial slab
        if ((g.N & 3) == 0) {
            store_tile64_fast<false, 0, false, true, false>(g, (char*)(g.ws + (long long)blockIdx.z * g.M * g.N), g.N, wave_lds, lane, acc, m_base, n_base, ntm);
            return;
        }
    } else if (aligned) {
        const bool bias = g.bias != nullptr, resid = g.resid != nullptr;
        const bool aux_ok = g.epi == MMAE_EPI_NONE || !g.aux_f32;
        if (!aux_ok && g.c_f32 && !resid && !g.accumulate) {      // exact-f32 mode: f32 C and f32 aux
            if (g.epi == MMAE_EPI_GELU && bias) { store_tile64_fast<true, MMAE_EPI_GELU, false, true, false, true>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm); return; }
            if (g.epi == MMAE_EPI_DGELU && !bias) {
                if (g.colpart) store_tile64_fast<false, MMAE_EPI_DGELU, false, true, false, true, true>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm);
                else store_tile64_fast<false, MMAE_EPI_DGELU, false, true, false, true>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm);
                return;
            }
        }
        if (aux_ok) {
            if (!g.c_f32 && !resid && !g.accumulate && g.wide_st && (g.N & 7) == 0 && (g.ldc & 7) == 0 && (((uintptr_t)Cz) & 15) == 0 &&
                (g.epi == MMAE_EPI_NONE || ((g.ldaux & 7) == 0 && (((uintptr_t)g.aux) & 15) == 0)) && (!bias || (((uintptr_t)g.bias) & 15) == 0)) {
                if (g.epi == MMAE_EPI_NONE) {
                    if (bias) store_tile64_bf16x8<true, MMAE_EPI_NONE, false>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm);
                    else store_tile64_bf16x8<false, MMAE_EPI_NONE, false>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm);
                    return;
                }
                if (g.epi == MMAE_EPI_GELU && bias) {
                    if (g.dbg == 0) { store_tile64_bf16x8<true, MMAE_EPI_GELU, false>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm); return; }
                    if (g.dbg == 1) { store_tile64_bf16x8<true, MMAE_EPI_GELU, false, 1>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm); return; }
                    if (g.dbg == 2) { store_tile64_bf16x8<true, MMAE_EPI_GELU, false, 2>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm); return; }
                    if (g.dbg == 3) { store_tile64_bf16x8<true, MMAE_EPI_GELU, false, 3>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm); return; }
                    return;
                }
                if (g.epi == MMAE_EPI_DGELU && !bias) {
                    if (g.colpart) store_tile64_bf16x8<false, MMAE_EPI_DGELU, true>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm);
                    else store_tile64_bf16x8<false, MMAE_EPI_DGELU, false>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm);
                    return;
                }
            }
            if (!g.c_f32 && !resid && !g.accumulate) {
                if (g.epi == MMAE_EPI_NONE) {
                    if (bias) { store_tile64_fast<true, 0, false, false, false>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm); return; }
                    store_tile64_fast<false, 0, false, false, false>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm); return;
                }
                if (g.epi == MMAE_EPI_GELU && bias) { store_tile64_fast<true, MMAE_EPI_GELU, false, false, false>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm); return; }
                if (g.epi == MMAE_EPI_DGELU && !bias) {
                    if (g.colpart) store_tile64_fast<false, MMAE_EPI_DGELU, false, false, false, false, true>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm);
                    else store_tile64_fast<false, MMAE_EPI_DGELU, false, false, false>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm);
                    return;
                }
            } else if (g.c_f32 && g.epi == MMAE_EPI_NONE) {
                if (bias && resid && !g.accumulate) { store_tile64_fast<true, 0, true, true, false>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm); return; }
                if (bias && !resid && !g.accumulate) { store_tile64_fast<true, 0, false, true, false>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm); return; }
                if (!bias && !resid && !g.accumulate) { store_tile64_fast<false, 0, false, true, false>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm); return; }
                if (!bias && !resid && g.accumulate) { store_tile64_fast<false, 0, false, true, true>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm); return; }
            }
        }
    }
    // generic path: any flag combination, ragged N, unaligned pointers
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        if (tm >= ntm) break;
        stage_acc_tile(wave_lds, lane, acc, tm);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 4 + (lane >> 4), c16 = lane & 15;
            const f32x4 v = *reinterpret_cast<const f32x4*>(wave_lds + r * 256 + ((c16 ^ (r & 15)) << 4));
            gemm_epilogue4(g, Cz, m_base + tm * 32 + r, n_base + c16 * 4, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Epilogue flavour fixed at COMPILE time (ping-pong kernel): the instantiation then contains one store routine instead of all
// of them -- the generic 320-row kernels hold 256 VGPRs and spill ~38 more, reloading them at the top of every output tile
// behind a full s_waitcnt vmcnt(0) (which also drains the next tile's prefetched K tiles); with one flavour they fit.
// gemm_flavour() applies exactly the conditions gemm_store_tile64 tests at run time; 0 = keep the generic kernel.
// ------------------------------------------------------------------------------------------------
enum { FL_GENERIC = 0, FL_BF16_BIAS = 1, FL_F32_BIAS_RESID = 2, FL_BF16_BIAS_GELU = 3, FL_BF16 = 4, FL_BF16_DGELU_CS = 5, FL_F32_BIAS = 6, FL_F32 = 7,
       FL_BF16_DGELU = 8,
       FL_BF16_BIAS_GELU_Q = 9, FL_BF16_DGELU_CS_Q = 10, FL_BF16_DGELU_Q = 11 };      // + MX-fp8 copy of the output (mxfp8.hip only)

static inline int gemm_flavour(const GemmArgs& g, int batch) {
    if (batch != 1 || g.splitk > 1 || !g.vec || (g.N & 3) || g.alpha != 1.0f || g.accumulate || g.dbg) return FL_GENERIC;
    const bool bias = g.bias != nullptr, resid = g.resid != nullptr;
    if (g.c_f32) {
        if (g.epi != MMAE_EPI_NONE) return FL_GENERIC;
        if (bias && resid) return FL_F32_BIAS_RESID;
        if (bias && !resid) return FL_F32_BIAS;
        if (!bias && !resid) return FL_F32;
        return FL_GENERIC;
    }
    if (resid) return FL_GENERIC;
    // the 8-column (dwordx4) bf16 routines
    const bool wide = g.wide_st && (g.N & 7) == 0 && (g.ldc & 7) == 0 && ((uintptr_t)g.C & 15) == 0 &&
                      (g.epi == MMAE_EPI_NONE || (!g.aux_f32 && (g.ldaux & 7) == 0 && ((uintptr_t)g.aux & 15) == 0)) &&
                      (!bias || ((uintptr_t)g.bias & 15) == 0);
    if (!wide) return FL_GENERIC;
    if (g.epi == MMAE_EPI_NONE) return bias ? FL_BF16_BIAS : FL_BF16;
    if (g.epi == MMAE_EPI_GELU && bias) return FL_BF16_BIAS_GELU;
    if (g.epi == MMAE_EPI_DGELU && !bias) return g.colpart ? FL_BF16_DGELU_CS : FL_BF16_DGELU;
    return FL_GENERIC;
}

template <int FL, bool H16 = false>
__device__ __forceinline__ void gemm_store_tile64_fl(const GemmArgs& g, char* Cz, char* wave_lds, int lane, f32x16 (&acc)[2][2], int m_base, int n_base,
                                                     int ntm = 2) {
    if (FL == FL_BF16_BIAS) store_tile64_bf16x8<true, MMAE_EPI_NONE, false, 0, 4, false, H16>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm);
    else if (FL == FL_BF16) store_tile64_bf16x8<false, MMAE_EPI_NONE, false, 0, 4, false, H16>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm);
    // the GELU flavours: one branch per tile on what aux holds (AUXG above), not a select per element
#define MMAE_AUXG(...) do { if (g.aux_grad) store_tile64_bf16x8<__VA_ARGS__, 1>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm); \
                            else store_tile64_bf16x8<__VA_ARGS__, 0>(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm); } while (0)
    else if (FL == FL_BF16_BIAS_GELU) MMAE_AUXG(true, MMAE_EPI_GELU, false, 0, 4, false, H16);
    // (with one flavour per instantiation there are registers to spare: all 8 pre-activation loads of a 64-row tile go out before its first use)
    else if (FL == FL_BF16_DGELU_CS) MMAE_AUXG(false, MMAE_EPI_DGELU, true, 0, 4, false, H16);
    else if (FL == FL_BF16_DGELU) MMAE_AUXG(false, MMAE_EPI_DGELU, false, 0, 8, false, H16);
    else if (FL == FL_BF16_BIAS_GELU_Q) MMAE_AUXG(true, MMAE_EPI_GELU, false, 0, 4, true, false);
    else if (FL == FL_BF16_DGELU_CS_Q) MMAE_AUXG(false, MMAE_EPI_DGELU, true, 0, 4, true, false);
    else if (FL == FL_BF16_DGELU_Q) MMAE_AUXG(false, MMAE_EPI_DGELU, false, 0, 8, true, false);
#undef MMAE_AUXG
    else if (FL == FL_F32_BIAS_RESID) store_tile64_fast<true, 0, true, true, false>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm);
    else if (FL == FL_F32_BIAS) store_tile64_fast<true, 0, false, true, false>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm);
    else if (FL == FL_F32) store_tile64_fast<false, 0, false, true, false>(g, Cz, g.ldc, wave_lds, lane, acc, m_base, n_base, ntm);
    else gemm_store_tile64(g, Cz, wave_lds, lane, acc, m_base, n_base, ntm);
}

