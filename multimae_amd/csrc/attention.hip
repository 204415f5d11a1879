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
// Replaces Attention.forward / CrossAttention.forward cores (multimae_utils.py:175-179, 206-210) + autograd.
#include "common.h"

#define LDS_AS __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(8))) short s16x8;

namespace {

constexpr unsigned OOB = 0x80000000u;

struct AttnArgs {
    const uint16_t *q, *k, *v, *o, *d_o;
    uint16_t *out, *dq, *dk, *dv;
    float* lse;
    int B, H, Nq, Nk, nqp, nkp;
    long long q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, dq_sb, dq_sr, dk_sb, dk_sr, dv_sb, dv_sr;
    float scale;
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
__device__ __forceinline__ bf16x8 load_frag_global(const __amdgpu_buffer_rsrc_t rs, bool ok, long long row, long long sr, int ks, int hi) {
    const unsigned off = ok ? (unsigned)((row * sr + ks * 16 + hi * 8) * 2) : OOB;
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
// store 4 consecutive head-dim values of one row
__device__ __forceinline__ void store4(uint16_t* p, float a, float b, float c, float d) {
    f32x4 t = {a, b, c, d};
    st4(p, t);
}

// -------------------------------------------------------------------------------------------------
template <int HD, int NT>
__global__ void __launch_bounds__(256, NT == 4 ? 3 : 1) attn_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    char* Ks = smem;
    char* Vs = smem + a.nkp * HD * 2;
    {
        TileLoader<HD, 256> lk, lv;
        lk.issue(a.k + b * a.k_sb + h * HD, a.k_sr, a.Nk, a.nkp, tid);
        lv.issue(a.v + b * a.v_sb + h * HD, a.v_sr, a.Nk, a.nkp, tid);
        lk.commit(Ks, a.nkp, tid);
        lv.commit(Vs, a.nkp, tid);
    }
    __syncthreads();
    const int nt = a.nkp >> 5, nqb = (a.Nq + 31) >> 5;
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(a.q + b * a.q_sb + h * HD), 0, 0x80000000, 0x00020000);
    uint16_t* ob = a.out + b * a.o_sb + h * HD;
    for (int qblk = wave; qblk < nqb; qblk += 4) {
        const int q = qblk * 32 + (lane & 31);
        const bool qok = q < a.Nq;
        bf16x8 qf[HD / 16];
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) qf[ks] = load_frag_global(rsQ, qok, q, a.q_sr, ks, hi);
        f32x16 s[NT];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < HD / 16; ++ks)
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Ks, t * 32, ks, lane), qf[ks], s[t], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float v = key < a.Nk ? s[t][r] * a.scale : -INFINITY;
                    s[t][r] = v;
                    m = fmaxf(m, v);
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float p = __expf(s[t][r] - m); s[t][r] = p; l += p; }
            }
        }
        l += __shfl_xor(l, 32, 64);
        f32x16 o[HD / 32];
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < nt) {
#pragma unroll
                for (int sI = 0; sI < 2; ++sI) {
                    const bf16x8 pf = pack8(s[t], sI);
#pragma unroll
                    for (int dt = 0; dt < HD / 32; ++dt)
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<HD>(Vs, dt * 32, t * 32 + 16 * sI, lane), pf, o[dt], 0, 0, 0);
                }
            }
        }
        if (qok) {
            const float inv = 1.0f / l;
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    store4(ob + q * a.o_sr + dt * 32 + rg * 8 + 4 * hi, o[dt][rg * 4] * inv, o[dt][rg * 4 + 1] * inv,
                           o[dt][rg * 4 + 2] * inv, o[dt][rg * 4 + 3] * inv);
            if (hi == 0) a.lse[((long long)b * a.H + h) * a.Nq + q] = m + __logf(l);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// 512 threads: after the shared prologue (tiles -> LDS, delta), waves 0-3 run pass 1 and waves 4-7 run
// pass 2 concurrently (the passes only read LDS and write disjoint outputs).
template <int HD>
__global__ void __launch_bounds__(512) attn_bwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    char* Qs = smem;
    char* dOs = Qs + a.nqp * HD * 2;
    char* Ks = dOs + a.nqp * HD * 2;
    char* Vs = Ks + a.nkp * HD * 2;
    float* lse_s = (float*)(Vs + a.nkp * HD * 2);
    float* delta_s = lse_s + a.nqp;
    const uint16_t* qg = a.q + b * a.q_sb + h * HD;
    const uint16_t* og = a.o + b * a.o_sb + h * HD;
    const uint16_t* dog = a.d_o + b * a.o_sb + h * HD;
    {
        TileLoader<HD, 512> lq, ld, lk, lv;
        lq.issue(qg, a.q_sr, a.Nq, a.nqp, tid);
        ld.issue(dog, a.o_sr, a.Nq, a.nqp, tid);
        lk.issue(a.k + b * a.k_sb + h * HD, a.k_sr, a.Nk, a.nkp, tid);
        lv.issue(a.v + b * a.v_sb + h * HD, a.v_sr, a.Nk, a.nkp, tid);
        for (int q = tid; q < a.nqp; q += 512) lse_s[q] = q < a.Nq ? a.lse[((long long)b * a.H + h) * a.Nq + q] : 0.f;
        lq.commit(Qs, a.nqp, tid);
        ld.commit(dOs, a.nqp, tid);
        lk.commit(Ks, a.nkp, tid);
        lv.commit(Vs, a.nkp, tid);
    }
    const int nt = a.nkp >> 5, nqb = a.nqp >> 5;
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)qg, 0, 0x80000000, 0x00020000);
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc((void*)og, 0, 0x80000000, 0x00020000);
    const auto rsdO = __builtin_amdgcn_make_buffer_rsrc((void*)dog, 0, 0x80000000, 0x00020000);
    // delta[q] = sum_d dO[q][d] * O[q][d]
    for (int qblk = wave; qblk < nqb; qblk += 8) {
        const int q = qblk * 32 + (lane & 31);
        const bool qok = q < a.Nq;
        float d = 0.f;
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) {
            const bf16x8 of = load_frag_global(rsO, qok, q, a.o_sr, ks, hi), df = load_frag_global(rsdO, qok, q, a.o_sr, ks, hi);
#pragma unroll
            for (int j = 0; j < 8; ++j) d += (float)of[j] * (float)df[j];
        }
        d += __shfl_xor(d, 32, 64);
        if (hi == 0) delta_s[q] = d;
    }
    __syncthreads();

    // ---- pass 1 (waves 0-3): lane = query row -> dQ
    for (int qblk = wave; wave < 4 && qblk < nqb; qblk += 4) {
        const int q = qblk * 32 + (lane & 31);
        const bool qok = q < a.Nq;
        bf16x8 qf[HD / 16], dof[HD / 16];
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) { qf[ks] = frag_rows<HD>(Qs, qblk * 32, ks, lane); dof[ks] = frag_rows<HD>(dOs, qblk * 32, ks, lane); }
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
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Ks, t * 32, ks, lane), qf[ks], st, 0, 0, 0);
                dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Vs, t * 32, ks, lane), dof[ks], dpt, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(st[r] * a.scale - lse_q);
                st[r] = p * (dpt[r] - delta_q) * a.scale;            // dS^T
            }
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) {
                const bf16x8 dsf = pack8(st, sI);
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<HD>(Ks, dt * 32, t * 32 + 16 * sI, lane), dsf, dq[dt], 0, 0, 0);
            }
        }
        if (qok) {
            uint16_t* dst = a.dq + b * a.dq_sb + h * HD + q * a.dq_sr;
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    store4(dst + dt * 32 + rg * 8 + 4 * hi, dq[dt][rg * 4], dq[dt][rg * 4 + 1], dq[dt][rg * 4 + 2], dq[dt][rg * 4 + 3]);
        }
    }

    // ---- pass 2 (waves 4-7): lane = key row -> dK, dV
    for (int kblk = wave - 4; wave >= 4 && kblk < nt; kblk += 4) {
        const int key = kblk * 32 + (lane & 31);
        bf16x8 kf[HD / 16], vf[HD / 16];
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks) { kf[ks] = frag_rows<HD>(Ks, kblk * 32, ks, lane); vf[ks] = frag_rows<HD>(Vs, kblk * 32, ks, lane); }
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
                sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(Qs, qt * 32, ks, lane), kf[ks], sm, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<HD>(dOs, qt * 32, ks, lane), vf[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float p = __expf(sm[r] * a.scale - lse_s[qr]);
                sm[r] = p;                                            // P
                dp[r] = p * (dp[r] - delta_s[qr]) * a.scale;          // dS
            }
#pragma unroll
            for (int sI = 0; sI < 2; ++sI) {
                const bf16x8 pf = pack8(sm, sI), dsf = pack8(dp, sI);
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<HD>(dOs, dt * 32, qt * 32 + 16 * sI, lane), pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<HD>(Qs, dt * 32, qt * 32 + 16 * sI, lane), dsf, dk[dt], 0, 0, 0);
                }
            }
        }
        if (key < a.Nk) {
            uint16_t* dkd = a.dk + b * a.dk_sb + h * HD + key * a.dk_sr;
            uint16_t* dvd = a.dv + b * a.dv_sb + h * HD + key * a.dv_sr;
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    store4(dkd + dt * 32 + rg * 8 + 4 * hi, dk[dt][rg * 4], dk[dt][rg * 4 + 1], dk[dt][rg * 4 + 2], dk[dt][rg * 4 + 3]);
                    store4(dvd + dt * 32 + rg * 8 + 4 * hi, dv[dt][rg * 4], dv[dt][rg * 4 + 1], dv[dt][rg * 4 + 2], dv[dt][rg * 4 + 3]);
                }
        }
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

int mmae_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                  int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr,
                  float scale, void* stream) {
    MMAE_REQUIRE(q && k && v && o && lse, "attn_fwd: null pointer");
    const long long st[] = {q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr};
    const int rc = check_common(B, H, Nq, Nk, hd, st, 8);
    MMAE_REQUIRE(rc == 0, rc == -2 ? "attn: head_dim must be 32 or 64" : (rc == -3 ? "attn: strides must be multiples of 8" : "attn: need 1 <= Nq, Nk <= 256"));
    MMAE_REQUIRE(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)o % 8 == 0), "attn_fwd: unaligned pointer");
    AttnArgs a = {};
    a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.out = (uint16_t*)o; a.lse = lse;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.nqp = (Nq + 31) / 32 * 32; a.nkp = (Nk + 31) / 32 * 32;
    a.q_sb = q_sb; a.q_sr = q_sr; a.k_sb = k_sb; a.k_sr = k_sr; a.v_sb = v_sb; a.v_sr = v_sr; a.o_sb = o_sb; a.o_sr = o_sr;
    a.scale = scale;
    const size_t lds = (size_t)2 * a.nkp * hd * 2;
    hipStream_t st_ = (hipStream_t)stream;
    dim3 grid(B * H), block(256);
#define LAUNCH_FWD(HD, NT)                                                                                                  \
    do {                                                                                                                     \
        hipFuncSetAttribute((const void*)attn_fwd_kernel<HD, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
        hipLaunchKernelGGL((attn_fwd_kernel<HD, NT>), grid, block, lds, st_, a);                                             \
    } while (0)
    const bool small = a.nkp <= 128;
    if (hd == 64) { if (small) LAUNCH_FWD(64, 4); else LAUNCH_FWD(64, 8); }
    else { if (small) LAUNCH_FWD(32, 4); else LAUNCH_FWD(32, 8); }
#undef LAUNCH_FWD
    return mmae_check_launch("attn_fwd");
}

int mmae_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq, void* dk,
                  void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb,
                  int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr, int64_t dk_sb, int64_t dk_sr, int64_t dv_sb,
                  int64_t dv_sr, float scale, void* stream) {
    MMAE_REQUIRE(q && k && v && o && d_o && lse && dq && dk && dv, "attn_bwd: null pointer");
    const long long st[] = {q_sb, q_sr, k_sb, k_sr, v_sb, v_sr, o_sb, o_sr, dq_sb, dq_sr, dk_sb, dk_sr, dv_sb, dv_sr};
    const int rc = check_common(B, H, Nq, Nk, hd, st, 14);
    MMAE_REQUIRE(rc == 0, rc == -2 ? "attn: head_dim must be 32 or 64" : (rc == -3 ? "attn: strides must be multiples of 8" : "attn: need 1 <= Nq, Nk <= 256"));
    AttnArgs a = {};
    a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.o = (const uint16_t*)o; a.d_o = (const uint16_t*)d_o;
    a.dq = (uint16_t*)dq; a.dk = (uint16_t*)dk; a.dv = (uint16_t*)dv; a.lse = (float*)lse;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.nqp = (Nq + 31) / 32 * 32; a.nkp = (Nk + 31) / 32 * 32;
    a.q_sb = q_sb; a.q_sr = q_sr; a.k_sb = k_sb; a.k_sr = k_sr; a.v_sb = v_sb; a.v_sr = v_sr; a.o_sb = o_sb; a.o_sr = o_sr;
    a.dq_sb = dq_sb; a.dq_sr = dq_sr; a.dk_sb = dk_sb; a.dk_sr = dk_sr; a.dv_sb = dv_sb; a.dv_sr = dv_sr;
    a.scale = scale;
    const size_t lds = (size_t)2 * (a.nqp + a.nkp) * hd * 2 + (size_t)2 * a.nqp * 4;
    hipStream_t st_ = (hipStream_t)stream;
    dim3 grid(B * H), block(512);
    if (hd == 64) {
        hipFuncSetAttribute((const void*)attn_bwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((attn_bwd_kernel<64>), grid, block, lds, st_, a);
    } else {
        hipFuncSetAttribute((const void*)attn_bwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((attn_bwd_kernel<32>), grid, block, lds, st_, a);
    }
    return mmae_check_launch("attn_bwd");
}

}  // extern "C"
