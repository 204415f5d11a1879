// MX-fp8 (OCP microscaling: e4m3 elements, one E8M0 scale per 32 consecutive K elements) products on the block-scaled
// MFMA of gfx950 (v_mfma_scale_f32_32x32x64_f8f6f4, 2x the bf16 MFMA rate) -- BASELINE.json configs[4] ("fp8 MFMA path").
//
//   mmae_mx_quant      act [rows][cols] (bf16 / f32) -> e4m3 [rows][cols] + packed scales       (operands of forward / dX products)
//   mmae_mx_quant_t    w   [n][k] (bf16 / f32)       -> e4m3 [k][n] (blocks along n) + scales   (the dX product's weight operand)
//   mmae_gemm (ab_dtype MMAE_MXFP8) -> mmae_gemm_mxfp8_impl: C[M][N] = A[M][K] . B[N][K]^T, both operands k-contiguous e4m3
//   mmae_probe_mx_mfma one scaled MFMA with caller-supplied registers: pins the operand / scale lane layout the kernels rely on
//
// Scale layout ("MFMA-ready"): u32 S[ceil(K / 256)][rows][2]; byte j of S[g][r][h] is the E8M0 exponent of the 32-element block
// K = 256 g + 64 j + 32 h .. + 32 of row r.  Measured register layout of the 32x32x64 MFMA (either operand; the ISA text does not
// spell it out): lane l < 32 holds K 0-15 (bytes 0-15 of its 8 VGPRs) and K 32-47 (bytes 16-31) of row l, lane l + 32 holds
// K 16-31 and K 48-63; the byte of lane l's scale VGPR selected by op_sel scales K 0-31 of row l (the first 16 bytes of both
// lanes), lane l + 32's scales K 32-63.  So a lane reads two 16-byte chunks of the 64-byte row (c and c + 2), one scale dword per
// lane covers four K tiles, and the selector is the K tile's index mod 4 -- an immediate in the K loop unrolled over the ring.
//
// The GEMM body is the 8-wave ping-pong structure of gemm_pp_body.h (see there for the barrier / ring reasoning) with a K tile
// of 64 BYTES per row as before -- now 64 elements: the LDS images, DMA pieces and their counted waits are byte-identical; a
// wave's two half-phases per K tile split its (TM x 2) scaled MFMAs (64 cycles each) by output rows instead of by K slice.
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include "gemm_common.h"

#define LDS_AS __attribute__((address_space(3)))
typedef __attribute__((ext_vector_type(8))) int i32x8;

int mmae_gemm_mxfp8_impl(const mmae_gemm_desc* d, const GemmArgs& g, hipStream_t st);

namespace {

constexpr unsigned OOB = 0x80000000u;

// ---------------------------------------------------------------------------------------------------------------------------
// quantisation
// ---------------------------------------------------------------------------------------------------------------------------
// (mx_shared_exp, mx_inv_scale, mx_cvt4_e4m3, mx_scale_addr: gemm_common.h)
__device__ __forceinline__ int cvt4_e4m3(float a, float b, float c, float d) { return mx_cvt4_e4m3(a, b, c, d); }

// 4 lanes per 32-element block (8 elements = 16 B of bf16 per lane): coalesced 1-KiB reads per wave instruction
template <typename T>
__global__ void __launch_bounds__(256) mx_quant_kernel(const T* __restrict__ x, long long ldx, int rows, int cols, unsigned char* __restrict__ q,
                                                       long long ldq, unsigned char* __restrict__ sc) {
    const int cpr = (cols + 7) >> 3;                     // 8-element chunks per row (cols % 32 == 0 -> whole blocks)
    const long long total = (long long)rows * cpr;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long r = idx / cpr;
        const int c8 = (int)(idx - r * cpr);
        float v[8];
        if (sizeof(T) == 2) {
            const i32x4 raw = *reinterpret_cast<const i32x4*>((const uint16_t*)x + r * ldx + c8 * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(((unsigned)raw[j]) << 16); v[2 * j + 1] = __uint_as_float(((unsigned)raw[j]) & 0xffff0000u); }
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>((const float*)x + r * ldx + c8 * 8);
            const f32x4 b = *reinterpret_cast<const f32x4*>((const float*)x + r * ldx + c8 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
        }
        float am = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(v[j]));
        am = fmaxf(am, __shfl_xor(am, 1, 64));
        am = fmaxf(am, __shfl_xor(am, 2, 64));
        const int e = mx_shared_exp(am);
        const float inv = mx_inv_scale(e);
        i32x2 o;
        o[0] = cvt4_e4m3(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
        o[1] = cvt4_e4m3(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv);
        *reinterpret_cast<i32x2*>(q + r * ldq + c8 * 8) = o;
        if ((c8 & 3) == 0) sc[mx_scale_addr(rows, r, c8 >> 2)] = (unsigned char)e;
    }
}

// weight [n][k] -> e4m3 [k][n], blocks of 32 along n.  One thread per (k, n-block): the 32 reads of a wave instruction are
// consecutive k (coalesced); the 32-byte result rows are strided by n.  Runs once per optimiser step on ~1e8 elements.
template <typename T>
__global__ void __launch_bounds__(256) mx_quant_t_kernel(const T* __restrict__ w, long long ldw, int n, int k, unsigned char* __restrict__ q,
                                                         long long ldq, unsigned char* __restrict__ sc) {
    const int nb = n >> 5;
    const long long total = (long long)nb * k;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int b = (int)(idx / k), kk = (int)(idx - (long long)b * k);
        float v[32];
        float am = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) { v[i] = ActT<T>::ld(w + (long long)(b * 32 + i) * ldw + kk); am = fmaxf(am, fabsf(v[i])); }
        const int e = mx_shared_exp(am);
        const float inv = mx_inv_scale(e);
        int o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = cvt4_e4m3(v[4 * i] * inv, v[4 * i + 1] * inv, v[4 * i + 2] * inv, v[4 * i + 3] * inv);
        i32x4* dst = reinterpret_cast<i32x4*>(q + (long long)kk * ldq + b * 32);
        dst[0] = i32x4{o[0], o[1], o[2], o[3]};
        dst[1] = i32x4{o[4], o[5], o[6], o[7]};
        sc[mx_scale_addr(k, kk, b)] = (unsigned char)e;
    }
}

// activation [M][C] (bf16) -> e4m3 [C][Mp], blocks of 32 along M (the contraction of the weight-gradient products dW = dY^T X), rows
// past M are zeros.  One workgroup per 128 x 64 tile: whole 128-byte lines in (8 lanes per row), a transposing pass through LDS
// (the column of a piece is rotated by 16 per 32-row block so the four blocks a wave reads sit in different banks), whole
// 128-byte lines out (four lanes write the four 32-byte blocks of one output row).
__global__ void __launch_bounds__(256) mx_quant_rows_t_kernel(const uint16_t* __restrict__ x, long long ldx, int M, int C, unsigned char* __restrict__ q,
                                                              long long ldq, unsigned char* __restrict__ sc) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[128][72];
    const int m0 = blockIdx.x * 128, c0 = blockIdx.y * 64, tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = p * 32 + (tid >> 3), cc = (tid & 7) * 8;
        i32x4 v = {0, 0, 0, 0};
        if (m0 + r < M) v = *reinterpret_cast<const i32x4*>(x + (long long)(m0 + r) * ldx + c0 + cc);
        *reinterpret_cast<i32x4*>(&tile[r][(cc + 16 * p) & 63]) = v;
    }
    __syncthreads();
    const int c = tid >> 2, blk = tid & 3;
    float v[32];
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { v[i] = bf16_bits_to_f32(tile[blk * 32 + i][(c + 16 * blk) & 63]); am = fmaxf(am, fabsf(v[i])); }
    const int e = mx_shared_exp(am);
    const float inv = mx_inv_scale(e);
    int o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = cvt4_e4m3(v[4 * i] * inv, v[4 * i + 1] * inv, v[4 * i + 2] * inv, v[4 * i + 3] * inv);
    i32x4* dst = reinterpret_cast<i32x4*>(q + (long long)(c0 + c) * ldq + m0 + blk * 32);
    dst[0] = i32x4{o[0], o[1], o[2], o[3]};
    dst[1] = i32x4{o[4], o[5], o[6], o[7]};
    sc[mx_scale_addr(C, c0 + c, (m0 >> 5) + blk)] = (unsigned char)e;
}

// up to 32 weights per launch (blockIdx.y = weight): the per-step refresh of a ViT-L encoder is 6 launches instead of 192
struct MxWeightTab { const void* w[32]; unsigned char* q[32]; unsigned char* s[32]; int n[32]; int k[32]; };

template <typename T>
__global__ void __launch_bounds__(256) mx_quant_batch_kernel(const MxWeightTab tab) {
    const int i = blockIdx.y;
    const T* __restrict__ x = (const T*)tab.w[i];
    unsigned char* __restrict__ q = tab.q[i];
    unsigned char* __restrict__ sc = tab.s[i];
    const int rows = tab.n[i], cols = tab.k[i], cpr = cols >> 3;
    const long long total = (long long)rows * cpr;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long r = idx / cpr;
        const int c8 = (int)(idx - r * cpr);
        float v[8];
        if (sizeof(T) == 2) {
            const i32x4 raw = *reinterpret_cast<const i32x4*>((const uint16_t*)x + r * cols + c8 * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(((unsigned)raw[j]) << 16); v[2 * j + 1] = __uint_as_float(((unsigned)raw[j]) & 0xffff0000u); }
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>((const float*)x + r * cols + c8 * 8);
            const f32x4 b = *reinterpret_cast<const f32x4*>((const float*)x + r * cols + c8 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
        }
        float am = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(v[j]));
        am = fmaxf(am, __shfl_xor(am, 1, 64));
        am = fmaxf(am, __shfl_xor(am, 2, 64));
        const int e = mx_shared_exp(am);
        const float inv = mx_inv_scale(e);
        i32x2 o;
        o[0] = cvt4_e4m3(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
        o[1] = cvt4_e4m3(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv);
        *reinterpret_cast<i32x2*>(q + r * cols + c8 * 8) = o;
        if ((c8 & 3) == 0) sc[mx_scale_addr(rows, r, c8 >> 2)] = (unsigned char)e;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) mx_quant_t_batch_kernel(const MxWeightTab tab) {
    const int i = blockIdx.y;
    const T* __restrict__ w = (const T*)tab.w[i];
    unsigned char* __restrict__ q = tab.q[i];
    unsigned char* __restrict__ sc = tab.s[i];
    const int n = tab.n[i], k = tab.k[i], nb = n >> 5;
    const long long total = (long long)nb * k;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int b = (int)(idx / k), kk = (int)(idx - (long long)b * k);
        float v[32];
        float am = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) { v[j] = ActT<T>::ld(w + (long long)(b * 32 + j) * k + kk); am = fmaxf(am, fabsf(v[j])); }
        const int e = mx_shared_exp(am);
        const float inv = mx_inv_scale(e);
        int o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = cvt4_e4m3(v[4 * j] * inv, v[4 * j + 1] * inv, v[4 * j + 2] * inv, v[4 * j + 3] * inv);
        i32x4* dst = reinterpret_cast<i32x4*>(q + (long long)kk * n + b * 32);
        dst[0] = i32x4{o[0], o[1], o[2], o[3]};
        dst[1] = i32x4{o[4], o[5], o[6], o[7]};
        sc[mx_scale_addr(k, kk, b)] = (unsigned char)e;
    }
}

__global__ void mx_scale_clear_kernel(unsigned* __restrict__ s, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s[i] = 0u;
}

// ---------------------------------------------------------------------------------------------------------------------------
// probe
// ---------------------------------------------------------------------------------------------------------------------------
template <int OA, int OB>
__global__ void probe_mx_kernel(const int* __restrict__ a, const int* __restrict__ b, const int* __restrict__ sa, const int* __restrict__ sb,
                                float* __restrict__ out) {
    const int l = threadIdx.x;
    i32x8 av, bv;
#pragma unroll
    for (int i = 0; i < 8; ++i) { av[i] = a[l * 8 + i]; bv[i] = b[l * 8 + i]; }
    f32x16 c;
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, OA, sa[l], OB, sb[l]);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}

// ---------------------------------------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int kc_off(int row, int c) {          // as gemm_pp_body.h: rows of 64 B (4 chunks), 4 rows per 256-B bank row
    return (row >> 2) * 256 + (((((row & 3) << 2) | c) ^ ((row >> 2) & 15)) << 4);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T>
__device__ __forceinline__ T* sgpr_ptr(T* p) {
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

template <int TM, int FL>
__global__ void __launch_bounds__(512) gemm_mxfp8_kernel(const GemmArgs g) {
    constexpr int WMR = TM * 32, BM = 2 * WMR, BN = 256, NW = 8;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int PA = A_BYTES / 1024, PB = B_BYTES / 1024;
    constexpr int LA = (PA + NW - 1) / NW, LB = PB / NW;
    constexpr int NST = 4, DUMP = NST * STAGE;
    constexpr int W = LA + LB + 2;
    constexpr int H0 = (TM + 1) / 2;                     // output row blocks (of 32) multiplied in the first half-phase
    constexpr int NS = TM + 2;                           // scale dwords a lane fetches per group of four K tiles
    static_assert(LB == 2 && LA >= 2 && LA <= 3, "piece schedule assumes 2 + (2|3) pieces per wave and tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const unsigned char* Az = sgpr_ptr((const unsigned char*)g.A);
    const unsigned char* Bz = sgpr_ptr((const unsigned char*)g.B);
    char* Cz = (char*)g.C;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, 0x80000000, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bz, 0, 0x80000000, 0x00020000);
    // K tiles of 64 elements; the host guarantees K % 256 == 0.  Split-K (the weight-gradient products: few output tiles, the whole
    // batch as contraction): virtual tile v = (slice, output tile); a slice covers kt_per_split K tiles (a multiple of 4, i.e. whole
    // scale groups) and writes its own dense f32 slab of the workspace, summed afterwards in a fixed order (mmae_splitk_reduce).
    const int T_all = g.K >> 6;
    const int groups = T_all >> 2;
    const int Tper = g.splitk > 1 ? g.kt_per_split : T_all;
    int T = T_all, gb = 0;                               // this virtual tile's K tiles and first scale group
    long long c_off = 0;                                 // ... and the byte offset of its slab
    const auto rsSA = __builtin_amdgcn_make_buffer_rsrc((void*)sgpr_ptr((const unsigned*)g.scA), 0, (int)((long long)groups * g.M * 8), 0x00020000);
    const auto rsSB = __builtin_amdgcn_make_buffer_rsrc((void*)sgpr_ptr((const unsigned*)g.scB), 0, (int)((long long)groups * g.N * 8), 0x00020000);

    unsigned a_cur[LA], b_cur[LB];
    unsigned sa_off[TM], sb_off[2];                      // byte offset of this lane's scale dword in group 0
    int m0 = 0, n0 = 0;
    auto set_tile = [&](int vv, int& tm0, int& tn0) {
        const int sk = __builtin_amdgcn_readfirstlane(vv / g.tiles_total), v = vv - sk * g.tiles_total;
        const unsigned k0 = (unsigned)(sk * Tper) * 64u;
        T = T_all - sk * Tper < Tper ? T_all - sk * Tper : Tper;
        gb = (sk * Tper) >> 2;
        c_off = (long long)sk * g.M * g.N * 4;
        const int tile = g.xcd_swizzle ? xcd_tile(v, g.tiles_total) : v;
        const int tile_m = __builtin_amdgcn_readfirstlane(tile / g.tiles_n);
        tm0 = tile_m * BM; tn0 = (tile - tile_m * g.tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int seg = i * NW + wave;
            const int b_abs = 4 * seg + (lane >> 4), j = (lane & 15) ^ (b_abs & 15);
            const int row = 4 * b_abs + (j >> 2), c = j & 3;
            a_cur[i] = (seg < PA && tm0 + row < g.M) ? (unsigned)(((long long)(tm0 + row)) * g.lda + c * 16) + k0 : OOB;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int seg = i * NW + wave;
            const int b_abs = 4 * seg + (lane >> 4), j = (lane & 15) ^ (b_abs & 15);
            const int row = 4 * b_abs + (j >> 2), c = j & 3;
            b_cur[i] = (tn0 + row < g.N) ? (unsigned)(((long long)(tn0 + row)) * g.ldb + c * 16) + k0 : OOB;
        }
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            int r = tm0 + wm * WMR + t * 32 + (lane & 31);
            r = r < g.M ? r : g.M - 1;                   // rows past M multiply zero data (out-of-range DMA): any finite scale will do
            sa_off[t] = (unsigned)(r * 8 + (lane >> 5) * 4);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int r = tn0 + wn * 64 + t * 32 + (lane & 31);
            r = r < g.N ? r : g.N - 1;
            sb_off[t] = (unsigned)(r * 8 + (lane >> 5) * 4);
        }
    };
    set_tile(blockIdx.x, m0, n0);

    auto dma_a = [&](int u, int i, int slot) {
        char* dst = (i * NW + wave < PA) ? smem + slot * STAGE + (i * NW + wave) * 1024 : smem + DUMP;
        const unsigned tail = u < T ? 0u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)dst, 16, (int)(a_cur[i] | tail), 0, 0, 0);
        a_cur[i] += 64u;
    };
    auto dma_b = [&](int u, int i, int slot) {
        char* dst = smem + slot * STAGE + A_BYTES + (i * NW + wave) * 1024;
        const unsigned tail = u < T ? 0u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)dst, 16, (int)(b_cur[i] | tail), 0, 0, 0);
        b_cur[i] += 64u;
    };
    const bool has3 = LA == 3 && 2 * NW + wave < PA;
    auto dma_first = [&](int u, int slot) { dma_a(u, 0, slot); dma_a(u, 1, slot); };
    auto dma_second = [&](int u, int slot) { dma_b(u, 0, slot); dma_b(u, 1, slot); if (LA == 3 && has3) dma_a(u, 2, slot); };
    // EXTRA: younger non-DMA loads (the scale prefetch of the next group) that may stay in flight at this wait
    auto wait_tile = [&](auto extra) {
        constexpr int X = decltype(extra)::value;
        if (LA == 3 && !has3) wait_vm<W - 1 + X>(); else wait_vm<W + X>();
    };

    f32x16 acc[2][TM];
    int fr = 0, fk = 0;
    auto derive = [&]() { fr = lane & 31; fk = lane >> 5; };
    derive();
    auto frag = [&](const char* base, int row0) -> i32x8 {
        // measured operand layout (tools/mx_probe_discover.py, tests/test_mxfp8_gpu.py): lane l < 32 holds K 0-15 and 32-47 of row l,
        // lane l + 32 holds K 16-31 and 48-63 -- 16-byte chunks fk and fk + 2 of the 64-byte row; lane l's scale byte covers K 0-31
        // (the first 16 bytes of BOTH lanes), lane l + 32's covers K 32-63
        const i32x4 lo = *reinterpret_cast<const i32x4*>(base + kc_off(row0 + fr, fk));
        const i32x4 hi = *reinterpret_cast<const i32x4*>(base + kc_off(row0 + fr, fk + 2));
        return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    i32x8 af[H0], bf[2];
    int sa[TM], sb[2], sa_n[TM], sb_n[2];
    auto load_scales = [&](int grp, int (&da)[TM], int (&db)[2]) {
        grp += gb;
        grp = grp < groups ? grp : groups - 1;                           // past the last group: any valid dwords (never used)
        const int go_a = grp * g.M * 8, go_b = grp * g.N * 8;            // wave-uniform group offset (soffset)
#pragma unroll
        for (int t = 0; t < TM; ++t) da[t] = __builtin_amdgcn_raw_buffer_load_b32(rsSA, (int)sa_off[t], go_a, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) db[t] = __builtin_amdgcn_raw_buffer_load_b32(rsSB, (int)sb_off[t], go_b, 0);
    };
    auto mem_phase = [&](int u, int kk, int slot) {
        const char* sa_ = smem + slot * STAGE;
        const char* sb_ = sa_ + A_BYTES;
        if (kk == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t) bf[t] = frag(sb_, wn * 64 + t * 32);
#pragma unroll
            for (int t = 0; t < H0; ++t) af[t] = frag(sa_, wm * WMR + t * 32);
            dma_second(u + 2, (slot + 2) & (NST - 1));
        } else {
#pragma unroll
            for (int t = H0; t < TM; ++t) af[t - H0] = frag(sa_, wm * WMR + t * 32);
            dma_first(u + 3, (slot + 3) & (NST - 1));
        }
    };
    auto mfma_phase = [&](int kk, auto jc) {
        constexpr int J = decltype(jc)::value;
        __builtin_amdgcn_s_setprio(1);
        if (kk == 0) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < H0; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bf[tn], af[tm], acc[tn][tm], 0, 0, J, sb[tn], J, sa[tm]);
        } else {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = H0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bf[tn], af[tm - H0], acc[tn][tm], 0, 0, J, sb[tn], J, sa[tm]);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using INS = std::integral_constant<int, NS>;

    load_scales(0, sa, sb);
    dma_first(0, 0); dma_second(0, 0);
    dma_first(1, 1); dma_second(1, 1);
    dma_first(2, 2);
    wait_tile(I0{});
    wg_barrier();

    char* stage = smem + 2 * STAGE + wave * 8192;
    static_assert(2 * STAGE >= 8 * 8192, "staging must fit in ring slots 2-3");
    const int n_virtual = g.tiles_total * (g.splitk > 1 ? g.splitk : 1);
    for (int v = blockIdx.x; v < n_virtual; v += gridDim.x) {
        asm volatile("" : "+v"(lane));
        derive();
        char* Cv = Cz + c_off;                           // (set_tile below moves c_off on to the next virtual tile)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        // one group = four K tiles = the four ring slots: slot and scale-byte selector are constants in every copy of the body
        auto tile_g0 = [&](int u, auto jc) {
            constexpr int J = decltype(jc)::value;
            if (J == 0) load_scales((u >> 2) + 1, sa_n, sb_n);
            mem_phase(u, 0, J);
            wg_barrier();
            mfma_phase(0, jc);
            wg_barrier();
            mem_phase(u, 1, J);
            wg_barrier();
            mfma_phase(1, jc);
            if (J == 0) wait_tile(INS{}); else wait_tile(I0{});          // K tile u + 1
            wg_barrier();
        };
        auto tile_g1 = [&](int u, auto jc) {
            constexpr int J = decltype(jc)::value;
            wg_barrier();
            if (J == 0) load_scales((u >> 2) + 1, sa_n, sb_n);
            mem_phase(u, 0, J);
            wg_barrier();
            mfma_phase(0, jc);
            wg_barrier();
            mem_phase(u, 1, J);
            if (J == 0) wait_tile(INS{}); else wait_tile(I0{});
            wg_barrier();
            mfma_phase(1, jc);
        };
        auto next_scales = [&]() {
#pragma unroll
            for (int t = 0; t < TM; ++t) sa[t] = sa_n[t];
#pragma unroll
            for (int t = 0; t < 2; ++t) sb[t] = sb_n[t];
        };
        if (wm == 0) {
            for (int u0 = 0; u0 < T; u0 += NST) {
                tile_g0(u0, std::integral_constant<int, 0>{});
                tile_g0(u0 + 1, std::integral_constant<int, 1>{});
                tile_g0(u0 + 2, std::integral_constant<int, 2>{});
                tile_g0(u0 + 3, std::integral_constant<int, 3>{});
                next_scales();
            }
        } else {
            for (int u0 = 0; u0 < T; u0 += NST) {
                tile_g1(u0, std::integral_constant<int, 0>{});
                tile_g1(u0 + 1, std::integral_constant<int, 1>{});
                tile_g1(u0 + 2, std::integral_constant<int, 2>{});
                tile_g1(u0 + 3, std::integral_constant<int, 3>{});
                next_scales();
            }
        }
        wait_vm<0>();
        __syncthreads();

        const int mw = m0 + wm * WMR, nw = n0 + wn * 64;
        const bool has_next = v + (int)gridDim.x < n_virtual;
        if (has_next) {
            asm volatile("" : "+v"(lane));
            set_tile(v + gridDim.x, m0, n0);
            load_scales(0, sa, sb);
            dma_first(0, 0); dma_second(0, 0);
            dma_first(1, 1); dma_second(1, 1);
        }
        {
            f32x16 sub[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
            gemm_store_tile64_fl<FL>(g, Cv, stage, lane, sub, mw, nw);
        }
        {
            f32x16 sub[2][2] = {{acc[0][2], acc[0][3]}, {acc[1][2], acc[1][3]}};
            gemm_store_tile64_fl<FL>(g, Cv, stage, lane, sub, mw + 64, nw);
        }
        if (TM & 1) {
            f32x16 sub[2][2] = {{acc[0][TM - 1], acc[0][TM - 1]}, {acc[1][TM - 1], acc[1][TM - 1]}};
            gemm_store_tile64_fl<FL>(g, Cv, stage, lane, sub, mw + (TM - 1) * 32, nw, 1);
        }
        if (has_next) {
            wait_vm<0>();
            __syncthreads();
            dma_first(2, 2);
        }
    }
}

template <int TM, int FL>
int launch_mx(const GemmArgs& g, hipStream_t st) {
    constexpr int BM = TM * 64, BN = 256;
    GemmArgs a = g;
    a.tiles_n = (g.N + BN - 1) / BN;
    a.tiles_total = ((g.M + BM - 1) / BM) * a.tiles_n;
    if (a.splitk > 1) { a.C = a.ws; a.ldc = a.N; a.accumulate = 0; }      // slices write dense f32 slabs [splitk][M][N]
    const int n_cu = mmae_cu_avail();
    const long long nv = (long long)a.tiles_total * (a.splitk > 1 ? a.splitk : 1);
    const int gx = nv > n_cu ? n_cu : (int)nv;
    const size_t lds = (size_t)4 * (BM + BN) * 64 + 1024;
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute((const void*)gemm_mxfp8_kernel<TM, FL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    hipLaunchKernelGGL((gemm_mxfp8_kernel<TM, FL>), dim3(gx), dim3(512), lds, st, a);
    return mmae_check_launch("gemm_mxfp8");
}

template <int TM>
int launch_mx_fl(const GemmArgs& g, int fl, hipStream_t st) {
    switch (fl) {
        case FL_BF16_BIAS: return launch_mx<TM, FL_BF16_BIAS>(g, st);
        case FL_BF16_BIAS_GELU: return launch_mx<TM, FL_BF16_BIAS_GELU>(g, st);
        case FL_F32_BIAS_RESID: return launch_mx<TM, FL_F32_BIAS_RESID>(g, st);
        case FL_F32_BIAS: return launch_mx<TM, FL_F32_BIAS>(g, st);
        case FL_BF16: return launch_mx<TM, FL_BF16>(g, st);
        case FL_BF16_DGELU_CS: return launch_mx<TM, FL_BF16_DGELU_CS>(g, st);
        case FL_BF16_DGELU: return launch_mx<TM, FL_BF16_DGELU>(g, st);
        case FL_F32: return launch_mx<TM, FL_F32>(g, st);
        case FL_BF16_BIAS_GELU_Q: return launch_mx<TM, FL_BF16_BIAS_GELU_Q>(g, st);
        case FL_BF16_DGELU_CS_Q: return launch_mx<TM, FL_BF16_DGELU_CS_Q>(g, st);
        case FL_BF16_DGELU_Q: return launch_mx<TM, FL_BF16_DGELU_Q>(g, st);
        default: mmae_set_error("gemm(mxfp8): epilogue combination without a compiled flavour (see gemm_flavour())"); return MMAE_ESUPPORT;
    }
}

}  // namespace

// C[M][N] = A[M][K] . B[N][K]^T on the scaled MFMA.  Called by mmae_gemm for ab_dtype MMAE_MXFP8.
int mmae_gemm_mxfp8_impl(const mmae_gemm_desc* d, const GemmArgs& g, hipStream_t st) {
    MMAE_REQUIRE(d->a_scale && d->b_scale, "gemm(mxfp8): operands need their scale arrays");
    MMAE_REQUIRE(!d->a_trans && !d->b_trans, "gemm(mxfp8): both operands must be k-contiguous (quantise the transposed copy)");
    MMAE_REQUIRE(d->batch == 1, "gemm(mxfp8): unbatched products only");
    MMAE_REQUIRE(g.splitk <= 1 || (g.kt_per_split % 4 == 0 && g.ws && !d->q_out && (long long)g.splitk * d->M * d->N * 4 < 0x7fffffffLL * 4LL),
                 "gemm(mxfp8): split_k slices must be whole scale groups and need the workspace");
    MMAE_REQUIRE(d->K % 256 == 0, "gemm(mxfp8): K must be a multiple of 256 (four 64-element K tiles per scale dword)");
    MMAE_REQUIRE(d->lda % 16 == 0 && d->ldb % 16 == 0 && (uintptr_t)d->A % 16 == 0 && (uintptr_t)d->B % 16 == 0, "gemm(mxfp8): operand rows must be 16-byte aligned");
    MMAE_REQUIRE((long long)d->M * d->lda < 0x7fffffffLL && (long long)d->N * d->ldb < 0x7fffffffLL, "gemm(mxfp8): operand larger than a 2 GiB buffer window");
    MMAE_REQUIRE((long long)(d->K / 256) * d->M * 8 < 0x7fffffffLL && (long long)(d->K / 256) * d->N * 8 < 0x7fffffffLL, "gemm(mxfp8): scale array too large");
    int fl = g.splitk > 1 ? (int)FL_F32 : gemm_flavour(g, d->batch);      // split-K slices: plain dense f32 slabs
    if (d->q_out) {                                       // fused quantisation of the output for the next MX product
        MMAE_REQUIRE(d->q_scale && d->N % 32 == 0 && d->ldq % 8 == 0 && (uintptr_t)d->q_out % 8 == 0, "gemm(mxfp8): q_out needs q_scale, N % 32 == 0 and 8-byte aligned rows");
        fl = fl == FL_BF16_BIAS_GELU ? FL_BF16_BIAS_GELU_Q : fl == FL_BF16_DGELU_CS ? FL_BF16_DGELU_CS_Q : fl == FL_BF16_DGELU ? FL_BF16_DGELU_Q : -1;
        if (fl < 0) { mmae_set_error("gemm(mxfp8): q_out is implemented for the bias + GELU and the dGELU epilogues only"); return MMAE_ESUPPORT; }
    }
    const long long nt = (d->N + 255) / 256;
    const long long t4 = ((d->M + 255) / 256) * nt, t5 = ((d->M + 319) / 320) * nt;
    const long long c4 = ((t4 + 255) / 256) * 256, c5 = ((t5 + 255) / 256) * 320;
    (void)c4; (void)c5;                                   // 320-row tiles spill ~35 registers in this body and measured 14 % slower:
#ifdef MMAE_EXPERIMENTS                                   // instantiated in experiment builds only (MMAE_MX_TM=5)
    static const int env_tm = mmae_env_int("MMAE_MX_TM", 0);
    if (env_tm == 5) return launch_mx_fl<5>(g, fl, st);
#endif
    return launch_mx_fl<4>(g, fl, st);
}

extern "C" {

int64_t mmae_mx_scale_bytes(int rows, int cols) { return (int64_t)((cols + 255) / 256) * rows * 8; }

int mmae_mx_quant(const void* x, int x_dtype, int64_t ldx, int rows, int cols, void* q, int64_t ldq, void* scales, void* stream) {
    MMAE_REQUIRE(x && q && scales && rows > 0 && cols > 0, "mx_quant: bad argument");
    MMAE_REQUIRE(x_dtype == MMAE_F32 || x_dtype == MMAE_BF16, "mx_quant: input must be f32 or bf16");
    MMAE_REQUIRE(cols % 32 == 0 && ldx % 8 == 0 && ldq % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)q % 8 == 0, "mx_quant: cols % 32, 16-byte aligned rows");
    hipStream_t st = (hipStream_t)stream;
    if (cols % 256) {                                    // blocks past the last column keep exponent 0
        const long long n = mmae_mx_scale_bytes(rows, cols) / 4;
        hipLaunchKernelGGL(mx_scale_clear_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, st, (unsigned*)scales, n);
    }
    const long long total = (long long)rows * (cols / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (x_dtype == MMAE_BF16)
        hipLaunchKernelGGL(mx_quant_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, (const uint16_t*)x, (long long)ldx, rows, cols, (unsigned char*)q, (long long)ldq, (unsigned char*)scales);
    else
        hipLaunchKernelGGL(mx_quant_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (long long)ldx, rows, cols, (unsigned char*)q, (long long)ldq, (unsigned char*)scales);
    return mmae_check_launch("mx_quant");
}

int mmae_mx_quant_t(const void* w, int w_dtype, int64_t ldw, int n, int k, void* q, int64_t ldq, void* scales, void* stream) {
    MMAE_REQUIRE(w && q && scales && n > 0 && k > 0, "mx_quant_t: bad argument");
    MMAE_REQUIRE(w_dtype == MMAE_F32 || w_dtype == MMAE_BF16, "mx_quant_t: input must be f32 or bf16");
    MMAE_REQUIRE(n % 32 == 0 && ldq % 16 == 0 && (uintptr_t)q % 16 == 0, "mx_quant_t: n % 32, 16-byte aligned output rows");
    hipStream_t st = (hipStream_t)stream;
    if (n % 256) {
        const long long m = mmae_mx_scale_bytes(k, n) / 4;
        hipLaunchKernelGGL(mx_scale_clear_kernel, dim3((unsigned)((m + 255) / 256 < 4096 ? (m + 255) / 256 : 4096)), dim3(256), 0, st, (unsigned*)scales, m);
    }
    const long long total = (long long)(n / 32) * k;
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (w_dtype == MMAE_BF16)
        hipLaunchKernelGGL(mx_quant_t_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, (const uint16_t*)w, (long long)ldw, n, k, (unsigned char*)q, (long long)ldq, (unsigned char*)scales);
    else
        hipLaunchKernelGGL(mx_quant_t_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)w, (long long)ldw, n, k, (unsigned char*)q, (long long)ldq, (unsigned char*)scales);
    return mmae_check_launch("mx_quant_t");
}

int mmae_mx_quant_rows_t(const void* x, int x_dtype, int64_t ldx, int M, int C, void* q, int64_t ldq, void* scales, void* stream) {
    MMAE_REQUIRE(x && q && scales && M > 0 && C > 0, "mx_quant_rows_t: bad argument");
    MMAE_REQUIRE(x_dtype == MMAE_BF16, "mx_quant_rows_t: bf16 activations only");
    MMAE_REQUIRE(C % 64 == 0 && ldx % 8 == 0 && (uintptr_t)x % 16 == 0 && ldq % 256 == 0 && ldq >= M && (uintptr_t)q % 16 == 0,
                 "mx_quant_rows_t: C % 64, 16-byte aligned rows, ldq a multiple of 256 >= M");
    hipLaunchKernelGGL(mx_quant_rows_t_kernel, dim3((unsigned)(ldq / 128), (unsigned)(C / 64)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (long long)ldx, M, C, (unsigned char*)q, (long long)ldq, (unsigned char*)scales);
    return mmae_check_launch("mx_quant_rows_t");
}

int mmae_mx_scale_clear(void* scales, int rows, int cols, void* stream) {
    MMAE_REQUIRE(scales && rows > 0 && cols > 0, "mx_scale_clear: bad argument");
    const long long n = mmae_mx_scale_bytes(rows, cols) / 4;
    hipLaunchKernelGGL(mx_scale_clear_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream, (unsigned*)scales, n);
    return mmae_check_launch("mx_scale_clear");
}

int64_t mmae_mx_tmp_bytes(int rows, int cols) {
    return ((int64_t)rows * cols + 255) / 256 * 256 + (mmae_mx_scale_bytes(rows, cols) + 255) / 256 * 256;
}

int mmae_mx_prepare_weights(int n, const void* const* w, int w_dtype, const int32_t* n_out, const int32_t* k_in, void* const* dst, void* stream) {
    MMAE_REQUIRE(n >= 0 && (n == 0 || (w && n_out && k_in && dst)), "mx_prepare_weights: null argument");
    MMAE_REQUIRE(w_dtype == MMAE_F32 || w_dtype == MMAE_BF16, "mx_prepare_weights: weights must be f32 or bf16");
    hipStream_t st = (hipStream_t)stream;
    bool batched = true;                                 // the batched kernels want whole scale groups in both orientations
    for (int i = 0; i < n; ++i) {
        MMAE_REQUIRE(w[i] && dst[4 * i] && dst[4 * i + 1] && dst[4 * i + 2] && dst[4 * i + 3], "mx_prepare_weights: null pointer");
        MMAE_REQUIRE(n_out[i] > 0 && k_in[i] > 0 && n_out[i] % 32 == 0 && k_in[i] % 32 == 0, "mx_prepare_weights: widths must be multiples of 32");
        batched = batched && n_out[i] % 256 == 0 && k_in[i] % 256 == 0 && (uintptr_t)w[i] % 16 == 0 && (uintptr_t)dst[4 * i] % 16 == 0 && (uintptr_t)dst[4 * i + 2] % 16 == 0;
    }
    if (!batched) {
        for (int i = 0; i < n; ++i) {
            int rc = mmae_mx_quant(w[i], w_dtype, k_in[i], n_out[i], k_in[i], dst[4 * i], k_in[i], dst[4 * i + 1], stream);
            if (rc) return rc;
            rc = mmae_mx_quant_t(w[i], w_dtype, k_in[i], n_out[i], k_in[i], dst[4 * i + 2], n_out[i], dst[4 * i + 3], stream);
            if (rc) return rc;
        }
        return 0;
    }
    for (int i0 = 0; i0 < n; i0 += 32) {
        const int m = n - i0 < 32 ? n - i0 : 32;
        MxWeightTab a = {}, t = {};
        for (int j = 0; j < m; ++j) {
            const int i = i0 + j;
            a.w[j] = t.w[j] = w[i]; a.n[j] = t.n[j] = n_out[i]; a.k[j] = t.k[j] = k_in[i];
            a.q[j] = (unsigned char*)dst[4 * i]; a.s[j] = (unsigned char*)dst[4 * i + 1];
            t.q[j] = (unsigned char*)dst[4 * i + 2]; t.s[j] = (unsigned char*)dst[4 * i + 3];
        }
        if (w_dtype == MMAE_BF16) {
            hipLaunchKernelGGL(mx_quant_batch_kernel<uint16_t>, dim3(128, m), dim3(256), 0, st, a);
            hipLaunchKernelGGL(mx_quant_t_batch_kernel<uint16_t>, dim3(128, m), dim3(256), 0, st, t);
        } else {
            hipLaunchKernelGGL(mx_quant_batch_kernel<float>, dim3(128, m), dim3(256), 0, st, a);
            hipLaunchKernelGGL(mx_quant_t_batch_kernel<float>, dim3(128, m), dim3(256), 0, st, t);
        }
    }
    return mmae_check_launch("mx_prepare_weights");
}

int mmae_probe_mx_mfma(const int32_t* a_64x8, const int32_t* b_64x8, const int32_t* scale_a_64, const int32_t* scale_b_64, int opsel_a, int opsel_b,
                       float* out_64x16, void* stream) {
    MMAE_REQUIRE(a_64x8 && b_64x8 && scale_a_64 && scale_b_64 && out_64x16, "probe_mx_mfma: null pointer");
    MMAE_REQUIRE(opsel_a >= 0 && opsel_a < 4 && opsel_b >= 0 && opsel_b < 4, "probe_mx_mfma: op_sel is a byte index 0..3");
    hipStream_t st = (hipStream_t)stream;
#define MMAE_PROBE_CASE(OA, OB) case OA * 4 + OB: hipLaunchKernelGGL((probe_mx_kernel<OA, OB>), dim3(1), dim3(64), 0, st, a_64x8, b_64x8, scale_a_64, scale_b_64, out_64x16); break;
    switch (opsel_a * 4 + opsel_b) {
        MMAE_PROBE_CASE(0, 0) MMAE_PROBE_CASE(0, 1) MMAE_PROBE_CASE(0, 2) MMAE_PROBE_CASE(0, 3)
        MMAE_PROBE_CASE(1, 0) MMAE_PROBE_CASE(1, 1) MMAE_PROBE_CASE(1, 2) MMAE_PROBE_CASE(1, 3)
        MMAE_PROBE_CASE(2, 0) MMAE_PROBE_CASE(2, 1) MMAE_PROBE_CASE(2, 2) MMAE_PROBE_CASE(2, 3)
        MMAE_PROBE_CASE(3, 0) MMAE_PROBE_CASE(3, 1) MMAE_PROBE_CASE(3, 2) MMAE_PROBE_CASE(3, 3)
    }
#undef MMAE_PROBE_CASE
    return mmae_check_launch("probe_mx_mfma");
}

}  // extern "C"
