// Exact-f32 MFMA GEMM (parity mode):  C[M,N] (+)= alpha * A[M,K] . B[N,K]^T (+ epilogue)
//
// v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain (MI355X guide, section 3), so
// this path reproduces an fp32 CPU reference up to summation order.  The f32 MFMA takes ONE
// f32 per lane per operand, so any operand storage order works with plain ds_read_b32:
// both operands are generic (row stride, k stride) views.  128x128 tile, 4 waves, BK = 16.
#include "gemm_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, NT = 256, PAD = 4;
constexpr int LDT = BM + PAD;                 // floats per k-row of a tile in LDS
constexpr int EPT = BM * BK / NT;             // elements per thread per operand tile (8)

__global__ void __launch_bounds__(NT) gemm_f32_kernel(const GemmArgs g, const long long sa_m, const long long sa_k,
                                                      const long long sb_n, const long long sb_k) {
    __shared__ float As[2][BK][LDT];
    __shared__ float Bs[2][BK][LDT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x;
    const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;
    const int z = blockIdx.y, zo = z / g.nb_inner, zi = z % g.nb_inner;
    const float* Az = (const float*)g.A + zo * g.sAo + zi * g.sAi;
    const float* Bz = (const float*)g.B + zo * g.sBo + zi * g.sBi;
    char* Cz = (char*)g.C + (zo * g.sCo + zi * g.sCi) * (g.c_f32 ? 4 : 2);

    // thread -> (row, k) mapping with the contiguous direction on consecutive lanes
    const bool a_kfast = (sa_k == 1), b_kfast = (sb_k == 1);
    int a_r[EPT], a_k[EPT], b_r[EPT], b_k[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * NT;
        a_r[i] = a_kfast ? (e / BK) : (e % BM);
        a_k[i] = a_kfast ? (e % BK) : (e / BM);
        b_r[i] = b_kfast ? (e / BK) : (e % BN);
        b_k[i] = b_kfast ? (e % BK) : (e / BN);
    }
    float ra[EPT], rb[EPT];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int m = m0 + a_r[i], k = k0 + a_k[i];
            ra[i] = (m < g.M && k < g.K) ? Az[m * sa_m + k * sa_k] : 0.f;
            const int n = n0 + b_r[i], kb = k0 + b_k[i];
            rb[i] = (n < g.N && kb < g.K) ? Bz[n * sb_n + kb * sb_k] : 0.f;
        }
    };
    auto lstore = [&](int s) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) { As[s][a_k[i]][a_r[i]] = ra[i]; Bs[s][b_k[i]][b_r[i]] = rb[i]; }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int fr = lane & 31, fk = lane >> 5;
    const int nkt_all = (g.K + BK - 1) / BK;
    const int kt_begin = blockIdx.z * g.kt_per_split;
    const int nkt = (kt_begin + g.kt_per_split < nkt_all) ? kt_begin + g.kt_per_split : nkt_all;
    gload(kt_begin); lstore(0); __syncthreads();
    for (int kt = kt_begin; kt < nkt; ++kt) {
        if (kt + 1 < nkt) gload(kt + 1);
        const int s = (kt - kt_begin) & 1;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[t] = As[s][kk * 2 + fk][wm * 64 + t * 32 + fr];
                bv[t] = Bs[s][kk * 2 + fk][wn * 64 + t * 32 + fr];
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[tn], av[tm], acc[tn][tm], 0, 0, 0);
        }
        if (kt + 1 < nkt) lstore((kt + 1 - kt_begin) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int m = m0 + wm * 64 + tm * 32 + (lane & 31);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = n0 + wn * 64 + tn * 32 + rg * 8 + (lane >> 5) * 4;
                f32x4 v = {acc[tn][tm][rg * 4 + 0], acc[tn][tm][rg * 4 + 1], acc[tn][tm][rg * 4 + 2], acc[tn][tm][rg * 4 + 3]};
                gemm_epilogue4(g, Cz, m, n, v);
            }
    }
}

}  // namespace

int mmae_gemm_f32_impl(const mmae_gemm_desc* d, const GemmArgs& g0, hipStream_t st) {
    GemmArgs g = g0;
    g.tiles_n = (d->N + BN - 1) / BN;
    const int tiles_m = (d->M + BM - 1) / BM;
    const long long sa_m = d->a_trans ? 1 : d->lda, sa_k = d->a_trans ? d->lda : 1;
    const long long sb_n = d->b_trans ? 1 : d->ldb, sb_k = d->b_trans ? d->ldb : 1;
    dim3 grid(tiles_m * g.tiles_n, d->batch, g.splitk), block(NT);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, block, 0, st, g, sa_m, sa_k, sb_n, sb_k);
    return mmae_check_launch("gemm_f32");
}
