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
    __shared__ __attribute__((aligned(16))) float lds_all[2 * 2 * BK * LDT];       // 33 KiB: operand ring, then epilogue staging
    float (*As)[BK][LDT] = reinterpret_cast<float (*)[BK][LDT]>(lds_all);
    float (*Bs)[BK][LDT] = reinterpret_cast<float (*)[BK][LDT]>(lds_all + 2 * BK * LDT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x;
    const int z = blockIdx.y, zo = z / g.nb_inner, zi = z % g.nb_inner;
    const float* Az = (const float*)g.A + zo * g.sAo + zi * g.sAi;
    const float* Bz = (const float*)g.B + zo * g.sBo + zi * g.sBi;
    char* Cz = (char*)g.C + (zo * g.sCo + zi * g.sCi) * (g.c_f32 ? 4 : 2);

    // thread -> (row, k) mapping with the contiguous direction on consecutive lanes.  When the contiguous
    // extent and the strides are multiples of 4 (g.vec bits, checked on the host) operands are fetched as
    // float4 (2 loads per operand per K tile instead of 8 scalar ones).
    const bool a_kfast = (sa_k == 1), b_kfast = (sb_k == 1);
    const bool a_vec = (g.tiles_n >> 30) & 1, b_vec = (g.tiles_n >> 29) & 1;
    const int tiles_n = g.tiles_n & 0x1fffffff;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    int a_r[EPT], a_k[EPT], b_r[EPT], b_k[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * NT;
        a_r[i] = a_kfast ? (e / BK) : (e % BM);
        a_k[i] = a_kfast ? (e % BK) : (e / BM);
        b_r[i] = b_kfast ? (e / BK) : (e % BN);
        b_k[i] = b_kfast ? (e % BK) : (e / BN);
    }
    // float4 mapping: 2 groups per thread; group e4 = tid + i*NT covers 4 consecutive elements along the
    // contiguous direction
    int a4_r[2], a4_k[2], b4_r[2], b4_k[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e4 = tid + i * NT;
        a4_r[i] = a_kfast ? (e4 / (BK / 4)) : ((e4 % (BM / 4)) * 4);
        a4_k[i] = a_kfast ? ((e4 % (BK / 4)) * 4) : (e4 / (BM / 4));
        b4_r[i] = b_kfast ? (e4 / (BK / 4)) : ((e4 % (BN / 4)) * 4);
        b4_k[i] = b_kfast ? ((e4 % (BK / 4)) * 4) : (e4 / (BN / 4));
    }
    float ra[EPT], rb[EPT];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
        if (a_vec) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = m0 + a4_r[i], k = k0 + a4_k[i];
                const f32x4 v = (m < g.M && k < g.K) ? ld4(Az + m * sa_m + k * sa_k) : f32x4{0.f, 0.f, 0.f, 0.f};
                ra[4 * i] = v[0]; ra[4 * i + 1] = v[1]; ra[4 * i + 2] = v[2]; ra[4 * i + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int m = m0 + a_r[i], k = k0 + a_k[i];
                ra[i] = (m < g.M && k < g.K) ? Az[m * sa_m + k * sa_k] : 0.f;
            }
        }
        if (b_vec) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = n0 + b4_r[i], k = k0 + b4_k[i];
                const f32x4 v = (n < g.N && k < g.K) ? ld4(Bz + n * sb_n + k * sb_k) : f32x4{0.f, 0.f, 0.f, 0.f};
                rb[4 * i] = v[0]; rb[4 * i + 1] = v[1]; rb[4 * i + 2] = v[2]; rb[4 * i + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int n = n0 + b_r[i], kb = k0 + b_k[i];
                rb[i] = (n < g.N && kb < g.K) ? Bz[n * sb_n + kb * sb_k] : 0.f;
            }
        }
    };
    auto lstore = [&](int s) {
        if (a_vec) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (a_kfast) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) As[s][a4_k[i] + j][a4_r[i]] = ra[4 * i + j];
                } else {
                    f32x4 v = {ra[4 * i], ra[4 * i + 1], ra[4 * i + 2], ra[4 * i + 3]};
                    *reinterpret_cast<f32x4*>(&As[s][a4_k[i]][a4_r[i]]) = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) As[s][a_k[i]][a_r[i]] = ra[i];
        }
        if (b_vec) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (b_kfast) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) Bs[s][b4_k[i] + j][b4_r[i]] = rb[4 * i + j];
                } else {
                    f32x4 v = {rb[4 * i], rb[4 * i + 1], rb[4 * i + 2], rb[4 * i + 3]};
                    *reinterpret_cast<f32x4*>(&Bs[s][b4_k[i]][b4_r[i]]) = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) Bs[s][b_k[i]][b_r[i]] = rb[i];
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
    static_assert(sizeof(float) * 2 * 2 * BK * LDT >= 4 * 8192, "epilogue staging needs 8 KiB per wave");
    gemm_store_tile64(g, Cz, (char*)lds_all + wave * 8192, lane, acc, m0 + wm * 64, n0 + wn * 64);   // (loop ended on a barrier)
}

}  // namespace

int mmae_gemm_f32_impl(const mmae_gemm_desc* d, const GemmArgs& g0, hipStream_t st) {
    GemmArgs g = g0;
    const int tiles_n = (d->N + BN - 1) / BN;
    const int tiles_m = (d->M + BM - 1) / BM;
    const long long sa_m = d->a_trans ? 1 : d->lda, sa_k = d->a_trans ? d->lda : 1;
    const long long sb_n = d->b_trans ? 1 : d->ldb, sb_k = d->b_trans ? d->ldb : 1;
    // float4 operand loads: contiguous extent, leading dimension, batch strides and base all 4-element aligned
    const bool a_vec = (d->lda % 4 == 0) && ((d->a_trans ? d->M : d->K) % 4 == 0) && ((uintptr_t)d->A % 16 == 0) &&
                       (d->sA_outer % 4 == 0) && (d->sA_inner % 4 == 0);
    const bool b_vec = (d->ldb % 4 == 0) && ((d->b_trans ? d->N : d->K) % 4 == 0) && ((uintptr_t)d->B % 16 == 0) &&
                       (d->sB_outer % 4 == 0) && (d->sB_inner % 4 == 0);
    g.tiles_n = tiles_n | (a_vec ? (1 << 30) : 0) | (b_vec ? (1 << 29) : 0);
    dim3 grid(tiles_m * tiles_n, d->batch, g.splitk), block(NT);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, block, 0, st, g, sa_m, sa_k, sb_n, sb_k);
    return mmae_check_launch("gemm_f32");
}
