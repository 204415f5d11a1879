// bf16 MFMA ping-pong GEMM on 64-wide K tiles (gemm_pp64_body.h: whole-cache-line LDS-DMA pieces), epilogue flavour fixed at
// compile time.  Forward Linear layers (A [M][K] x W [N][K]) and their dX products (dY [M][N] x W [N][K] read through the
// transposing LDS path).  Replaces nn.Linear forward / backward-input of the reference's Block: multimae/multimae_utils.py:138-155,
// 158-182, 217-232.
#include <mutex>
#include "gemm_pp64_body.h"

namespace {

template <int TM, bool AKS, bool BKS, int FL, int VAR = 0>
__global__ void __launch_bounds__(512) gemm_bf16_pp64_kernel(const GemmArgs g) {
    pp64_body<TM, AKS, BKS, FL, VAR % 10, (VAR >= 10)>(g, blockIdx.x, gridDim.x);      // VAR >= 10: with the L2 prefetch (measured slower)
}

template <int TM, bool AKS, bool BKS, int FL, int VAR = 0>
int launch64(const GemmArgs& g, int batch, hipStream_t st) {
    constexpr int BM = TM * 64, BN = 256;
    constexpr size_t LDS = (size_t)2 * (BM + BN) * 128;
    const int tiles_m = (g.M + BM - 1) / BM;
    GemmArgs a = g;
    a.tiles_n = (g.N + BN - 1) / BN;
    a.tiles_total = tiles_m * a.tiles_n;                 // kt_per_split already counts 64-wide K tiles (runtime.hip)
    const int n_cu = mmae_cu_count();
    const int gx = a.tiles_total > n_cu ? n_cu : a.tiles_total;      // one resident workgroup per CU walks the tile list
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp64_kernel<TM, AKS, BKS, FL, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    });
    hipLaunchKernelGGL((gemm_bf16_pp64_kernel<TM, AKS, BKS, FL, VAR>), dim3(gx, batch, a.splitk), dim3(512), LDS, st, a);
    return mmae_check_launch("gemm_bf16_pp64");
}

template <int TM, int VAR = 0>
int dispatch64(const mmae_gemm_desc* d, const GemmArgs& g, int fl, hipStream_t st) {
    if (!d->b_trans) {
        switch (fl) {
            case FL_BF16_BIAS: return launch64<TM, false, false, FL_BF16_BIAS, VAR>(g, d->batch, st);
            case FL_BF16_BIAS_GELU: return launch64<TM, false, false, FL_BF16_BIAS_GELU, VAR>(g, d->batch, st);
            case FL_F32_BIAS_RESID: return launch64<TM, false, false, FL_F32_BIAS_RESID, VAR>(g, d->batch, st);
            case FL_F32_BIAS: return launch64<TM, false, false, FL_F32_BIAS, VAR>(g, d->batch, st);
            default: return MMAE_ESUPPORT;
        }
    }
    switch (fl) {
        case FL_BF16: return launch64<TM, false, true, FL_BF16, VAR>(g, d->batch, st);
        case FL_BF16_DGELU_CS: return launch64<TM, false, true, FL_BF16_DGELU_CS, VAR>(g, d->batch, st);
        case FL_BF16_DGELU: return launch64<TM, false, true, FL_BF16_DGELU, VAR>(g, d->batch, st);
        case FL_F32: return launch64<TM, false, true, FL_F32, VAR>(g, d->batch, st);
        default: return MMAE_ESUPPORT;
    }
}

}  // namespace

// tile codes: 13 = 256 x 256, 14 = 320 x 256 on 64-wide K tiles.  MMAE_ESUPPORT: not one of the instantiated flavours / shapes
// (K % 64, k-strided A, split-K, batched): the caller falls back to the 32-wide-K-tile kernel (codes 9 / 10).
int mmae_gemm_bf16_pp64_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st) {
    if (d->a_trans || (g.K & 63) || (g.M & 7) || (g.N & 7) || g.splitk > 1) return MMAE_ESUPPORT;
    const int fl = gemm_flavour(g, d->batch);
    if (!fl) return MMAE_ESUPPORT;
    switch (code) {
        case 13: return dispatch64<4>(d, g, fl, st);
        case 14: return dispatch64<5>(d, g, fl, st);
#ifdef MMAE_EXPERIMENTS
        case 64: return dispatch64<5, 10>(d, g, fl, st);      // production schedule + L2 prefetch three K tiles ahead
        case 74: return dispatch64<5, 15>(d, g, fl, st);
        case 84: return dispatch64<5, 16>(d, g, fl, st);
        case 24: return dispatch64<5, 1>(d, g, fl, st);       // dissection builds: code = 14 + 10 x VAR
        case 34: return dispatch64<5, 2>(d, g, fl, st);
        case 44: return dispatch64<5, 3>(d, g, fl, st);
        case 54: return dispatch64<5, 4>(d, g, fl, st);
#endif
        default: return MMAE_ESUPPORT;
    }
}

int mmae_gemm_bf16_duo_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st);

// codes 13 / 14 and their dissection builds (x4): this file; 11 / 12 and (x1, x2): the duo kernel
int mmae_gemm_bf16_ext_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st) {
    const int unit = code % 10;
    if (unit == 3 || unit == 4) return mmae_gemm_bf16_pp64_impl(d, g, code, st);
    if (unit == 1 || unit == 2) return mmae_gemm_bf16_duo_impl(d, g, code, st);
    return MMAE_ESUPPORT;
}
