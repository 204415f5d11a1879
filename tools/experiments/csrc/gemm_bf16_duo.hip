// bf16 MFMA "duo" GEMM (gemm_duo_body.h): 4-wave workgroups on 128 x 256 tiles, two resident per CU, with the epilogue flavour
// fixed at compile time (gemm_common.h, gemm_flavour()).  Instantiated for the products of the ViT step: forward Linear
// layers (A [M][K] x W [N][K]) and their dX products (dY [M][N] x W [N][K] through the transposing LDS read).
// Replaces nn.Linear forward / backward-input under autocast: multimae/multimae_utils.py:138-155 (Mlp), 158-182 (Attention
// qkv / proj), 217-232 (Block) of the reference.
#include <mutex>
#ifdef MMAE_EXPERIMENTS
// phase trace (experiments build): workgroup b, tile t, event e (0 loop start, 1 loop end, 2 epilogue end) -> 100 MHz time stamp
__device__ unsigned g_duo_trace[2048 * 8 * 3];
#define DUO_TRACE(tile_no, what) do { if (threadIdx.x == 0 && (tile_no) < 8 && blockIdx.x < 2048 && blockIdx.y == 0 && blockIdx.z == 0) \
    g_duo_trace[(blockIdx.x * 8 + (tile_no)) * 3 + (what)] = (unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#include "gemm_duo_body.h"

namespace {

#ifdef MMAE_EXPERIMENTS
// residency census (experiments build): per workgroup {HW_ID, XCC_ID, start, end} with the 100 MHz real-time counter
__device__ unsigned g_duo_census[4 * 2048];
#endif

template <int NW, bool AKS, bool BKS, int FL, int VAR = 0>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) gemm_bf16_duo_kernel(const GemmArgs g) {
#ifdef MMAE_EXPERIMENTS
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#endif
    duo_body<NW, AKS, BKS, FL, true, VAR>(g, blockIdx.x, gridDim.x);
#ifdef MMAE_EXPERIMENTS
    if (threadIdx.x == 0 && blockIdx.x < 2048 && blockIdx.y == 0 && blockIdx.z == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        unsigned* c = g_duo_census + 4 * blockIdx.x;
        c[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        c[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        c[2] = (unsigned)t0; c[3] = (unsigned)t1;
    }
#endif
}

int duo_cu_count() { return mmae_cu_count(); }

template <int NW, bool AKS, bool BKS, int FL, int VAR = 0>
int duo_launch(const GemmArgs& g, int batch, hipStream_t st) {
    constexpr int BM = (NW / 4) * 128, BN = 256, NST = NW == 4 ? 3 : 4;
    constexpr size_t LDS = (size_t)NST * (BM + BN) * 64;
    const int tiles_m = (g.M + BM - 1) / BM;
    GemmArgs a = g;
    a.tiles_n = (g.N + BN - 1) / BN;
    a.kt_per_split = g.kt_per_split * 2;                 // runtime.hip counts 64-wide K tiles; this kernel steps by 32
    a.tiles_total = tiles_m * a.tiles_n;
    const int slots = duo_cu_count() * (NW == 4 ? 2 : 1);       // resident workgroups: each walks the tile list
    const int gx = a.tiles_total > slots ? slots : a.tiles_total;
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_duo_kernel<NW, AKS, BKS, FL, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    });
    hipLaunchKernelGGL((gemm_bf16_duo_kernel<NW, AKS, BKS, FL, VAR>), dim3(gx, batch, a.splitk), dim3(NW * 64), LDS, st, a);
    return mmae_check_launch("gemm_bf16_duo");
}

template <int NW, int VAR = 0>
int duo_dispatch(const mmae_gemm_desc* d, const GemmArgs& g, int fl, hipStream_t st) {
    if (!d->b_trans) {
        switch (fl) {
            case FL_BF16_BIAS: return duo_launch<NW, false, false, FL_BF16_BIAS, VAR>(g, d->batch, st);
            case FL_BF16_BIAS_GELU: return duo_launch<NW, false, false, FL_BF16_BIAS_GELU, VAR>(g, d->batch, st);
            case FL_F32_BIAS_RESID: return duo_launch<NW, false, false, FL_F32_BIAS_RESID, VAR>(g, d->batch, st);
            case FL_F32_BIAS: return duo_launch<NW, false, false, FL_F32_BIAS, VAR>(g, d->batch, st);
            default: return MMAE_ESUPPORT;
        }
    }
    switch (fl) {
        case FL_BF16: return duo_launch<NW, false, true, FL_BF16, VAR>(g, d->batch, st);
        case FL_BF16_DGELU_CS: return duo_launch<NW, false, true, FL_BF16_DGELU_CS, VAR>(g, d->batch, st);
        case FL_BF16_DGELU: return duo_launch<NW, false, true, FL_BF16_DGELU, VAR>(g, d->batch, st);
        case FL_F32: return duo_launch<NW, false, true, FL_F32, VAR>(g, d->batch, st);
        default: return MMAE_ESUPPORT;
    }
}

}  // namespace

// tile codes: 11 = 128 x 256, two workgroups per CU; 12 = 256 x 256, one 8-wave workgroup per CU on the same schedule.
// MMAE_ESUPPORT: not one of the instantiated flavours / shapes -- the caller falls back to the ping-pong kernel.
int mmae_gemm_bf16_duo_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st) {
    if (d->a_trans || (g.K & 31) || g.splitk > 1) return MMAE_ESUPPORT;
    const int fl = gemm_flavour(g, d->batch);
    if (!fl) return MMAE_ESUPPORT;
    switch (code) {
        case 11: return duo_dispatch<4>(d, g, fl, st);
        case 12: return duo_dispatch<8>(d, g, fl, st);
#ifdef MMAE_EXPERIMENTS
        // schedule experiments / dissection builds (make EXTRA=-DMMAE_EXPERIMENTS): code = 11 | 12 + 10 x VAR
        case 21: return duo_dispatch<4, 1>(d, g, fl, st);
        case 22: return duo_dispatch<8, 1>(d, g, fl, st);
        case 31: return duo_dispatch<4, 2>(d, g, fl, st);
        case 32: return duo_dispatch<8, 2>(d, g, fl, st);
        case 41: return duo_dispatch<4, 3>(d, g, fl, st);
        case 42: return duo_dispatch<8, 3>(d, g, fl, st);
        case 51: return duo_dispatch<4, 4>(d, g, fl, st);
        case 52: return duo_dispatch<8, 4>(d, g, fl, st);
#endif
        default: return MMAE_ESUPPORT;
    }
}

extern "C" int mmae_gemm_duo_occupancy(int tile) {
    int n = -1;
    hipError_t e;
    if (tile == 12) {
        constexpr size_t LDS = (size_t)4 * (256 + 256) * 64;
        (void)hipFuncSetAttribute((const void*)gemm_bf16_duo_kernel<8, false, false, FL_BF16_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_bf16_duo_kernel<8, false, false, FL_BF16_BIAS>, 512, LDS);
    } else {
        constexpr size_t LDS = (size_t)3 * (128 + 256) * 64;
        (void)hipFuncSetAttribute((const void*)gemm_bf16_duo_kernel<4, false, false, FL_BF16_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_bf16_duo_kernel<4, false, false, FL_BF16_BIAS>, 256, LDS);
    }
    if (e != hipSuccess) { mmae_set_error(hipGetErrorString(e)); return -1; }
    return n;
}

#ifdef MMAE_EXPERIMENTS
extern "C" int mmae_debug_duo_census(unsigned* out_host, int n_words) {
    if (n_words > 4 * 2048) n_words = 4 * 2048;
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_duo_census), (size_t)n_words * 4) == hipSuccess ? 0 : -1;
}
extern "C" int mmae_debug_duo_trace(unsigned* out_host, int n_words) {
    if (n_words > 2048 * 8 * 3) n_words = 2048 * 8 * 3;
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_duo_trace), (size_t)n_words * 4) == hipSuccess ? 0 : -1;
}
#endif
