// bf16 MFMA ping-pong GEMM: instantiations with the epilogue flavour fixed at compile time (gemm_common.h, gemm_flavour()).
// Only the flavours the MultiMAE step actually launches: forward Linear layers (bias -> bf16 / bias + GELU -> bf16 x 2 / bias
// [+ residual] -> f32) and their dX products (bf16, bf16 x dGELU [+ column sums], f32).  Anything else returns MMAE_ESUPPORT
// and the caller launches the generic kernel.
#include "gemm_pp_body.h"

// -DMMAE_NO_KF builds the same flavours on the general address walk (A/B library: make NOKF=1, MMAE_LIB=...)
#ifdef MMAE_NO_KF
constexpr bool KFV = false;
#else
constexpr bool KFV = true;
#endif

int mmae_gemm_bf16_pp_fl_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, int fl, hipStream_t st) {
    const bool bks = d->b_trans != 0;
    const bool t10 = code == 10;
    if (g.h16 && !d->a_trans && bks && fl == FL_BF16 && (g.K & 31))    // out_proj's dX of the semseg adapter contracts over C * ph * pw = 2128 columns:
        return launch<4, false, true, FL_BF16, false, true>(g, d->batch, st);   // the general address walk (zero-filled K tail)
    if (d->a_trans || (KFV && (g.K & 31))) return MMAE_ESUPPORT;       // these instantiations carry the K % 32 == 0 address walk (KF)
    if (g.ln_out) {                               // LayerNorm / cast side output: the f32 flavours of the 256 x 256 tile, N == 256 (runtime.hip checks)
        if (bks || t10 || g.N != 256) return MMAE_ESUPPORT;
        if (g.h16) {
            if (fl == FL_F32_BIAS_RESID) return launch<4, false, false, FL_F32_BIAS_RESID, KFV, true, true>(g, d->batch, st);
            if (fl == FL_F32_BIAS) return launch<4, false, false, FL_F32_BIAS, KFV, true, true>(g, d->batch, st);
            return MMAE_ESUPPORT;
        }
        if (fl == FL_F32_BIAS_RESID) return launch<4, false, false, FL_F32_BIAS_RESID, KFV, false, true>(g, d->batch, st);
        if (fl == FL_F32_BIAS) return launch<4, false, false, FL_F32_BIAS, KFV, false, true>(g, d->batch, st);
        return MMAE_ESUPPORT;
    }
    if (g.h16) {                                  // fp16 storage (MMAE_F16): the 256 x 256 tile, the flavours an output adapter launches
        if (t10) return MMAE_ESUPPORT;
        if (!bks) {
            switch (fl) {
                case FL_BF16_BIAS: return launch<4, false, false, FL_BF16_BIAS, KFV, true>(g, d->batch, st);
                case FL_BF16_BIAS_GELU: return launch<4, false, false, FL_BF16_BIAS_GELU, KFV, true>(g, d->batch, st);
                case FL_F32_BIAS_RESID: return launch<4, false, false, FL_F32_BIAS_RESID, KFV, true>(g, d->batch, st);
                case FL_F32_BIAS: return launch<4, false, false, FL_F32_BIAS, KFV, true>(g, d->batch, st);
                default: return MMAE_ESUPPORT;
            }
        }
        switch (fl) {
            case FL_BF16: return launch<4, false, true, FL_BF16, KFV, true>(g, d->batch, st);
            case FL_BF16_DGELU_CS: return launch<4, false, true, FL_BF16_DGELU_CS, KFV, true>(g, d->batch, st);
            case FL_BF16_DGELU: return launch<4, false, true, FL_BF16_DGELU, KFV, true>(g, d->batch, st);
            case FL_F32: return launch<4, false, true, FL_F32, KFV, true>(g, d->batch, st);
            default: return MMAE_ESUPPORT;
        }
    }
    if (!bks) {                                   // forward products: A [M][K], W [N][K]
        if (t10) {
            switch (fl) {
                case FL_BF16_BIAS: return launch<5, false, false, FL_BF16_BIAS, KFV>(g, d->batch, st);
                case FL_BF16_BIAS_GELU: return launch<5, false, false, FL_BF16_BIAS_GELU, KFV>(g, d->batch, st);
                case FL_F32_BIAS_RESID: return launch<5, false, false, FL_F32_BIAS_RESID, KFV>(g, d->batch, st);
                case FL_F32_BIAS: return launch<5, false, false, FL_F32_BIAS, KFV>(g, d->batch, st);
                default: return MMAE_ESUPPORT;
            }
        }
        switch (fl) {
            case FL_BF16_BIAS: return launch<4, false, false, FL_BF16_BIAS, KFV>(g, d->batch, st);
            case FL_BF16_BIAS_GELU: return launch<4, false, false, FL_BF16_BIAS_GELU, KFV>(g, d->batch, st);
            case FL_F32_BIAS_RESID: return launch<4, false, false, FL_F32_BIAS_RESID, KFV>(g, d->batch, st);
            case FL_F32_BIAS: return launch<4, false, false, FL_F32_BIAS, KFV>(g, d->batch, st);
            default: return MMAE_ESUPPORT;
        }
    }
    if (t10) {                                    // dX products: dY [M][N], W [N][K] read through the transposing LDS path
        switch (fl) {
            case FL_BF16: return launch<5, false, true, FL_BF16, KFV>(g, d->batch, st);
            case FL_BF16_DGELU_CS: return launch<5, false, true, FL_BF16_DGELU_CS, KFV>(g, d->batch, st);
            case FL_BF16_DGELU: return launch<5, false, true, FL_BF16_DGELU, KFV>(g, d->batch, st);
            case FL_F32: return launch<5, false, true, FL_F32, KFV>(g, d->batch, st);
            default: return MMAE_ESUPPORT;
        }
    }
    switch (fl) {
        case FL_BF16: return launch<4, false, true, FL_BF16, KFV>(g, d->batch, st);
        case FL_BF16_DGELU_CS: return launch<4, false, true, FL_BF16_DGELU_CS, KFV>(g, d->batch, st);
        case FL_BF16_DGELU: return launch<4, false, true, FL_BF16_DGELU, KFV>(g, d->batch, st);
        case FL_F32: return launch<4, false, true, FL_F32, KFV>(g, d->batch, st);
        default: return MMAE_ESUPPORT;
    }
}

#ifdef MMAE_PP_TRACE
// phase stamps of the last flavoured ping-pong launch (workgroup 0, waves 0 and 4): tools/pp_trace.py
extern "C" int mmae_debug_pp_trace(long long* out_host_256) { return (int)hipMemcpyFromSymbol(out_host_256, HIP_SYMBOL(g_pp_trace), 2 * 128 * 8); }
extern "C" int mmae_debug_pp_wg(long long* out_host_4096) { return (int)hipMemcpyFromSymbol(out_host_4096, HIP_SYMBOL(g_pp_wg), 1024 * 4 * 8); }
#endif
