// ABI bookkeeping (version, per-thread error string) and the GEMM entry point's argument
// validation / dtype dispatch.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "gemm_common.h"

static thread_local char g_err[256] = "";

void mmae_set_error(const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

int mmae_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

// compute units the persistent GEMM grids leave free (mmae_gemm_cu_reserve): a launch POLICY read on the host when a grid is sized
// Two independent requesters, one slot each (a save / restore of ONE shared value let an enclosing context wipe what the
// gradient reducer had set mid-backward): `reserve` belongs to the data-parallel gradient exchange (dist.GradAllReducer),
// `share` to the output adapters' side-by-side experiment (functions._adapter_cu_share); the grids leave max(reserve, share) free.
static std::atomic<int> g_cu_reserve{0};
static std::atomic<int> g_cu_share{0};
static std::atomic<int> g_side_cus{0};
int mmae_cu_avail() {
    const int r = g_cu_reserve.load(std::memory_order_relaxed), sh = g_cu_share.load(std::memory_order_relaxed);
    const int n = mmae_cu_count(), k = (r > sh ? r : sh) + g_side_cus.load(std::memory_order_relaxed);
    const int a = n - k;
    return a < 16 ? (n < 16 ? n : 16) : a;
}
// experiment (mmae_gemm_side_cus): k > 0 splits the chip between the compute stream's persistent GEMM grids (n_cu - reserve - k workgroups)
// and the grouped weight-gradient launches of the side stream (sized for k CUs), so that both are resident at once
int mmae_cu_side() { return g_side_cus.load(std::memory_order_relaxed); }
extern "C" int mmae_gemm_side_cus(int k) {
    const int prev = g_side_cus.load(std::memory_order_relaxed);
    if (k >= 0) g_side_cus.store(k, std::memory_order_relaxed);
    return prev;
}
extern "C" int mmae_gemm_cu_reserve(int k) {
    const int prev = g_cu_reserve.load(std::memory_order_relaxed);
    if (k >= 0) g_cu_reserve.store(k, std::memory_order_relaxed);
    return prev;
}
extern "C" int mmae_gemm_cu_share(int k) {
    const int prev = g_cu_share.load(std::memory_order_relaxed);
    if (k >= 0) g_cu_share.store(k, std::memory_order_relaxed);
    return prev;
}

int mmae_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return MMAE_ELAUNCH;
    }
    return 0;
}

int mmae_gemm_bf16_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st);
int mmae_gemm_bf16_pp_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st);
int mmae_gemm_f32_impl(const mmae_gemm_desc* d, const GemmArgs& g, hipStream_t st);
int mmae_gemm_f32x3_impl(const mmae_gemm_desc* d, const GemmArgs& g, hipStream_t st);
int mmae_gemm_mxfp8_impl(const mmae_gemm_desc* d, const GemmArgs& g, hipStream_t st);
int mmae_splitk_reduce(const float* ws, float* C, int M, int N, long long ldc, int splits, int accumulate, hipStream_t st);
int mmae_acs_reduce(const float* part, int splits, int M, float* out, int accumulate, hipStream_t st);

// ---- optional launch timing (mmae_gemm_timing_*) ----------------------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
struct TimedLaunch { hipEvent_t a, b; double flop; int cls; double bytes; };
std::mutex g_tmu;
bool g_timing = false;
std::vector<TimedLaunch> g_timed;
}  // namespace

hipEvent_t mmae_timing_begin(hipStream_t st) {
    if (!g_timing) return nullptr;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    (void)hipEventRecord(e, st);
    return e;
}
// bytes: the launch's ALGORITHMIC HBM bytes (operands once, outputs / epilogue streams once) -- the figure the counter traffic is read against
void mmae_timing_end(hipEvent_t a, hipStream_t st, double flop, int cls, double bytes) {
    if (!a) return;
    hipEvent_t b = nullptr;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
    (void)hipEventRecord(b, st);
    std::lock_guard<std::mutex> lk(g_tmu);
    g_timed.push_back({a, b, flop, cls, bytes});
}

namespace {

bool split_eligible(const mmae_gemm_desc* d) {
    return d->c_dtype == MMAE_F32 && d->epi == MMAE_EPI_NONE && !d->bias && !d->resid && d->batch == 1 && d->alpha == 1.0f &&
           d->N % 4 == 0 && d->ldc % 4 == 0 && !d->colsum_part;
}

// Kernel variant + split-K choice for one product (256 CUs; the ping-pong kernel holds one workgroup per CU, the
// 128 x 128 kernels about three).
void gemm_plan(const mmae_gemm_desc* d, int* tile_out, int* split_out) {
    static const int env_tile = mmae_env_int("MMAE_GEMM_TILE", 0);
    const int bk = d->ab_dtype == MMAE_F32 ? 16 : 64;
    const int nkt = (d->K + bk - 1) / bk;
    const bool can_split = split_eligible(d) && nkt >= 32;
    int tile = d->tile ? d->tile : env_tile;
    const long long nt256 = (d->N + 255) / 256;
    if (tile == 0) {
        tile = 3;
        if (d->ab_dtype == MMAE_BF16 && d->batch == 1) {
            const long long t4 = ((d->M + 255) / 256) * nt256, t5 = ((d->M + 319) / 320) * nt256;
            static const int env_min_k = mmae_env_int("MMAE_PP_MIN_K", 128);
            if (d->M >= 2048 && d->N >= 192 && d->K >= env_min_k) {
                // whole rounds of 256 workgroups x rows per tile: 320-row tiles when they waste less of the last round
                const long long c4 = ((t4 + 255) / 256) * 256, c5 = ((t5 + 255) / 256) * 320;
                static const int env_t10cs = mmae_env_int("MMAE_GEMM_T10_CS", 1);     // 0: 256-row tiles for the column-sum epilogue (A/B)
                tile = (c5 <= c4 && !d->a_trans && (env_t10cs || !d->colsum_part)) ? 10 : 9;
                // 64-wide K tiles (whole-cache-line DMA pieces, gemm_pp64_body.h): in isolation the products with a plain bf16
                // epilogue run 3-6 % faster on them (qkv forward, dX of fc1 / qkv / proj, profiles/r03_pp64_ab.txt), inside the
                // training step the same switch measured +0.2 ms (33.52 vs 33.32 ms, encoder step 18.24 vs 18.10): off.
                static const int env_pp64 = mmae_env_int("MMAE_PP64", 0);
                if (env_pp64 && tile == 10 && d->c_dtype == MMAE_BF16 && d->epi == MMAE_EPI_NONE && !d->resid && !d->accumulate &&
                    d->alpha == 1.0f && (d->K % 64) == 0 && (d->M % 8) == 0 && (d->N % 8) == 0 && d->split_k <= 1)
                    tile = 14;
            } else if (t4 >= 4 && can_split && d->K >= 4096) {
                tile = 9;                                   // dW-shaped: few tiles, split along K below
            }
        }
    }
    // experiments: the two-workgroups-per-CU structure (tile 11) for short contractions (the output adapters' D = 256 products)
    static const int env_duo_k = mmae_env_int("MMAE_DUO_MAX_K", 0);
    if (env_duo_k && !d->tile && (tile == 9 || tile == 10) && d->K <= env_duo_k && !d->a_trans && d->split_k <= 1 && (d->K % 32) == 0) tile = 11;
    int s = 1;
    // workgroups a split product should reach (ping-pong kernel: one per CU).  Below 256 the dW launches leave CUs to the dX
    // chain they run beside, write fewer partial slabs and run longer K loops; tunable for experiments.
    static const int env_wgs = mmae_env_int("MMAE_SPLITK_WGS", 256);
    if (can_split) {
        if (tile == 9 || tile == 10) {
            const long long t = ((d->M + (tile == 9 ? 255 : 319)) / (tile == 9 ? 256 : 320)) * nt256;
            if (t * 2 <= env_wgs) s = (int)(env_wgs / t);
        } else {
            const long long t = (long long)((d->M + 127) / 128) * ((d->N + 127) / 128);
            if (t < 384) s = (int)((1024 + t - 1) / t);
        }
        const int max_split = nkt / 8;                      // keep >= 8 K tiles per slice
        if (s > max_split) s = max_split;
        if (s > 64) s = 64;
        if (s < 1) s = 1;
    }
    *tile_out = tile;
    *split_out = s;
}

}  // namespace

extern "C" {

int mmae_abi_version(void) { return MMAE_ABI_VERSION; }
int mmae_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(mmae_gemm_desc);
        case 1: return (int)sizeof(mmae_block_desc);
        case 2: return (int)sizeof(mmae_stack_desc);
        case 3: return (int)sizeof(mmae_adapter_desc);
        case 4: return (int)sizeof(mmae_opt_desc);
        case 5: return (int)sizeof(mmae_patch_src);
        case 6: return (int)sizeof(mmae_dw_group_desc);
        case 7: return (int)sizeof(mmae_colsum_job);
        default: return -1;
    }
}
const char* mmae_last_error(void) { return g_err; }

int mmae_gemm(const mmae_gemm_desc* d, void* stream) { return mmae_gemm_ex(d, stream, -1, 1.0); }

}  // extern "C"

// operands once, C once, every epilogue stream once (aux written or read, residual read, LayerNorm side output written)
static double gemm_algorithmic_bytes(const mmae_gemm_desc* d) {
    const double eab = (d->ab_dtype == MMAE_F32 || d->ab_dtype == MMAE_F32X3 || d->ab_dtype == MMAE_F32F16) ? 4.0 : (d->ab_dtype == MMAE_MXFP8 ? 1.0 : 2.0);
    const double ec = d->c_dtype == MMAE_F32 ? 4.0 : 2.0;
    const double mn = (double)d->M * d->N * d->batch;
    double b = ((double)d->M * d->K + (double)d->N * d->K) * d->batch * eab + mn * ec;
    if (d->accumulate) b += mn * ec;
    if (d->aux && d->epi != MMAE_EPI_NONE) b += mn * (d->aux_dtype == MMAE_F32 ? 4.0 : 2.0);
    if (d->resid) b += mn * 4.0;
    if (d->ln_out) b += mn * 2.0;
    return b;
}

// timing_cls >= 0 / flop_scale: how the launch is booked by mmae_gemm_timing_* (a pre-split x3 product is one bf16 launch over 3 K
// that belongs to the f32 / split class with a third of its MFMA work as algorithmic FLOPs)
int mmae_gemm_ex(const mmae_gemm_desc* d, void* stream, int timing_cls, double flop_scale) {
    MMAE_REQUIRE(d && d->A && d->B && d->C, "gemm: null operand");
    MMAE_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: empty problem");
    MMAE_REQUIRE(d->batch >= 1 && d->batch <= 65535 && d->batch_inner >= 1, "gemm: bad batch");
    MMAE_REQUIRE(d->ab_dtype == MMAE_F32 || d->ab_dtype == MMAE_BF16 || d->ab_dtype == MMAE_F32X3 || d->ab_dtype == MMAE_F32F16 || d->ab_dtype == MMAE_MXFP8 ||
                 d->ab_dtype == MMAE_F16, "gemm: bad ab_dtype");
    const bool h16 = d->ab_dtype == MMAE_F16;
    MMAE_REQUIRE(!d->a_amax || d->ab_dtype == MMAE_F32F16 || (h16 && d->c_dtype == MMAE_F32), "gemm: a_amax is an MMAE_F32F16 option (MMAE_F16: with an f32 C)");
    MMAE_REQUIRE(d->c_dtype == MMAE_F32 || d->c_dtype == (h16 ? MMAE_F16 : MMAE_BF16), "gemm: bad c_dtype (16-bit C: the operands' format)");
    MMAE_REQUIRE(!d->aux || d->aux_dtype == MMAE_F32 || d->aux_dtype == (h16 ? MMAE_F16 : MMAE_BF16), "gemm: bad aux_dtype (16-bit aux: the operands' format)");
    if (h16 && (d->batch != 1 || d->split_k > 1 || d->a_colsum || d->a_trans)) { mmae_set_error("gemm(f16): unbatched, unsplit products with k-contiguous A only"); return MMAE_ESUPPORT; }
    MMAE_REQUIRE(!(d->accumulate && d->c_dtype != MMAE_F32), "gemm: accumulate needs f32 C");
    MMAE_REQUIRE(!(d->epi != MMAE_EPI_NONE && !d->aux), "gemm: epilogue needs aux");
    MMAE_REQUIRE(!(d->resid && d->batch != 1), "gemm: residual is unbatched only");
    GemmArgs g;
    g.A = d->A; g.B = d->B; g.C = d->C;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
    g.nb_inner = d->batch_inner;
    g.sAo = d->sA_outer; g.sAi = d->sA_inner; g.sBo = d->sB_outer; g.sBi = d->sB_inner;
    g.sCo = d->sC_outer; g.sCi = d->sC_inner;
    g.bias = d->bias; g.resid = d->resid; g.ldr = d->ldr;
    g.aux = d->aux; g.ldaux = d->ldaux;
    g.c_f32 = d->c_dtype == MMAE_F32; g.aux_f32 = d->aux_dtype == MMAE_F32;
    MMAE_REQUIRE(d->epi >= MMAE_EPI_NONE && d->epi <= MMAE_EPI_MUL, "gemm: bad epi");
    // GELU_G / MUL are GELU / DGELU with aux holding the derivative: same kernels and flavours, one flag
    g.aux_grad = (d->epi == MMAE_EPI_GELU_G || d->epi == MMAE_EPI_MUL) ? 1 : 0;
    g.epi = d->epi == MMAE_EPI_GELU_G ? MMAE_EPI_GELU : (d->epi == MMAE_EPI_MUL ? MMAE_EPI_DGELU : d->epi);
    g.accumulate = d->accumulate; g.alpha = d->alpha;
    g.tiles_n = 0;
    g.colpart = d->colsum_part;
    static const int env_swz = mmae_env_int("MMAE_GEMM_XCD", 1);
    g.xcd_swizzle = env_swz;
    static const int env_wide = mmae_env_int("MMAE_EPI_WIDE", 1);
    g.wide_st = env_wide;
    static const int env_dbg = mmae_env_int("MMAE_EPI_DBG", 0);
    g.dbg = env_dbg;
    static const int env_dephase = mmae_env_int("MMAE_PP_DEPHASE", 0);
    g.dephase = env_dephase;
    static const int env_dephase_sel = mmae_env_int("MMAE_PP_DEPHASE_SEL", 0);     // 1: only the two-stream epilogues of the encoder (fc1 + GELU, fc2-dX x aux)
    if (env_dephase_sel == 1 && (d->epi == MMAE_EPI_NONE || d->K < 512)) g.dephase = 0;
    g.scA = d->a_scale; g.scB = d->b_scale;
    g.a_amax = d->a_amax;
    g.ln_g = d->ln_gamma; g.ln_b = d->ln_beta; g.ln_out = d->ln_out; g.ln_mean = d->ln_mean; g.ln_rstd = d->ln_rstd; g.ln_eps = d->ln_eps;
    if (d->ln_out) {
        const bool ok = (d->ab_dtype == MMAE_BF16 || h16) && d->c_dtype == MMAE_F32 && d->N == 256 && d->ldc == 256 && d->bias && d->epi == MMAE_EPI_NONE &&
                        !d->accumulate && d->batch == 1 && d->split_k <= 1 && !d->a_trans && !d->b_trans && (d->K % 32) == 0 && d->alpha == 1.0f &&
                        !d->a_amax && (!d->ln_gamma || (d->ln_beta && d->ln_mean && d->ln_rstd)) && ((uintptr_t)d->ln_out % 16) == 0 &&
                        (!d->ln_gamma || (((uintptr_t)d->ln_gamma | (uintptr_t)d->ln_beta) % 16) == 0);
        if (!ok) { mmae_set_error("gemm: ln_out needs a 16-bit x 16-bit -> f32 product with bias [+ residual], N = ldc = 256, unbatched, unsplit, k-contiguous operands, K % 32 == 0"); return MMAE_ESUPPORT; }
    }
    g.h16 = h16 ? 1 : 0;
    g.qout = (unsigned char*)d->q_out; g.qsc = (unsigned char*)d->q_scale; g.ldq = d->ldq;
    MMAE_REQUIRE(!d->q_out || d->ab_dtype == MMAE_MXFP8, "gemm: q_out is an MX-fp8 product option");
    MMAE_REQUIRE(!d->colsum_part || ((d->epi == MMAE_EPI_DGELU || d->epi == MMAE_EPI_MUL) && !d->bias && !d->resid && !d->accumulate && d->batch == 1 && d->split_k <= 1 &&
                                     d->alpha == 1.0f && d->N % 4 == 0) ,
                 "gemm: colsum_part is only supported with the plain dGELU epilogue");
    // vector (4-element) epilogue accesses need every touched row start 4-element aligned
    bool vec = (d->ldc % 4 == 0) && (d->sC_outer % 4 == 0) && (d->sC_inner % 4 == 0) &&
               ((uintptr_t)d->C % 16 == 0);
    if (d->bias) vec = vec && ((uintptr_t)d->bias % 16 == 0);
    if (d->resid) vec = vec && (d->ldr % 4 == 0) && ((uintptr_t)d->resid % 16 == 0);
    if (d->aux) vec = vec && (d->ldaux % 4 == 0) && ((uintptr_t)d->aux % 16 == 0);
    g.vec = vec ? 1 : 0;
    MMAE_REQUIRE(!d->colsum_part || (vec && ((uintptr_t)d->colsum_part % 16 == 0)), "gemm: colsum_part needs 4-element aligned C/aux");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t t_ev = mmae_timing_begin(st);
    struct TimingGuard {            // the bracket closes on every return path
        hipEvent_t a; hipStream_t st; double flop; int cls; double bytes;
        ~TimingGuard() { mmae_timing_end(a, st, flop, cls, bytes); }
    } t_guard{t_ev, st, 2.0 * d->M * d->N * d->K * d->batch * flop_scale,
              timing_cls >= 0 ? timing_cls : (d->ab_dtype == MMAE_BF16 ? 0 : (d->ab_dtype == MMAE_MXFP8 ? 2 : 1)),      // (MMAE_F16: the fp32 adapters' class)
              gemm_algorithmic_bytes(d)};
    // split-K: the dW-type products (small M x N, K = all rows of the batch) would otherwise occupy
    // a handful of the 256 CUs.  Each K slice writes a dense f32 partial slab into the caller's
    // workspace; a second launch sums the slabs into C in a fixed order (deterministic).
    const int bk = d->ab_dtype == MMAE_F32 ? 16 : 64;
    const int nkt = (d->K + bk - 1) / bk;
    int splitk = d->split_k > 1 ? d->split_k : 1;
    if (splitk > 1) {
        MMAE_REQUIRE(d->c_dtype == MMAE_F32 && d->epi == MMAE_EPI_NONE && !d->bias && !d->resid && d->batch == 1 && d->alpha == 1.0f,
                     "gemm: split_k needs a plain unbatched f32 C");
        MMAE_REQUIRE(d->ws && d->ws_elems >= (int64_t)splitk * d->M * d->N, "gemm: split_k workspace too small");
        MMAE_REQUIRE((uintptr_t)d->ws % 16 == 0, "gemm: split_k workspace unaligned");
    }
    if (splitk > nkt) splitk = nkt;
    g.kt_per_split = (nkt + splitk - 1) / splitk;
    g.splitk = (nkt + g.kt_per_split - 1) / g.kt_per_split;
    g.ws = (float*)d->ws;
    int code = 0, unused = 0;
    gemm_plan(d, &code, &unused);
    if (d->ln_out) code = 9;                                // the 256 x 256 ping-pong tile: the only kernel with the LayerNorm side output
    g.acs = nullptr;
    if (h16) return mmae_gemm_bf16_pp_impl(d, g, 9, st);    // fp16 storage: the 256 x 256 ping-pong tile, compiled flavours only
    if (d->a_colsum) {
        MMAE_REQUIRE(d->ab_dtype == MMAE_BF16 && d->a_trans && d->batch == 1 && code == 9,
                     "gemm: a_colsum needs a bf16, a_trans, unbatched product on the 256x256 ping-pong kernel (see mmae_gemm_plan)");
        const int64_t slab = g.splitk > 1 ? (int64_t)g.splitk * d->M * d->N : 0;
        MMAE_REQUIRE(d->ws && d->ws_elems >= slab + (int64_t)g.splitk * d->M, "gemm: a_colsum workspace too small");
        g.acs = (float*)d->ws + slab;
    }
    if (d->ab_dtype == MMAE_MXFP8) {
        MMAE_REQUIRE(!d->a_colsum, "gemm(mxfp8): no a_colsum");
        if (splitk > 1) {                                   // slices of whole scale groups (four K tiles)
            g.kt_per_split = (g.kt_per_split + 3) / 4 * 4;
            g.splitk = (nkt + g.kt_per_split - 1) / g.kt_per_split;
        }
        const int rc = mmae_gemm_mxfp8_impl(d, g, st);
        if (rc || g.splitk <= 1) return rc;
        return mmae_splitk_reduce(g.ws, (float*)d->C, d->M, d->N, d->ldc, g.splitk, d->accumulate, st);
    }
    int rc = (d->ab_dtype == MMAE_BF16) ? mmae_gemm_bf16_impl(d, g, code, st)
           : ((d->ab_dtype == MMAE_F32X3 || d->ab_dtype == MMAE_F32F16) ? mmae_gemm_f32x3_impl(d, g, st) : mmae_gemm_f32_impl(d, g, st));
    if (rc) return rc;
    if (g.acs) {
        rc = mmae_acs_reduce(g.acs, g.splitk, d->M, d->a_colsum, d->a_colsum_acc, st);
        if (rc) return rc;
    }
    if (g.splitk <= 1) return 0;
    return mmae_splitk_reduce(g.ws, (float*)d->C, d->M, d->N, d->ldc, g.splitk, d->accumulate, st);
}

extern "C" {

int mmae_gemm_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(g_tmu);
    for (auto& t : g_timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    g_timed.clear();
    g_timing = on != 0;
    return 0;
}

int mmae_gemm_timing_read_bytes(double* bytes3) {
    MMAE_REQUIRE(bytes3, "gemm_timing_read_bytes: null pointer");
    std::lock_guard<std::mutex> lk(g_tmu);
    for (int c = 0; c < 3; ++c) bytes3[c] = 0.0;
    for (auto& t : g_timed) bytes3[t.cls] += t.bytes;
    return 0;
}

int mmae_gemm_timing_read(double* ms2, double* flop2, int64_t* calls2) {
    MMAE_REQUIRE(ms2 && flop2 && calls2, "gemm_timing_read: null pointer");
    std::lock_guard<std::mutex> lk(g_tmu);
    for (int c = 0; c < 3; ++c) { ms2[c] = 0.0; flop2[c] = 0.0; calls2[c] = 0; }
    for (auto& t : g_timed) {
        if (hipEventSynchronize(t.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) continue;
        ms2[t.cls] += ms; flop2[t.cls] += t.flop; calls2[t.cls] += 1;
    }
    return 0;
}

int mmae_gemm_plan(const mmae_gemm_desc* d, int* tile, int* split_k) {
    MMAE_REQUIRE(d && tile && split_k, "gemm_plan: null argument");
    MMAE_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm_plan: empty problem");
    gemm_plan(d, tile, split_k);
    return 0;
}

int mmae_gemm_auto_splitk(int M, int N, int K, int ab_dtype) {
    mmae_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = M; d.N = N; d.K = K; d.ab_dtype = ab_dtype; d.c_dtype = MMAE_F32;
    d.a_trans = d.b_trans = 1; d.batch = d.batch_inner = 1; d.alpha = 1.0f;
    d.ldc = (N + 3) / 4 * 4;
    int tile = 0, s = 1;
    gemm_plan(&d, &tile, &s);
    return (N % 4 == 0) ? s : 1;
}

}  // extern "C"
