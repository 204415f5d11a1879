// bf16 MFMA ping-pong GEMM: the generic instantiations (every epilogue flavour selected at run time) and the grouped
// weight-gradient kernel.  The kernel body lives in gemm_pp_body.h; gemm_bf16_pp_fl.hip holds the instantiations with the
// epilogue flavour fixed at compile time.
#include "gemm_pp_body.h"
#include <mutex>

namespace {

// Grouped weight-gradient launch: up to 8 dW[N_out][K_in] = dY^T X products that share the reduction length (the rows of the
// batch) and the number of K slices, in ONE grid -- blockIdx.x walks the concatenated tile lists, blockIdx.z is the K slice.
// Every workgroup computes one 256 x 256 tile over its slice of the rows: the four weight gradients of a transformer block fill
// the chip with 2 slices (216 workgroups at ViT-B) instead of 7-28 slices per product, i.e. 1/4-1/14 of the partial-slab
// traffic, K loops 4-14x longer, and one reduction launch per block instead of four (+ the bias reductions).
struct DwProblem {
    const void* A; const void* B; float* ws; float* acs;
    int M, N;                    // M = N_out (columns of dY), N = K_in (columns of X)
    long long lda, ldb;
    int tiles_n, tile_begin;
};
struct DwGroupArgs {
    int n, K, splitk, kt_per_split, tiles_total, xcd;
    DwProblem p[8];
};

template <bool KF, bool WIDE = false, bool H16 = false>
__global__ void __launch_bounds__(512) gemm_bf16_pp_dwgroup_kernel(const DwGroupArgs ga) {
    // XCD-aware order (workgroup b runs on XCD b % 8): every XCD takes a contiguous range of the concatenated tile lists, so the
    // tiles resident on one XCD share dY / X column panels in its L2 instead of every XCD streaming every panel
    const int tile = __builtin_amdgcn_readfirstlane(ga.xcd ? xcd_tile((int)blockIdx.x, ga.tiles_total) : (int)blockIdx.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) pi += (i < ga.n && tile >= ga.p[i].tile_begin) ? 1 : 0;
    pi = __builtin_amdgcn_readfirstlane(pi);
    const DwProblem& pr = ga.p[pi];
    GemmArgs g;
    g.A = pr.A; g.B = pr.B; g.C = nullptr;
    g.M = pr.M; g.N = pr.N; g.K = ga.K;
    g.lda = pr.lda; g.ldb = pr.ldb; g.ldc = pr.N;
    g.nb_inner = 1;
    g.sAo = g.sAi = g.sBo = g.sBi = g.sCo = g.sCi = 0;
    g.bias = nullptr; g.resid = nullptr; g.ldr = 0; g.aux = nullptr; g.ldaux = 0;
    g.c_f32 = 1; g.aux_f32 = 0; g.epi = MMAE_EPI_NONE; g.accumulate = 0; g.vec = 1;
    g.alpha = 1.0f;
    g.tiles_n = pr.tiles_n;
    const int local = tile - pr.tile_begin;
    g.tiles_total = local + 1;                            // exactly one tile for this workgroup
    g.splitk = 2;                                         // always through the partial slabs (slice z -> ws[z][M][N]); > 1 only selects that epilogue
    g.kt_per_split = ga.kt_per_split;
    g.ws = pr.ws;
    g.xcd_swizzle = 0;
    g.acs = pr.acs;
    g.colpart = nullptr;
    g.wide_st = 0;
    g.dbg = 0; g.dephase = 0;
    g.h16 = H16 ? 1 : 0; g.a_amax = nullptr;              // (the partial slabs stay in the operands' units; the reduction applies 1/S)
    pp_body<4, true, true, 0, KF, KF, WIDE, H16>(g, local, 1 << 30);
}

}  // namespace

int mmae_gemm_bf16_pp_fl_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, int fl, hipStream_t st);
hipEvent_t mmae_timing_begin(hipStream_t st);
void mmae_timing_end(hipEvent_t a, hipStream_t st, double flop, int cls, double bytes);

// tile codes: 9 = 256 x 256 (TM = 4), 10 = 320 x 256 (TM = 5; k-contiguous A only)
int mmae_gemm_bf16_pp_impl(const mmae_gemm_desc* d, const GemmArgs& g, int code, hipStream_t st) {
    const bool aks = d->a_trans != 0, bks = d->b_trans != 0;
    // epilogue flavour known before the launch: use the instantiation that contains only that epilogue (fewer live registers:
    // no spills in the 320-row kernels, whose per-tile scratch reloads cost a full vmcnt(0) drain of the prefetched K tiles)
    static const int env_fl = mmae_env_int("MMAE_PP_FL", 1);
    if (g.h16) {                                  // fp16 storage: only the compiled flavours of the 256 x 256 tile (gemm_bf16_pp_fl.hip)
        const int fl = (code == 9 && !aks) ? gemm_flavour(g, d->batch) : 0;
        const int rc = fl ? mmae_gemm_bf16_pp_fl_impl(d, g, 9, fl, st) : MMAE_ESUPPORT;
        if (rc == MMAE_ESUPPORT) mmae_set_error("gemm(f16): this operand layout / epilogue has no fp16 instantiation (k-contiguous A, K % 32 == 0, 8-aligned widths)");
        return rc;
    }
    if (env_fl && !aks) {
        const int fl = gemm_flavour(g, d->batch);
        if (fl) {
            const int rc = mmae_gemm_bf16_pp_fl_impl(d, g, code, fl, st);
            if (rc != MMAE_ESUPPORT) return rc;
        }
    }
    if (g.ln_out) { mmae_set_error("gemm: no instantiation with the LayerNorm side output for this product"); return MMAE_ESUPPORT; }
    if (code == 10 && !aks) {
        return bks ? launch<5, false, true>(g, d->batch, st) : launch<5, false, false>(g, d->batch, st);
    }
    if (!aks && !bks) return launch<4, false, false>(g, d->batch, st);
    if (!aks && bks) return launch<4, false, true>(g, d->batch, st);
    if (aks && !bks) return launch<4, true, false>(g, d->batch, st);
    return launch<4, true, true>(g, d->batch, st);
}

// ------------------------------------------------------------------------------------------------------------------
// grouped weight gradients (mmae_gemm_dw_group)
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct DwReduceProblem { const float* ws; float* C; const float* acs; float* bias; long long mn; int M; int blk_begin; };
struct DwReduceArgs { int n, splits, accumulate; const float* unscale; DwReduceProblem p[8]; };

// C_p (+)= sum_z ws_p[z] (fixed order: deterministic); bias_p (+)= sum_z acs_p[z].
// A workgroup belongs to ONE problem (its descriptor comes through scalar loads) and a thread owns four float4 a workgroup-stride apart:
// 4 x splits independent 16-byte loads in flight per lane.  (The first form indexed the problem table per element with a per-lane index -- a
// dependent vector load of the descriptor in front of every data load -- and ran at 2.5 TB/s: 33 us per encoder block, 0.94 ms per step.)
__global__ void __launch_bounds__(256) dw_group_reduce_kernel(const DwReduceArgs ra) {
    const float us = h16_grad_unscale(ra.unscale);         // fp16-storage gradients: the slabs are in scaled units (1 otherwise)
    const int blk = (int)blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) pi += (k < ra.n && blk >= ra.p[k].blk_begin) ? 1 : 0;
    pi = __builtin_amdgcn_readfirstlane(pi);
    const DwReduceProblem& pr = ra.p[pi];
    const long long mn4 = pr.mn >> 2;
    const long long i0 = (long long)(blk - pr.blk_begin) * 1024 + threadIdx.x;
    f32x4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < ra.splits; ++z) {
        const float* slab = pr.ws + (long long)z * pr.mn;
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long long i = i0 + 256 * u; v[u] = i < mn4 ? ld4(slab + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[u][j] += v[u][j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = i0 + 256 * u;
        if (i < mn4) {
            if (ra.unscale) {
#pragma unroll
                for (int j = 0; j < 4; ++j) a[u][j] *= us;
            }
            if (ra.accumulate) {
                const f32x4 c0 = ld4(pr.C + i * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[u][j] += c0[j];
            }
            st4(pr.C + i * 4, a[u]);
        }
    }
    if (blk == pr.blk_begin && pr.bias) {                  // the problem's first workgroup also reduces its bias partials
        for (int m = threadIdx.x; m < pr.M; m += 256) {
            float t = 0.f;
            for (int z = 0; z < ra.splits; ++z) t += pr.acs[(long long)z * pr.M + m];
            if (ra.unscale) t *= us;
            pr.bias[m] = ra.accumulate ? pr.bias[m] + t : t;
        }
    }
}

int dw_group_splits(const mmae_dw_group_desc* d, long long* tiles_out) {
    const int n_cu = mmae_cu_side() > 0 ? mmae_cu_side() : mmae_cu_avail();
    long long tiles = 0;
    for (int i = 0; i < d->n; ++i) tiles += (long long)((d->p[i].n_out + 255) / 256) * ((d->p[i].k_in + 255) / 256);
    static const int env_split = mmae_env_int("MMAE_DW_SPLIT", 0);     // experiments: slices when the group has > 64 tiles
    int s = d->split_k > 0 ? d->split_k : ((env_split > 0 && tiles > 64) ? env_split : (int)(n_cu / (tiles > 0 ? tiles : 1)));
    // (Filling whole rounds of the chip -- ViT-L's 192 tiles as 4 slices = 768 workgroups instead of 192 on 256 CUs -- measured
    // +1.8 % on the bf16 cfg5 step and -1.3 % on the mxfp8 one: the launch shares the chip with the dX chain of the main stream,
    // which uses the CUs a single slice leaves free.  Kept at n_cu / tiles; MMAE_DW_SPLIT overrides for experiments.)
    const int nkt = (d->rows + 31) / 32;                      // 32-wide K tiles
    if (s > nkt / 16) s = nkt / 16;                            // >= 16 K tiles per slice
    if (s < 1) s = 1;
    const int kt_per = (nkt + s - 1) / s;
    s = (nkt + kt_per - 1) / kt_per;
    if (tiles_out) *tiles_out = tiles;
    return s;
}

}  // namespace

extern "C" int64_t mmae_gemm_dw_group_ws_elems(const mmae_dw_group_desc* d) {
    if (!d || d->n < 1 || d->n > 8 || d->rows <= 0) return -1;
    const int s = dw_group_splits(d, nullptr);
    long long elems = 0;
    for (int i = 0; i < d->n; ++i) elems += (long long)s * ((long long)d->p[i].n_out * d->p[i].k_in + d->p[i].n_out);
    return elems;
}

extern "C" int mmae_gemm_dw_group(const mmae_dw_group_desc* d, void* stream) {
    MMAE_REQUIRE(d && d->n >= 1 && d->n <= 8 && d->rows > 0, "dw_group: bad descriptor");
    if (d->ab_dtype != MMAE_BF16 && d->ab_dtype != MMAE_F16) { mmae_set_error("dw_group: bf16 / fp16 operands only"); return MMAE_ESUPPORT; }
    MMAE_REQUIRE(!d->unscale || d->ab_dtype == MMAE_F16, "dw_group: unscale is an MMAE_F16 option");
    const bool h16 = d->ab_dtype == MMAE_F16;
    for (int i = 0; i < d->n; ++i) {
        const mmae_dw_problem& q = d->p[i];
        MMAE_REQUIRE(q.dy && q.x && q.dw && q.n_out > 0 && q.k_in > 0, "dw_group: null / empty problem");
        if ((q.n_out % 8) || (q.k_in % 8) || (q.ldy % 8) || (q.ldx % 8) || ((uintptr_t)q.dy % 16) || ((uintptr_t)q.x % 16) || ((uintptr_t)q.dw % 16) ||
            (q.db && ((uintptr_t)q.db % 4))) { mmae_set_error("dw_group: widths / leading dimensions must be multiples of 8, bases 16-byte aligned"); return MMAE_ESUPPORT; }
    }
    long long tiles = 0;
    const int s = dw_group_splits(d, &tiles);
    const int nkt = (d->rows + 31) / 32;
    const int kt_per = (nkt + s - 1) / s;
    MMAE_REQUIRE(d->ws && ((uintptr_t)d->ws % 16) == 0, "dw_group: workspace missing / unaligned");
    MMAE_REQUIRE(d->ws_elems >= mmae_gemm_dw_group_ws_elems(d), "dw_group: workspace too small (mmae_gemm_dw_group_ws_elems)");
    DwGroupArgs ga = {};
    DwReduceArgs ra = {};
    ga.n = d->n; ga.K = d->rows; ga.splitk = s; ga.kt_per_split = kt_per;
    ra.n = d->n; ra.splits = s; ra.accumulate = d->accumulate; ra.unscale = d->unscale;
    float* w = d->ws;
    int tb = 0;
    int rblk = 0;                                            // reduction workgroups: 1024 float4 each, per problem
    for (int i = 0; i < d->n; ++i) {
        const mmae_dw_problem& q = d->p[i];
        const long long mn = (long long)q.n_out * q.k_in;
        DwProblem& p = ga.p[i];
        p.A = q.dy; p.B = q.x; p.ws = w; p.acs = q.db ? w + (long long)s * mn : nullptr;
        p.M = q.n_out; p.N = q.k_in; p.lda = q.ldy; p.ldb = q.ldx;
        p.tiles_n = (q.k_in + 255) / 256; p.tile_begin = tb;
        tb += ((q.n_out + 255) / 256) * p.tiles_n;
        DwReduceProblem& r = ra.p[i];
        r.ws = w; r.C = q.dw; r.acs = p.acs; r.bias = q.db; r.mn = mn; r.M = q.n_out; r.blk_begin = rblk;
        rblk += (int)((mn / 4 + 1023) / 1024);
        w += (long long)s * (mn + q.n_out);      // n_out, k_in multiples of 8: every slab base stays 16-byte aligned
    }
    ga.tiles_total = tb;
    static const int env_xcd = mmae_env_int("MMAE_DW_XCD", 1);
    ga.xcd = env_xcd;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)4 * (256 + 256) * 64 + 1024;
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp_dwgroup_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp_dwgroup_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp_dwgroup_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp_dwgroup_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp_dwgroup_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
#ifdef MMAE_NO_KF
    static const int env_kf = 0;
#else
    static const int env_kf = mmae_env_int("MMAE_PP_KF", 1);
#endif
    double flop = 0.0, abytes = 0.0;             // algorithmic bytes: dY and X once (16-bit), dW once (f32; read too when accumulating) -- not the partial slabs
    for (int i = 0; i < d->n; ++i) {
        flop += 2.0 * d->rows * d->p[i].n_out * d->p[i].k_in;
        abytes += 2.0 * d->rows * ((double)d->p[i].n_out + d->p[i].k_in) + (d->accumulate ? 8.0 : 4.0) * d->p[i].n_out * d->p[i].k_in;
    }
    hipEvent_t t_ev = mmae_timing_begin(st);
    static const int env_wide = mmae_env_int("MMAE_DW_WIDE", 0);      // one phase pair per K tile: measured neutral (421 vs 411-436 us per block), off
    if (h16) {
        if (env_kf && (d->rows & 31) == 0) hipLaunchKernelGGL((gemm_bf16_pp_dwgroup_kernel<true, false, true>), dim3(tb, 1, s), dim3(512), lds, st, ga);
        else hipLaunchKernelGGL((gemm_bf16_pp_dwgroup_kernel<false, false, true>), dim3(tb, 1, s), dim3(512), lds, st, ga);
    } else
    if (env_kf && env_wide && (d->rows & 31) == 0) hipLaunchKernelGGL((gemm_bf16_pp_dwgroup_kernel<true, true>), dim3(tb, 1, s), dim3(512), lds, st, ga);
    else if (env_kf && (d->rows & 31) == 0) hipLaunchKernelGGL(gemm_bf16_pp_dwgroup_kernel<true>, dim3(tb, 1, s), dim3(512), lds, st, ga);
    else hipLaunchKernelGGL(gemm_bf16_pp_dwgroup_kernel<false>, dim3(tb, 1, s), dim3(512), lds, st, ga);
    int rc = mmae_check_launch("gemm_bf16_pp_dwgroup");
    if (rc) { mmae_timing_end(t_ev, st, flop, h16 ? 1 : 0, abytes); return rc; }
    hipLaunchKernelGGL(dw_group_reduce_kernel, dim3((unsigned)(rblk < 1 ? 1 : rblk)), dim3(256), 0, st, ra);
    mmae_timing_end(t_ev, st, flop, h16 ? 1 : 0, abytes);
    return mmae_check_launch("dw_group_reduce");
}

#ifdef MMAE_PP_TRACE
// phase stamps of the last launch from THIS translation unit (the generic and the grouped weight-gradient ping-pong kernels): tools/pp_trace.py
extern "C" int mmae_debug_pp_trace_dw(long long* out_host_256) { return (int)hipMemcpyFromSymbol(out_host_256, HIP_SYMBOL(g_pp_trace), 2 * 128 * 8); }
extern "C" int mmae_debug_pp_wg_dw(long long* out_host_4096) { return (int)hipMemcpyFromSymbol(out_host_4096, HIP_SYMBOL(g_pp_wg), 1024 * 4 * 8); }
#endif
