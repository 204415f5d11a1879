// Composite entry points: a whole transformer block per call (see mmae.h, mmae_block_desc).
//
// Nothing here is a new kernel: the functions below enqueue the same launches, through the same public C entry points, that
// the Python side (multimae_amd/functions.py::block_fwd / block_bwd) issues one by one.  The point is the host: at B = 256 a
// ViT-B step is ~1 000 launches and Python spends 20-25 us on each; the four output adapters' backward passes (short
// kernels, ~125 launches each) were paced by the host, not by the GPU.  One call per block per direction brings a cfg3
// step from ~1 040 host round trips to ~500.
#include <mutex>
#include "gemm_common.h"

namespace {

struct Lin {
    const mmae_block_desc* d;
    int ab() const { return d->act_dtype == MMAE_BF16 ? MMAE_BF16 : d->f32_gemm; }
};

// events that order the side stream behind the main stream.  A wait captures the event's state when it is enqueued, so
// the ring only has to outlive the hipStreamWaitEvent call that follows each record.
hipEvent_t next_event() {
    static std::mutex mu;
    static hipEvent_t ring[64];
    static int made = 0, pos = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (made < 64) { if (hipEventCreateWithFlags(&ring[made], hipEventDisableTiming) != hipSuccess) return nullptr; ++made; pos = made - 1; return ring[pos]; }
    pos = (pos + 1) & 63;
    return ring[pos];
}

int fork_to(hipStream_t from, hipStream_t to) {      // `to` continues after everything enqueued so far on `from`
    if (from == to) return 0;
    hipEvent_t e = next_event();
    if (!e || hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) {
        mmae_set_error("block: could not order the side stream behind the compute stream");
        return MMAE_ELAUNCH;
    }
    return 0;
}

// out[M,N] = x[M,K] w[N,K]^T (+ bias, epilogue, residual) -- ops.linear_fwd
int lin_fwd(const mmae_block_desc* b, const void* x, const void* w, const float* bias, void* out, int out_dtype, int M, int N, int K,
            const float* resid, void* aux, int epi, hipStream_t st) {
    mmae_gemm_desc g = {};
    g.A = x; g.B = w; g.C = out;
    g.ab_dtype = b->act_dtype == MMAE_BF16 ? MMAE_BF16 : b->f32_gemm;
    g.c_dtype = out_dtype;
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldb = K; g.ldc = N;
    g.batch = g.batch_inner = 1;
    g.bias = bias; g.resid = resid; g.ldr = N;
    g.aux = aux; g.ldaux = N; g.aux_dtype = b->act_dtype;
    g.epi = epi; g.alpha = 1.0f;
    int tile = 0, split = 1;
    int rc = mmae_gemm_plan(&g, &tile, &split);
    if (rc) return rc;
    g.tile = tile; g.split_k = split;
    if (split > 1) {
        if ((int64_t)split * M * N > b->ws_main_elems) { mmae_set_error("block: ws_main too small"); return MMAE_EINVAL; }
        g.ws = b->ws_main; g.ws_elems = b->ws_main_elems;
    }
    return mmae_gemm(&g, st);
}

// out[M,K] = dy[M,N] w[N,K] (+ dGELU epilogue with column-sum partials) -- ops.linear_dx
int lin_dx(const mmae_block_desc* b, const void* dy, const void* w, void* out, int out_dtype, int M, int N, int K, void* aux, int epi,
           float* colsum_part, hipStream_t st) {
    mmae_gemm_desc g = {};
    g.A = dy; g.B = w; g.C = out;
    g.ab_dtype = b->act_dtype == MMAE_BF16 ? MMAE_BF16 : b->f32_gemm;
    g.c_dtype = out_dtype;
    g.M = M; g.N = K; g.K = N;
    g.lda = N; g.ldb = K; g.ldc = K;
    g.b_trans = 1;
    g.batch = g.batch_inner = 1;
    g.aux = aux; g.ldaux = K; g.aux_dtype = b->act_dtype;
    g.epi = epi; g.alpha = 1.0f;
    g.colsum_part = colsum_part;
    int tile = 0, split = 1;
    int rc = mmae_gemm_plan(&g, &tile, &split);
    if (rc) return rc;
    g.tile = tile; g.split_k = split;
    if (split > 1) {
        if ((int64_t)split * M * K > b->ws_main_elems) { mmae_set_error("block: ws_main too small"); return MMAE_EINVAL; }
        g.ws = b->ws_main; g.ws_elems = b->ws_main_elems;
    }
    return mmae_gemm(&g, st);
}

// dw[N,K] (+)= dy[M,N]^T x[M,K]; db[N] (+)= column sums of dy (inside the GEMM where the kernel can) -- ops.linear_dw
int lin_dw(const mmae_block_desc* b, const void* dy, const void* x, float* dw, float* db, int M, int N, int K, hipStream_t st) {
    if (!dw && !db) return 0;
    const int acc = b->grad_acc;
    int64_t ws_used = 0;
    if (dw) {
        mmae_gemm_desc g = {};
        g.A = dy; g.B = x; g.C = dw;
        g.ab_dtype = b->act_dtype == MMAE_BF16 ? MMAE_BF16 : b->f32_gemm;
        g.c_dtype = MMAE_F32;
        g.M = N; g.N = K; g.K = M;
        g.lda = N; g.ldb = K; g.ldc = K;
        g.a_trans = g.b_trans = 1;
        g.batch = g.batch_inner = 1;
        g.accumulate = acc; g.alpha = 1.0f;
        int tile = 0, split = 1;
        int rc = mmae_gemm_plan(&g, &tile, &split);
        if (rc) return rc;
        g.tile = tile; g.split_k = split;
        const bool fused_db = db && tile == 9 && g.ab_dtype == MMAE_BF16;
        ws_used = (split > 1 ? (int64_t)split * N * K : 0) + (fused_db ? (int64_t)(split > 1 ? split : 1) * N : 0);
        if (ws_used > b->ws_side_elems) { mmae_set_error("block: ws_side too small"); return MMAE_EINVAL; }
        if (ws_used) { g.ws = b->ws_side; g.ws_elems = b->ws_side_elems; }
        if (fused_db) { g.a_colsum = db; g.a_colsum_acc = acc; db = nullptr; }
        rc = mmae_gemm(&g, st);
        if (rc) return rc;
    }
    if (db) {
        if (mmae_colsum_ws_elems(M, N) > b->ws_side_elems) { mmae_set_error("block: ws_side too small"); return MMAE_EINVAL; }
        return mmae_colsum(dy, b->act_dtype, M, N, N, db, acc, b->ws_side, st);
    }
    return 0;
}

int scatter3(const mmae_block_desc* b, const float* part, int rows, int seg_w, float* d0, float* d1, float* d2, hipStream_t st) {
    if (!d0 && !d1 && !d2) return 0;
    float* dsts[3] = {d0, d1, d2};
    if (mmae_colsum_ws_elems(rows, 3 * seg_w) > b->ws_side_elems) { mmae_set_error("block: ws_side too small"); return MMAE_EINVAL; }
    return mmae_colsum_scatter(part, MMAE_F32, rows, 3 * seg_w, 3 * seg_w, seg_w, dsts, 3, b->grad_acc, b->ws_side, st);
}

int check_desc(const mmae_block_desc* d) {
    MMAE_REQUIRE(d, "block: null descriptor");
    MMAE_REQUIRE(d->B > 0 && d->N > 0 && d->D > 0 && d->heads > 0 && d->Hd > 0 && d->D % d->heads == 0, "block: bad geometry");
    MMAE_REQUIRE(d->act_dtype == MMAE_BF16 || (d->act_dtype == MMAE_F32 && d->f32_gemm == MMAE_F32X3),
                 "block: activations must be bf16, or f32 with split-bf16 (MMAE_F32X3) products");
    const int hd = d->D / d->heads;
    if ((hd != 32 && hd != 64) || d->N > 256) { mmae_set_error("block: geometry outside the fused attention kernel (head_dim 32/64, N <= 256)"); return MMAE_ESUPPORT; }
    MMAE_REQUIRE(d->qkv_w && d->proj_w && d->fc1_w && d->fc2_w && d->n1_w && d->n1_b && d->qkv_b && d->proj_b && d->n2_w && d->n2_b &&
                 d->fc1_b && d->fc2_b, "block: null parameter");
    MMAE_REQUIRE(d->x0 && d->ln1 && d->mean1 && d->rstd1 && d->qkv && d->lse && d->ao && d->x1 && d->ln2 && d->mean2 && d->rstd2 &&
                 d->hpre && d->hact, "block: null activation buffer");
    return 0;
}

int attn_strides_fwd(const mmae_block_desc* d, hipStream_t st) {
    const int D = d->D, N = d->N, hd = D / d->heads;
    const size_t es = d->act_dtype == MMAE_BF16 ? 2 : 4;
    const char* qkv = (const char*)d->qkv;
    const float scale = 1.0f / sqrtf((float)hd);
    auto fn = d->act_dtype == MMAE_BF16 ? mmae_attn_fwd : mmae_attn_fwd_f32x3;
    return fn(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, d->ao, d->lse, d->B, d->heads, N, N, hd, (int64_t)N * 3 * D, 3 * D,
              (int64_t)N * 3 * D, 3 * D, (int64_t)N * 3 * D, 3 * D, (int64_t)N * D, D, scale, st);
}

}  // namespace

extern "C" {

int mmae_block_fwd(const mmae_block_desc* d, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    MMAE_REQUIRE(d->x2, "block_fwd: null output");
    hipStream_t st = (hipStream_t)stream;
    const int R = d->B * d->N, D = d->D, Hd = d->Hd, act = d->act_dtype;
    if ((rc = mmae_layernorm_fwd(d->x0, d->n1_w, d->n1_b, d->ln1, act, d->mean1, d->rstd1, R, D, d->eps, st))) return rc;
    if ((rc = lin_fwd(d, d->ln1, d->qkv_w, d->qkv_b, d->qkv, act, R, 3 * D, D, nullptr, nullptr, MMAE_EPI_NONE, st))) return rc;
    if ((rc = attn_strides_fwd(d, st))) return rc;
    if ((rc = lin_fwd(d, d->ao, d->proj_w, d->proj_b, d->x1, MMAE_F32, R, D, D, d->x0, nullptr, MMAE_EPI_NONE, st))) return rc;
    if ((rc = mmae_layernorm_fwd(d->x1, d->n2_w, d->n2_b, d->ln2, act, d->mean2, d->rstd2, R, D, d->eps, st))) return rc;
    if ((rc = lin_fwd(d, d->ln2, d->fc1_w, d->fc1_b, d->hact, act, R, Hd, D, nullptr, d->hpre, MMAE_EPI_GELU, st))) return rc;
    return lin_fwd(d, d->hact, d->fc2_w, d->fc2_b, d->x2, MMAE_F32, R, D, Hd, d->x1, nullptr, MMAE_EPI_NONE, st);
}

int mmae_block_bwd(const mmae_block_desc* d, void* stream, void* side_stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    MMAE_REQUIRE(d->dx && d->dx_act && d->dx0 && d->d_hpre && d->d_ln2 && d->d_ao && d->d_qkv && d->d_ln1 && d->dx1 && d->part1 && d->part2,
                 "block_bwd: null gradient / temporary buffer");
    const int act = d->act_dtype;
    MMAE_REQUIRE(act == MMAE_F32 || (d->dx1_act && d->dx0_act), "block_bwd: bf16 activations need dx1_act / dx0_act");
    MMAE_REQUIRE(!d->g_fc1_b || d->part_h, "block_bwd: part_h needed for the fc1 bias gradient");
    hipStream_t st = (hipStream_t)stream;
    hipStream_t sd = side_stream ? (hipStream_t)side_stream : st;
    const int R = d->B * d->N, D = d->D, Hd = d->Hd, N = d->N, hd = D / d->heads;
    const int nblk = mmae_layernorm_bwd_nblk(R);
    const int hrows = (R + 63) / 64;
    // ---- MLP
    float* part_h = d->g_fc1_b ? d->part_h : nullptr;
    if ((rc = lin_dx(d, d->dx_act, d->fc2_w, d->d_hpre, act, R, D, Hd, d->hpre, MMAE_EPI_DGELU, part_h, st))) return rc;
    if ((rc = fork_to(st, sd))) return rc;                          // dx_act, d_hpre ready for the weight-gradient stream
    if ((rc = lin_dw(d, d->dx_act, d->hact, d->g_fc2_w, d->fc2_b_done ? nullptr : d->g_fc2_b, R, D, Hd, sd))) return rc;
    if ((rc = lin_dx(d, d->d_hpre, d->fc1_w, d->d_ln2, act, R, Hd, D, nullptr, MMAE_EPI_NONE, nullptr, st))) return rc;
    if ((rc = lin_dw(d, d->d_hpre, d->ln2, d->g_fc1_w, nullptr, R, Hd, D, sd))) return rc;
    if (part_h) {
        float* dst[1] = {d->g_fc1_b};
        if (mmae_colsum_ws_elems(hrows, Hd) > d->ws_side_elems) { mmae_set_error("block: ws_side too small"); return MMAE_EINVAL; }
        if ((rc = mmae_colsum_scatter(part_h, MMAE_F32, hrows, Hd, Hd, Hd, dst, 1, d->grad_acc, d->ws_side, sd))) return rc;
    }
    void* dx1_act = act == MMAE_F32 ? nullptr : d->dx1_act;
    if ((rc = mmae_layernorm_bwd(d->d_ln2, act, d->x1, d->n2_w, d->mean2, d->rstd2, d->dx, d->dx1, dx1_act, act, d->part2, R, D, st))) return rc;
    const void* dx1a = act == MMAE_F32 ? (const void*)d->dx1 : (const void*)d->dx1_act;
    // ---- attention
    if ((rc = lin_dx(d, dx1a, d->proj_w, d->d_ao, act, R, D, D, nullptr, MMAE_EPI_NONE, nullptr, st))) return rc;
    if ((rc = fork_to(st, sd))) return rc;                          // part2, dx1_act
    if ((rc = scatter3(d, d->part2, nblk, D, d->g_n2_w, d->g_n2_b, d->g_proj_b, sd))) return rc;
    if ((rc = lin_dw(d, dx1a, d->ao, d->g_proj_w, nullptr, R, D, D, sd))) return rc;
    {
        const size_t es = act == MMAE_BF16 ? 2 : 4;
        const char* qkv = (const char*)d->qkv;
        char* dq = (char*)d->d_qkv;
        const int64_t sb3 = (int64_t)N * 3 * D, sb1 = (int64_t)N * D;
        auto fn = act == MMAE_BF16 ? mmae_attn_bwd : mmae_attn_bwd_f32x3;
        if ((rc = fn(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, d->ao, d->d_ao, d->lse, dq, dq + (size_t)D * es, dq + (size_t)2 * D * es,
                     d->B, d->heads, N, N, hd, sb3, 3 * D, sb3, 3 * D, sb3, 3 * D, sb1, D, sb3, 3 * D, sb3, 3 * D, sb3, 3 * D,
                     1.0f / sqrtf((float)hd), st))) return rc;
    }
    if ((rc = lin_dx(d, d->d_qkv, d->qkv_w, d->d_ln1, act, R, 3 * D, D, nullptr, MMAE_EPI_NONE, nullptr, st))) return rc;
    if ((rc = fork_to(st, sd))) return rc;                          // d_qkv
    if ((rc = lin_dw(d, d->d_qkv, d->ln1, d->g_qkv_w, d->g_qkv_b, R, 3 * D, D, sd))) return rc;
    void* dx0_act = act == MMAE_F32 ? nullptr : d->dx0_act;
    if ((rc = mmae_layernorm_bwd(d->d_ln1, act, d->x0, d->n1_w, d->mean1, d->rstd1, d->dx1, d->dx0, dx0_act, act, d->part1, R, D, st))) return rc;
    if ((rc = fork_to(st, sd))) return rc;                          // part1
    return scatter3(d, d->part1, nblk, D, d->g_n1_w, d->g_n1_b, d->g_cs, sd);
}

}  // extern "C"
