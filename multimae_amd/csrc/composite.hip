// Composite entry points: a transformer block, a stack of blocks, a whole SpatialOutputAdapter, the optimiser step -- one
// library call per direction each (see mmae.h: mmae_block_desc, mmae_stack_desc, mmae_adapter_desc, mmae_opt_desc).
//
// No GEMM / attention / LayerNorm kernel is new here: the functions below enqueue the same launches, through the same public C
// entry points, that the Python side (multimae_amd/functions.py) can issue one by one.  The point is the host: at B = 256 a
// cfg3 step is ~1 000 launches; issued from Python (20-60 us each, ~520 ctypes calls with the per-block composites of round 1)
// the host needed 20-33 ms per step and paced the four output adapters' backward passes.  With one call per stack / adapter a
// step is ~40 library calls.
#include <math.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "gemm_common.h"

namespace {

// what the linear helpers need to know about the call they serve
struct Ctx {
    int act_dtype, f32_gemm, grad_acc;
    float* ws_main; int64_t ws_main_elems;
    float* ws_side; int64_t ws_side_elems;
    void* mx_tmp = nullptr; int64_t mx_tmp_bytes = 0;      // scratch of the MX-fp8 activation operands (main stream only)
    const void* const* x3_w = nullptr; int x3_n = 0;       // pre-split weights of an f32 call: triples {f32 weight, [n][3k], [3n][k]}
    void* x3_tmp = nullptr; int64_t x3_tmp_bytes = 0;      // scratch of the pre-split activation operand (main stream only)
    const float* dy_amax = nullptr;                        // MMAE_F32F16: device scalar that pre-scales gradient operands (mmae_gemm_desc.a_amax)
    bool b16() const { return act_dtype != MMAE_F32; }     // 16-bit activations: bf16, or fp16 storage (MMAE_F16: an fp32 adapter in 'h16' mode)
    bool h16() const { return act_dtype == MMAE_F16; }
    int ab() const { return b16() ? act_dtype : f32_gemm; }
    // products whose A operand is a GRADIENT: fp16 operands only with the loss gradient's amax at hand, else the split-bf16 form
    int ab_grad() const { return b16() ? act_dtype : ((f32_gemm == MMAE_F32F16 && !dy_amax) ? MMAE_F32X3 : f32_gemm); }
    const float* amax_for(int ab) const { return ab == MMAE_F32F16 ? dy_amax : nullptr; }
    // fp16 storage: every gradient tensor is stored scaled by S(dy_amax); what leaves as f32 (parameter gradients, d_enc) gets 1/S
    const float* unscale() const { return h16() ? dy_amax : nullptr; }
    size_t es() const { return b16() ? 2 : 4; }
};
inline bool is16(int act) { return act != MMAE_F32; }
inline size_t es_of(int act) { return act == MMAE_F32 ? 4 : 2; }

Ctx ctx_of(const mmae_block_desc* d) {
    Ctx c{d->act_dtype, d->f32_gemm, d->grad_acc, d->ws_main, d->ws_main_elems, d->ws_side, d->ws_side_elems};
    c.mx_tmp = d->mx_tmp; c.mx_tmp_bytes = d->mx_tmp_bytes;
    c.x3_w = d->x3_w; c.x3_n = d->x3_n; c.x3_tmp = d->x3_tmp; c.x3_tmp_bytes = d->x3_tmp_bytes;
    c.dy_amax = d->dy_amax;
    return c;
}

// Pre-split x3 product (mmae.h, mmae_x3_split): if this f32 call carries a pre-split copy of weight w and the contraction is a
// multiple of 32, split the activation operand [M][kc] (row stride ldx) into the scratch and return the weight triple.
const void* const* x3_operand(const Ctx& c, const void* x, int64_t ldx, const void* w, int M, int kc, const void** a3, hipStream_t st, int* rc) {
    static const bool on = (mmae_env_int("MMAE_X3_PRESPLIT", 1) != 0);    // the host side only passes x3_w when asked to (ops.py)
    *rc = 0;
    if (!on || c.act_dtype != MMAE_F32 || c.f32_gemm != MMAE_F32X3 || !c.x3_w || !c.x3_tmp || (kc % 32) || (ldx % 4)) return nullptr;
    const void* const* t3 = nullptr;
    for (int i = 0; i < c.x3_n; ++i) if (c.x3_w[3 * i] == w) { t3 = c.x3_w + 3 * i; break; }
    if (!t3 || mmae_x3_tmp_bytes(M, kc) > c.x3_tmp_bytes) return nullptr;
    *rc = mmae_x3_split((const float*)x, ldx, M, kc, c.x3_tmp, 3LL * kc, kc, 2, st);
    *a3 = c.x3_tmp;
    return *rc ? nullptr : t3;
}

// The MX scratch is two halves: a product reads its quantised activation operand from one while its epilogue (fc1 forward,
// fc2 dX) may already write the next product's operand into the other.
// mx_slot: where the e4m3 bytes [M][K] and the packed scales of an operand live in half `which`
int mx_slot(const Ctx& c, int which, int M, int K, void** q, void** s) {
    const int64_t half = c.mx_tmp_bytes / 2 / 256 * 256, qb = ((int64_t)M * K + 255) / 256 * 256;
    if (!c.mx_tmp || qb + mmae_mx_scale_bytes(M, K) > half) { mmae_set_error("composite: mx_tmp too small"); return MMAE_EINVAL; }
    *q = (char*)c.mx_tmp + which * half; *s = (char*)*q + qb;
    return 0;
}
// quantise the activation operand of an MX-fp8 product into half `which`
int mx_operand(const Ctx& c, const void* x, int64_t ldx, int M, int K, int which, const void** q, const void** s, hipStream_t st) {
    void *qq, *ss;
    int rc = mx_slot(c, which, M, K, &qq, &ss);
    if (rc) return rc;
    *q = qq; *s = ss;
    return mmae_mx_quant(x, c.act_dtype, ldx, M, K, qq, K, ss, st);
}
// operand of an MX product: already quantised in half mx_in (by the producing kernel), or quantised here into the half the
// epilogue does not write
int mx_in_operand(const Ctx& c, const void* x, int64_t ldx, int M, int K, int mx_in, int mx_out, const void** q, const void** s, hipStream_t st) {
    if (mx_in < 0) return mx_operand(c, x, ldx, M, K, mx_out == 0 ? 1 : 0, q, s, st);
    void *qq, *ss;
    int rc = mx_slot(c, mx_in, M, K, &qq, &ss);
    *q = qq; *s = ss;
    return rc;
}

// events that order one stream behind another.  A wait captures the event's state when it is enqueued, so a ring entry only
// has to outlive the hipStreamWaitEvent call that consumes it.  Two rings: `fork` entries are consumed immediately; `held`
// entries are kept across a few more launches (side-stream progress marks of mmae_stack_bwd) and get a longer ring.
struct EventRing {
    std::mutex mu;
    hipEvent_t ring[256];
    int made = 0, pos = 0;
    hipEvent_t next() {
        std::lock_guard<std::mutex> lk(mu);
        if (made < 256) { if (hipEventCreateWithFlags(&ring[made], hipEventDisableTiming) != hipSuccess) return nullptr; ++made; pos = made - 1; return ring[pos]; }
        pos = (pos + 1) & 255;
        return ring[pos];
    }
};
EventRing g_fork_ring, g_held_ring;

int fork_to(hipStream_t from, hipStream_t to) {      // `to` continues after everything enqueued so far on `from`
    if (from == to) return 0;
    hipEvent_t e = g_fork_ring.next();
    if (!e || hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) {
        mmae_set_error("composite: could not order one stream behind the other");
        return MMAE_ELAUNCH;
    }
    return 0;
}

// out[M,N] = x[M,K] w[N,K]^T (+ bias, epilogue, residual) -- ops.linear_fwd
// mxw: {e4m3 weight [N][K], its scales, ...} of mmae_mx_prepare_weights, or NULL for the act-dtype product.  mx_in >= 0: x is
// already quantised in that half of the MX scratch; mx_out >= 0: the epilogue also leaves the quantised output there.
// LayerNorm (g != NULL) or plain 16-bit cast (g == NULL) of the f32 rows a Linear product completes, written by the product's own
// epilogue (mmae_gemm_desc.ln_out): the D = 256 decoder products, whose 256-column tile spans the row
struct LnSide { const float* g; const float* b; void* out; float* mean; float* rstd; float eps; };
std::atomic<int> g_ln_fuse{1};
bool ln_side_ok(const Ctx& c, int N, int K, int out_dtype) {
    return g_ln_fuse.load(std::memory_order_relaxed) != 0 && is16(c.act_dtype) && out_dtype == MMAE_F32 && N == 256 && (K % 32) == 0 && !c.x3_w;
}

int lin_fwd(const Ctx& c, const void* x, const void* w, const float* bias, void* out, int out_dtype, int M, int N, int K,
            const float* resid, void* aux, int epi, hipStream_t st, const void* const* mxw = nullptr, int mx_in = -1, int mx_out = -1,
            const LnSide* ln = nullptr) {
    mmae_gemm_desc g = {};
    if (ln) { g.ln_gamma = ln->g; g.ln_beta = ln->b; g.ln_out = ln->out; g.ln_mean = ln->mean; g.ln_rstd = ln->rstd; g.ln_eps = ln->eps; }
    g.A = x; g.B = w; g.C = out;
    g.ab_dtype = c.ab();
    g.c_dtype = out_dtype;
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldb = K; g.ldc = N;
    g.batch = g.batch_inner = 1;
    g.bias = bias; g.resid = resid; g.ldr = N;
    g.aux = aux; g.ldaux = N; g.aux_dtype = c.act_dtype;
    g.epi = epi; g.alpha = 1.0f;
    bool x3p = false;
    if (!mxw) {
        int rc3 = 0;
        const void* a3 = nullptr;
        const void* const* t3 = x3_operand(c, x, K, w, M, K, &a3, st, &rc3);
        if (rc3) return rc3;
        if (t3) { g.A = a3; g.B = t3[1]; g.K = 3 * K; g.lda = g.ldb = 3LL * K; g.ab_dtype = MMAE_BF16; x3p = true; }
    }
    if (mxw) {
        int rc = mx_in_operand(c, x, K, M, K, mx_in, mx_out, &g.A, &g.a_scale, st);
        if (rc) return rc;
        if (mx_out >= 0) { if ((rc = mx_slot(c, mx_out, M, N, &g.q_out, &g.q_scale))) return rc; g.ldq = N; }
        g.B = mxw[0]; g.b_scale = mxw[1]; g.ab_dtype = MMAE_MXFP8; g.split_k = 1;
        return mmae_gemm(&g, st);
    }
    int tile = 0, split = 1;
    int rc = mmae_gemm_plan(&g, &tile, &split);
    if (rc) return rc;
    g.tile = tile; g.split_k = split;
    if (split > 1) {
        if ((int64_t)split * M * N > c.ws_main_elems) { mmae_set_error("composite: ws_main too small"); return MMAE_EINVAL; }
        g.ws = c.ws_main; g.ws_elems = c.ws_main_elems;
    }
    return x3p ? mmae_gemm_ex(&g, st, 1, 1.0 / 3.0) : mmae_gemm(&g, st);
}

// out[M,K] = dy[M,N] w[N,K] (+ dGELU epilogue with column-sum partials) -- ops.linear_dx.  ldy: row stride of dy.
int lin_dx(const Ctx& c, const void* dy, int64_t ldy, const void* w, void* out, int out_dtype, int M, int N, int K, void* aux, int epi,
           float* colsum_part, hipStream_t st, const void* const* mxw = nullptr, int mx_in = -1, int mx_out = -1) {
    mmae_gemm_desc g = {};
    g.A = dy; g.B = w; g.C = out;
    g.ab_dtype = c.ab_grad(); g.a_amax = c.amax_for(g.ab_dtype);
    if (c.h16() && out_dtype == MMAE_F32) g.a_amax = c.unscale();
    g.c_dtype = out_dtype;
    g.M = M; g.N = K; g.K = N;
    g.lda = ldy; g.ldb = K; g.ldc = K;
    g.b_trans = 1;
    g.batch = g.batch_inner = 1;
    g.aux = aux; g.ldaux = K; g.aux_dtype = c.act_dtype;
    g.epi = epi; g.alpha = 1.0f;
    g.colsum_part = colsum_part;
    bool x3p = false;
    if (!mxw) {                                           // dy pre-split along n; the weight's [3 N][K] copy
        int rc3 = 0;
        const void* a3 = nullptr;
        const void* const* t3 = x3_operand(c, dy, ldy, w, M, N, &a3, st, &rc3);
        if (rc3) return rc3;
        if (t3) { g.A = a3; g.lda = 3LL * N; g.B = t3[2]; g.K = 3 * N; g.ab_dtype = MMAE_BF16; x3p = true; }
    }
    if (mxw) {                                            // dy quantised along n; the weight's transposed copy [K][N], blocks along n
        int rc = mx_in_operand(c, dy, ldy, M, N, mx_in, mx_out, &g.A, &g.a_scale, st);
        if (rc) return rc;
        if (mx_out >= 0) { if ((rc = mx_slot(c, mx_out, M, K, &g.q_out, &g.q_scale))) return rc; g.ldq = K; }
        g.lda = N; g.B = mxw[2]; g.b_scale = mxw[3]; g.ldb = N; g.b_trans = 0; g.ab_dtype = MMAE_MXFP8; g.split_k = 1;
        return mmae_gemm(&g, st);
    }
    int tile = 0, split = 1;
    int rc = mmae_gemm_plan(&g, &tile, &split);
    if (rc) return rc;
    g.tile = tile; g.split_k = split;
    if (split > 1) {
        if ((int64_t)split * M * K > c.ws_main_elems) { mmae_set_error("composite: ws_main too small"); return MMAE_EINVAL; }
        g.ws = c.ws_main; g.ws_elems = c.ws_main_elems;
    }
    return x3p ? mmae_gemm_ex(&g, st, 1, 1.0 / 3.0) : mmae_gemm(&g, st);
}

// The MLP's activation pair.  bf16 activations: fc1's epilogue stores GELU'(pre-activation) in the `hpre` buffer (it has the erf terms
// in registers anyway) and fc2-dX's epilogue multiplies by it -- no transcendental work in the backward epilogue, the largest product
// of a block.  f32 activations (the exact parity mode) keep the pre-activation and re-evaluate, as autograd does.
std::atomic<int> g_gelu_grad_aux{1};                                 // mmae_gelu_grad_aux(): must not change between a forward and its backward
inline int epi_gelu(int act) { return (is16(act) && g_gelu_grad_aux.load(std::memory_order_relaxed)) ? MMAE_EPI_GELU_G : MMAE_EPI_GELU; }
inline int epi_dgelu(int act) { return (is16(act) && g_gelu_grad_aux.load(std::memory_order_relaxed)) ? MMAE_EPI_MUL : MMAE_EPI_DGELU; }

// Parameter-gradient column sums collected into ONE launch (mmae_colsum_batch): the LayerNorm partial blocks, the dGELU epilogue's
// partials and the bias gradients that no GEMM carries.  add() queues; flush() launches on the weight-gradient stream once every
// source is final there.  A full batch flushes itself, so add() must only be called when the sources queued SO FAR are ordered
// before `st` -- the callers add right behind a fork_to(compute, side).
struct ColBatch {
    mmae_colsum_job j[MMAE_COLSUM_MAX_JOBS];
    int n = 0;
    int add(const Ctx& c, const void* src, int dtype, int64_t rows, int cols, int64_t ld, int seg_w, float* const* dsts, int nseg, hipStream_t st) {
        bool any = false;
        for (int i = 0; i < nseg; ++i) any = any || dsts[i];
        if (!any) return 0;
        if (n == MMAE_COLSUM_MAX_JOBS) { const int rc = flush(c, st); if (rc) return rc; }
        mmae_colsum_job& q = j[n++];
        q = mmae_colsum_job{};
        q.src = src; q.dtype = dtype; q.rows = rows; q.cols = cols; q.ld = ld; q.seg_w = seg_w; q.nseg = nseg;
        q.unscale = c.unscale();                           // every source queued by a backward pass is a gradient
        for (int i = 0; i < nseg; ++i) q.dst[i] = dsts[i];
        return 0;
    }
    int add3(const Ctx& c, const float* part, int rows, int seg_w, float* d0, float* d1, float* d2, hipStream_t st) {
        float* dsts[3] = {d0, d1, d2};
        return add(c, part, MMAE_F32, rows, 3 * seg_w, 3 * seg_w, seg_w, dsts, 3, st);
    }
    int add1(const Ctx& c, const void* src, int dtype, int64_t rows, int cols, int64_t ld, float* dst, hipStream_t st) {
        float* dsts[1] = {dst};
        return add(c, src, dtype, rows, cols, ld, cols, dsts, 1, st);
    }
    int flush(const Ctx& c, hipStream_t st) {
        if (n == 0) return 0;
        const int64_t need = mmae_colsum_batch_ws_elems(j, n);
        if (need > c.ws_side_elems) { mmae_set_error("composite: ws_side too small for the batched column sums"); return MMAE_EINVAL; }
        const int rc = mmae_colsum_batch(j, n, c.grad_acc, c.ws_side, c.ws_side_elems, st);
        n = 0;
        return rc;
    }
};

// dw[N,K] (+)= dy[M,N]^T x[M,K]; db[N] (+)= column sums of dy (inside the GEMM where the kernel can) -- ops.linear_dw
// cb: where a bias gradient that no GEMM kernel carries is queued (one batched column-sum launch later) instead of its own launches
int lin_dw(const Ctx& c, const void* dy, int64_t ldy, const void* x, float* dw, float* db, int M, int N, int K, hipStream_t st, ColBatch* cb = nullptr) {
    if (!dw && !db) return 0;
    const int acc = c.grad_acc;
    if (c.h16() && (dw || !cb)) { mmae_set_error("composite: an fp16-storage weight gradient outside the grouped launch (widths must be multiples of 8)"); return MMAE_ESUPPORT; }
    if (dw) {
        mmae_gemm_desc g = {};
        g.A = dy; g.B = x; g.C = dw;
        g.ab_dtype = c.ab_grad(); g.a_amax = c.amax_for(g.ab_dtype);
        g.c_dtype = MMAE_F32;
        g.M = N; g.N = K; g.K = M;
        g.lda = ldy; g.ldb = K; g.ldc = K;
        g.a_trans = g.b_trans = 1;
        g.batch = g.batch_inner = 1;
        g.accumulate = acc; g.alpha = 1.0f;
        int tile = 0, split = 1;
        int rc = mmae_gemm_plan(&g, &tile, &split);
        if (rc) return rc;
        g.tile = tile; g.split_k = split;
        const bool fused_db = db && tile == 9 && g.ab_dtype == MMAE_BF16;
        const int64_t ws_used = (split > 1 ? (int64_t)split * N * K : 0) + (fused_db ? (int64_t)(split > 1 ? split : 1) * N : 0);
        if (ws_used > c.ws_side_elems) { mmae_set_error("composite: ws_side too small"); return MMAE_EINVAL; }
        if (ws_used) { g.ws = c.ws_side; g.ws_elems = c.ws_side_elems; }
        if (fused_db) { g.a_colsum = db; g.a_colsum_acc = acc; db = nullptr; }
        rc = mmae_gemm(&g, st);
        if (rc) return rc;
    }
    if (db) {
        if (cb) return cb->add1(c, dy, c.act_dtype, M, N, ldy, db, st);
        if (mmae_colsum_ws_elems(M, N) > c.ws_side_elems) { mmae_set_error("composite: ws_side too small"); return MMAE_EINVAL; }
        return mmae_colsum(dy, c.act_dtype, M, N, ldy, db, acc, c.ws_side, st);
    }
    return 0;
}

// weight gradients of one block collected into ONE grouped launch (mmae_gemm_dw_group); bf16 only.  MMAE_DW_GROUP=0 restores
// one split-K launch (+ reduce) per product.
bool dw_group_enabled() {
    static const bool on = (mmae_env_int("MMAE_DW_GROUP", 1) != 0);
    return on;
}
struct DwGroup {
    mmae_dw_group_desc g;
    bool on;
    DwGroup(const Ctx& c, int rows) : g{}, on(c.b16() && (dw_group_enabled() || c.h16())) {
        g.rows = rows; g.ab_dtype = c.act_dtype; g.accumulate = c.grad_acc; g.unscale = c.unscale();
    }
    // queue dw (+ db); returns false if this product must be issued on its own (group off / full / no weight gradient wanted)
    bool add(const void* dy, int64_t ldy, const void* x, int64_t ldx, float* dw, float* db, int n_out, int k_in) {
        if (!on || !dw || g.n >= 8 || (n_out % 8) || (k_in % 8) || (ldy % 8) || (ldx % 8)) return false;
        mmae_dw_problem& q = g.p[g.n++];
        q.dy = dy; q.ldy = ldy; q.x = x; q.ldx = ldx; q.dw = dw; q.db = db; q.n_out = n_out; q.k_in = k_in;
        return true;
    }
    int flush(const Ctx& c, hipStream_t st) {
        if (g.n == 0) return 0;
        g.ws = c.ws_side; g.ws_elems = c.ws_side_elems;
        if (mmae_gemm_dw_group_ws_elems(&g) > c.ws_side_elems) { mmae_set_error("composite: ws_side too small for the grouped weight gradients"); return MMAE_EINVAL; }
        const int rc = mmae_gemm_dw_group(&g, st);
        g.n = 0;
        return rc;
    }
};

// MX-fp8 weight gradient of one Linear (BASELINE.json configs[4]): dw[n][k] (+)= sum_m dy[m][n] x[m][k] on the scaled MFMA.  Both operands
// are re-quantised with their 32-element blocks along m (mmae_mx_quant_rows_t: e4m3 [n][Mp] and [k][Mp], Mp = m rounded up to 256, zero
// rows past m), the product is the k-contiguous MX GEMM with the contraction split into slices of whole scale groups.  Scratch = the
// tail of ws_side behind the f32 slabs.  Returns MMAE_ESUPPORT-free: false when the shape is outside the path (caller falls back to bf16).
std::atomic<int> g_mx_wgrad{1};
bool mx_lin_dw(const Ctx& c, const void* dy, int64_t ldy, const void* x, int64_t ldx, float* dw, int M, int n_out, int k_in, hipStream_t st, int* rc) {
    *rc = 0;
    if (!g_mx_wgrad.load(std::memory_order_relaxed) || c.act_dtype != MMAE_BF16 || !dw || (n_out % 64) || (k_in % 64) || (ldy % 8) || (ldx % 8) || M <= 256) return false;
    const int64_t Mp = ((int64_t)M + 255) / 256 * 256;
    const int groups = (int)(Mp / 256);
    const int64_t tiles = (int64_t)((n_out + 255) / 256) * ((k_in + 255) / 256);
    int split = (int)(256 / tiles);                                       // one workgroup per CU
    if (split > groups / 2) split = groups / 2;                           // two scale groups (8 K tiles) per slice ...
    if (split < 2) split = 2;                                             // ... or one each for 257 .. 1024 rows: the slabs + reduce carry the accumulate
    if (split > groups) return false;
    const int64_t slab = (int64_t)split * n_out * k_in;                    // f32 elements
    const int64_t qa = (int64_t)n_out * Mp, qb = (int64_t)k_in * Mp;
    const int64_t sa = mmae_mx_scale_bytes(n_out, (int)Mp), sb = mmae_mx_scale_bytes(k_in, (int)Mp);
    const int64_t need = slab * 4 + ((qa + 255) / 256 + (qb + 255) / 256 + (sa + 255) / 256 + (sb + 255) / 256) * 256;
    if (need > c.ws_side_elems * 4) return false;
    char* base = (char*)c.ws_side + slab * 4;
    unsigned char* a_q = (unsigned char*)base; base += (qa + 255) / 256 * 256;
    unsigned char* b_q = (unsigned char*)base; base += (qb + 255) / 256 * 256;
    unsigned char* a_s = (unsigned char*)base; base += (sa + 255) / 256 * 256;
    unsigned char* b_s = (unsigned char*)base;
    if ((*rc = mmae_mx_quant_rows_t(dy, MMAE_BF16, ldy, M, n_out, a_q, Mp, a_s, st))) return true;
    if ((*rc = mmae_mx_quant_rows_t(x, MMAE_BF16, ldx, M, k_in, b_q, Mp, b_s, st))) return true;
    mmae_gemm_desc g = {};
    g.A = a_q; g.B = b_q; g.C = dw;
    g.a_scale = a_s; g.b_scale = b_s;
    g.ab_dtype = MMAE_MXFP8; g.c_dtype = MMAE_F32;
    g.M = n_out; g.N = k_in; g.K = (int)Mp;
    g.lda = Mp; g.ldb = Mp; g.ldc = k_in;
    g.batch = g.batch_inner = 1;
    g.accumulate = c.grad_acc; g.alpha = 1.0f;
    g.split_k = split; g.ws = c.ws_side; g.ws_elems = slab;
    *rc = mmae_gemm(&g, st);
    return true;
}

int cast_to_act(int act, const float* src, void* dst, int64_t n, hipStream_t st) {       // f32 -> 16-bit act dtype, same units (f32 callers alias)
    return act == MMAE_BF16 ? mmae_cast_f32_to_bf16(src, dst, n, st) : (act == MMAE_F16 ? mmae_cast_f32_to_f16(src, dst, n, nullptr, st) : 0);
}
int cast_from_act(int act, const void* src, float* dst, int64_t n, hipStream_t st) {
    return act == MMAE_BF16 ? mmae_cast_bf16_to_f32(src, dst, n, st) : (act == MMAE_F16 ? mmae_cast_f16_to_f32(src, dst, n, nullptr, st) : 0);
}

int check_desc(const mmae_block_desc* d) {
    MMAE_REQUIRE(d, "block: null descriptor");
    MMAE_REQUIRE(d->B > 0 && d->N > 0 && d->D > 0 && d->heads > 0 && d->Hd > 0 && d->D % d->heads == 0, "block: bad geometry");
    MMAE_REQUIRE(d->act_dtype == MMAE_BF16 || d->act_dtype == MMAE_F16 || (d->act_dtype == MMAE_F32 && (d->f32_gemm == MMAE_F32X3 || d->f32_gemm == MMAE_F32F16)),
                 "block: activations must be bf16, fp16 (MMAE_F16), or f32 with split-bf16 (MMAE_F32X3) / fp16-operand (MMAE_F32F16) products");
    if (d->act_dtype == MMAE_F16 && (d->dp1 || d->dp2 || d->mx_w || (d->D % 32) || (d->Hd % 32))) {
        mmae_set_error("block: fp16 storage needs D, Hd multiples of 32, no stochastic depth, no MX weights"); return MMAE_ESUPPORT;
    }
    const int hd = d->D / d->heads;
    if ((hd != 32 && hd != 64) || d->N > 256) { mmae_set_error("block: geometry outside the fused attention kernel (head_dim 32/64, N <= 256)"); return MMAE_ESUPPORT; }
    MMAE_REQUIRE(d->qkv_w && d->proj_w && d->fc1_w && d->fc2_w && d->n1_w && d->n1_b && d->qkv_b && d->proj_b && d->n2_w && d->n2_b &&
                 d->fc1_b && d->fc2_b, "block: null parameter");
    MMAE_REQUIRE(d->x0 && d->ln1 && d->mean1 && d->rstd1 && d->qkv && d->lse && d->ao && d->x1 && d->ln2 && d->mean2 && d->rstd2 &&
                 d->hpre && d->hact, "block: null activation buffer");
    if (d->mx_w) {
        MMAE_REQUIRE(d->act_dtype == MMAE_BF16, "block: MX-fp8 products need bf16 activations");
        if (d->D % 256 || d->Hd % 256) { mmae_set_error("block: MX-fp8 products need D and Hd to be multiples of 256"); return MMAE_ESUPPORT; }
        const int wide = d->Hd > 3 * d->D ? d->Hd : 3 * d->D;
        MMAE_REQUIRE(d->mx_tmp && d->mx_tmp_bytes >= 2 * mmae_mx_tmp_bytes(d->B * d->N, wide), "block: mx_tmp too small");
        for (int i = 0; i < 16; ++i) MMAE_REQUIRE(d->mx_w[i], "block: null MX weight pointer");
    }
    return 0;
}

// mx_q / mx_s: also leave the MX-fp8 copy of the attention output there (bf16 activations only)
int attn_strides_fwd(const mmae_block_desc* d, hipStream_t st, void* mx_q = nullptr, void* mx_s = nullptr) {
    const int D = d->D, N = d->N, hd = D / d->heads;
    const size_t es = es_of(d->act_dtype);
    const char* qkv = (const char*)d->qkv;
    const float scale = 1.0f / sqrtf((float)hd);
    if (mx_q)
        return mmae_attn_fwd_mx(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, d->ao, d->lse, d->B, d->heads, N, N, hd, (int64_t)N * 3 * D, 3 * D,
                                (int64_t)N * 3 * D, 3 * D, (int64_t)N * 3 * D, 3 * D, (int64_t)N * D, D, scale, mx_q, mx_s, st);
    auto fn = d->act_dtype == MMAE_BF16 ? mmae_attn_fwd : (d->act_dtype == MMAE_F16 ? mmae_attn_fwd_f16 : (d->f32_gemm == MMAE_F32F16 ? mmae_attn_fwd_f32f16 : mmae_attn_fwd_f32x3));
    return fn(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, d->ao, d->lse, d->B, d->heads, N, N, hd, (int64_t)N * 3 * D, 3 * D,
              (int64_t)N * 3 * D, 3 * D, (int64_t)N * 3 * D, 3 * D, (int64_t)N * D, D, scale, st);
}

// byte-slab carving, 256-byte aligned
struct Carver {             // base 0: the returned "pointers" are plain byte offsets (layout queries)
    uintptr_t base; int64_t off;
    explicit Carver(const void* b) : base((uintptr_t)b), off(0) {}
    void* take(int64_t bytes) { void* p = (void*)(base + (uintptr_t)off); off += (bytes + 255) / 256 * 256; return p; }
    template <typename T> T* takeT(int64_t n) { return (T*)take(n * (int64_t)sizeof(T)); }
};

// ------------------------------------------------------------------------------------------------------------------
// stack layouts
// ------------------------------------------------------------------------------------------------------------------
struct BlockAct {       // saved activations of one block (inside the act slab)
    void *ln1, *qkv, *ao, *ln2, *hpre, *hact;
    float *mean1, *rstd1, *mean2, *rstd2, *lse, *x1, *x2;
};
BlockAct carve_block_act(Carver& cv, int B, int N, int D, int heads, int Hd, size_t es) {
    const int64_t R = (int64_t)B * N;
    BlockAct a;
    a.ln1 = cv.take(R * D * es); a.qkv = cv.take(R * 3 * D * es); a.ao = cv.take(R * D * es); a.ln2 = cv.take(R * D * es);
    a.hpre = cv.take(R * Hd * es); a.hact = cv.take(R * Hd * es);
    a.mean1 = cv.takeT<float>(R); a.rstd1 = cv.takeT<float>(R); a.mean2 = cv.takeT<float>(R); a.rstd2 = cv.takeT<float>(R);
    a.lse = cv.takeT<float>((int64_t)B * heads * N);
    a.x1 = cv.takeT<float>(R * D); a.x2 = cv.takeT<float>(R * D);
    return a;
}

struct BlockTmp {       // backward temporaries of one block
    void *d_hpre, *d_ln2, *d_ao, *d_qkv, *d_ln1, *dx1_act, *dx0_act, *dxs_act;
    float *dx1, *dx0, *part_h, *part1, *part2;
};
BlockTmp carve_block_tmp(Carver& cv, int B, int N, int D, int Hd, size_t es, bool bf, bool dp) {
    const int64_t R = (int64_t)B * N;
    const int nblk = mmae_layernorm_bwd_nblk(R);
    BlockTmp t;
    t.d_hpre = cv.take(R * Hd * es); t.d_ln2 = cv.take(R * D * es); t.d_ao = cv.take(R * D * es); t.d_qkv = cv.take(R * 3 * D * es);
    t.d_ln1 = cv.take(R * D * es);
    t.dx1 = cv.takeT<float>(R * D); t.dx0 = cv.takeT<float>(R * D);
    t.dx1_act = bf ? cv.take(R * D * es) : nullptr;
    t.dx0_act = bf ? cv.take(R * D * es) : nullptr;
    t.dxs_act = dp ? cv.take(R * D * es) : nullptr;
    t.part_h = cv.takeT<float>((R + 31) / 32 * Hd);
    t.part1 = cv.takeT<float>((int64_t)nblk * 3 * D); t.part2 = cv.takeT<float>((int64_t)nblk * 3 * D);
    return t;
}

constexpr int NSET = 3;      // backward temporary sets of a stack (block l uses set l % NSET)

int64_t stack_mx_tmp_bytes(const mmae_stack_desc* d) {
    return 2 * mmae_mx_tmp_bytes(d->B * d->N, d->Hd > 3 * d->D ? d->Hd : 3 * d->D);
}

bool stack_has_dp(const mmae_stack_desc* d) {
    if (!d->dp) return false;
    for (int i = 0; i < 2 * d->L; ++i) if (d->dp[i]) return true;
    return false;
}

int check_stack(const mmae_stack_desc* d) {
    MMAE_REQUIRE(d, "stack: null descriptor");
    MMAE_REQUIRE(d->L > 0 && d->B > 0 && d->N > 0 && d->D > 0 && d->heads > 0 && d->Hd > 0 && d->D % d->heads == 0, "stack: bad geometry");
    MMAE_REQUIRE(d->act_dtype == MMAE_BF16 || (d->act_dtype == MMAE_F32 && (d->f32_gemm == MMAE_F32X3 || d->f32_gemm == MMAE_F32F16)),
                 "stack: activations must be bf16, or f32 with split-bf16 (MMAE_F32X3) / fp16-operand (MMAE_F32F16) products");
    const int hd = d->D / d->heads;
    if ((hd != 32 && hd != 64) || d->N > 256) { mmae_set_error("stack: geometry outside the fused attention kernel (head_dim 32/64, N <= 256)"); return MMAE_ESUPPORT; }
    MMAE_REQUIRE(d->w && d->p && d->x && d->act, "stack: null pointer");
    if (d->mx_w) {
        MMAE_REQUIRE(d->act_dtype == MMAE_BF16, "stack: MX-fp8 products need bf16 activations");
        if (d->D % 256 || d->Hd % 256) { mmae_set_error("stack: MX-fp8 products need D and Hd to be multiples of 256"); return MMAE_ESUPPORT; }
    }
    return 0;
}

void fill_block_params(mmae_block_desc& b, int B, int N, int D, int heads, int Hd, int act, int f32g, float eps, const void* const* w,
                       const float* const* p) {
    b.B = B; b.N = N; b.D = D; b.heads = heads; b.Hd = Hd;
    b.act_dtype = act; b.f32_gemm = f32g; b.eps = eps;
    b.qkv_w = w[0]; b.proj_w = w[1]; b.fc1_w = w[2]; b.fc2_w = w[3];
    b.n1_w = p[0]; b.n1_b = p[1]; b.qkv_b = p[2]; b.proj_b = p[3]; b.n2_w = p[4]; b.n2_b = p[5]; b.fc1_b = p[6]; b.fc2_b = p[7];
}
void fill_block_act(mmae_block_desc& b, const float* x0, const BlockAct& a) {
    b.x0 = x0; b.ln1 = a.ln1; b.mean1 = a.mean1; b.rstd1 = a.rstd1; b.qkv = a.qkv; b.lse = a.lse; b.ao = a.ao; b.x1 = a.x1;
    b.ln2 = a.ln2; b.mean2 = a.mean2; b.rstd2 = a.rstd2; b.hpre = a.hpre; b.hact = a.hact; b.x2 = a.x2;
}
void fill_block_tmp(mmae_block_desc& b, const BlockTmp& t) {
    b.d_hpre = t.d_hpre; b.d_ln2 = t.d_ln2; b.d_ao = t.d_ao; b.d_qkv = t.d_qkv; b.d_ln1 = t.d_ln1; b.dx1 = t.dx1; b.dx1_act = t.dx1_act;
    b.dx0 = t.dx0; b.dx0_act = t.dx0_act; b.part_h = t.part_h; b.part1 = t.part1; b.part2 = t.part2; b.dxs_act = t.dxs_act;
}
void fill_block_grads(mmae_block_desc& b, float* const* g) {
    b.g_n1_w = g[0]; b.g_n1_b = g[1]; b.g_qkv_w = g[2]; b.g_qkv_b = g[3]; b.g_proj_w = g[4]; b.g_proj_b = g[5];
    b.g_n2_w = g[6]; b.g_n2_b = g[7]; b.g_fc1_w = g[8]; b.g_fc1_b = g[9]; b.g_fc2_w = g[10]; b.g_fc2_b = g[11];
}

// backward of blocks hi-1 ... lo of a stack whose saved activations are acts[], with temporaries tmps[l % nset].
//   dx_top / dx_top_act: gradient of block hi-1's output.
//   extra[l] (or NULL): gradient arriving at block l's output from outside (added before block l runs), l < hi-1.
//   cs_first: destination of colsum(d(stack input)) -- the bias gradient of the Linear that produced the stack input -- or NULL.
//   first_fc2_b_done: the producer of dx_top already delivered the last block's fc2 bias gradient.
// Returns through *dx_out / *dx_out_act the gradient of block lo's input (buffers of tmps[lo % nset], or dx_final when given).
struct StackRun {
    int B, N, D, heads, Hd, act, f32g, grad_acc;
    const void* const* w; const float* const* p; const float* const* dp; float* const* g;
    const float* x_in; const BlockAct* acts; BlockTmp* tmps; int nset;
    float* ws_main; int64_t ws_main_elems; float* ws_side; int64_t ws_side_elems;
    const void* const* mx_w = nullptr; void* mx_tmp = nullptr; int64_t mx_tmp_bytes = 0;
    const void* const* x3_w = nullptr; int x3_n = 0; void* x3_tmp = nullptr; int64_t x3_tmp_bytes = 0;
    const float* dy_amax = nullptr;
};

int run_blocks_bwd(const StackRun& s, int lo, int hi, const float* dx_top, const void* dx_top_act, bool top_in_sets, bool first_fc2_b_done,
                   const float* const* extra, float* cs_first, float* dx_final, hipStream_t st, hipStream_t sd,
                   const float** dx_out, const void** dx_out_act) {
    const int64_t R = (int64_t)s.B * s.N;
    const float* dx = dx_top;
    const void* dx_act = dx_top_act;
    bool fc2_done = first_fc2_b_done;
    hipEvent_t side_done[4] = {nullptr, nullptr, nullptr, nullptr};       // side_done[l % nset]: side work of block l enqueued
    int rc;
    for (int l = hi - 1; l >= lo; --l) {
        BlockTmp t = s.tmps[l % s.nset];
        // Block l overwrites set l % nset.  The SIDE stream reads, for block k: its own set k % nset and its input gradient,
        // which block k + 1 left in set (k + 1) % nset.  So the side work of block l + nset (own set) and of block l + nset - 1
        // (input gradient; only if that gradient lives in the sets: block l + nset exists, or the top gradient was carried
        // over from the previous call, top_in_sets) must have drained.  Both sit on one in-order stream: wait for the later.
        if (sd != st && l + s.nset < hi + (top_in_sets ? 1 : 0) && side_done[(l + s.nset - 1) % s.nset]) {
            if (hipStreamWaitEvent(st, side_done[(l + s.nset - 1) % s.nset], 0) != hipSuccess) { mmae_set_error("stack_bwd: stream wait failed"); return MMAE_ELAUNCH; }
        }
        if (extra && l < hi - 1 && extra[l]) {                 // more gradient for this block's output: dx += extra (in place: dx is ours)
            if ((rc = mmae_axpy_f32((float*)dx, extra[l], 1.0f, R * s.D, st))) return rc;
            if ((rc = cast_to_act(s.act, dx, (void*)dx_act, R * s.D, st))) return rc;
            fc2_done = false;
        }
        mmae_block_desc b = {};
        fill_block_params(b, s.B, s.N, s.D, s.heads, s.Hd, s.act, s.f32g, 0.f, s.w + 4 * l, s.p + 8 * l);
        fill_block_act(b, l == 0 ? s.x_in : s.acts[l - 1].x2, s.acts[l]);
        if (l == lo && dx_final) { t.dx0 = dx_final; }
        fill_block_tmp(b, t);
        b.dx = dx; b.dx_act = dx_act;
        float* none[12] = {};
        fill_block_grads(b, s.g ? s.g + 12 * l : none);
        b.dp1 = s.dp ? s.dp[2 * l] : nullptr; b.dp2 = s.dp ? s.dp[2 * l + 1] : nullptr;
        // the bias gradient of the Linear that produced this block's input = colsum(dx0): rides along with the LayerNorm-1
        // reduction -- unless more gradient joins dx0 before the producer sees it, or the producer's output was rescaled
        const bool below_clean = l > 0 && !(extra && extra[l - 1]) && !(s.dp && s.dp[2 * (l - 1) + 1]);
        b.g_cs = l > 0 ? ((below_clean && s.g) ? s.g[12 * (l - 1) + 11] : nullptr) : cs_first;
        b.grad_acc = s.grad_acc; b.fc2_b_done = fc2_done ? 1 : 0;
        b.ws_main = s.ws_main; b.ws_main_elems = s.ws_main_elems; b.ws_side = s.ws_side; b.ws_side_elems = s.ws_side_elems;
        if (s.mx_w) { b.mx_w = s.mx_w + 16 * l; b.mx_tmp = s.mx_tmp; b.mx_tmp_bytes = s.mx_tmp_bytes; }
        b.x3_w = s.x3_w; b.x3_n = s.x3_n; b.x3_tmp = s.x3_tmp; b.x3_tmp_bytes = s.x3_tmp_bytes;
        b.dy_amax = s.dy_amax;
        if ((rc = mmae_block_bwd(&b, st, sd))) return rc;
        if (sd != st) {
            hipEvent_t e = g_held_ring.next();
            if (!e || hipEventRecord(e, sd) != hipSuccess) { mmae_set_error("stack_bwd: event record failed"); return MMAE_ELAUNCH; }
            side_done[l % s.nset] = e;
        }
        fc2_done = b.g_cs != nullptr;
        dx = t.dx0;
        dx_act = is16(s.act) ? (const void*)t.dx0_act : (const void*)t.dx0;
    }
    *dx_out = dx; *dx_out_act = dx_act;
    return 0;
}

}  // namespace

int mmae_decoder_build_rows(const float* ctx, const int64_t* ids_keep, const int64_t* ids_restore, const float* mask_token,
                            const float* const* task_emb_rows, const float* pos, const int32_t* task_offsets_host, int T, int q_task, int B,
                            int n_keep, int G, int D, int n_q, float* queries, float* context, void* stream);      // tokens.hip
int mmae_decoder_build_rows_ln(const float* ctx, const int64_t* ids_keep, const int64_t* ids_restore, const float* mask_token,
                               const float* const* task_emb_rows, const float* pos, const int32_t* task_offsets_host, int T, int q_task, int B,
                               int n_keep, int G, int D, int n_q, float* queries, float* context, const BuildLn* ln, int ln_dtype, void* stream);

extern "C" {

static int block_fwd_impl(const mmae_block_desc* d, void* stream, bool ln1_done, const LnSide* next);
int mmae_block_fwd(const mmae_block_desc* d, void* stream) { return block_fwd_impl(d, stream, false, nullptr); }

// ln1_done: the producer of x0 already wrote norm1(x0) and its statistics into d->ln1 / mean1 / rstd1 (LayerNorm side output of its
// epilogue); next: what the fc2 product should write beside x2 -- the next block's norm1, or the 16-bit copy the following Linear reads
static int block_fwd_impl(const mmae_block_desc* d, void* stream, bool ln1_done, const LnSide* next) {
    int rc = check_desc(d);
    if (rc) return rc;
    MMAE_REQUIRE(d->x2, "block_fwd: null output");
    MMAE_REQUIRE(!(d->dp1 || d->dp2) || d->branch, "block_fwd: stochastic depth needs the branch scratch buffer");
    hipStream_t st = (hipStream_t)stream;
    const Ctx c = ctx_of(d);
    const int R = d->B * d->N, D = d->D, Hd = d->Hd, act = d->act_dtype;
    const void* const* mx = (d->mx_w && act == MMAE_BF16) ? d->mx_w : nullptr;
    // MX mode: the LayerNorms leave the quantised copy of their output in half 0 of the scratch; fc1's epilogue leaves the
    // quantised GELU output in half 1 (the separate passes cost 7 % of the step, profiles/r02_mxfp8_*)
    static const bool mx_fuse = (mmae_env_int("MMAE_MX_FUSE", 1) != 0);
    const int pre = (mx && mx_fuse) ? 0 : -1;
    auto ln = [&](const float* x, const float* w, const float* b, void* y, float* mu, float* rs) -> int {
        if (pre < 0) return mmae_layernorm_fwd(x, w, b, y, act, mu, rs, R, D, d->eps, st);
        void *q, *s;
        const int r0 = mx_slot(c, 0, R, D, &q, &s);
        return r0 ? r0 : mmae_layernorm_fwd_mx(x, w, b, y, mu, rs, R, D, d->eps, q, s, st);
    };
    if (!ln1_done && (rc = ln(d->x0, d->n1_w, d->n1_b, d->ln1, d->mean1, d->rstd1))) return rc;
    if ((rc = lin_fwd(c, d->ln1, d->qkv_w, d->qkv_b, d->qkv, act, R, 3 * D, D, nullptr, nullptr, MMAE_EPI_NONE, st, mx, pre))) return rc;
    // norm2 as the side output of the proj product's epilogue (D = 256: the tile spans the row), next's norm1 / cast of the fc2 product's
    const bool side = !mx && !d->dp1 && !d->dp2 && ln_side_ok(c, D, D, MMAE_F32) && (Hd % 32) == 0;
    const LnSide ln2s = {d->n2_w, d->n2_b, d->ln2, d->mean2, d->rstd2, d->eps};
    if (pre >= 0) {                                       // the attention kernel leaves the quantised copy of its output in half 0
        void *aq, *as;
        if ((rc = mx_slot(c, 0, R, D, &aq, &as)) || (rc = attn_strides_fwd(d, st, aq, as))) return rc;
    } else if ((rc = attn_strides_fwd(d, st))) return rc;
    if (d->dp1) {
        if ((rc = lin_fwd(c, d->ao, d->proj_w, d->proj_b, d->branch, MMAE_F32, R, D, D, nullptr, nullptr, MMAE_EPI_NONE, st, mx ? mx + 4 : nullptr, pre))) return rc;
        if ((rc = mmae_rowscale_add(d->x0, d->branch, d->dp1, d->x1, R, d->N, D, st))) return rc;
    } else {
        if ((rc = lin_fwd(c, d->ao, d->proj_w, d->proj_b, d->x1, MMAE_F32, R, D, D, d->x0, nullptr, MMAE_EPI_NONE, st, mx ? mx + 4 : nullptr, pre, -1,
                          side ? &ln2s : nullptr))) return rc;
    }
    if (!side && (rc = ln(d->x1, d->n2_w, d->n2_b, d->ln2, d->mean2, d->rstd2))) return rc;
    const int hq = pre < 0 ? -1 : 1;                     // quantised GELU output: half 1
    if ((rc = lin_fwd(c, d->ln2, d->fc1_w, d->fc1_b, d->hact, act, R, Hd, D, nullptr, d->hpre, epi_gelu(act), st, mx ? mx + 8 : nullptr, pre, hq))) return rc;
    if (d->dp2) {
        if ((rc = lin_fwd(c, d->hact, d->fc2_w, d->fc2_b, d->branch, MMAE_F32, R, D, Hd, nullptr, nullptr, MMAE_EPI_NONE, st, mx ? mx + 12 : nullptr, hq))) return rc;
        return mmae_rowscale_add(d->x1, d->branch, d->dp2, d->x2, R, d->N, D, st);
    }
    return lin_fwd(c, d->hact, d->fc2_w, d->fc2_b, d->x2, MMAE_F32, R, D, Hd, d->x1, nullptr, MMAE_EPI_NONE, st, mx ? mx + 12 : nullptr, hq, -1,
                   (side && next) ? next : nullptr);
}

// can block_fwd_impl's fc2 product carry a side output for this geometry?  (the adapter asks before it skips its own LayerNorm / cast)
static bool block_side_ok(const mmae_block_desc& b) {
    const Ctx c = ctx_of(&b);
    return !(b.mx_w && b.act_dtype == MMAE_BF16) && !b.dp1 && !b.dp2 && ln_side_ok(c, b.D, b.D, MMAE_F32) && (b.Hd % 32) == 0;
}

std::atomic<int> g_xattn_fuse{0};
int mmae_xattn_fuse(int on) {
    const int prev = g_xattn_fuse.load(std::memory_order_relaxed);
    if (on >= 0) g_xattn_fuse.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}

int mmae_ln_fuse(int on) {
    const int prev = g_ln_fuse.load(std::memory_order_relaxed);
    if (on >= 0) g_ln_fuse.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}

int mmae_gelu_grad_aux(int on) {
    const int prev = g_gelu_grad_aux.load(std::memory_order_relaxed);
    if (on >= 0) g_gelu_grad_aux.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}

int mmae_mx_wgrad(int on) {
    const int prev = g_mx_wgrad.load(std::memory_order_relaxed);
    if (on >= 0) g_mx_wgrad.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}

int mmae_block_bwd(const mmae_block_desc* d, void* stream, void* side_stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    MMAE_REQUIRE(d->dx && d->dx_act && d->dx0 && d->d_hpre && d->d_ln2 && d->d_ao && d->d_qkv && d->d_ln1 && d->dx1 && d->part1 && d->part2,
                 "block_bwd: null gradient / temporary buffer");
    const int act = d->act_dtype;
    MMAE_REQUIRE(act == MMAE_F32 || (d->dx1_act && d->dx0_act), "block_bwd: 16-bit activations need dx1_act / dx0_act");
    MMAE_REQUIRE(act != MMAE_F16 || d->dy_amax, "block_bwd: fp16 storage needs dy_amax (the scale its gradients are stored in)");
    MMAE_REQUIRE(!d->g_fc1_b || d->part_h, "block_bwd: part_h needed for the fc1 bias gradient");
    MMAE_REQUIRE(!(d->dp1 || d->dp2) || d->dxs_act, "block_bwd: stochastic depth needs the dxs_act scratch buffer");
    MMAE_REQUIRE(!(d->dp2 && d->fc2_b_done), "block_bwd: with a scaled MLP branch the fc2 bias gradient cannot come from the producer of dx");
    hipStream_t st = (hipStream_t)stream;
    hipStream_t sd = side_stream ? (hipStream_t)side_stream : st;
    const Ctx c = ctx_of(d);
    const int R = d->B * d->N, D = d->D, Hd = d->Hd, N = d->N, hd = D / d->heads;
    const int nblk = mmae_layernorm_bwd_nblk(R);
    const int hrows = (R + 31) / 32;
    // ---- MLP: x2 = x1 + dp2 * mlp(norm2(x1))
    const void* dm_act = d->dx_act;                                  // gradient of the MLP branch output, act dtype
    if (d->dp2) {
        if ((rc = mmae_rowscale_cast(d->dx, d->dp2, d->dxs_act, act, R, N, D, st))) return rc;
        dm_act = d->dxs_act;
    }
    float* part_h = d->g_fc1_b ? d->part_h : nullptr;
    const void* const* mx = (d->mx_w && act == MMAE_BF16) ? d->mx_w : nullptr;
    DwGroup grp(c, R);                                               // the block's four weight gradients: one launch at the end
    if (d->dp1 || d->dp2) grp.on = false;                           // stochastic depth re-uses dxs_act between the two branches
    // the block's parameter-gradient column sums (two LayerNorm partial blocks, the dGELU partials, bias gradients no GEMM
    // carries): ONE launch at the end of the block instead of two each (round 4; 72 of a cfg3 step's 176 column-sum launches
    // were the encoder's).  Not for sources that are rewritten inside the block (the rescaled copies of stochastic depth).
    ColBatch cbat;
    ColBatch* const cb_defer = (d->dp1 || d->dp2) ? nullptr : &cbat;
    // one weight gradient (+ bias gradient): MX-fp8 product in MX mode where the shape allows, else the grouped bf16 launch, else its own
    auto wgrad = [&](const void* dy, int64_t ldy, const void* x, int64_t ldx, float* dw, float* db, int n_out, int k_in) -> int {
        int r = 0;
        if (mx && mx_lin_dw(c, dy, ldy, x, ldx, dw, R, n_out, k_in, sd, &r)) {
            if (r || !db) return r;
            if (mmae_colsum_ws_elems(R, n_out) > c.ws_side_elems) { mmae_set_error("composite: ws_side too small"); return MMAE_EINVAL; }
            return mmae_colsum(dy, c.act_dtype, R, n_out, ldy, db, c.grad_acc, c.ws_side, sd);
        }
        if (grp.add(dy, ldy, x, ldx, dw, db, n_out, k_in)) return 0;
        return lin_dw(c, dy, ldy, x, dw, db, R, n_out, k_in, sd, cb_defer);
    };
    static const bool mx_fuse = (mmae_env_int("MMAE_MX_FUSE", 1) != 0);
    const int hq = (mx && mx_fuse) ? 1 : -1;                 // fc2's dX epilogue leaves the quantised d_hpre in half 1 for fc1's dX
    if ((rc = lin_dx(c, dm_act, D, d->fc2_w, d->d_hpre, act, R, D, Hd, d->hpre, epi_dgelu(act), part_h, st, mx ? mx + 12 : nullptr, -1, hq))) return rc;
    if ((rc = fork_to(st, sd))) return rc;                          // dx_act, d_hpre ready for the weight-gradient stream
    if ((rc = wgrad(dm_act, D, d->hact, Hd, d->g_fc2_w, d->fc2_b_done ? nullptr : d->g_fc2_b, D, Hd))) return rc;
    if ((rc = lin_dx(c, d->d_hpre, Hd, d->fc1_w, d->d_ln2, act, R, Hd, D, nullptr, MMAE_EPI_NONE, nullptr, st, mx ? mx + 8 : nullptr, hq))) return rc;
    if ((rc = wgrad(d->d_hpre, Hd, d->ln2, D, d->g_fc1_w, nullptr, Hd, D))) return rc;
    if (part_h && (rc = cbat.add1(c, part_h, MMAE_F32, hrows, Hd, Hd, d->g_fc1_b, sd))) return rc;
    void* dx1_act = act == MMAE_F32 ? nullptr : d->dx1_act;
    if ((rc = mmae_layernorm_bwd(d->d_ln2, act, d->x1, d->n2_w, d->mean2, d->rstd2, d->dx, d->dx1, dx1_act, act, d->part2, R, D, st))) return rc;
    // ---- attention: x1 = x0 + dp1 * attn(norm1(x0))
    const void* da_act = act == MMAE_F32 ? (const void*)d->dx1 : (const void*)d->dx1_act;
    if (d->dp1) {
        if (d->dp2 && (rc = fork_to(sd, st))) return rc;           // dxs_act is still being read by the fc2 weight gradient
        if ((rc = mmae_rowscale_cast(d->dx1, d->dp1, d->dxs_act, act, R, N, D, st))) return rc;
        da_act = d->dxs_act;
    }
    if ((rc = lin_dx(c, da_act, D, d->proj_w, d->d_ao, act, R, D, D, nullptr, MMAE_EPI_NONE, nullptr, st, mx ? mx + 4 : nullptr))) return rc;
    if ((rc = fork_to(st, sd))) return rc;                          // part2, dx1_act
    // proj's bias gradient: colsum(dx1) from the LayerNorm partials, or colsum of the rescaled copy under stochastic depth
    if ((rc = cbat.add3(c, d->part2, nblk, D, d->g_n2_w, d->g_n2_b, d->dp1 ? nullptr : d->g_proj_b, sd))) return rc;
    if ((rc = wgrad(da_act, D, d->ao, D, d->g_proj_w, d->dp1 ? d->g_proj_b : nullptr, D, D))) return rc;
    {
        const size_t es = es_of(act);
        const char* qkv = (const char*)d->qkv;
        char* dq = (char*)d->d_qkv;
        const int64_t sb3 = (int64_t)N * 3 * D, sb1 = (int64_t)N * D;
        auto fn = act == MMAE_BF16 ? mmae_attn_bwd : (act == MMAE_F16 ? mmae_attn_bwd_f16 : mmae_attn_bwd_f32x3);
        if (act == MMAE_F32 && c.ab_grad() == MMAE_F32F16) {            // fp16-operand products, dO scaled by the loss gradient's amax
            if ((rc = mmae_attn_bwd_f32f16(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, d->ao, d->d_ao, d->lse, dq, dq + (size_t)D * es,
                                           dq + (size_t)2 * D * es, d->B, d->heads, N, N, hd, sb3, 3 * D, sb3, 3 * D, sb3, 3 * D, sb1, D, sb3, 3 * D,
                                           sb3, 3 * D, sb3, 3 * D, 1.0f / sqrtf((float)hd), c.dy_amax, st))) return rc;
        } else
        if (hq >= 0) {                                    // MX mode: the kernel leaves the quantised d_qkv in half 0 for the qkv dX product
            void *gq, *gs;
            if ((rc = mx_slot(c, 0, R, 3 * D, &gq, &gs))) return rc;
            if ((rc = mmae_attn_bwd_mx(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, d->ao, d->d_ao, d->lse, dq, dq + (size_t)D * es,
                                       dq + (size_t)2 * D * es, d->B, d->heads, N, N, hd, sb3, 3 * D, sb3, 3 * D, sb3, 3 * D, sb1, D, sb3, 3 * D, sb3, 3 * D,
                                       sb3, 3 * D, 1.0f / sqrtf((float)hd), gq, gs, st))) return rc;
        } else if ((rc = fn(qkv, qkv + (size_t)D * es, qkv + (size_t)2 * D * es, d->ao, d->d_ao, d->lse, dq, dq + (size_t)D * es, dq + (size_t)2 * D * es,
                     d->B, d->heads, N, N, hd, sb3, 3 * D, sb3, 3 * D, sb3, 3 * D, sb1, D, sb3, 3 * D, sb3, 3 * D, sb3, 3 * D,
                     1.0f / sqrtf((float)hd), st))) return rc;
    }
    if ((rc = lin_dx(c, d->d_qkv, 3 * D, d->qkv_w, d->d_ln1, act, R, 3 * D, D, nullptr, MMAE_EPI_NONE, nullptr, st, mx, hq >= 0 ? 0 : -1))) return rc;
    if ((rc = fork_to(st, sd))) return rc;                          // d_qkv
    if ((rc = wgrad(d->d_qkv, 3 * D, d->ln1, D, d->g_qkv_w, d->g_qkv_b, 3 * D, D))) return rc;
    if ((rc = grp.flush(c, sd))) return rc;
    void* dx0_act = act == MMAE_F32 ? nullptr : d->dx0_act;
    if ((rc = mmae_layernorm_bwd(d->d_ln1, act, d->x0, d->n1_w, d->mean1, d->rstd1, d->dx1, d->dx0, dx0_act, act, d->part1, R, D, st))) return rc;
    if ((rc = fork_to(st, sd))) return rc;                          // part1
    if ((rc = cbat.add3(c, d->part1, nblk, D, d->g_n1_w, d->g_n1_b, d->g_cs, sd))) return rc;
    return cbat.flush(c, sd);
}

// ------------------------------------------------------------------------------------------------------------------
// stack of blocks
// ------------------------------------------------------------------------------------------------------------------
int64_t mmae_stack_act_bytes(const mmae_stack_desc* d) {
    if (!d || d->L <= 0) return 0;
    Carver cv(nullptr);
    const size_t es = d->act_dtype == MMAE_BF16 ? 2 : 4;
    for (int l = 0; l < d->L; ++l) (void)carve_block_act(cv, d->B, d->N, d->D, d->heads, d->Hd, es);
    if (stack_has_dp(d)) (void)cv.takeT<float>((int64_t)d->B * d->N * d->D);
    if (d->mx_w) (void)cv.take(stack_mx_tmp_bytes(d));
    return cv.off;
}

int64_t mmae_stack_out_offset(const mmae_stack_desc* d, int l) {
    if (!d || l < 0 || l >= d->L) return -1;
    Carver cv(nullptr);
    const size_t es = d->act_dtype == MMAE_BF16 ? 2 : 4;
    BlockAct a = {};
    for (int i = 0; i <= l; ++i) a = carve_block_act(cv, d->B, d->N, d->D, d->heads, d->Hd, es);
    return (int64_t)(uintptr_t)a.x2;
}

int64_t mmae_stack_tmp_bytes(const mmae_stack_desc* d) {
    if (!d || d->L <= 0) return 0;
    Carver cv(nullptr);
    const size_t es = d->act_dtype == MMAE_BF16 ? 2 : 4;
    const bool bf = d->act_dtype == MMAE_BF16;
    const int nset = d->L < NSET ? d->L : NSET;
    for (int s = 0; s < nset; ++s) (void)carve_block_tmp(cv, d->B, d->N, d->D, d->Hd, es, bf, stack_has_dp(d));
    if (bf) (void)cv.take((int64_t)d->B * d->N * d->D * es);       // act-dtype copy of the incoming gradient
    if (d->mx_w) (void)cv.take(stack_mx_tmp_bytes(d));
    return cv.off;
}

int mmae_stack_fwd(const mmae_stack_desc* d, void* stream) {
    int rc = check_stack(d);
    if (rc) return rc;
    MMAE_REQUIRE(d->act_bytes >= mmae_stack_act_bytes(d), "stack_fwd: activation slab too small");
    hipStream_t st = (hipStream_t)stream;
    Carver cv(d->act);
    const size_t es = d->act_dtype == MMAE_BF16 ? 2 : 4;
    const float* x = d->x;
    BlockAct acts[64];
    MMAE_REQUIRE(d->L <= 64, "stack: at most 64 blocks");
    for (int l = 0; l < d->L; ++l) acts[l] = carve_block_act(cv, d->B, d->N, d->D, d->heads, d->Hd, es);
    float* branch = stack_has_dp(d) ? cv.takeT<float>((int64_t)d->B * d->N * d->D) : nullptr;
    void* mx_tmp = d->mx_w ? cv.take(stack_mx_tmp_bytes(d)) : nullptr;
    for (int l = 0; l < d->L; ++l) {
        mmae_block_desc b = {};
        fill_block_params(b, d->B, d->N, d->D, d->heads, d->Hd, d->act_dtype, d->f32_gemm, d->eps, d->w + 4 * l, d->p + 8 * l);
        fill_block_act(b, x, acts[l]);
        b.dp1 = d->dp ? d->dp[2 * l] : nullptr; b.dp2 = d->dp ? d->dp[2 * l + 1] : nullptr; b.branch = branch;
        b.ws_main = d->ws_main; b.ws_main_elems = d->ws_main_elems;
        if (d->mx_w) { b.mx_w = d->mx_w + 16 * l; b.mx_tmp = mx_tmp; b.mx_tmp_bytes = stack_mx_tmp_bytes(d); }
        if ((rc = mmae_block_fwd(&b, st))) return rc;
        x = acts[l].x2;
    }
    return 0;
}

int mmae_stack_bwd(const mmae_stack_desc* d, void* stream, void* side_stream) {
    int rc = check_stack(d);
    if (rc) return rc;
    MMAE_REQUIRE(d->L <= 64, "stack: at most 64 blocks");
    MMAE_REQUIRE(d->tmp && d->tmp_bytes >= mmae_stack_tmp_bytes(d) && d->act_bytes >= mmae_stack_act_bytes(d), "stack_bwd: slab too small");
    MMAE_REQUIRE(0 <= d->l_begin && d->l_begin < d->l_end && d->l_end <= d->L, "stack_bwd: bad block range");
    MMAE_REQUIRE(d->d_out && d->d_out[d->L - 1], "stack_bwd: the gradient of the last block's output is required");
    MMAE_REQUIRE(d->l_begin > 0 || d->dx, "stack_bwd: null dx");
    hipStream_t st = (hipStream_t)stream;
    hipStream_t sd = side_stream ? (hipStream_t)side_stream : st;
    const size_t es = d->act_dtype == MMAE_BF16 ? 2 : 4;
    const bool bf = d->act_dtype == MMAE_BF16;
    const int64_t RD = (int64_t)d->B * d->N * d->D;
    Carver ca(d->act);
    BlockAct acts[64];
    for (int l = 0; l < d->L; ++l) acts[l] = carve_block_act(ca, d->B, d->N, d->D, d->heads, d->Hd, es);
    Carver ct(d->tmp);
    const int nset = d->L < NSET ? d->L : NSET;
    BlockTmp tmps[NSET];
    for (int s = 0; s < nset; ++s) tmps[s] = carve_block_tmp(ct, d->B, d->N, d->D, d->Hd, es, bf, stack_has_dp(d));
    void* top_act = bf ? ct.take(RD * es) : nullptr;
    void* mx_tmp = d->mx_w ? ct.take(stack_mx_tmp_bytes(d)) : nullptr;
    // a continuation call (l_end < L) may overwrite temporaries that the previous call's side-stream work still reads
    if (d->l_end < d->L && (rc = fork_to(sd, st))) return rc;
    const float* dx; const void* dx_act;
    bool fc2_done = false;
    if (d->l_end == d->L) {
        dx = d->d_out[d->L - 1];
        if (bf) { if ((rc = mmae_cast_f32_to_bf16(dx, top_act, RD, st))) return rc; dx_act = top_act; } else dx_act = dx;
    } else {
        // input gradient of block l_end, left in its temporary set by the previous call
        const BlockTmp& t = tmps[d->l_end % nset];
        dx = t.dx0; dx_act = bf ? (const void*)t.dx0_act : (const void*)t.dx0;
        const int le = d->l_end;
        const bool clean = !(d->d_out[le - 1]) && !(d->dp && d->dp[2 * (le - 1) + 1]);
        fc2_done = clean && d->g && d->g[12 * (le - 1) + 11];
        if (d->d_out[le - 1]) {
            if ((rc = mmae_axpy_f32((float*)dx, d->d_out[le - 1], 1.0f, RD, st))) return rc;
            if (bf && (rc = mmae_cast_f32_to_bf16(dx, (void*)dx_act, RD, st))) return rc;
        }
    }
    StackRun s = {d->B, d->N, d->D, d->heads, d->Hd, d->act_dtype, d->f32_gemm, d->grad_acc, d->w, d->p, d->dp, d->g, d->x, acts, tmps, nset,
                  d->ws_main, d->ws_main_elems, d->ws_side, d->ws_side_elems};
    if (d->mx_w) { s.mx_w = d->mx_w; s.mx_tmp = mx_tmp; s.mx_tmp_bytes = stack_mx_tmp_bytes(d); }
    const float* o; const void* oa;
    return run_blocks_bwd(s, d->l_begin, d->l_end, dx, dx_act, d->l_end < d->L, fc2_done, d->d_out, nullptr, d->l_begin == 0 ? d->dx : nullptr, st, sd, &o, &oa);
}

// ------------------------------------------------------------------------------------------------------------------
// SpatialOutputAdapter
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct AdapterAct {
    float *ctx_tok, *te, *queries, *context, *qmean, *qrstd, *cmean, *crstd, *lse, *x, *omean, *orstd, *x1, *pat;
    void *enc_act, *qn, *cn, *q, *kv, *xo, *on, *hpre, *hact, *h_act;
    BlockAct blocks[8];
    void* x3_tmp;
};
int kp_of(const mmae_adapter_desc* d) { return d->C * d->ph * d->pw; }
// scratch for the widest pre-split operand of an f32 adapter (rows: the B * n_q query rows; columns: the widest contraction)
int64_t adapter_x3_bytes(const mmae_adapter_desc* d) {
    if (!d->x3_w || d->act_dtype != MMAE_F32) return 0;
    int wide = d->Hd > 3 * d->D ? d->Hd : 3 * d->D;
    if (d->Denc > wide) wide = d->Denc;
    if (kp_of(d) % 32 == 0 && kp_of(d) > wide) wide = kp_of(d);
    const int64_t rq = (int64_t)d->B * d->n_q, rc = (int64_t)d->B * d->NC;
    return mmae_x3_tmp_bytes(rq > rc ? rq : rc, wide);
}

AdapterAct carve_adapter_act(Carver& cv, const mmae_adapter_desc* d) {
    const size_t es = es_of(d->act_dtype);
    const int64_t Rq = (int64_t)d->B * d->n_q, Rc = (int64_t)d->B * d->NC;
    const int D = d->D;
    AdapterAct a;
    a.enc_act = (is16(d->act_dtype) && !d->enc_act) ? cv.take(Rc * d->Denc * es) : nullptr;
    a.ctx_tok = cv.takeT<float>(Rc * D); a.te = cv.takeT<float>((int64_t)d->T * D);
    a.queries = cv.takeT<float>(Rq * D); a.context = cv.takeT<float>(Rc * D);
    a.qn = cv.take(Rq * D * es); a.cn = cv.take(Rc * D * es);
    a.qmean = cv.takeT<float>(Rq); a.qrstd = cv.takeT<float>(Rq); a.cmean = cv.takeT<float>(Rc); a.crstd = cv.takeT<float>(Rc);
    a.q = cv.take(Rq * D * es); a.kv = cv.take(Rc * 2 * D * es); a.xo = cv.take(Rq * D * es);
    a.lse = cv.takeT<float>((int64_t)d->B * d->heads * d->n_q);
    a.x = cv.takeT<float>(Rq * D); a.on = cv.take(Rq * D * es); a.omean = cv.takeT<float>(Rq); a.orstd = cv.takeT<float>(Rq);
    a.hpre = cv.take(Rq * d->Hd * es); a.hact = cv.take(Rq * d->Hd * es); a.x1 = cv.takeT<float>(Rq * D);
    for (int l = 0; l < d->depth; ++l) a.blocks[l] = carve_block_act(cv, d->B, d->n_q, D, d->heads, d->Hd, es);
    a.h_act = is16(d->act_dtype) ? cv.take(Rq * D * es) : nullptr;
    a.pat = d->pat ? d->pat : cv.takeT<float>(Rq * kp_of(d));
    a.x3_tmp = adapter_x3_bytes(d) ? cv.take(adapter_x3_bytes(d)) : nullptr;
    return a;
}

struct AdapterTmp {
    void *d_pat, *dh_act, *d_hpre, *d_on, *dx_act, *d_xo, *d_q, *d_kv, *d_qn, *d_cn, *d_ctx_act;
    float *dh, *part_h, *dx, *part_o, *d_queries, *part_q, *d_context, *part_c, *d_ctx, *part_b;
    BlockTmp blocks[8];
    void* x3_tmp;
};
int64_t ldpat_of(const mmae_adapter_desc* d) { return (kp_of(d) + 7) / 8 * 8; }

AdapterTmp carve_adapter_tmp(Carver& cv, const mmae_adapter_desc* d) {
    const size_t es = es_of(d->act_dtype);
    const bool bf = is16(d->act_dtype);
    const int64_t Rq = (int64_t)d->B * d->n_q, Rc = (int64_t)d->B * d->NC;
    const int D = d->D;
    AdapterTmp t;
    t.d_pat = cv.take(Rq * ldpat_of(d) * es);
    t.dh_act = cv.take(Rq * D * es);
    t.dh = bf ? cv.takeT<float>(Rq * D) : nullptr;
    for (int l = 0; l < d->depth; ++l) t.blocks[l] = carve_block_tmp(cv, d->B, d->n_q, D, d->Hd, es, bf, false);
    t.part_h = cv.takeT<float>((Rq + 31) / 32 * d->Hd);
    t.d_hpre = cv.take(Rq * d->Hd * es); t.d_on = cv.take(Rq * D * es);
    t.dx = cv.takeT<float>(Rq * D); t.dx_act = bf ? cv.take(Rq * D * es) : nullptr;
    t.part_o = cv.takeT<float>((int64_t)mmae_layernorm_bwd_nblk(Rq) * 3 * D);
    t.d_xo = cv.take(Rq * D * es); t.d_q = cv.take(Rq * D * es); t.d_kv = cv.take(Rc * 2 * D * es);
    t.d_qn = cv.take(Rq * D * es); t.d_cn = cv.take(Rc * D * es);
    t.d_queries = cv.takeT<float>(Rq * D); t.part_q = cv.takeT<float>((int64_t)mmae_layernorm_bwd_nblk(Rq) * 3 * D);
    t.d_context = cv.takeT<float>(Rc * D); t.part_c = cv.takeT<float>((int64_t)mmae_layernorm_bwd_nblk(Rc) * 3 * D);
    t.d_ctx = cv.takeT<float>(Rc * D); t.part_b = cv.takeT<float>((int64_t)mmae_decoder_build_bwd_nblk(d->B) * (d->T + 1) * D);
    t.d_ctx_act = bf ? cv.take(Rc * D * es) : nullptr;
    t.x3_tmp = adapter_x3_bytes(d) ? cv.take(adapter_x3_bytes(d)) : nullptr;
    return t;
}

int check_adapter(const mmae_adapter_desc* d) {
    MMAE_REQUIRE(d, "adapter: null descriptor");
    MMAE_REQUIRE(d->B > 0 && d->NC > 0 && d->Denc > 0 && d->D > 0 && d->heads > 0 && d->Hd > 0 && d->D % d->heads == 0 && d->depth >= 0 &&
                 d->depth <= 8 && d->T >= 1 && d->T <= 7 && d->q_task >= -1 && d->q_task < d->T && d->G >= 0 && d->n_q > 0 && d->NC > d->G,
                 "adapter: bad geometry");
    MMAE_REQUIRE(d->C > 0 && d->nh > 0 && d->nw > 0 && d->ph > 0 && d->pw > 0 && d->nh * d->nw == d->n_q, "adapter: bad patch geometry");
    MMAE_REQUIRE(d->act_dtype == MMAE_BF16 || d->act_dtype == MMAE_F16 || (d->act_dtype == MMAE_F32 && (d->f32_gemm == MMAE_F32X3 || d->f32_gemm == MMAE_F32F16)),
                 "adapter: activations must be bf16, fp16 (MMAE_F16), or f32 with split-bf16 (MMAE_F32X3) / fp16-operand (MMAE_F32F16) products");
    if (d->act_dtype == MMAE_F16 && ((d->D % 32) || (d->Denc % 32) || (d->Hd % 32) || (kp_of(d) % 8))) {
        mmae_set_error("adapter: fp16 storage needs D, Denc, Hd multiples of 32 and C * ph * pw a multiple of 8"); return MMAE_ESUPPORT;
    }
    const int hd = d->D / d->heads;
    if ((hd != 32 && hd != 64) || d->n_q > 256 || d->NC > 256) { mmae_set_error("adapter: geometry outside the fused attention kernel"); return MMAE_ESUPPORT; }
    if (d->act_dtype == MMAE_F32) {
        const int64_t qp = (d->n_q + 31) / 32 * 32, kp = (d->NC + 31) / 32 * 32;
        if (4 * (qp + kp) * hd * 2 + 8 * qp > 160 * 1024) { mmae_set_error("adapter: f32 attention tiles exceed the LDS"); return MMAE_ESUPPORT; }
    }
    if ((d->D % 8) || (d->Denc % 8) || (d->Hd % 8) || (kp_of(d) % 4)) { mmae_set_error("adapter: widths must be multiples of 8 (patch row of 4)"); return MMAE_ESUPPORT; }
    MMAE_REQUIRE(d->task_offsets_host && d->w && d->p && d->mask_token && d->task_emb && d->pos && d->enc && d->ids_keep && d->ids_restore && d->act,
                 "adapter: null pointer");
    MMAE_REQUIRE(is16(d->act_dtype) || d->enc_act == nullptr || d->enc_act == (const void*)d->enc, "adapter: enc_act must alias enc for f32 activations");
    return 0;
}

Ctx ctx_of(const mmae_adapter_desc* d, void* x3_tmp) {
    Ctx c{d->act_dtype, d->f32_gemm, d->grad_acc, d->ws_main, d->ws_main_elems, d->ws_side, d->ws_side_elems};
    if (x3_tmp) { c.x3_w = d->x3_w; c.x3_n = d->x3_n; c.x3_tmp = x3_tmp; c.x3_tmp_bytes = adapter_x3_bytes(d); }
    c.dy_amax = d->dy_amax;
    return c;
}

}  // namespace

int64_t mmae_adapter_act_bytes(const mmae_adapter_desc* d) {
    if (!d || d->depth < 0 || d->depth > 8) return 0;
    Carver cv(nullptr);
    (void)carve_adapter_act(cv, d);
    return cv.off;
}
int64_t mmae_adapter_tmp_bytes(const mmae_adapter_desc* d) {
    if (!d || d->depth < 0 || d->depth > 8) return 0;
    Carver cv(nullptr);
    (void)carve_adapter_tmp(cv, d);
    return cv.off;
}
int64_t mmae_adapter_pat_offset(const mmae_adapter_desc* d) {
    if (!d || d->depth < 0 || d->depth > 8) return -1;
    Carver cv(nullptr);
    AdapterAct a = carve_adapter_act(cv, d);
    return d->pat ? -1 : (int64_t)(uintptr_t)a.pat;       // caller-owned rows: not in the slab
}

int mmae_adapter_fwd(const mmae_adapter_desc* d, void* stream) {
    int rc = check_adapter(d);
    if (rc) return rc;
    MMAE_REQUIRE(d->act_bytes >= mmae_adapter_act_bytes(d), "adapter_fwd: activation slab too small");
    hipStream_t st = (hipStream_t)stream;
    const int act = d->act_dtype, D = d->D, Hd = d->Hd, B = d->B, NC = d->NC, n_q = d->n_q, T = d->T, depth = d->depth;
    const int Rq = B * n_q, Rc = B * NC, hd = D / d->heads, KP = kp_of(d);
    Carver cv(d->act);
    AdapterAct a = carve_adapter_act(cv, d);
    const Ctx c = ctx_of(d, a.x3_tmp);
    const size_t es = c.es();
    const void* const* w = d->w; const float* const* p = d->p;
    const void *qw = w[0], *kvw = w[1], *pw = w[2], *f1w = w[3], *f2w = w[4], *ow = w[5 + 4 * depth], *pcw = w[6 + 4 * depth];
    const float *qb = p[0], *kvb = p[1], *pb = p[2], *cnw = p[3], *cnb = p[4], *qnw = p[5], *qnb = p[6], *onw = p[7], *onb = p[8], *f1b = p[9],
                *f2b = p[10], *ob = p[11 + 8 * depth], *pcb = p[12 + 8 * depth];
    const void* enc_act = act == MMAE_F32 ? (const void*)d->enc : d->enc_act;
    if (!enc_act) {
        if ((rc = cast_to_act(act, d->enc, a.enc_act, (int64_t)Rc * d->Denc, st))) return rc;
        enc_act = a.enc_act;
    }
    if ((rc = lin_fwd(c, enc_act, pcw, pcb, a.ctx_tok, MMAE_F32, Rc, D, d->Denc, nullptr, nullptr, MMAE_EPI_NONE, st))) return rc;      // :258
    // the task-embedding rows are read where the parameters live (no staging copies)
    if (D <= 256) {
        // queries / context AND their query_norm / context_norm (:259-260) in one launch: a wave holds the row it assembled (round 5)
        const BuildLn ln = {qnw, qnb, cnw, cnb, a.qn, a.cn, a.qmean, a.qrstd, a.cmean, a.crstd, d->eps};
        if ((rc = mmae_decoder_build_rows_ln(a.ctx_tok, d->ids_keep, d->ids_restore, d->mask_token, (const float* const*)d->task_emb, d->pos, d->task_offsets_host,
                                             T, d->q_task, B, NC - d->G, d->G, D, n_q, a.queries, a.context, &ln, act, st))) return rc;    // :183-234
    } else {
        if ((rc = mmae_decoder_build_rows(a.ctx_tok, d->ids_keep, d->ids_restore, d->mask_token, (const float* const*)d->task_emb, d->pos, d->task_offsets_host, T,
                                          d->q_task, B, NC - d->G, d->G, D, n_q, a.queries, a.context, st))) return rc;                    // :183-234
        if ((rc = mmae_layernorm_fwd(a.queries, qnw, qnb, a.qn, act, a.qmean, a.qrstd, Rq, D, d->eps, st))) return rc;
        if ((rc = mmae_layernorm_fwd(a.context, cnw, cnb, a.cn, act, a.cmean, a.crstd, Rc, D, d->eps, st))) return rc;
    }
    // round 6: the two projections inside the attention launch (attention.hip: xattn_fwd_fused_kernel) -- policy switch, off by default
    const bool xfuse = g_xattn_fuse.load(std::memory_order_relaxed) != 0 && act == MMAE_BF16 && D == 256 && d->heads == 8 && NC <= 128 && !c.x3_w;
    if (xfuse) {
        if ((rc = mmae_xattn_fwd_fused(a.qn, a.cn, qw, qb, kvw, kvb, a.q, a.kv, a.xo, a.lse, B, d->heads, n_q, NC, D, 1.0f / sqrtf((float)hd), st))) return rc;
    } else {
    if ((rc = lin_fwd(c, a.qn, qw, qb, a.q, act, Rq, D, D, nullptr, nullptr, MMAE_EPI_NONE, st))) return rc;
    if ((rc = lin_fwd(c, a.cn, kvw, kvb, a.kv, act, Rc, 2 * D, D, nullptr, nullptr, MMAE_EPI_NONE, st))) return rc;
    {
        auto fn = act == MMAE_BF16 ? mmae_attn_fwd : (act == MMAE_F16 ? mmae_attn_fwd_f16 : (d->f32_gemm == MMAE_F32F16 ? mmae_attn_fwd_f32f16 : mmae_attn_fwd_f32x3));
        const char* kv = (const char*)a.kv;
        if ((rc = fn(a.q, kv, kv + (size_t)D * es, a.xo, a.lse, B, d->heads, n_q, NC, hd, (int64_t)n_q * D, D, (int64_t)NC * 2 * D, 2 * D,
                     (int64_t)NC * 2 * D, 2 * D, (int64_t)n_q * D, D, 1.0f / sqrtf((float)hd), st))) return rc;
    }
    }
    // every LayerNorm (and the final 16-bit cast) of the D = 256 decoder is the side output of the Linear product in front of it (round 5):
    // out_norm of the cross-attention's proj, norm1 of a block of the product that completes its input (the MLP's fc2, the block before's
    // fc2), norm2 of the block's own proj, the copy out_proj reads of the last fc2 -- 20 LayerNorm launches + 4 casts of a cfg3 step gone
    const bool side = ln_side_ok(c, D, D, MMAE_F32) && (Hd % 32) == 0;
    const LnSide ons = {onw, onb, a.on, a.omean, a.orstd, d->eps};
    if ((rc = lin_fwd(c, a.xo, pw, pb, a.x, MMAE_F32, Rq, D, D, nullptr, nullptr, MMAE_EPI_NONE, st, nullptr, -1, -1, side ? &ons : nullptr))) return rc;   // :265
    if (!side && (rc = mmae_layernorm_fwd(a.x, onw, onb, a.on, act, a.omean, a.orstd, Rq, D, d->eps, st))) return rc;
    if ((rc = lin_fwd(c, a.on, f1w, f1b, a.hact, act, Rq, Hd, D, nullptr, a.hpre, epi_gelu(act), st))) return rc;
    // what follows the product that writes stack position l's input (l = depth: the stack's output)
    auto after = [&](int l, LnSide* s) -> bool {
        if (!side) return false;
        if (l < depth) {
            const float* const* pl = p + 11 + 8 * l;
            *s = LnSide{pl[0], pl[1], a.blocks[l].ln1, a.blocks[l].mean1, a.blocks[l].rstd1, d->eps};
            return true;
        }
        if (!is16(act)) return false;
        *s = LnSide{nullptr, nullptr, a.h_act, nullptr, nullptr, 0.f};                 // the 16-bit copy out_proj reads
        return true;
    };
    LnSide nx;
    bool have = after(0, &nx);
    if ((rc = lin_fwd(c, a.hact, f2w, f2b, a.x1, MMAE_F32, Rq, D, Hd, a.x, nullptr, MMAE_EPI_NONE, st, nullptr, -1, -1, have ? &nx : nullptr))) return rc;   // :266
    const float* h = a.x1;
    bool h_act_done = have && depth == 0;
    for (int l = 0; l < depth; ++l) {                                                                                                // :271
        mmae_block_desc b = {};
        fill_block_params(b, B, n_q, D, d->heads, Hd, act, d->f32_gemm, d->eps, w + 5 + 4 * l, p + 11 + 8 * l);
        fill_block_act(b, h, a.blocks[l]);
        b.ws_main = d->ws_main; b.ws_main_elems = d->ws_main_elems;
        b.x3_w = c.x3_w; b.x3_n = c.x3_n; b.x3_tmp = c.x3_tmp; b.x3_tmp_bytes = c.x3_tmp_bytes;
        const bool ln1_done = have;                      // written by the product that completed h
        LnSide nl;
        have = block_side_ok(b) && after(l + 1, &nl);
        if ((rc = block_fwd_impl(&b, st, ln1_done, have ? &nl : nullptr))) return rc;
        h = a.blocks[l].x2;
        if (l == depth - 1) h_act_done = have;
    }
    const void* h_act = h;
    if (is16(act)) { if (!h_act_done && (rc = cast_to_act(act, h, a.h_act, (int64_t)Rq * D, st))) return rc; h_act = a.h_act; }
    if ((rc = lin_fwd(c, h_act, ow, ob, a.pat, MMAE_F32, Rq, KP, D, nullptr, nullptr, MMAE_EPI_NONE, st))) return rc;                  // :274
    if (d->img) return mmae_unpatchify(a.pat, d->img, B, d->C, d->nh, d->nw, d->ph, d->pw, st);                                       // :277-280
    return 0;
}

int mmae_adapter_bwd(const mmae_adapter_desc* d, void* stream, void* side_stream) {
    int rc = check_adapter(d);
    if (rc) return rc;
    MMAE_REQUIRE(d->tmp && d->tmp_bytes >= mmae_adapter_tmp_bytes(d) && d->act_bytes >= mmae_adapter_act_bytes(d), "adapter_bwd: slab too small");
    MMAE_REQUIRE((d->d_img != nullptr) != (d->d_pat != nullptr), "adapter_bwd: exactly one of d_img / d_pat");
    MMAE_REQUIRE(d->g && d->d_enc, "adapter_bwd: null gradient destination table / d_enc");
    hipStream_t st = (hipStream_t)stream;
    hipStream_t sd = side_stream ? (hipStream_t)side_stream : st;
    const int act = d->act_dtype, D = d->D, Hd = d->Hd, B = d->B, NC = d->NC, n_q = d->n_q, T = d->T, depth = d->depth;
    const int Rq = B * n_q, Rc = B * NC, hd = D / d->heads, KP = kp_of(d);
    const bool bf = is16(act);                               // 16-bit activations (bf16 / fp16 storage)
    MMAE_REQUIRE(act != MMAE_F16 || (d->d_pat && d->dy_amax), "adapter_bwd: fp16 storage takes the loss gradient as fp16 patch rows (d_pat) in the units of dy_amax");
    Carver ca(d->act);
    AdapterAct a = carve_adapter_act(ca, d);
    Carver ct(d->tmp);
    AdapterTmp t = carve_adapter_tmp(ct, d);
    const Ctx c = ctx_of(d, t.x3_tmp);
    const size_t es = c.es();
    const void* const* w = d->w; const float* const* p = d->p;
    const void *qw = w[0], *kvw = w[1], *pw = w[2], *f1w = w[3], *f2w = w[4], *ow = w[5 + 4 * depth], *pcw = w[6 + 4 * depth];
    const float *cnw = p[3], *qnw = p[5], *onw = p[7];
    float* const* g = d->g;
    float* g_mask = g[0];
    float* const* g_temb = g + 1;
    float* const* gb = g + 1 + T;          // q_w q_b kv_w kv_b proj_w proj_b ctxn_w ctxn_b qn_w qn_b outn_w outn_b fc1_w fc1_b fc2_w fc2_b
    float* const* gblk = gb + 16;
    float* const* gtail = gblk + 12 * depth;   // out_proj_w out_proj_b pc_w pc_b
    const void* enc_act = act == MMAE_F32 ? (const void*)d->enc : (d->enc_act ? d->enc_act : a.enc_act);
    const void* h_act = depth > 0 ? (bf ? a.h_act : (void*)a.blocks[depth - 1].x2) : (bf ? a.h_act : (void*)a.x1);

    // ---- out_proj
    const void* d_pat = d->d_pat; int64_t ldp = d->ld_pat;
    if (!d_pat) {
        ldp = ldpat_of(d);
        if (ldp != KP && hipMemsetAsync(t.d_pat, 0, (size_t)Rq * ldp * es, st) != hipSuccess) { mmae_set_error("adapter_bwd: memset failed"); return MMAE_ELAUNCH; }
        if ((rc = mmae_patchify(d->d_img, t.d_pat, act, ldp, B, d->C, d->nh, d->nw, d->ph, d->pw, st))) return rc;
        d_pat = t.d_pat;
    }
    MMAE_REQUIRE(ldp >= KP && ldp % 4 == 0, "adapter_bwd: bad ld_pat");
    // the adapter's own weight gradients: two grouped launches at the end (products over the B * n_q query rows / over the
    // B * NC context rows); whatever a group cannot take (f32 activations, odd widths) is issued on its own right away
    DwGroup grp_q(c, Rq), grp_c(c, Rc);
    ColBatch cbat;                                                   // the adapter's own parameter-gradient column sums: one launch at the end
    if ((rc = fork_to(st, sd))) return rc;
    if (!grp_q.add(d_pat, ldp, h_act, D, gtail[0], gtail[1], KP, D) && (rc = lin_dw(c, d_pat, ldp, h_act, gtail[0], gtail[1], Rq, KP, D, sd, &cbat))) return rc;
    if ((rc = lin_dx(c, d_pat, ldp, ow, t.dh_act, act, Rq, KP, D, nullptr, MMAE_EPI_NONE, nullptr, st))) return rc;
    const float* dh = (const float*)t.dh_act;
    if (bf) { if ((rc = cast_from_act(act, t.dh_act, t.dh, (int64_t)Rq * D, st))) return rc; dh = t.dh; }
    const void* dh_act = t.dh_act;
    // ---- decoder_transformer blocks
    bool fc2_done = false;
    if (depth > 0) {
        // parameter / gradient tables of the blocks in stack order
        StackRun s = {B, n_q, D, d->heads, Hd, act, d->f32_gemm, d->grad_acc, w + 5, p + 11, nullptr, gblk, a.x1, a.blocks, t.blocks, depth,
                      d->ws_main, d->ws_main_elems, d->ws_side, d->ws_side_elems};
        s.x3_w = c.x3_w; s.x3_n = c.x3_n; s.x3_tmp = c.x3_tmp; s.x3_tmp_bytes = c.x3_tmp_bytes;
        s.dy_amax = c.dy_amax;
        const float* o; const void* oa;
        if ((rc = run_blocks_bwd(s, 0, depth, dh, dh_act, false, false, nullptr, gb[15], nullptr, st, sd, &o, &oa))) return rc;
        dh = o; dh_act = oa;
        fc2_done = gb[15] != nullptr;
    }
    // ---- x1 = x + mlp(out_norm(x))
    if ((rc = lin_dx(c, dh_act, D, f2w, t.d_hpre, act, Rq, D, Hd, a.hpre, epi_dgelu(act), gb[13] ? t.part_h : nullptr, st))) return rc;
    if ((rc = fork_to(st, sd))) return rc;
    if (!grp_q.add(dh_act, D, a.hact, Hd, gb[14], fc2_done ? nullptr : gb[15], D, Hd) &&
        (rc = lin_dw(c, dh_act, D, a.hact, gb[14], fc2_done ? nullptr : gb[15], Rq, D, Hd, sd, &cbat))) return rc;
    if ((rc = lin_dx(c, t.d_hpre, Hd, f1w, t.d_on, act, Rq, Hd, D, nullptr, MMAE_EPI_NONE, nullptr, st))) return rc;
    if (!grp_q.add(t.d_hpre, Hd, a.on, D, gb[12], nullptr, Hd, D) && (rc = lin_dw(c, t.d_hpre, Hd, a.on, gb[12], nullptr, Rq, Hd, D, sd))) return rc;
    if (gb[13] && (rc = cbat.add1(c, t.part_h, MMAE_F32, (Rq + 31) / 32, Hd, Hd, gb[13], sd))) return rc;
    if ((rc = mmae_layernorm_bwd(t.d_on, act, a.x, onw, a.omean, a.orstd, dh, t.dx, bf ? t.dx_act : nullptr, act, t.part_o, Rq, D, st))) return rc;
    const void* dx_act = bf ? (const void*)t.dx_act : (const void*)t.dx;
    // ---- x = proj(attn(q, k, v))
    if ((rc = lin_dx(c, dx_act, D, pw, t.d_xo, act, Rq, D, D, nullptr, MMAE_EPI_NONE, nullptr, st))) return rc;
    if ((rc = fork_to(st, sd))) return rc;
    if ((rc = cbat.add3(c, t.part_o, mmae_layernorm_bwd_nblk(Rq), D, gb[10], gb[11], gb[5], sd))) return rc;      // outn_w, outn_b, proj_b
    if (!grp_q.add(dx_act, D, a.xo, D, gb[4], nullptr, D, D) && (rc = lin_dw(c, dx_act, D, a.xo, gb[4], nullptr, Rq, D, D, sd))) return rc;
    {
        auto fn = act == MMAE_BF16 ? mmae_attn_bwd : (act == MMAE_F16 ? mmae_attn_bwd_f16 : mmae_attn_bwd_f32x3);
        const char* kv = (const char*)a.kv; char* dkv = (char*)t.d_kv;
        if (!bf && c.ab_grad() == MMAE_F32F16) {
            if ((rc = mmae_attn_bwd_f32f16(a.q, kv, kv + (size_t)D * es, a.xo, t.d_xo, a.lse, t.d_q, dkv, dkv + (size_t)D * es, B, d->heads, n_q, NC, hd,
                                           (int64_t)n_q * D, D, (int64_t)NC * 2 * D, 2 * D, (int64_t)NC * 2 * D, 2 * D, (int64_t)n_q * D, D, (int64_t)n_q * D, D,
                                           (int64_t)NC * 2 * D, 2 * D, (int64_t)NC * 2 * D, 2 * D, 1.0f / sqrtf((float)hd), c.dy_amax, st))) return rc;
        } else
        if ((rc = fn(a.q, kv, kv + (size_t)D * es, a.xo, t.d_xo, a.lse, t.d_q, dkv, dkv + (size_t)D * es, B, d->heads, n_q, NC, hd,
                     (int64_t)n_q * D, D, (int64_t)NC * 2 * D, 2 * D, (int64_t)NC * 2 * D, 2 * D, (int64_t)n_q * D, D, (int64_t)n_q * D, D,
                     (int64_t)NC * 2 * D, 2 * D, (int64_t)NC * 2 * D, 2 * D, 1.0f / sqrtf((float)hd), st))) return rc;
    }
    if ((rc = lin_dx(c, t.d_q, D, qw, t.d_qn, act, Rq, D, D, nullptr, MMAE_EPI_NONE, nullptr, st))) return rc;
    if ((rc = lin_dx(c, t.d_kv, 2 * D, kvw, t.d_cn, act, Rc, 2 * D, D, nullptr, MMAE_EPI_NONE, nullptr, st))) return rc;
    if ((rc = fork_to(st, sd))) return rc;
    if (!grp_q.add(t.d_q, D, a.qn, D, gb[0], gb[1], D, D) && (rc = lin_dw(c, t.d_q, D, a.qn, gb[0], gb[1], Rq, D, D, sd, &cbat))) return rc;
    if (!grp_c.add(t.d_kv, 2 * D, a.cn, D, gb[2], gb[3], 2 * D, D) && (rc = lin_dw(c, t.d_kv, 2 * D, a.cn, gb[2], gb[3], Rc, 2 * D, D, sd, &cbat))) return rc;
    if ((rc = mmae_layernorm_bwd(t.d_qn, act, a.queries, qnw, a.qmean, a.qrstd, nullptr, t.d_queries, nullptr, MMAE_F32, t.part_q, Rq, D, st))) return rc;
    if ((rc = mmae_layernorm_bwd(t.d_cn, act, a.context, cnw, a.cmean, a.crstd, nullptr, t.d_context, nullptr, MMAE_F32, t.part_c, Rc, D, st))) return rc;
    if ((rc = mmae_decoder_build_bwd(t.d_queries, t.d_context, d->ids_keep, d->ids_restore, d->task_offsets_host, T, d->q_task, B, NC - d->G, d->G, D,
                                     n_q, t.d_ctx, t.part_b, st))) return rc;
    const void* d_ctx_act = t.d_ctx;
    if (bf) { if ((rc = cast_to_act(act, t.d_ctx, t.d_ctx_act, (int64_t)Rc * D, st))) return rc; d_ctx_act = t.d_ctx_act; }
    if ((rc = fork_to(st, sd))) return rc;
    if ((rc = cbat.add3(c, t.part_q, mmae_layernorm_bwd_nblk(Rq), D, gb[8], gb[9], nullptr, sd))) return rc;      // qn_w, qn_b
    if ((rc = cbat.add3(c, t.part_c, mmae_layernorm_bwd_nblk(Rc), D, gb[6], gb[7], nullptr, sd))) return rc;      // ctxn_w, ctxn_b
    {
        float* dst[8];
        for (int i = 0; i < T; ++i) dst[i] = g_temb[i];
        dst[T] = g_mask;
        if ((rc = cbat.add(c, t.part_b, MMAE_F32, mmae_decoder_build_bwd_nblk(B), (T + 1) * D, (T + 1) * D, D, dst, T + 1, sd))) return rc;
    }
    if (!grp_c.add(d_ctx_act, D, enc_act, d->Denc, gtail[2], gtail[3], D, d->Denc) &&
        (rc = lin_dw(c, d_ctx_act, D, enc_act, gtail[2], gtail[3], Rc, D, d->Denc, sd, &cbat))) return rc;
    if ((rc = grp_q.flush(c, sd))) return rc;
    if ((rc = grp_c.flush(c, sd))) return rc;
    if ((rc = cbat.flush(c, sd))) return rc;
    return lin_dx(c, d_ctx_act, D, pcw, d->d_enc, MMAE_F32, Rc, D, d->Denc, nullptr, MMAE_EPI_NONE, nullptr, st);
}

}  // extern "C"
