// Fused patch embedding + positional embedding (BASELINE.json north_star: "fused patch-embed+pos-embed").
//
//   tok[b][r][:] = W_t . patch(b, sel[b][r]) + bias_t + pos_t[p]      (t, p = owner task / patch of token sel[b][r])
//   tok[b][n_sel + g][:] = global_tok[g][:]
//
// replaces PatchedInputAdapter.forward / SemSegInputAdapter.forward (multimae/input_adapters.py:97-119, 215-241: Conv2d with
// kernel = stride = patch, + pos_emb) and the token selection of MultiMAE.forward (multimae/multimae.py:340-347: gather of the
// kept tokens, global tokens appended) in ONE kernel: gather-first (only the kept patches are ever read), the image pixels
// (or class embeddings) go HBM -> registers -> bf16 -> LDS, the product runs on the bf16 MFMA against the task's projection
// weight, and the epilogue adds bias and position rows and writes the encoder's f32 token rows.  Nothing intermediate reaches
// HBM on the forward path; the zero-padded bf16 patch rows the weight-gradient products of the backward pass contract over are
// written on the side from the same registers (optional).
//
// One workgroup (8 waves) per image.  The image's kept tokens are grouped by task in LDS (their order inside a task does not
// matter: every output row depends on its own patch only), each task is walked in groups of 64 token rows (two 32-row MFMA
// blocks), K in chunks of 128 elements double-buffered in LDS.  A wave owns output column blocks {w, w + 8, w + 16 (, w + 24)}
// of 32 columns: its weight fragments are K-contiguous 16-byte pieces read straight from L2 into registers, prefetched one
// unit (4 or 2 k-steps) ahead -- the weight stream (all tasks' weights once per image, ~3 MB for ViT-B) is what bounds the
// kernel (L2 -> CU at 64 B / clk), the MFMA work is about as long, the pixel gather hides behind both.
#include <mutex>
#include <type_traits>
#include "common.h"

namespace {

#ifdef MMAE_EMBED_TRACE
__device__ long long g_embed_trace[256];
#define EMB_STAMP() do { if (blockIdx.x == 0 && threadIdx.x == 0 && tr_n < 255) g_embed_trace[1 + tr_n++] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define EMB_STAMP() do {} while (0)
#endif

constexpr int MAX_TASKS = 8;
constexpr int KC = 128;                 // K elements per LDS chunk
constexpr int RG = 64;                  // token rows per group
constexpr int RS = KC * 2 + 16;         // LDS row stride (bytes): 16-byte pad keeps the 32 fragment rows off each other's banks
constexpr int MAX_SEL = 1024;
constexpr int MAX_PP = 64;              // semseg: pixels per patch whose class ids are cached in LDS

struct EmbSrc {
    const void* data; const float* emb; const uint16_t* w; const float* bias; const float* pos;
    int kind, C, H, W, ph, pw, k_off, k_len, n_cls, tok_off, n_tok;
};
struct EmbArgs {
    EmbSrc s[MAX_TASKS];
    const long long* sel; const float* global_tok; float* tok; uint16_t* rows;
    int T, B, n_sel, G, D, Ktot;
};

// NCB: 32-column blocks per wave (3: D <= 768, 4: D <= 1024); UB: bytes of every weight row per staging unit (128 or 64)
template <int NCB, int UB>
__global__ void __launch_bounds__(512) patch_embed_kernel(const EmbArgs a) {
    constexpr int PPR = UB / 16, RPB = 256 / UB, UKS = UB / 32, NQ = NCB * UB / 32, UPC = KC * 2 / UB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char (*s_a)[RG * RS] = reinterpret_cast<char (*)[RG * RS]>(smem);                      // [2][RG * RS]
    short* s_cls = reinterpret_cast<short*>(smem + 2 * RG * RS);                            // [RG][MAX_PP] class ids of the group's patches
    char* sbw = smem + 2 * RG * RS + RG * MAX_PP * 2 + (threadIdx.x >> 6) * (NCB * 32 * UB);  // this wave's weight image
    uint16_t* s_emb = reinterpret_cast<uint16_t*>(smem + 2 * RG * RS + RG * MAX_PP * 2 + 8 * NCB * 32 * UB);   // bf16 class-embedding table of the semseg task
    __shared__ short s_pos[MAX_SEL];
    __shared__ int s_cnt[MAX_TASKS], s_start[MAX_TASKS + 1], s_r[RG], s_p[RG];

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef MMAE_EMBED_TRACE
    int tr_n = 0;
#endif
    EMB_STAMP();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int D = a.D, n_sel = a.n_sel;
    const long long* selb = a.sel + (long long)b * n_sel;
    float* tokb = a.tok + (long long)b * (n_sel + a.G) * D;
    const auto rsT = __builtin_amdgcn_make_buffer_rsrc((void*)tokb, 0, 0x7fffffff, 0x00020000);

    auto task_of = [&](int idx) -> int {
        int t = 0;
#pragma unroll
        for (int i = 1; i < MAX_TASKS; ++i) if (i < a.T && idx >= a.s[i].tok_off) t = i;
        return t;
    };

    // ---- global tokens, and the kept tokens grouped by task ----
    for (int i = tid * 4; i < a.G * D; i += 2048) st4(tokb + (long long)n_sel * D + i, ld4(a.global_tok + i));
    if (tid < MAX_TASKS) s_cnt[tid] = 0;
    __syncthreads();
    for (int r = tid; r < n_sel; r += 512) atomicAdd(&s_cnt[task_of((int)selb[r])], 1);
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int t = 0; t < MAX_TASKS; ++t) { s_start[t] = acc; acc += s_cnt[t]; s_cnt[t] = 0; }
        s_start[MAX_TASKS] = acc;
    }
    __syncthreads();
    for (int r = tid; r < n_sel; r += 512) {
        const int t = task_of((int)selb[r]);
        s_pos[s_start[t] + atomicAdd(&s_cnt[t], 1)] = (short)r;
    }
    __syncthreads();
    EMB_STAMP();

    for (int t = 0; t < a.T; ++t) {
        const EmbSrc& s = a.s[t];
        const int n_t = s_start[t + 1] - s_start[t], st0 = s_start[t];
        const int nwp = s.W / s.pw, pp = s.ph * s.pw, k_len = s.k_len;
        const bool vec = s.kind == 0 && (s.pw & 7) == 0 && (s.W & 3) == 0;
        const int nch = (k_len + KC - 1) / KC, nunits = k_len * 2 / UB;
        // 32-bit addressing through buffer resources (one VGPR of offset per stream instead of a 64-bit pointer per access)
        const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)s.w, 0, D * k_len * 2, 0x00020000);
        const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)s.pos, 0, 0x7fffffff, 0x00020000);
        const auto rsI = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)s.data + (long long)b * s.C * s.H * s.W), 0, 0x7fffffff, 0x00020000);

        for (int g0 = 0; g0 < n_t; g0 += RG) {
            const int nrows = n_t - g0 < RG ? n_t - g0 : RG;
            const int nrb = (nrows + 31) >> 5;
            __syncthreads();                                  // the previous group's MFMA reads of s_a / epilogue reads of s_r, s_p
            if (tid < RG) {
                const int r = tid < nrows ? (int)s_pos[st0 + g0 + tid] : 0;
                s_r[tid] = r;
                s_p[tid] = (int)selb[r] - s.tok_off;
            }
            __syncthreads();
            if (a.rows) {                                     // the zero part of the side rows: every 16-byte piece outside this task's K segment
                const int zp = (a.Ktot - k_len) >> 3;
                for (int e = tid; e < nrows * zp; e += 512) {
                    const int row = e / zp, c8 = (e - row * zp) * 8;
                    *reinterpret_cast<i32x4*>(a.rows + ((long long)b * n_sel + s_r[row]) * a.Ktot + (c8 < s.k_off ? c8 : c8 + k_len)) = i32x4{0, 0, 0, 0};
                }
            }
            if (s.kind == 1) {
                if (g0 == 0) {                                // (a second semseg task would overwrite the table: the group barrier above orders that)
                    for (int e = tid; e < s.n_cls * s.C; e += 512) s_emb[e] = f32_to_bf16_bits(s.emb[e]);
                }
                for (int e = tid; e < nrows * pp; e += 512) {
                    const int row = e / pp, ij = e - row * pp, i = ij / s.pw, j = ij - i * s.pw;
                    const int p = s_p[row], py = p / nwp, px = p - py * nwp;
                    const long long c = ((const long long*)s.data)[((long long)b * s.H + py * s.ph + i) * s.W + px * s.pw + j];
                    s_cls[row * MAX_PP + ij] = (c < 0 || c >= s.n_cls) ? (short)-1 : (short)c;
                }
                __syncthreads();
            }

            EMB_STAMP();
            // The group body, compiled once per (gather kind, number of 32-row blocks): straight-line inside.
            auto run_group = [&](auto kind_c, auto nrb_c) {
                constexpr int KIND = decltype(kind_c)::value;     // 0: image, 8-pixel vector gather; 1: class embeddings from LDS; 2: image, element by element
                constexpr int NRB = decltype(nrb_c)::value;
                // 8 consecutive K elements kk .. kk + 7 of group row `row` as bf16 (zeros past the group / past K)
                auto gather8 = [&](int row, int kk) -> i32x4 {
                    if (row >= nrows || kk >= k_len) return i32x4{0, 0, 0, 0};
                    const int p = s_p[row], py = p / nwp, px = p - py * nwp;
                    i32x4 o;
                    if (KIND == 0) {
                        const int c = kk / pp, ij = kk - c * pp, i = ij / s.pw, j = ij - i * s.pw;
                        const int off = (((c * s.H + py * s.ph + i) * s.W) + px * s.pw + j) * 4;
                        const f32x4 x0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, off, 0, 0));
                        const f32x4 x1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, off + 16, 0, 0));
                        o[0] = (int)pack_bf16x2(x0[0], x0[1]); o[1] = (int)pack_bf16x2(x0[2], x0[3]);
                        o[2] = (int)pack_bf16x2(x1[0], x1[1]); o[3] = (int)pack_bf16x2(x1[2], x1[3]);
                    } else if (KIND == 1) {                       // ids and table are in LDS
                        unsigned h[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int ke = kk + e, c = ke / pp, ij = ke - c * pp;
                            const int cls = s_cls[row * MAX_PP + ij];
                            h[e] = cls < 0 ? 0u : (unsigned)s_emb[cls * s.C + c];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (int)(h[2 * e] | (h[2 * e + 1] << 16));
                    } else {                                      // patch rows that are not 8-pixel multiples (rare geometry, not tuned)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v2[2];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const int ke = kk + 2 * e + q, c = ke / pp, ij = ke - c * pp, i = ij / s.pw, j = ij - i * s.pw;
                                v2[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsI, (((c * s.H + py * s.ph + i) * s.W) + px * s.pw + j) * 4, 0, 0));
                            }
                            o[e] = (int)pack_bf16x2(v2[0], v2[1]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (a.rows) *reinterpret_cast<i32x4*>(a.rows + ((long long)b * n_sel + s_r[row]) * a.Ktot + s.k_off + kk) = o;
                    return o;
                };
                // this thread's 16-byte pieces of a chunk: piece v = tid + 512 q -> row v / 16, K offset (v % 16) * 8
                const int g_row0 = tid >> 4, g_row1 = (tid + 512) >> 4, g_kq = (tid & 15) * 8;
                i32x4 gq0, gq1 = i32x4{0, 0, 0, 0};
                auto gather_chunk = [&](int c) {
                    gq0 = gather8(g_row0, c * KC + g_kq);
                    if (NRB > 1) gq1 = gather8(g_row1, c * KC + g_kq);
                };
                auto store_chunk = [&](int buf) {
                    *reinterpret_cast<i32x4*>(&s_a[buf][g_row0 * RS + g_kq * 2]) = gq0;
                    if (NRB > 1) *reinterpret_cast<i32x4*>(&s_a[buf][g_row1 * RS + g_kq * 2]) = gq1;
                };

                // The wave's weight rows (NCB blocks of 32 output columns) go L2 -> registers -> the wave's private LDS image one
                // UNIT (UB bytes of every row) at a time: each load instruction covers whole rows of the unit (64 / PPR rows x UB
                // contiguous bytes -- with 16-byte K-contiguous fragments loaded straight into the MFMA layout every 128-byte line
                // was touched by four separate instructions and fell out of L1 in between: 4x the L2 traffic, 190 us instead of 40),
                // the fragments are then read back with ds_read_b128.  16-byte slot of (row, piece): piece ^ (row / RPB mod PPR)
                // -- conflict-free for the row-major writes and for the 32-rows-one-piece fragment reads.  Private to the wave: no
                // barrier, LDS operations of a wave execute in order.
                i32x4 breg[NQ];
                const int fcol = lane & 31, fh = lane >> 5;
                const int lrow = lane / PPR, lpiece = lane % PPR;
                const int wv = lrow * k_len * 2 + lpiece * 16;
                auto load_b = [&](int u) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        constexpr int QPB = PPR / 2;                          // load instructions per 32-row block
                        const int cbase = (wave + 8 * (q / QPB)) * 32;        // a column block past D re-reads block 0 (never stored)
                        const int row0 = (cbase < D ? cbase : 0) + (q % QPB) * (64 / PPR);
                        breg[q] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wv, row0 * k_len * 2 + u * UB, 0);
                    }
                };
                auto store_b = [&]() {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int row = q * (64 / PPR) + lrow;
                        *reinterpret_cast<i32x4*>(sbw + row * UB + ((lpiece ^ ((row / RPB) & (PPR - 1))) << 4)) = breg[q];
                    }
                };
                f32x16 acc[NRB][NCB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
                auto mma_unit = [&](const char* sa, int uu) {
#pragma unroll
                    for (int ks = 0; ks < UKS; ++ks) {
                        const int kin = (uu * UKS + ks) * 32;                 // byte offset of the k-step inside the chunk row
                        bf16x8 af[NRB], bfr[NCB];
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb) af[rb] = *reinterpret_cast<const bf16x8*>(sa + (rb * 32 + fcol) * RS + kin + fh * 16);
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb) {
                            const int row = cb * 32 + fcol;
                            bfr[cb] = *reinterpret_cast<const bf16x8*>(sbw + row * UB + (((2 * ks + fh) ^ ((row / RPB) & (PPR - 1))) << 4));
                        }
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                            for (int rb = 0; rb < NRB; ++rb)
                                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb], bfr[cb], acc[rb][cb], 0, 0, 0);
                    }
                };

                load_b(0);
                gather_chunk(0);
                store_chunk(0);
                __syncthreads();
                for (int c = 0; c < nch; ++c) {
                    if (c + 1 < nch) gather_chunk(c + 1);
                    const char* sa = s_a[c & 1];
#pragma unroll
                    for (int uu = 0; uu < UPC; ++uu) {
                        const int u = c * UPC + uu;
                        if (u < nunits) {
                            store_b();
                            if (u + 1 < nunits) load_b(u + 1);
                            mma_unit(sa, uu);
                        }
                    }
                    if (c + 1 < nch) store_chunk((c + 1) & 1);
                    __syncthreads();
                }

                EMB_STAMP();
                // epilogue: accumulator register r of lane l is output row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31 of the block.
                // The position rows of a whole 32-row block are fetched together (16 x NCB loads in flight; a few at a time left the
                // epilogue waiting on one L2 round trip after another, 24 of them per group).
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    float pe[16][NCB];
                    int orow[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int lrow = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        const bool ok = lrow < nrows;
                        const int prow = ok ? s_p[lrow] : 0;
                        orow[r] = ok ? s_r[lrow] * D * 4 : -1;
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb) {
                            const int col = (wave + 8 * cb) * 32 + fcol;
                            pe[r][cb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsP, (prow * D + (col < D ? col : 0)) * 4, 0, 0));
                        }
                    }
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        const int col = (wave + 8 * cb) * 32 + fcol;
                        const float bias = col < D ? s.bias[col] : 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (col < D && orow[r] >= 0)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (acc[rb][cb][r] + bias) + pe[r][cb]), rsT, orow[r] + col * 4, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>;
            if (s.kind == 1) { if (nrb > 1) run_group(C1{}, C2{}); else run_group(C1{}, C1{}); }
            else if (vec)    { if (nrb > 1) run_group(C0{}, C2{}); else run_group(C0{}, C1{}); }
            else             { if (nrb > 1) run_group(C2{}, C2{}); else run_group(C2{}, C1{}); }
            EMB_STAMP();
        }
    }
#ifdef MMAE_EMBED_TRACE
    if (blockIdx.x == 0 && threadIdx.x == 0) g_embed_trace[0] = tr_n;
#endif
}

size_t lds_bytes(const mmae_patch_src* srcs, int T, int D) {
    size_t emb_bytes = 0;
    for (int t = 0; t < T; ++t)
        if (srcs[t].kind == 1 && (size_t)srcs[t].n_cls * srcs[t].C * 2 > emb_bytes) emb_bytes = (size_t)srcs[t].n_cls * srcs[t].C * 2;
    const size_t wimg = D <= 768 ? (size_t)8 * 3 * 32 * 128 : (size_t)8 * 4 * 32 * 64;
    return (size_t)2 * RG * RS + (size_t)RG * MAX_PP * 2 + wimg + ((emb_bytes + 15) & ~(size_t)15);
}

int check_geometry(const mmae_patch_src* srcs, int T, int n_sel, int D) {
    if (T < 1 || T > MAX_TASKS || n_sel < 1 || n_sel > MAX_SEL || D < 32 || D > 1024 || D % 32) return 0;
    const int kq = D <= 768 ? 64 : 32;                     // whole staging units: no read past the end of a weight row
    for (int t = 0; t < T; ++t) {
        const mmae_patch_src& s = srcs[t];
        if (s.ph <= 0 || s.pw <= 0 || s.H % s.ph || s.W % s.pw) return 0;
        const int k = s.C * s.ph * s.pw;
        if (k % kq || s.k_off % 8) return 0;
        if (s.kind == 1 && (s.ph * s.pw > MAX_PP || s.n_cls <= 0 || s.n_cls > 32767)) return 0;
    }
    return lds_bytes(srcs, T, D) + 3 * 1024 <= 160 * 1024;     // + the static arrays
}

}  // namespace

extern "C" {

#ifdef MMAE_EMBED_TRACE
int mmae_debug_embed_trace(long long* out_host_256) { return (int)hipMemcpyFromSymbol(out_host_256, HIP_SYMBOL(g_embed_trace), 256 * 8); }
#endif

int mmae_patch_embed_supported(const mmae_patch_src* srcs_host, int T, int n_sel, int D) {
    return srcs_host ? check_geometry(srcs_host, T, n_sel, D) : 0;
}

int mmae_patch_embed_fwd(const mmae_patch_src* srcs_host, const void* const* w_bf16_host, const float* const* bias_host, const float* const* pos_host,
                         const int32_t* task_offsets_host, int T, const int64_t* sel, const float* global_tok, float* tok, void* rows_bf16,
                         int B, int n_sel, int G, int D, int Ktot, void* stream) {
    MMAE_REQUIRE(srcs_host && w_bf16_host && bias_host && pos_host && task_offsets_host && sel && tok && (G == 0 || global_tok), "patch_embed_fwd: null pointer");
    MMAE_REQUIRE(B > 0 && G >= 0 && check_geometry(srcs_host, T, n_sel, D), "patch_embed_fwd: geometry outside the fused kernel (mmae_patch_embed_supported)");
    MMAE_REQUIRE(!rows_bf16 || (Ktot % 8 == 0 && (uintptr_t)rows_bf16 % 16 == 0), "patch_embed_fwd: side rows need Ktot % 8 == 0 and a 16-byte aligned base");
    MMAE_REQUIRE((uintptr_t)tok % 16 == 0 && (G == 0 || (uintptr_t)global_tok % 16 == 0), "patch_embed_fwd: tok / global_tok must be 16-byte aligned");
    EmbArgs a = {};
    for (int t = 0; t < T; ++t) {
        const mmae_patch_src& s = srcs_host[t];
        MMAE_REQUIRE(s.data && w_bf16_host[t] && bias_host[t] && pos_host[t] && (s.kind == 0 || s.emb), "patch_embed_fwd: null task pointer");
        MMAE_REQUIRE((uintptr_t)w_bf16_host[t] % 16 == 0, "patch_embed_fwd: weights must be 16-byte aligned");
        MMAE_REQUIRE(s.kind != 0 || (uintptr_t)s.data % 16 == 0, "patch_embed_fwd: images must be 16-byte aligned");
        const int n_tok = task_offsets_host[t + 1] - task_offsets_host[t];
        MMAE_REQUIRE((s.H / s.ph) * (s.W / s.pw) == n_tok, "patch_embed_fwd: patch count != task tokens");
        MMAE_REQUIRE(!rows_bf16 || s.k_off + s.C * s.ph * s.pw <= Ktot, "patch_embed_fwd: task K segment outside the side rows");
        a.s[t] = EmbSrc{s.data, s.emb, (const uint16_t*)w_bf16_host[t], bias_host[t], pos_host[t], s.kind, s.C, s.H, s.W, s.ph, s.pw, s.k_off,
                        s.C * s.ph * s.pw, s.n_cls, task_offsets_host[t], n_tok};
    }
    a.sel = (const long long*)sel; a.global_tok = global_tok; a.tok = tok; a.rows = (uint16_t*)rows_bf16;
    a.T = T; a.B = B; a.n_sel = n_sel; a.G = G; a.D = D; a.Ktot = Ktot;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = lds_bytes(srcs_host, T, D);
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute((const void*)patch_embed_kernel<3, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 3 * 1024);
        (void)hipFuncSetAttribute((const void*)patch_embed_kernel<4, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 3 * 1024);
    });
    if (D <= 768) hipLaunchKernelGGL((patch_embed_kernel<3, 128>), dim3(B), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((patch_embed_kernel<4, 64>), dim3(B), dim3(512), lds, st, a);
    return mmae_check_launch("patch_embed_fwd");
}

}  // extern "C"
