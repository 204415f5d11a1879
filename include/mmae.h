/*
 * mmae.h -- C ABI of libmmae_hip.so, the MI355X (gfx950) kernel library under the
 * MultiMAE pre-training engine.
 *
 * The reference (EPFL-VILAB MultiMAE) has no FFI: every arithmetic step of its hot
 * path is a PyTorch ATen call made from multimae/*.py.  Each entry point below is
 * therefore the HIP replacement for a group of ATen calls, and cites the reference
 * call site it stands in for (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C: raw device pointers, sizes, a hipStream_t passed as void*.  No torch
 *     types, no C++ ABI coupling.
 *   - return 0 on success, a negative MMAE_E* code on a rejected argument; never
 *     throws, never allocates device memory, never synchronises.
 *   - re-entrant and thread-safe (autograd calls backward from another thread).
 *   - "act" tensors are bf16 (dtype code 1) or f32 (dtype code 0): MMAE_BF16 is the
 *     speed mode, MMAE_F32 the exact-f32 parity mode (f32-input MFMA).
 *   - all row-major; "ld*" are leading dimensions in ELEMENTS.
 */
#ifndef MMAE_H
#define MMAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMAE_ABI_VERSION 7

#define MMAE_F32  0
#define MMAE_BF16 1
#define MMAE_F32X3 2   /* GEMM only: f32 operands in memory, multiplied as split bf16 (hi+lo) on the bf16 MFMA:
                        a.b ~= ah.bh + ah.bl + al.bh, fp32 accumulate (~16 operand mantissa bits, > TF32) */
#define MMAE_F32F16 3  /* GEMM only (round 4): f32 operands in memory, each rounded to fp16 -- TF32's 11-bit significand -- for ONE product
                          per tile step on v_mfma_f32_32x32x16_f16, fp32 accumulate.  The operand precision the reference's fp32 output
                          adapters ran at on A100 (torch 1.10: allow_tf32), at a third of MMAE_F32X3's MFMA work.  Values beyond the fp16
                          range round to +-inf (never silently clamped: the inf reaches the loss / gradient norm and mmae_opt_step skips
                          and counts the update, the GradScaler contract of fp16 training); a gradient operand needs mmae_gemm_desc.a_amax (see there). */
#define MMAE_MXFP8 4   /* GEMM only: OCP MX operands -- e4m3 elements [rows][K] plus one E8M0 scale per 32 consecutive K elements in the
                          packed layout of mmae_mx_quant (a_scale / b_scale of the descriptor); block-scaled MFMA, fp32 accumulation */

#define MMAE_F16 5     /* fp16 STORAGE (round 4, the fp32 output adapters' 'h16' mode): 16-bit activations / operands / saved tensors like
                          MMAE_BF16, but IEEE half -- an 11-bit significand, TF32's, what every matmul input of the reference's fp32 adapters
                          was rounded to on A100 -- so such an adapter runs on the bf16 pipeline's kernels (v_mfma_f32_32x32x16_f16, 16-bit
                          HBM traffic) instead of f32 tensors in memory.  Residual stream, LayerNorm statistics, softmax, losses and
                          every parameter gradient stay f32.  Values beyond +-65504 are stored as +-inf -- not clamped -- so an overflow ends in a skipped,
                          counted optimiser step (mmae_opt_step's non-finite test), as under fp16 autocast.  GRADIENT tensors are stored multiplied
                          by S = 2^(4 - floor(log2 m)), m = the device scalar `dy_amax` the loss backward wrote (an upper bound of
                          |dL/dprediction|): the largest element sits in [16, 32), 11 bits of head-room above and 2^-18..2^-28 of it
                          below; the f32 sinks (weight / bias / LayerNorm / token gradients, d_enc) multiply by 1/S.  Accepted by:
                          mmae_gemm (ab/c/aux dtype, ping-pong kernels with a compiled epilogue flavour only), mmae_gemm_dw_group,
                          mmae_layernorm_*, mmae_colsum*, mmae_cast_f32_to_f16 / _f16_to_f32, mmae_attn_*_f16, mmae_masked_ce_pat_bwd (the
                          one loss whose gradient is bounded before it is written; any other gradient enters through the scaled cast),
                          mmae_block_* and mmae_adapter_* (not mmae_stack_*: the encoder stays bf16 / MX; adapter_bwd takes d_pat, not d_img). */

#define MMAE_EINVAL   (-1)   /* bad argument (shape / alignment / dtype)        */
#define MMAE_ELAUNCH  (-2)   /* hipLaunchKernel reported an error               */
#define MMAE_ESUPPORT (-3)   /* combination not implemented by this build       */

int mmae_abi_version(void);
/* sizeof the descriptor structs as this build of the library sees them (0 gemm, 1 block, 2 stack, 3 adapter, 4 opt,
 * 5 patch_src, 6 dw_group; -1 for an unknown index): a binding checks its own struct mirrors against these at load time. */
int mmae_struct_size(int which);
/* last HIP error string seen by this thread's launches (static storage). */
const char* mmae_last_error(void);

/* ------------------------------------------------------------------------- *
 * GEMM  C[M,N] (+)= alpha * A[M,K] . B[N,K]^T  with fused epilogue.
 * Replaces: nn.Linear / F.linear (multimae_utils.py:143-153,165-180,192-212;
 * output_adapters.py:138-139,154,258,274), nn.Conv2d-as-patch-linear
 * (input_adapters.py:88-91,110,206-209,232), q@k^T and attn@v
 * (multimae_utils.py:175-179,206-210) and every autograd dX / dW product of them.
 *
 * a_trans = 0: A stored [M][K] (k contiguous);  1: stored [K][M] (m contiguous).
 * b_trans = 0: B stored [N][K] (k contiguous);  1: stored [K][N] (n contiguous).
 * Batched: z in [0, batch): zo = z / batch_inner, zi = z % batch_inner;
 *          X_z = X + zo * sX_outer + zi * sX_inner  (elements).
 * Epilogue (in this order): v = alpha*acc; v += bias[n]; epi; v += resid[m,n];
 *          C = accumulate ? C + v : v.
 *   epi = MMAE_EPI_GELU : aux[m,n] = v (pre-activation, act dtype); v = gelu_erf(v)
 *   epi = MMAE_EPI_DGELU: v *= gelu_erf'(aux[m,n])
 *   epi = MMAE_EPI_GELU_G: aux[m,n] = gelu_erf'(v) (the DERIVATIVE at the pre-activation, act dtype); v = gelu_erf(v)
 *   epi = MMAE_EPI_MUL  : v *= aux[m,n]
 *         -- the same pair with the derivative evaluated once, in the forward epilogue that has the erf terms in registers
 *            anyway, so that the backward epilogue is a multiply (Mlp.forward / its backward, multimae_utils.py:137-149;
 *            the composite encoder / decoder calls use this pair for bf16 activations).  Same kernels and restrictions.
 * bf16 operands need lda/ldb % 8 == 0 and 16-byte aligned bases; a k-contiguous
 * operand may have any K as long as the bytes up to the next multiple of 8 along
 * k are readable and finite*0-safe (the engine zero-pads).
 * ------------------------------------------------------------------------- */
#define MMAE_EPI_NONE  0
#define MMAE_EPI_GELU  1
#define MMAE_EPI_DGELU 2
#define MMAE_EPI_GELU_G 3
#define MMAE_EPI_MUL   4

typedef struct mmae_gemm_desc {
    const void* A; const void* B; void* C;
    int32_t ab_dtype;            /* MMAE_F32 | MMAE_BF16 | MMAE_F32X3 | MMAE_MXFP8 */
    int32_t c_dtype;             /* MMAE_F32 | MMAE_BF16 */
    int32_t M, N, K;
    int32_t a_trans, b_trans;
    int64_t lda, ldb, ldc;
    int32_t batch, batch_inner;
    int64_t sA_outer, sA_inner, sB_outer, sB_inner, sC_outer, sC_inner;
    const float* bias;           /* [N] or NULL */
    const float* resid;          /* f32 [M][ldr] or NULL (unbatched only) */
    int64_t ldr;
    void* aux;                   /* act-dtype [M][ldaux] or NULL */
    int64_t ldaux;
    int32_t aux_dtype;
    int32_t epi;
    int32_t accumulate;          /* C += v (c_dtype must be f32) */
    float alpha;
    int32_t tile;                /* bf16 kernel variant.  0 = let the library choose (mmae_gemm_plan; env MMAE_GEMM_TILE overrides);
                                    1/2 = 128x128 / 256x128 2-stage LDS-DMA, 3/4 = same tiles, VGPR-staged, 5..8 = LDS-DMA ring
                                    with counted vmcnt (BK = 32), 9/10 = 8-wave ping-pong on 256x256 / 320x256 tiles,
                                    11 .. 14 = the round-3 experiment structures ("duo": two 4-wave workgroups per CU; 64-wide K tiles):
                                    compiled only into experiment builds (-DMMAE_EXPERIMENTS); the production library runs them on 9 / 10 */
    int32_t split_k;             /* <= 1: off; n: n K-slices, each writing a dense f32 [M][N] partial into ws,
                                    then summed into C in a fixed order (plain unbatched f32 C only).  The library
                                    never allocates: ask mmae_gemm_auto_splitk() and pass a workspace. */
    void* ws;                    /* f32 workspace of ws_elems >= split_k * M * N elements (split_k > 1) */
    int64_t ws_elems;
    float* colsum_part;          /* optional f32 [ceil(M/32)][N]: per-32-row-block column sums of the epilogue output
                                    (bias gradient of the next Linear for free); dGELU epilogue only, else must be NULL */
    float* a_colsum;             /* optional f32 [M]: receives sum_k A[k][m] -- for a dW product (A = dy, k-strided) that is the
                                    bias gradient, taken from the operand tiles already in LDS instead of a second pass over dy.
                                    Only for bf16, a_trans = 1, batch = 1 products that mmae_gemm_plan maps to tile 9; needs
                                    max(split_k, 1) * M extra workspace floats after the split-K slabs. */
    int32_t a_colsum_acc;        /* a_colsum += (else =) */
    const void* a_scale;         /* MMAE_MXFP8 only: packed E8M0 scales of A ([M] rows) and B ([N] rows) as mmae_mx_quant writes them. */
    const void* b_scale;         /* Such a product needs a_trans = b_trans = 0 (both operands [rows][K], K contiguous: quantise the
                                    transposed copy of a weight for its dX product, mmae_mx_quant_t), K % 256 == 0, batch = 1, no split_k,
                                    and one of the epilogue combinations of the training step (bias -> bf16, bias + GELU, bias
                                    [+ residual] -> f32, plain bf16 / f32, dGELU [+ colsum_part]); MMAE_ESUPPORT otherwise. */
    void* q_out; void* q_scale;  /* MMAE_MXFP8 products with a bf16 C and the bias + GELU or the dGELU epilogue: also write the MX-fp8 quantisation
                                    of C (what mmae_mx_quant would produce from it: e4m3 [M][ldq] + packed scales for M rows, N cols) -- the
                                    operand of the next MX product without a separate pass.  N % 32 == 0.  NULL = off. */
    int64_t ldq;
    const float* a_amax;         /* MMAE_F32F16 only, optional: DEVICE pointer to one f32, an upper bound m > 0 of |A| (the loss gradient's
                                    largest element, written by the masked-loss backward kernels).  The A operand is multiplied by
                                    2^-floor(log2 m) before its fp16 rounding and the accumulators by the inverse afterwards: gradient
                                    operands (1e-6 .. 1e-3) keep their 11 bits instead of falling into fp16's subnormals.  NULL = as is.
                                    MMAE_F16 products with an f32 C (a gradient leaving the fp16-storage domain, e.g. d_enc): the dy_amax scalar
                                    of MMAE_F16 above; C is multiplied by 1/S.  NULL = as is. */
    /* LayerNorm side output (round 5; 16-bit operands, f32 C, bias [+ residual] epilogue, N == 256, unbatched, unsplit, k-contiguous
       operands, K % 32 == 0: the D = 256 decoder products whose 256-column tile spans the row).  Besides C the kernel writes
       ln_out[M][N] in the operands' 16-bit format: LayerNorm(C) with ln_gamma / ln_beta / ln_eps and the row statistics into ln_mean /
       ln_rstd [M] -- the nn.LayerNorm that follows the Linear (multimae_utils.py:229-232, output_adapters.py:265-266) without its own
       pass -- or, with ln_gamma == NULL, the plain 16-bit cast of C (the autocast cast in front of the next Linear).  ln_out == NULL = off.
       MMAE_ESUPPORT for any other product. */
    const float* ln_gamma; const float* ln_beta; void* ln_out; float* ln_mean; float* ln_rstd; float ln_eps;
} mmae_gemm_desc;

int mmae_gemm(const mmae_gemm_desc* d, void* stream);
/* What mmae_gemm would pick for this descriptor on a 256-CU MI355X: *tile = bf16 kernel variant (as mmae_gemm_desc.tile;
 * d->tile / env MMAE_GEMM_TILE honoured), *split_k = number of K slices worth using (1 = do not split; > 1 only for
 * products the split path supports).  The caller sizes the workspace from *split_k and passes both back in the desc. */
int mmae_gemm_plan(const mmae_gemm_desc* d, int* tile, int* split_k);
/* suggested number of K slices for a dW-shaped (both operands k-strided) [M,N,K] product (1 = do not split) */
int mmae_gemm_auto_splitk(int M, int N, int K, int ab_dtype);
/* Compute units the library's persistent GEMM grids (bf16 ping-pong, MX-fp8, grouped weight gradients) leave free: launches
 * enqueued after the call are at most n_cu - k workgroups wide (never fewer than 16).  Those kernels hold one workgroup with the
 * whole register file and 147 KiB of LDS per CU, so nothing else can co-reside with them; a data-parallel caller reserves a few CUs
 * for RCCL's channel kernels while gradient buckets are in flight (multimae_amd/dist.py) instead of letting a full-width
 * statically strided persistent grid run its last k workgroups as a second round.  A host-side launch policy (not stream
 * ordered, process wide).  k < 0 only reads.  Returns the previous value. */
int mmae_gemm_cu_reserve(int k);
/* A second, independent slot of the same policy for callers that want narrower grids for their own reasons (the output adapters'
 * side-by-side experiment, multimae_amd/functions.py): the grids leave max(reserve, share) CUs free, so saving / restoring one slot
 * never disturbs what the other requester set.  k < 0 only reads.  Returns the previous value. */
int mmae_gemm_cu_share(int k);
/* A/B switch (default 0 = off): k > 0 splits the chip between the compute stream's persistent GEMM grids (k fewer workgroups) and the
 * grouped weight-gradient launches (sized for k CUs: at k = the number of their output tiles they run unsplit), so that a block's dX
 * chain and its weight gradients are resident side by side instead of time-slicing.  k < 0 only reads.  Returns the previous value. */
int mmae_gemm_side_cus(int k);
#ifdef MMAE_EXPERIMENTS
/* experiment builds only: resident workgroups per CU the HIP runtime reports for the duo kernel of tile code 11 / 12 with its
 * dynamic LDS size (11 must report 2: the kernel's design is two independent workgroups per CU); < 0 on error */
int mmae_gemm_duo_occupancy(int tile);
#endif

/* ------------------------------------------------------------------------- *
 * MX-fp8 operands (BASELINE.json configs[4], "fp8 MFMA path"; OCP Microscaling v1.0: e4m3 elements, one power-of-two E8M0
 * scale per 32 consecutive elements along the contraction, elements rounded to nearest even; shared exponent =
 * floor(log2(max|x|)) - 8, plus one when max|x| / 2^floor(log2 max|x|) > 1.75 so that no element saturates -- the
 * specification's plain floor rule clips the largest element of such blocks by up to 12.5 %).  The reference has no such path (it trains under fp16 autocast,
 * run_pretraining_multimae.py:514-516); these entry points replace the casts autocast inserts in front of nn.Linear.
 *   scales: uint32 S[ceil(cols/256)][rows][2]; byte j of S[g][r][h] = biased exponent of the block cols 256g + 64j + 32h .. +32
 *           of row r (exponent 0 for blocks past the last column) -- the layout the 32x32x64 scaled MFMA consumes with one dword
 *           load per lane and four K tiles (lane l < 32 supplies the scale of K 0-31 of row l, lane l + 32 that of K 32-63).  mmae_mx_scale_bytes(rows, cols) bytes.
 *   mmae_mx_quant:   x [rows][cols] (f32 / bf16, row stride ldx) -> q [rows][cols] e4m3 (row stride ldq) + scales.  cols % 32 == 0.
 *   mmae_mx_quant_t: w [n][k] -> q [k][n] e4m3 with the blocks along n (+ scales for k rows, n cols): the operand of the dX
 *                    product dx[m][k] = sum_n dy[m][n] w[n][k].  n % 32 == 0.
 *   mmae_probe_mx_mfma: one v_mfma_scale_f32_32x32x64_f8f6f4 on caller-supplied registers (8 operand dwords and one scale dword
 *                    per lane, op_sel byte selectors) -- the test-suite pins the lane layout the kernels assume with it.
 * ------------------------------------------------------------------------- */
int64_t mmae_mx_scale_bytes(int rows, int cols);
int mmae_mx_quant(const void* x, int x_dtype, int64_t ldx, int rows, int cols, void* q, int64_t ldq, void* scales, void* stream);
int mmae_mx_quant_t(const void* w, int w_dtype, int64_t ldw, int n, int k, void* q, int64_t ldq, void* scales, void* stream);
/* activation [M][C] (bf16, row stride ldx) -> q [C][ldq] e4m3 with the blocks along M (+ scales for C rows, ldq cols; rows M .. ldq - 1
 * are zeros): an operand of the weight-gradient product dW[n][k] = sum_m dy[m][n] x[m][k], whose contraction runs over the rows.
 * C % 64 == 0, ldq % 256 == 0, ldq >= M.  With both operands quantised this way mmae_gemm (MMAE_MXFP8, split_k > 1 + workspace:
 * slices of whole 256-row scale groups write dense f32 slabs that are summed in a fixed order) computes dW on the scaled MFMA. */
int mmae_mx_quant_rows_t(const void* x, int x_dtype, int64_t ldx, int M, int C, void* q, int64_t ldq, void* scales, void* stream);
/* Process-wide policy for the encoder stack's weight gradients in MX-fp8 mode (mmae_stack_bwd with mx_w set): 1 = the four dW
 * products of a block run on the scaled MFMA from row-blocked quantised copies of dy and x (scratch carved from ws_side),
 * 0 = they stay bf16 (grouped launch).  Default 1.  Returns the previous value; on < 0 only queries. */
int mmae_mx_wgrad(int on);
/* Process-wide policy of the composite encoder / decoder calls with bf16 activations: 1 (default) = the MLP's fc1 epilogue stores
 * GELU'(pre-activation) (MMAE_EPI_GELU_G) and fc2's dX epilogue multiplies by it (MMAE_EPI_MUL); 0 = it stores the pre-activation and
 * the backward epilogue re-evaluates the derivative (MMAE_EPI_GELU / MMAE_EPI_DGELU), as autograd does.  Must not change between a
 * forward call and its backward.  Returns the previous value; on < 0 only queries. */
int mmae_gelu_grad_aux(int on);
/* Policy of the composite calls (default on): in a D = 256 output adapter with 16-bit activations every nn.LayerNorm forward (and the final
 * 16-bit cast) is the LayerNorm side output of the Linear product in front of it (mmae_gemm_desc.ln_out) instead of its own launch.
 * on < 0 only reads.  Returns the previous value. */
int mmae_ln_fuse(int on);
/* Policy of mmae_adapter_fwd (default off -- measured: profiles/r06_xattn_fused_ab.txt): the cross-attention of a D = 256 bf16 output adapter
 * as ONE launch, mmae_xattn_fwd_fused (q-projection + kv-projection + softmax + PV), instead of two Linear products and the attention core.
 * on < 0 only reads.  Returns the previous value. */
int mmae_xattn_fuse(int on);
/* zero a packed scale array (only needed by producers that write the blocks of a width that is not a multiple of 256) */
int mmae_mx_scale_clear(void* scales, int rows, int cols, void* stream);
/* nn.LayerNorm forward (mmae_layernorm_fwd, bf16 y) that also emits the MX-fp8 quantisation of y -- bit-identical to
 * mmae_mx_quant applied to y -- for the Linear that follows (multimae_utils.py:230-231).  D % 32 == 0. */
int mmae_layernorm_fwd_mx(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd, int64_t R, int D,
                          float eps, void* q, void* scales, void* stream);
/* mmae_attn_fwd / mmae_attn_bwd (bf16) that also emit the MX-fp8 quantisation of their outputs -- bit-identical to mmae_mx_quant
 * of them -- for the projection / qkv-dX product that follows: forward o must be dense [B * Nq][H * head_dim]; backward dq, dk, dv
 * must be the column slices of one packed [B * N][3 * H * head_dim] tensor (self-attention), whose quantisation is written. */
int mmae_attn_fwd_mx(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                     int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr,
                     float scale, void* mx_q, void* mx_scale, void* stream);
int mmae_attn_bwd_mx(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq, void* dk,
                     void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb,
                     int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr, int64_t dk_sb, int64_t dk_sr, int64_t dv_sb,
                     int64_t dv_sr, float scale, void* mx_q, void* mx_scale, void* stream);
/* scratch a composite call needs to quantise one [rows][cols] activation operand (bytes + scales, 256-byte aligned parts) */
int64_t mmae_mx_tmp_bytes(int rows, int cols);
/* Quantise n weights w[i] = [n_out[i]][k_in[i]] (f32 / bf16, contiguous) for both products that read them: dst[4 i] e4m3
 * [n_out][k_in] + dst[4 i + 1] its scales (forward: contraction over k_in), dst[4 i + 2] e4m3 of the transpose [k_in][n_out] +
 * dst[4 i + 3] its scales (dX: contraction over n_out).  Sizes: n_out * k_in bytes and mmae_mx_scale_bytes(n_out, k_in) /
 * mmae_mx_scale_bytes(k_in, n_out).  Once per optimiser step. */
int mmae_mx_prepare_weights(int n, const void* const* w, int w_dtype, const int32_t* n_out, const int32_t* k_in, void* const* dst, void* stream);
int mmae_probe_mx_mfma(const int32_t* a_64x8, const int32_t* b_64x8, const int32_t* scale_a_64, const int32_t* scale_b_64, int opsel_a, int opsel_b,
                       float* out_64x16, void* stream);

/* ------------------------------------------------------------------------- *
 * Launch timing of the MFMA GEMM entry points (mmae_gemm, mmae_gemm_dw_group -- also when they are reached through the
 * composite calls): while enabled, every call is bracketed by two hipEvents recorded on ITS launch stream; mmae_gemm_timing_read
 * waits for them and returns, per operand class (0: bf16, 1: f32 / split-bf16, 2: MX-fp8 -- arrays of THREE entries), the summed
 * launch time [ms], the algorithmic FLOPs (2 M N K) and the number of calls.  For bench.py's roofline line (run the step single-stream while measuring: with
 * other streams active a bracket also contains the time the launch shares the CUs).  Off by default; costs nothing then.
 * ------------------------------------------------------------------------- */
int mmae_gemm_timing_enable(int on);
int mmae_gemm_timing_read(double* ms3, double* flop3, int64_t* calls3);
/* ... and the ALGORITHMIC HBM bytes of the same launches per class (operands once, C and every epilogue stream -- aux, residual, LayerNorm
 * side output -- once; a grouped weight gradient: dY and X once, dW once, not its partial slabs): what bench.py prints beside the counter
 * traffic (roofline.algorithmic_bytes_per_launch).  ABI v7. */
int mmae_gemm_timing_read_bytes(double* bytes3);

/* ------------------------------------------------------------------------- *
 * Grouped weight gradients: up to 8 products dw_i[n_out_i][k_in_i] (+)= dy_i[rows][n_out_i]^T . x_i[rows][k_in_i] (the dW of
 * nn.Linear layers that saw the same rows, e.g. the four of a transformer block, multimae_utils.py:143-153,165-180) in ONE
 * MFMA launch + ONE reduction launch; db_i[n_out_i] (+)= column sums of dy_i ride along (NULL = not wanted).  bf16 / fp16 operands
 * (MMAE_ESUPPORT otherwise), widths and leading dimensions multiples of 8, dw contiguous.  Every product is cut into the
 * same number of row slices -- as many as fill the chip once (split_k = 0), at least 16 x 32 rows each -- whose f32 partials
 * go through the caller's workspace (mmae_gemm_dw_group_ws_elems) and are summed in a fixed order (deterministic).
 * ------------------------------------------------------------------------- */
typedef struct mmae_dw_problem {
    const void* dy; int64_t ldy;
    const void* x; int64_t ldx;
    float* dw; float* db;
    int32_t n_out, k_in;
} mmae_dw_problem;

typedef struct mmae_dw_group_desc {
    int32_t n, rows, ab_dtype, accumulate, split_k;
    mmae_dw_problem p[8];
    float* ws; int64_t ws_elems;
    const float* unscale;        /* MMAE_F16 operands: the dy_amax device scalar -- dy is stored scaled by S (see MMAE_F16), dw / db get 1/S.  NULL = as is */
} mmae_dw_group_desc;

int64_t mmae_gemm_dw_group_ws_elems(const mmae_dw_group_desc* d);
int mmae_gemm_dw_group(const mmae_dw_group_desc* d, void* stream);

/* ------------------------------------------------------------------------- *
 * LayerNorm (biased variance, eps inside rsqrt), rows of width D.
 * Replaces nn.LayerNorm(eps=1e-6): multimae_utils.py:222,225,230-231;
 * output_adapters.py:120-122,265-266.
 *   fwd:  y = (x-mean)*rstd*gamma+beta;   x f32 [R][D]; y act dtype; saves mean,rstd
 *   bwd:  dx_out = (dx_in ? dx_in : 0) + LN'(dy);  also writes an act-dtype copy
 *         dx_act (may be NULL); writes per-block partial sums part[nblk][3][D]
 *         (nblk = mmae_layernorm_bwd_nblk(R)): dgamma, dbeta and the column sums of dx_out
 *         (= bias gradient of the Linear that produced this residual-stream value),
 *         reduced by mmae_colsum / mmae_colsum_partials.
 * ------------------------------------------------------------------------- */
int mmae_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                       float* mean, float* rstd, int64_t R, int D, float eps, void* stream);
int mmae_layernorm_bwd_nblk(int64_t R);
int mmae_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma,
                       const float* mean, const float* rstd, const float* dx_in, float* dx_out,
                       void* dx_act, int dx_act_dtype, float* part, int64_t R, int D, void* stream);
/* out[c] (+)= sum_r part[r][c], part f32 [nrows][ncols] */
int mmae_colsum_partials(const float* part, float* out, int nrows, int ncols, int accumulate, void* stream);

/* column sums of an activation-gradient matrix: out[n] (+)= sum_m dy[m][n]
 * (bias gradients of every Linear).  dy act dtype [M][ld].  ws: f32 scratch of
 * mmae_colsum_ws_elems(M, N) elements. */
int64_t mmae_colsum_ws_elems(int64_t M, int N);
int mmae_colsum(const void* dy, int dtype, int64_t M, int N, int64_t ld, float* out, int accumulate,
                float* ws, void* stream);
/* same reduction, scattered: column c is delivered to dsts_host[c / seg_w][c % seg_w] (host array of nseg <= 8
 * device pointers, NULL = drop that segment).  One pass over a partial block feeds several parameter
 * gradients, e.g. LayerNorm's [dgamma | dbeta | bias gradient of the producing Linear]
 * (autograd's three AccumulateGrad nodes of multimae_utils.py:222-232). */
int mmae_colsum_scatter(const void* dy, int dtype, int64_t M, int N, int64_t ld, int seg_w, const void* dsts_host,
                        int nseg, int accumulate, float* ws, void* stream);

/* Up to MMAE_COLSUM_MAX_JOBS such reductions in ONE pair of launches (round 4: the parameter-gradient column sums of a transformer
 * block -- LayerNorm partial blocks, the dGELU epilogue's partials, f32 bias gradients -- were 6 launches of 5-6 us, 176 per cfg3 step).
 * A job is a mmae_colsum_scatter call: src act dtype [rows][ld], `cols` columns, column c delivered to dst[c / seg_w][c % seg_w].
 * Launch 1: row slices of every job -> ws; launch 2: the slices of every 256-column group summed in a fixed order and scattered.
 * (A one-launch form with a last-workgroup reduction behind __threadfence() was built first and measured 140 us per call: an
 * agent-scope release on gfx950 writes back the WHOLE L2 of the XCD, once per workgroup.)  Deterministic.
 * ws: mmae_colsum_batch_ws_elems() floats.  All jobs share `accumulate`. */
#define MMAE_COLSUM_MAX_JOBS 8
typedef struct mmae_colsum_job {
    const void* src; int32_t dtype; int32_t cols; int64_t rows; int64_t ld;
    int32_t seg_w; int32_t nseg; float* dst[8];
    const float* unscale;        /* the source holds gradients of an MMAE_F16 adapter (stored scaled by S, see MMAE_F16): its dy_amax device scalar,
                                    the sums are multiplied by 1/S.  NULL = as is */
} mmae_colsum_job;
int64_t mmae_colsum_batch_ws_elems(const mmae_colsum_job* jobs, int n);
int mmae_colsum_batch(const mmae_colsum_job* jobs, int n, int accumulate, float* ws, int64_t ws_elems, void* stream);

/* ------------------------------------------------------------------------- *
 * Row softmax over materialised attention scores (unfused attention path and the
 * f32 parity mode).  Replaces attn.softmax(dim=-1), multimae_utils.py:176,207.
 *   fwd: P[r][0:n] = softmax(scale * S[r][0:n]); P[r][n:ldp] = 0.  S f32, P act dtype.
 *   bwd: dS = scale * P .* (dP - sum_j dP_j P_j); dS[r][n:ld] = 0.  dP f32, dS act dtype.
 * ------------------------------------------------------------------------- */
int mmae_softmax_fwd(const float* S, int64_t lds_, void* P, int p_dtype, int64_t ldp, int64_t rows, int n,
                     float scale, void* stream);
int mmae_softmax_bwd(const void* P, int p_dtype, int64_t ldp, const float* dP, int64_t lddp, void* dS,
                     int64_t ldds, int64_t rows, int n, float scale, void* stream);

/* ------------------------------------------------------------------------- *
 * Fused attention, bf16, one workgroup per (batch, head); head_dim 32 or 64; 1 <= Nq, Nk <= 256
 * (the whole K and V of a head live in LDS; scores never reach HBM).  Replaces the
 * Attention.forward / CrossAttention.forward cores, multimae_utils.py:175-179, 206-210, and their
 * autograd.  Operands are addressed as base + b*s_b + row*s_r + h*head_dim (elements), so q/k/v
 * may be column slices of a packed qkv / kv activation.  lse f32 [B][H][Nq] = max + log(sum) of
 * the scaled scores (saved by fwd, consumed by bwd).  d_o shares o's strides.  (o is no longer read by bwd: since round 4
 * delta = sum_j P_j dP_j is formed from the kernel's own fp32 P and dP -- the flash-attention shortcut rowsum(dO . O) with the
 * stored 16-bit O cost 22 % on dQ of the cfg5 decoders, profiles/r04_xattn_delta_probe.txt; the argument stays for ABI stability.)
 * ------------------------------------------------------------------------- */
/* The fusion BASELINE.json's north star names, for CrossAttention.forward (multimae_utils.py:199-214) at D = H * 32 = 256, Nk <= 128, bf16:
 * q = qn Wq^T + bq, [k | v] = cn Wkv^T + bkv, out = softmax(q k^T * scale) v per head, one workgroup per (image, head), the context rows staged
 * in LDS once and projected there.  qn [B * Nq][D], cn [B * Nk][D] dense rows; wq [D][D], wkv [2 D][D] in nn.Linear layout.  Writes q [B * Nq][D]
 * and kv [B * Nk][2 D] (the tensors mmae_attn_bwd and the weight gradients read), out [B * Nq][D] and lse [B][H][Nq] -- bit-identical to
 * mmae_gemm (bias -> bf16) x 2 + mmae_attn_fwd.  MMAE_ESUPPORT for any other geometry.  ABI v7. */
int mmae_xattn_fwd_fused(const void* qn, const void* cn, const void* wq, const float* bq, const void* wkv, const float* bkv, void* q, void* kv,
                         void* out, float* lse, int B, int H, int Nq, int Nk, int D, float scale, void* stream);
int mmae_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                  int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb,
                  int64_t o_sr, float scale, void* stream);
int mmae_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq,
                  void* dk, void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb,
                  int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr,
                  int64_t dk_sb, int64_t dk_sr, int64_t dv_sb, int64_t dv_sr, float scale, void* stream);
/* The same two kernels for f32 q/k/v/o/dO/dq/dk/dv (fp32_output_adapters in speed mode): operands split into bf16 hi + lo
 * on the fly, every product as a_hi.b_hi + a_hi.b_lo + a_lo.b_hi with fp32 accumulation (the MMAE_F32X3 precision).
 * Strides in f32 elements.  backward needs 4 * (Nq_pad + Nk_pad) * head_dim * 2 bytes of LDS <= 160 KB, else MMAE_ESUPPORT. */
int mmae_attn_fwd_f32x3(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                  int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb,
                  int64_t o_sr, float scale, void* stream);
int mmae_attn_bwd_f32x3(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq,
                  void* dk, void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb,
                  int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr,
                  int64_t dk_sb, int64_t dk_sr, int64_t dv_sb, int64_t dv_sr, float scale, void* stream);
/* f32 tensors again, operands rounded to fp16 (TF32's significand; the MMAE_F32F16 precision): one MFMA per product.  bwd: dy_amax =
 * device scalar (or NULL) whose power of two pre-scales dO on its way into LDS; dq / dk / dv are scaled back at their store. */
int mmae_attn_fwd_f32f16(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                  int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb,
                  int64_t o_sr, float scale, void* stream);
int mmae_attn_bwd_f32f16(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq,
                  void* dk, void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb,
                  int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr,
                  int64_t dk_sb, int64_t dk_sr, int64_t dv_sb, int64_t dv_sr, float scale, const float* dy_amax, void* stream);
/* fp16 tensors in memory (MMAE_F16 storage; strides in fp16 elements): the bf16 kernels' data movement with fp16 MFMA products.  The
 * backward takes d_o and returns dq / dk / dv in the adapter's scaled gradient units (see MMAE_F16): nothing is rescaled here. */
int mmae_attn_fwd_f16(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, int hd,
                      int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb, int64_t v_sr, int64_t o_sb, int64_t o_sr,
                      float scale, void* stream);
int mmae_attn_bwd_f16(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, void* dq, void* dk,
                      void* dv, int B, int H, int Nq, int Nk, int hd, int64_t q_sb, int64_t q_sr, int64_t k_sb, int64_t k_sr, int64_t v_sb,
                      int64_t v_sr, int64_t o_sb, int64_t o_sr, int64_t dq_sb, int64_t dq_sr, int64_t dk_sb, int64_t dk_sr, int64_t dv_sb,
                      int64_t dv_sr, float scale, void* stream);


/* ------------------------------------------------------------------------- *
 * One pre-LN transformer block (multimae_utils.py:217-232: x + attn(norm1(x)), then x + mlp(norm2(x))) as ONE call per
 * direction: the library enqueues the block's whole kernel sequence (forward 7 launches; backward ~20, with the
 * weight-gradient GEMMs and the parameter-gradient reductions on `side_stream`, ordered against `stream` with events).
 * Same kernels, same order and same results as issuing the individual entry points; what it removes is ~25 host
 * round trips per block (the host, not the GPU, paced the adapters' backward passes).
 * Every buffer is the caller's: activations (forward writes, backward reads), backward temporaries, gradient
 * destinations, and two f32 workspaces (split-K slabs / reduction scratch) private to the two streams.
 * act_dtype MMAE_BF16, or MMAE_F32 with f32_gemm = MMAE_F32X3 (fp32 adapters in speed mode).  Needs the fused attention
 * kernel's geometry (head_dim 32 / 64, N <= 256); otherwise MMAE_ESUPPORT and the caller issues the steps itself.
 * ------------------------------------------------------------------------- */
typedef struct mmae_block_desc {
    int32_t B, N, D, heads, Hd;                  /* R = B*N rows of width D; Hd = MLP hidden width */
    int32_t act_dtype, f32_gemm;
    float eps;
    const void *qkv_w, *proj_w, *fc1_w, *fc2_w;  /* act dtype: [3D][D], [D][D], [Hd][D], [D][Hd] */
    const float *n1_w, *n1_b, *qkv_b, *proj_b, *n2_w, *n2_b, *fc1_b, *fc2_b;
    const float* x0;                             /* block input, f32 [R][D] */
    void* ln1; float* mean1; float* rstd1; void* qkv; float* lse; void* ao; float* x1;
    void* ln2; float* mean2; float* rstd2; void* hpre; void* hact;
    float* x2;                                   /* block output, f32 [R][D] */
    const float* dx; const void* dx_act;         /* backward in: d(x2) f32 and its act-dtype copy (== dx when act is f32) */
    float* dx0; void* dx0_act;                   /* backward out: d(x0) (dx0_act NULL when act is f32) */
    void *d_hpre, *d_ln2, *d_ao, *d_qkv, *d_ln1; float* dx1; void* dx1_act;     /* backward temporaries */
    float *part_h, *part1, *part2;               /* [ceil(R/32)][Hd], [nblk][3D], [nblk][3D]; nblk = mmae_layernorm_bwd_nblk(R) */
    float *g_n1_w, *g_n1_b, *g_qkv_w, *g_qkv_b, *g_proj_w, *g_proj_b, *g_n2_w, *g_n2_b, *g_fc1_w, *g_fc1_b, *g_fc2_w, *g_fc2_b;
    float* g_cs;                                 /* NULL or f32 [D]: receives colsum(dx0), the bias gradient of the Linear that produced x0 */
    int32_t grad_acc;                            /* gradients are added to (1) or stored into (0) their destinations */
    int32_t fc2_b_done;                          /* the producer of dx already delivered fc2's bias gradient: skip it */
    float* ws_main; int64_t ws_main_elems;       /* f32 scratch used by launches on `stream` */
    float* ws_side; int64_t ws_side_elems;       /* f32 scratch used by launches on `side_stream` */
    /* stochastic depth (DropPath, multimae_utils.py:105-135,229-232): per-sample scales f32 [B] = mask_b / keep_prob of the
     * attention branch (dp1) and the MLP branch (dp2); NULL = branch kept for every sample.  x1 = x0 + dp1[b] * attn(..),
     * x2 = x1 + dp2[b] * mlp(..).  With a scale set, `branch` (f32 [R][D], forward) and `dxs_act` (act [R][D], backward)
     * are scratch the caller provides. */
    const float* dp1; const float* dp2;
    float* branch; void* dxs_act;
    /* MX-fp8 products (optional; bf16 activations only): mx_w = host array of 16 device pointers, for qkv, proj, fc1, fc2 in turn
     * {e4m3 [n_out][k_in], its scales, e4m3 of the transpose [k_in][n_out], its scales} as mmae_mx_prepare_weights writes them.
     * The four forward products and the four dX products then run on the block-scaled MFMA: their activation operand is
     * quantised into mx_tmp (two halves of mmae_mx_tmp_bytes(B * N, max(Hd, 3 D)) bytes each) right before each product -- or by
     * the kernel that produces it: both LayerNorms and the GELU / dGELU epilogues emit the e4m3 copy themselves (env
     * MMAE_MX_FUSE=0: separate passes everywhere); weight gradients stay bf16.  D, 3 D and Hd must be multiples of 256.  NULL = bf16 products. */
    const void* const* mx_w; void* mx_tmp; int64_t mx_tmp_bytes;
    /* f32 activations with MMAE_F32X3 products: optional pre-split weights -- host array of x3_n triples {the f32 weight pointer
     * as it appears above, its [n_out][3 k_in] copy, its [3 n_out][k_in] copy} (mmae_x3_prepare_weights) -- and a scratch of
     * mmae_x3_tmp_bytes(B * N, max(Hd, 3 D)) bytes.  Forward and dX products whose contraction is a multiple of 32 then run as
     * one bf16 product over 3 K on the ping-pong kernel; weight gradients keep the MMAE_F32X3 kernel.  NULL = off. */
    const void* const* x3_w; int32_t x3_n; void* x3_tmp; int64_t x3_tmp_bytes;
    const float* dy_amax;                        /* f32 activations with f32_gemm = MMAE_F32F16: device scalar for the gradient operands of the
                                                    dX / dW products (mmae_gemm_desc.a_amax); NULL: those products run as MMAE_F32X3 */
} mmae_block_desc;

int mmae_block_fwd(const mmae_block_desc* d, void* stream);
/* side_stream may equal stream (or be NULL): everything then runs in order on `stream`. */
int mmae_block_bwd(const mmae_block_desc* d, void* stream, void* side_stream);

/* ------------------------------------------------------------------------- *
 * A stack of L pre-LN transformer blocks (the ViT encoder, multimae.py:94-98,350, or an adapter's decoder_transformer,
 * output_adapters.py:127-133,271) as ONE call per direction: the library lays the saved activations of all blocks out in
 * one caller-provided slab and enqueues every block's kernel sequence (mmae_block_fwd / mmae_block_bwd).  A ViT-B encoder
 * step is then 2 host round trips instead of 24 (or ~600 single launches).
 *   w   host array [L][4]  act-dtype weights  qkv_w, proj_w, fc1_w, fc2_w
 *   p   host array [L][8]  f32                n1_w, n1_b, qkv_b, proj_b, n2_w, n2_b, fc1_b, fc2_b
 *   g   host array [L][12] gradient destinations in the order n1_w n1_b qkv_w qkv_b proj_w proj_b n2_w n2_b fc1_w fc1_b
 *       fc2_w fc2_b (NULL entries = not wanted); grad_acc: add to (1) / store into (0) them
 *   dp  NULL or host array [L][2] of device f32 [B] stochastic-depth scales (see mmae_block_desc.dp1 / dp2)
 *   act slab of mmae_stack_act_bytes() bytes: forward writes, backward reads; block l's output (f32 [R][D]) lives at byte
 *       offset mmae_stack_out_offset(d, l) -- the stack's result is the last one.
 * Backward handles blocks l_end-1 ... l_begin (a data-parallel caller splits the stack at its gradient-bucket borders
 * and launches a bucket's all-reduce between two calls; otherwise l_begin = 0, l_end = L).  d_out[l] (host array [L], f32
 * [R][D] or NULL) is the gradient arriving at block l's output from outside the stack; d_out[L-1] is required.  The
 * gradient of the stack input is written to dx when l_begin == 0.  tmp: slab of mmae_stack_tmp_bytes() bytes that must
 * stay untouched between the calls of one backward pass and alive until side_stream has drained.
 * Geometry / dtype limits are those of mmae_block_fwd (MMAE_ESUPPORT otherwise).
 * ------------------------------------------------------------------------- */
typedef struct mmae_stack_desc {
    int32_t L, B, N, D, heads, Hd;
    int32_t act_dtype, f32_gemm;
    float eps;
    int32_t grad_acc;
    const void* const* w;
    const float* const* p;
    const float* const* dp;
    const float* x;                              /* stack input, f32 [R][D] */
    void* act; int64_t act_bytes;
    float* const* g;
    const float* const* d_out;
    float* dx;
    void* tmp; int64_t tmp_bytes;
    int32_t l_begin, l_end;
    float* ws_main; int64_t ws_main_elems;
    float* ws_side; int64_t ws_side_elems;
    const void* const* mx_w;     /* optional host array [16 L]: mmae_block_desc.mx_w of every block (NULL = bf16 products).  Set it
                                    BEFORE asking for the slab sizes: the quantisation scratch is carved from act (forward) / tmp (backward) */
} mmae_stack_desc;

int64_t mmae_stack_act_bytes(const mmae_stack_desc* d);
int64_t mmae_stack_tmp_bytes(const mmae_stack_desc* d);
int64_t mmae_stack_out_offset(const mmae_stack_desc* d, int l);
int mmae_stack_fwd(const mmae_stack_desc* d, void* stream);
int mmae_stack_bwd(const mmae_stack_desc* d, void* stream, void* side_stream);

/* ------------------------------------------------------------------------- *
 * SpatialOutputAdapter.forward (output_adapters.py:236-282) and its autograd as ONE call per direction: proj_context,
 * query / context build (:183-234), the cross-attention layer with its MLP (:262-266), the decoder_transformer blocks
 * (:271), out_proj (:274) and the patch -> image rearrangement (:277-280).  Same kernels and order as the individual
 * entry points; ~45 host round trips forward and ~125 backward become one each.
 *   enc      f32 [B][NC][Denc] encoder tokens (NC = n_keep + G); enc_act: their act-dtype copy (== enc when act is f32)
 *   w        host array [6 + 4*depth] act-dtype weights: q, kv, proj, fc1, fc2, then per block qkv, proj, fc1, fc2,
 *            then out_proj; proj_context LAST
 *   p        host array [12 + 8*depth + 2] f32: q_b kv_b proj_b ctxn_w ctxn_b qn_w qn_b outn_w outn_b fc1_b fc2_b
 *            (11 entries), then per block the 8 of mmae_stack_desc.p, then out_proj_b, proj_context_b
 *   mask_token f32 [D]; task_emb host array [T] of f32 [D] (NULL = task without embedding -> zeros)
 *   pos      f32 [n_q][D] decoder position table (all tasks share the grid)
 *   act / tmp slabs as in mmae_stack_desc; pat (f32 [B*n_q][C*ph*pw]) is kept in act for a patch-domain loss
 *   (mmae_adapter_pat_offset); img f32 [B][C][nh*ph][nw*pw] is the API result.
 * Backward: d_img (image-domain gradient) or d_pat (act dtype [B*n_q][ld_pat], already in patch layout) -- exactly one;
 *   g host array in the parameter order  mask_token, task_emb[0..T-1], q_w q_b kv_w kv_b proj_w proj_b ctxn_w ctxn_b qn_w
 *   qn_b outn_w outn_b fc1_w fc1_b fc2_w fc2_b, 12 per block, out_proj_w out_proj_b, proj_context_w proj_context_b;
 *   d_enc f32 [B][NC][Denc] receives the gradient of the encoder tokens.
 * ------------------------------------------------------------------------- */
typedef struct mmae_adapter_desc {
    int32_t B, NC, Denc, D, heads, Hd, depth, T, q_task, G, n_q;
    int32_t C, nh, nw, ph, pw;
    int32_t act_dtype, f32_gemm;
    float eps;
    int32_t grad_acc;
    const int32_t* task_offsets_host;            /* [T+1] */
    const void* const* w;
    const float* const* p;
    const float* mask_token;
    const float* const* task_emb;
    const float* pos;
    const float* enc; const void* enc_act;
    const int64_t* ids_keep; const int64_t* ids_restore;
    void* act; int64_t act_bytes;
    float* img;                                  /* forward result; NULL = do not materialise the image */
    /* backward */
    const float* d_img; const void* d_pat; int64_t ld_pat;
    float* const* g;
    float* d_enc;
    void* tmp; int64_t tmp_bytes;
    float* ws_main; int64_t ws_main_elems;
    float* ws_side; int64_t ws_side_elems;
    const void* const* x3_w; int32_t x3_n;       /* optional pre-split weights (triples, as mmae_block_desc.x3_w) of an f32 adapter; set them
                                                    BEFORE asking for the slab sizes: the operand scratch is carved from act / tmp */
    const float* dy_amax;                        /* as mmae_block_desc.dy_amax, for the adapter's own products and its blocks */
    float* pat;                                  /* optional caller-owned home of the prediction rows f32 [B * n_q][C * ph * pw] (out_proj's output, what
                                                    the patch-domain losses and a deferred unpatchify read); NULL = carved from `act`.  Set it BEFORE asking
                                                    for the slab sizes.  With its own allocation the rows outlive the activation slab (a lazily
                                                    written prediction image no longer pins the whole slab until it is read or dropped) */
} mmae_adapter_desc;

int64_t mmae_adapter_act_bytes(const mmae_adapter_desc* d);
int64_t mmae_adapter_tmp_bytes(const mmae_adapter_desc* d);
int64_t mmae_adapter_pat_offset(const mmae_adapter_desc* d);
int mmae_adapter_fwd(const mmae_adapter_desc* d, void* stream);
int mmae_adapter_bwd(const mmae_adapter_desc* d, void* stream, void* side_stream);

/* ------------------------------------------------------------------------- *
 * Casts.  f32 master weights -> act-dtype shadows (optionally transposed so that
 * dX = dY.W is again an "NT" product).
 * ------------------------------------------------------------------------- */
int mmae_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int mmae_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* fp16 storage (MMAE_F16).  scale_amax (optional): the dy_amax device scalar of a gradient tensor -- f32 -> fp16 multiplies by S on the
 * way (an f32 loss gradient entering the fp16-storage domain), fp16 -> f32 by 1/S. */
int mmae_cast_f32_to_f16(const float* src, void* dst, int64_t n, const float* scale_amax, void* stream);
int mmae_cast_f16_to_f32(const void* src, float* dst, int64_t n, const float* scale_amax, void* stream);
/* dst[c][r] = src[r][c], src f32 [rows][cols]; dst act dtype [cols][rows] */
int mmae_transpose_cast(const float* src, void* dst, int dst_dtype, int rows, int cols, void* stream);
/* mean pooling over tokens, LinearOutputAdapter.forward (output_adapters.py:346-347):  y[b][:] = mean_n x[b][n][:]
 * (x f32 [B][N][D], y f32 [B][D]) and its backward dx[b][n][:] = dy[b][:] / N. */
int mmae_token_mean_fwd(const float* x, float* y, int B, int N, int D, void* stream);
int mmae_token_mean_bwd(const float* dy, float* dx, int B, int N, int D, void* stream);
/* y (+)= a*x elementwise, f32 */
int mmae_axpy_f32(float* y, const float* x, float a, int64_t n, void* stream);
/* out = in[0] + ... + in[n_in - 1] (f32, summed in index order; in_host: host array of 1 <= n_in <= 8 device pointers; n % 4 == 0; out may be in[0]).
 * The sum autograd forms of the output adapters' gradients with respect to the shared encoder tokens (multimae.py:352-381), in one pass.  ABI v7. */
int mmae_add_n_f32(float* out, const float* const* in_host, int n_in, int64_t n, void* stream);
/* stochastic depth (multimae_utils.py:105-122) on 2-D activations, rows grouped N per sample, s f32 [R/N]:
 *   rowscale_add:  out[r][:] = resid[r][:] + s[r / N] * y[r][:]                 (f32, out may alias resid)
 *   rowscale_cast: out[r][:] = cast(s[r / N] * x[r][:])   (x f32; out act dtype) */
int mmae_rowscale_add(const float* resid, const float* y, const float* s, float* out, int64_t R, int N, int D, void* stream);
int mmae_rowscale_cast(const float* x, const float* s, void* out, int out_dtype, int64_t R, int N, int D, void* stream);
/* nn.Dropout (Mlp.drop, attn_drop, proj_drop: multimae_utils.py:138-155, 158-214; the mask itself is drawn by the caller) as ONE
 * elementwise pass, forward and backward alike:
 *   out[i] = (resid ? resid[i] : 0) + (keep[i] ? x[i] * scale * (s ? s[i / per] : 1) : 0)
 * x f32 / bf16 / fp16 (x_dtype), keep uint8 0 / 1, scale = 1 / (1 - p); s: optional per-sample stochastic-depth scale f32 (`per`
 * elements per sample) so that x + drop_path(dropout(y)) is still one pass; resid f32 (then out_dtype must be MMAE_F32); out may
 * alias x when the types agree.  n % 4 == 0. */
int mmae_dropout(const void* x, int x_dtype, const void* keep, float scale, const float* s, int64_t per, const float* resid, void* out,
                 int out_dtype, int64_t n, void* stream);

/* ------------------------------------------------------------------------- *
 * Mask sampler: deterministic core of MultiMAE.generate_random_masks
 * (multimae/multimae.py:191-216).  The random draws are INPUTS (the reference draws
 * Dirichlet on the CPU generator, :187, and the noise with torch.rand on the
 * device, :195,204) so a CPU-generated stream reproduces the reference bit for bit.
 *   samples_per_task  int64 [B][T]      round(p * num_encoded) (:189)
 *   task_noise        f32   [B][Ntot]   per-task noise, tasks concatenated
 *   all_noise         f32   [B][Ntot]
 *   task_offsets      int32 [T+1]       host array: start of each task in Ntot
 * outputs (int64): mask_all [B][Ntot] (0 = visible), ids_keep [B][n_keep],
 *                  ids_restore [B][Ntot].  Ties broken by lower index first.
 * ------------------------------------------------------------------------- */
int mmae_mask_sample(const int64_t* samples_per_task, const float* task_noise, const float* all_noise,
                     const int32_t* task_offsets_host, int T, int B, int Ntot, int n_keep,
                     int64_t* mask_all, int64_t* ids_keep, int64_t* ids_restore, void* stream);

/* ------------------------------------------------------------------------- *
 * Gather-first patch embedding.  Replaces PatchedInputAdapter.forward
 * (input_adapters.py:97-119) / SemSegInputAdapter.forward (:215-241) followed by
 * torch.gather(ids_keep) and the global-token concat (multimae.py:340-347), for the
 * kept tokens only.
 *
 * mmae_patch_rows builds, for every kept token (b, r), one row of width Ktot =
 * sum_t C_t*ph_t*pw_t: the flattened patch in Conv2d-weight column order (c, i, j)
 * placed in the owner task's segment [k_off, k_off + C*ph*pw), zeros elsewhere.  One
 * GEMM against the K-concatenated projection weights then embeds all modalities.
 *   kind 0: data = f32 [B][C][H][W];  kind 1: data = int64 class ids [B][H][W]
 *   looked up in emb f32 [n_cls][C] (C = dim_class_emb).
 * task_offsets_host int32 [T+1]: first token index of each task in the concatenated
 * token axis (host memory; baked into the launch).
 * ------------------------------------------------------------------------- */
typedef struct mmae_patch_src {
    const void* data;
    const float* emb;
    int32_t kind, C, H, W, ph, pw, k_off;
    int32_t n_cls;               /* kind 1: rows of emb; class ids outside [0, n_cls) embed as zeros and receive no gradient
                                    (nn.Embedding raises a device assert there); 0 = unchecked */
} mmae_patch_src;

int mmae_patch_rows(const mmae_patch_src* srcs_host, const int32_t* task_offsets_host, int T, const int64_t* sel,
                    void* rows, int rows_dtype, int B, int n_sel, int Ktot, void* stream);
/* d_emb[cls][e] += d_rows[row][k_off + e*ph*pw + i*pw + j] over the selected tokens in
 * [tok_off, tok_off + n_patches); d_rows act dtype [B*n_sel][ld].  (float atomics) */
int mmae_semseg_emb_bwd(const void* d_rows, int rows_dtype, int64_t ld, const int64_t* cls, const int64_t* sel, float* d_emb,
                        int B, int H, int W, int E, int ph, int pw, int n_sel, int k_off, int tok_off, int n_patches, int n_cls,
                        void* stream);
/* The same gradient without atomics: every table entry is summed in a fixed order (per-thread columns of workgroup-private LDS tables, partial
 * tables per workgroup in ws, a second launch that sums them in index order) -- bit-identical from run to run.  ws: f32 scratch of
 * mmae_semseg_emb_bwd_ws_elems() elements (-1: geometry not supported); accumulate 0: d_emb = gradient (no zero-fill needed), 1: d_emb += gradient.
 * rows_dtype MMAE_BF16 / MMAE_F32.  ABI v7. */
int64_t mmae_semseg_emb_bwd_ws_elems(int B, int n_sel, int E, int n_cls);
int mmae_semseg_emb_bwd_det(const void* d_rows, int rows_dtype, int64_t ld, const int64_t* cls, const int64_t* sel, float* d_emb,
                            int B, int H, int W, int E, int ph, int pw, int n_sel, int k_off, int tok_off, int n_patches, int n_cls,
                            float* ws, int64_t ws_elems, int accumulate, void* stream);
/* tok[b][r][:] = proj[b*n_sel+r][:] + bias_t[:] + pos_t[p][:] (t, p = owner task / patch of
 * sel[b][r]); tok[b][n_sel+g][:] = global_tok[g][:].  tok f32 [B][n_sel+G][D];
 * bias / pos: host arrays of T device pointers (pos_t f32 [n_patches_t][D]). */
int mmae_tokens_assemble(float* tok, const float* proj, const float* const* bias_host, const float* const* pos_host,
                         const int32_t* task_offsets_host, int T, const int64_t* sel, const float* global_tok, int B, int n_sel,
                         int G, int D, void* stream);
/* backward: d_proj[b*n_sel+r][:] = d_tok[b][r][:] (act dtype); part f32 [nblk][T+G][D] holds
 * per-workgroup column sums (rows 0..T-1: bias_t, rows T..T+G-1: global tokens), to be reduced
 * with mmae_colsum_partials.  nblk = mmae_tokens_assemble_bwd_nblk(B); needs T+G <= 8. */
int mmae_tokens_assemble_bwd_nblk(int B);
int mmae_tokens_assemble_bwd(const float* d_tok, void* d_proj, int proj_dtype, const int32_t* task_offsets_host, int T,
                             const int64_t* sel, float* part, int B, int n_sel, int G, int D, void* stream);

/* ------------------------------------------------------------------------- *
 * Fused patch embedding + positional embedding (csrc/embed.hip) -- the forward of
 * PatchedInputAdapter / SemSegInputAdapter (input_adapters.py:97-119, 215-241: Conv2d with
 * kernel = stride = patch, + pos_emb) and of the token selection of MultiMAE.forward
 * (multimae.py:340-347) in ONE kernel, bf16 MFMA with f32 accumulation:
 *   tok[b][r][:] = W_t . patch(b, sel[b][r]) + bias_t + pos_t[p];  tok[b][n_sel+g][:] = global_tok[g][:]
 * = mmae_patch_rows + one mmae_gemm per task + mmae_tokens_assemble without their HBM
 * intermediates.  w_bf16_host: T device pointers to bf16 [D][C*ph*pw] (Conv2d weight, flattened);
 * bias_host / pos_host as for mmae_tokens_assemble.  rows_bf16 (optional, may be NULL): the
 * zero-padded bf16 patch rows [B*n_sel][Ktot] of mmae_patch_rows, written on the side for the
 * weight-gradient products of the backward pass.
 * mmae_patch_embed_supported: 1 if the fused kernel takes the geometry -- exactly what csrc/embed.hip: check_geometry tests:
 * 32 <= D <= 1024 and D % 32 == 0; 1 <= n_sel <= 1024; for every task H % ph == 0, W % pw == 0, k_off % 8 == 0 and
 * C*ph*pw a multiple of the kernel's staging unit (64 elements for D <= 768, 32 above); semseg: ph*pw <= 64 and at most
 * 32 767 classes; and the workgroup's LDS image (token ids + the double-buffered gather chunk + the per-wave weight images +
 * the semseg class tables in bf16) within the CU's 160 KiB.  Otherwise use the three calls above.  (The framework's dispatch,
 * multimae_amd/functions.py: EmbedFn, additionally keeps ViT-L width on the three passes, where they measured faster.)
 * ------------------------------------------------------------------------- */
int mmae_patch_embed_supported(const mmae_patch_src* srcs_host, int T, int n_sel, int D);
int mmae_patch_embed_fwd(const mmae_patch_src* srcs_host, const void* const* w_bf16_host, const float* const* bias_host,
                         const float* const* pos_host, const int32_t* task_offsets_host, int T, const int64_t* sel,
                         const float* global_tok, float* tok, void* rows_bf16, int B, int n_sel, int G, int D, int Ktot, void* stream);

/* ------------------------------------------------------------------------- *
 * Decoder query / context builder.  Replaces SpatialOutputAdapter.
 * get_queries_and_context + generate_context_embeddings, output_adapters.py:160-234
 * (use_task_queries path), without materialising the (B, Ntot, D) tensor.
 *   ctx      f32 [B][n_keep+G][D]   proj_context output
 *   queries  f32 [B][n_q][D]:  q[b][j] = (vis ? ctx[b][ids_restore[b][q_off+j]] : mask_token)
 *                                        + task_emb[q_task] + pos[j]
 *   context  f32 [B][n_keep+G][D]: c[b][r] = ctx[b][r] + task_emb[task(ids_keep[b][r])]
 *                                        + pos[ids_keep[b][r] - off(task)];  global rows copied.
 *   q_task = -1: pure mask-token queries (output_adapters.py:213-220 -- the adapter's task is not among the encoder
 *  inputs, or use_task_queries=False): q[b][j] = mask_token + pos[j]; the caller adds the adapter's own task embedding
 *  (if it has one) to the mask_token vector it passes.  n_q = N_H * N_W then.
 *   task_emb f32 [T][D]; pos f32 [n_patch][D] (same table for every task: all tasks share
 *  the decoder's N_H x N_W grid, output_adapters.py:172-175); task_offsets int32 [T+1] (host).
 * ------------------------------------------------------------------------- */
int mmae_decoder_build(const float* ctx, const int64_t* ids_keep, const int64_t* ids_restore,
                       const float* mask_token, const float* task_emb, const float* pos,
                       const int32_t* task_offsets_host, int T, int q_task, int B, int n_keep, int G, int D,
                       int n_q, float* queries, float* context, void* stream);
/* backward: d_ctx (f32, overwritten) from d_queries/d_context; parameter gradients are
 * accumulated through partials: part f32 [nblk][T+1][D] (rows 0..T-1 task_emb, row T
 * mask_token; with q_task = -1 row T is the sum of ALL query-row gradients), nblk = mmae_decoder_build_bwd_nblk(B). */
int mmae_decoder_build_bwd_nblk(int B);
int mmae_decoder_build_bwd(const float* d_queries, const float* d_context, const int64_t* ids_keep,
                           const int64_t* ids_restore, const int32_t* task_offsets_host, int T, int q_task,
                           int B, int n_keep, int G, int D, int n_q, float* d_ctx, float* part, void* stream);

/* ------------------------------------------------------------------------- *
 * Patch <-> image layout.  Replaces einops.rearrange 'b (nh nw) (c ph pw) ->
 * b c (nh ph) (nw pw)', output_adapters.py:277-280, and its transpose for autograd.
 * ------------------------------------------------------------------------- */
int mmae_unpatchify(const float* patches, float* img, int B, int C, int nh, int nw, int ph, int pw, void* stream);
/* patches act dtype [B*nh*nw][ld], ld >= C*ph*pw (columns beyond C*ph*pw are left untouched) */
int mmae_patchify(const float* img, void* patches, int patches_dtype, int64_t ld, int B, int C, int nh, int nw, int ph,
                  int pw, void* stream);

/* ------------------------------------------------------------------------- *
 * Masked losses.  Replace MaskedMSELoss / MaskedL1Loss (criterion.py:84-114,
 * 141-171, incl. norm_pix :88-95) and MaskedCrossEntropyLoss (:37-57).
 *   pred/target f32 [B][C][H][W] (CE: logits [B][C][H][W], target int64 [B][H][W]);
 *   mask int64 [B][nh*nw] (1 = contributes), patch = H/nh.
 *   per_sample f32 [B][2]: (sum of masked per-pixel errors, number of masked pixels)
 *   loss f32 [2] = (mean over samples with count>0 of sum/count  (nanmean semantics),
 *                   number of such samples);
 *   stats f32 [B][nh*nw][2]: (mean, rstd) of the target patch when norm_pix.
 * kind: 0 = MSE, 1 = L1.
 * bwd: d_pred = upstream * d loss / d pred (zeros on unmasked pixels).
 * ------------------------------------------------------------------------- */
int mmae_loss_split(void);   /* workgroups per sample: partial is f32 [B][mmae_loss_split()] */
int mmae_masked_pixel_loss_fwd(const float* pred, const float* target, const int64_t* mask, int kind, int norm_pix,
                               int B, int C, int H, int W, int patch, float* stats, float* partial, float* per_sample,
                               float* loss, void* stream);
int mmae_masked_pixel_loss_bwd(const float* pred, const float* target, const int64_t* mask, int kind, int norm_pix,
                               int B, int C, int H, int W, int patch, const float* stats, const float* per_sample,
                               const float* loss, const float* upstream, float* d_pred, void* stream);
/* label_smoothing as F.cross_entropy's (criterion.py:47): (1 - eps) * nll(target) + eps * mean over classes of -log p_c */
int mmae_masked_ce_fwd(const float* logits, const int64_t* target, const int64_t* mask, int B, int C, int H, int W,
                       int patch, float label_smoothing, float* lse, float* partial, float* per_sample, float* loss, void* stream);
int mmae_masked_ce_bwd(const float* logits, const int64_t* target, const int64_t* mask, int B, int C, int H, int W,
                       int patch, float label_smoothing, const float* lse, const float* per_sample, const float* loss,
                       const float* upstream, float* d_logits, void* stream);

/* The same losses evaluated on the adapters' patch rows pat f32 [B*nh*nw][C*patch*patch] (column order c, i, j -- out_proj's
 * output, output_adapters.py:274, before the rearrangement of :277-280): identical arithmetic per pixel, but the gradient is
 * written straight back as patch rows d_pat (act dtype, row stride ld_pat >= C*patch*patch, pad columns zeroed) for
 * mmae_adapter_bwd -- no f32 image-domain gradient and no patchify pass.  target / mask / stats / per_sample / loss as above;
 * lse_pat f32 [B*nh*nw][patch*patch].  CE needs patch*patch a power of two <= 64 (MMAE_ESUPPORT otherwise). */
int mmae_masked_pixel_loss_pat_fwd(const float* pat, const float* target, const int64_t* mask, int kind, int norm_pix, int B, int C, int H,
                                   int W, int patch, float* stats, float* partial, float* per_sample, float* loss, void* stream);
int mmae_masked_pixel_loss_pat_bwd(const float* pat, const float* target, const int64_t* mask, int kind, int norm_pix, int B, int C, int H,
                                   int W, int patch, const float* stats, const float* per_sample, const float* loss, const float* upstream,
                                   void* d_pat, int d_pat_dtype, int64_t ld_pat, float* amax, void* stream);
int mmae_masked_ce_pat_fwd(const float* pat, const int64_t* target, const int64_t* mask, int B, int C, int H, int W, int patch,
                           float label_smoothing, float* lse_pat, float* partial, float* per_sample, float* loss, void* stream);
int mmae_masked_ce_pat_bwd(const float* pat, const int64_t* target, const int64_t* mask, int B, int C, int H, int W, int patch,
                           float label_smoothing, const float* lse_pat, const float* per_sample, const float* loss, const float* upstream,
                           void* d_pat, int d_pat_dtype, int64_t ld_pat, float* amax, void* stream);
/* (d_pat_dtype MMAE_F16, cross entropy only: amax is required and is WRITTEN with m = max over samples of |upstream / (n_samples * n_pixels_b)|,
 * the bound of every element; d_pat holds the gradient times S(m) -- the units an MMAE_F16 adapter's backward runs in, mmae_adapter_desc.dy_amax.)
 * (amax, optional: device f32 the kernel raises -- atomic max -- to the largest |element| it writes into d_pat; the caller zeroes it.
 * It is what an adapter with f32 activations and MMAE_F32F16 products scales its gradient operands by: mmae_adapter_desc.dy_amax.) */

/* ------------------------------------------------------------------------- *
 * Optimiser step on flat arenas.  Replaces get_grad_norm_ / clip_grad_norm_
 * (utils/native_scaler.py:22-35,49-62) and torch.optim.AdamW.step as configured by
 * utils/optim_factory.py:138-174 (decoupled weight decay on every tensor).
 *   mmae_sumsq: out[0] (+)= sum x^2   (ws: f32 scratch >= 1024 elements)
 *   mmae_adamw: p,g,m,v f32 [n]; grad_scale multiplies g first (clipping);
 *               optional act-dtype shadow of the updated parameters.
 *               If skip_flag (device int32) is non-null and *skip_flag != 0 the step
 *               is a no-op (non-finite / skip_grad handling without a host sync).
 * ------------------------------------------------------------------------- */
int mmae_sumsq(const float* x, int64_t n, float* out, float* ws, void* stream);
int mmae_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
               float eps, float weight_decay, int step, const float* grad_scale_dev, const int32_t* skip_flag,
               void* shadow, int shadow_dtype, void* stream);
/* same update with the step-dependent scalars read from device memory: hyper_dev f32[4] =
 * { lr, weight_decay, 1 - beta1^step, sqrt(1 - beta2^step) }.  A captured hipGraph of the training step replays
 * with the schedule's current values (the host refreshes the 16 bytes before every replay). */
int mmae_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev, float beta1,
                   float beta2, float eps, const float* grad_scale_dev, const int32_t* skip_flag, void* shadow,
                   int shadow_dtype, void* stream);

/* ------------------------------------------------------------------------- *
 * The whole optimiser step of the training loop as one call: gradient 2-norm (get_grad_norm_,
 * utils/native_scaler.py:49-62), clip / skip decision (:22-35), the non-finite guards (GradScaler.step skipping an
 * update with inf / NaN gradients; the loop's isfinite(loss) check, run_pretraining_multimae.py:529-531) and AdamW --
 * with every decision taken ON THE DEVICE (no host synchronisation):
 *   pre    = grad_prescale / (grad_scale_dev ? *grad_scale_dev : 1)   (grad_prescale = 1 / world_size for SUMMED data-parallel
 *            gradients; grad_scale_dev = torch.amp.GradScaler's loss scale when the arena still holds SCALED gradients -- the
 *            un-scaling is folded into the step's existing multiply, no pass over the arena)
 *   norm   = pre * |g|_2
 *   skip   = !finite(norm) || (skip_grad > 0 && norm >= skip_grad) || (loss_dev && !finite(*loss_dev))
 *            || (found_inf_dev && *found_inf_dev > 0)       (GradScaler's own inf / NaN verdict, counted apart from a bad loss)
 *   scale  = pre * (clip_grad > 0 ? min(1, clip_grad / (norm + 1e-6)) : 1)
 *   !skip: step += 1;  AdamW with lr, weight_decay (host values, or lrwd_dev[0..1] when given -- the per-iteration cosine
 *          tables of run_pretraining_multimae.py:474-480) and the bias corrections of the DEVICE step counter, so a skipped
 *          iteration does not advance Adam's step (the reference never calls optimizer.step() for it).
 * state  f32 [8]:  [0] sum of squares  [1] norm  [2] scale  [3..6] lr, weight_decay, 1 - beta1^t, sqrt(1 - beta2^t)
 * istate i32 [8]:  [0] skip flag of this step  [1] t = updates applied so far  [2] steps whose loss was not finite
 *                  [3] steps skipped for any reason  [4] steps skipped because found_inf_dev was set (AMP overflow)
 *                  [5] steps whose gradient norm was not finite  [6..7] reserved.     ws: f32 scratch >= 1024.
 * ------------------------------------------------------------------------- */
typedef struct mmae_opt_desc {
    float* p; const float* g; float* m; float* v; int64_t n;
    void* shadow; int32_t shadow_dtype;          /* optional act-dtype copy of the updated parameters */
    float lr, weight_decay, beta1, beta2, eps;
    const float* lrwd_dev;
    float clip_grad, skip_grad, grad_prescale;
    const float* loss_dev;
    float* state; int32_t* istate; float* ws;
    const float* found_inf_dev;                  /* optional: GradScaler.step()'s found_inf (> 0: skip, istate[4] += 1) */
    const float* grad_scale_dev;                 /* optional: the loss scale the gradients still carry */
} mmae_opt_desc;
int mmae_opt_step(const mmae_opt_desc* d, void* stream);

/* ------------------------------------------------------------------------- *
 * Truncated depth standardisation, the step right before the model in the training loop
 * (run_pretraining_multimae.py:487-492): per sample, mean / unbiased variance of the values of rank [lo, hi) among
 * the n = c*h*w values (the reference: torch.sort, slice [int(0.1 n), int(0.9 n)), mean, var), then
 * y = (x - mean) / sqrt(var + eps) over the whole map.  x, y f32 [B][n] (y may alias x).  Rank selection by radix
 * select, no sort; ties at the cuts are counted by rank like the sorted slice.
 * ------------------------------------------------------------------------- */
int mmae_depth_standardize(const float* x, float* y, int B, int n, int lo, int hi, float eps, void* stream);

/* hardware probes used by tests/ to pin instruction semantics the kernels rely on */
/* SemSegInputAdapter(interpolate_class_emb=True) (input_adapters.py:192-198): the class-embedding image resized by 1 / patch with
 * nn.Upsample(bilinear) -- per token the mean of the centre taps -- as out f32 [B][E][H/ph][W/pw] (then projected like a 1 x 1 patch
 * image), and its gradient into class_emb (float atomics; pad_idx = nn.Embedding's padding_idx or -1).  mmae_rows_to_image is the
 * data gradient of the patch embedding for an image-like input: the selected tokens' row gradients back into d_img (zeroed by
 * the caller). */
int mmae_semseg_avg_emb_fwd(const int64_t* x, const float* class_emb, float* out, int B, int H, int W, int E, int ph, int pw, int n_cls, void* stream);
int mmae_semseg_avg_emb_bwd(const float* d_img, const int64_t* x, float* d_class_emb, int B, int H, int W, int E, int ph, int pw, int n_cls, int pad_idx,
                            void* stream);
int mmae_rows_to_image(const float* d_rows, int64_t ldr, int k_off, const int64_t* sel, float* d_img, int B, int n_sel, int64_t tok_off, int n_patches,
                       int C, int H, int W, int ph, int pw, void* stream);
/* Gradient of LEARNABLE positional embeddings (PatchedInputAdapter / SemSegInputAdapter with learnable_pos_emb=True,
 * input_adapters.py:75-78,183-186): d_pos[sel[b][j]][:] += d_tok[b][j][:] over the selected tokens (d_tok f32 [B][n_sel + G][D] as
 * mmae_tokens_assemble lays tokens out; sel as given to it; d_pos f32 [n_pos][D], all tasks' position tables stacked in task order,
 * zeroed by the caller).  The resize of the parameter to the token grid (F.interpolate) stays with the caller. */
int mmae_pos_emb_bwd(const float* d_tok, const int64_t* sel, float* d_pos, int B, int n_sel, int G, int D, int n_pos, void* stream);
/* ------------------------------------------------------------------------- *
 * Pre-split operands for the split-bf16 ("x3") products of fp32_output_adapters (multimae.py:367-377) in speed mode.
 * a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi is ONE bf16 product over a contraction of 3 K when the operands are stored as
 * A' = [hi | hi | lo] and B' = [hi | lo | hi]: it then runs on the ping-pong bf16 kernel (2-3x the rate of the MMAE_F32X3 kernel,
 * which splits in registers).  mmae_x3_split writes such an operand: out bf16 [rows][out_ld], segment s (cols elements) at
 * element offset s * seg_stride of the row (or, with seg_stride = rows * cols and out_ld = cols, stacked along the rows);
 * segment lo_seg gets bf16(x - bf16(x)), the other two bf16(x).  mmae_x3_prepare_weights does both weight layouts:
 * dst[2 i] = [n_out][3 k_in] = [hi | lo | hi] for the forward product, dst[2 i + 1] = [3 n_out][k_in] = [hi ; lo ; hi] for dX.
 * The composite calls use them when the descriptor's x3_w table is set (mmae_adapter_desc / mmae_block_desc).
 * ------------------------------------------------------------------------- */
int mmae_x3_split(const float* x, int64_t ldx, int64_t rows, int cols, void* out, int64_t out_ld, int64_t seg_stride, int lo_seg, void* stream);
int mmae_x3_prepare_weights(int n, const void* const* w, const int32_t* n_out, const int32_t* k_in, void* const* dst, void* stream);
/* scratch for one pre-split activation operand [rows][3 cols] bf16 */
int64_t mmae_x3_tmp_bytes(int64_t rows, int cols);
int mmae_probe_tr16(const uint16_t* lds_image_1024, const uint32_t* lane_byte_addr_64, uint16_t* out_64x4,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMAE_H */
