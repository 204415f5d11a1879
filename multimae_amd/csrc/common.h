// Shared device/host helpers for the gfx950 kernels of libmmae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmae.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define MMAE_WAVE 64

void mmae_set_error(const char* msg);
int mmae_check_launch(const char* what);
int mmae_cu_count();                                          // runtime.hip: compute units of the CURRENT device (cached per device)
int mmae_cu_side();                                           // experiment: CUs set aside for the side stream's grouped weight gradients (0 = off)
int mmae_cu_avail();                                          // ... minus the ones mmae_gemm_cu_reserve() keeps free: width of a persistent GEMM grid

// A/B switches of the experiments (environment variables) exist only in builds with -DMMAE_EXPERIMENTS (make EXTRA=-DMMAE_EXPERIMENTS);
// the production library reads no environment: every switch is its default.
#include <stdlib.h>
static inline int mmae_env_int(const char* name, int dflt) {
#ifdef MMAE_EXPERIMENTS
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}
int mmae_gemm_ex(const mmae_gemm_desc* d, void* stream, int timing_cls, double flop_scale);   // runtime.hip

// query_norm / context_norm outputs of the fused decoder build (tokens.hip: decoder_build_kernel<.., true>; composite.hip: mmae_adapter_fwd)
struct BuildLn { const float *qg, *qb, *cg, *cb; void *qn, *cn; float *qmean, *qrstd, *cmean, *crstd; float eps; };

#define MMAE_REQUIRE(cond, msg) do { if (!(cond)) { mmae_set_error(msg); return MMAE_EINVAL; } } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
}

// ---- fp16 <-> f32 (round to nearest even): the 16-bit format of the fp32 output adapters' activations (MMAE_F16) ----
// NOT saturating (round 5): a value beyond +-65504 becomes +-inf (and a NaN stays a NaN), exactly as under fp16 autocast.
// The inf then reaches the loss / the gradient norm, mmae_opt_step's non-finite test skips the update and counts it
// (FusedAdamW.counters()['skipped']) -- the GradScaler contract of the reference's own fp16 runs -- instead of a silently clamped
// activation training on.  f16_cvt() is the one conversion every fp16 store and operand rounding of the library goes through.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ _Float16 f16_cvt(float f) { return (_Float16)f; }
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
    return __builtin_bit_cast(uint16_t, f16_cvt(f));
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
struct h16_t { uint16_t v; };
// Gradient scale of an fp16-storage adapter (mmae.h, MMAE_F16): S = 2^(4 - floor(log2 m)), m = *amax (the loss backward's bound of
// |dL/dprediction|); exact powers of two from the exponent field, 1 when the scalar is missing / zero / not finite.
__device__ __forceinline__ float h16_grad_scale(const float* amax) {
    if (!amax) return 1.0f;
    const unsigned e = (__float_as_uint(*amax) >> 23) & 0xffu;
    return (e >= 8 && e <= 253) ? __uint_as_float((258u - e) << 23) : 1.0f;
}
__device__ __forceinline__ float h16_grad_unscale(const float* amax) {
    if (!amax) return 1.0f;
    const unsigned e = (__float_as_uint(*amax) >> 23) & 0xffu;
    return (e >= 8 && e <= 253) ? __uint_as_float((e - 4u) << 23) : 1.0f;
}                                 // element type tag of fp16 tensors in the typed row kernels

// typed element access: T is float or uint16_t (bf16 bits)
template <typename T> struct ActT;
template <> struct ActT<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ActT<uint16_t> {
    static __device__ __forceinline__ float ld(const uint16_t* p) { return bf16_bits_to_f32(*p); }
    static __device__ __forceinline__ void st(uint16_t* p, float v) { *p = f32_to_bf16_bits(v); }
};

template <> struct ActT<h16_t> {
    static __device__ __forceinline__ float ld(const h16_t* p) { return f16_bits_to_f32(p->v); }
    static __device__ __forceinline__ void st(h16_t* p, float v) { p->v = f32_to_f16_bits(v); }
};

// load/store 4 consecutive act elements (16-byte / 8-byte aligned)
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const uint16_t* p) {
    i32x2 r = *reinterpret_cast<const i32x2*>(p);
    f32x4 o;
    o[0] = __uint_as_float(((uint32_t)r[0]) << 16); o[1] = __uint_as_float(((uint32_t)r[0]) & 0xffff0000u);
    o[2] = __uint_as_float(((uint32_t)r[1]) << 16); o[3] = __uint_as_float(((uint32_t)r[1]) & 0xffff0000u);
    return o;
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(uint16_t* p, f32x4 v) {
    i32x2 r; r[0] = (int)pack_bf16x2(v[0], v[1]); r[1] = (int)pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<i32x2*>(p) = r;
}

__device__ __forceinline__ f32x4 ld4(const h16_t* p) {
    i32x2 r = *reinterpret_cast<const i32x2*>(p);
    f32x4 o;
    o[0] = f16_bits_to_f32((uint16_t)((uint32_t)r[0] & 0xffffu)); o[1] = f16_bits_to_f32((uint16_t)((uint32_t)r[0] >> 16));
    o[2] = f16_bits_to_f32((uint16_t)((uint32_t)r[1] & 0xffffu)); o[3] = f16_bits_to_f32((uint16_t)((uint32_t)r[1] >> 16));
    return o;
}
__device__ __forceinline__ void st4(h16_t* p, f32x4 v) {
    i32x2 r;
    r[0] = (int)((uint32_t)f32_to_f16_bits(v[0]) | ((uint32_t)f32_to_f16_bits(v[1]) << 16));
    r[1] = (int)((uint32_t)f32_to_f16_bits(v[2]) | ((uint32_t)f32_to_f16_bits(v[3]) << 16));
    *reinterpret_cast<i32x2*>(p) = r;
}

// ---- exact-form (erf) GELU as nn.GELU() computes it --------------------------------------
// Phi(x) = 0.5 erfc(-x/sqrt2) through the Abramowitz-Stegun 7.1.26 rational form (one v_exp, one v_rcp,
// five FMAs); measured max |error| vs fp64: Phi 3.0e-7, gelu 4.2e-7, gelu' 3.2e-7 -- fp32 round-off class,
// so the same code serves the exact-f32 parity mode.  (libm erff in the GEMM epilogue cost 2.4x the GEMM.)
__device__ __forceinline__ void gelu_cdf_exp(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    float p = 1.061405429f;
    p = p * t - 1.453152027f; p = p * t + 1.421413741f; p = p * t - 0.284496736f; p = p * t + 0.254829592f;
    e = __expf(-z * z);
    const float h = 0.5f * p * t * e;
    cdf = x < 0.f ? h : 1.0f - h;
}
__device__ __forceinline__ float gelu_erf(float x) { float c, e; gelu_cdf_exp(x, c, e); return x * c; }
__device__ __forceinline__ void gelu_both(float x, float& y, float& dy) { float c, e; gelu_cdf_exp(x, c, e); y = x * c; dy = c + x * e * 0.39894228040143268f; }
__device__ __forceinline__ float gelu_erf_grad(float x) { float c, e; gelu_cdf_exp(x, c, e); return c + x * e * 0.39894228040143268f; }

// ---- the same pair for bf16 OUTPUTS: no transcendental, packed fp32 math ---------------------------------------------------------
// Phi(x) - 1/2 and GELU'(x) - 1/2 are odd: x Q(x^2) and x R(x^2) with degree-8 minimax polynomials in s = x^2 on |x| <= 4, evaluated at
// the argument clamped to that range (t).  Inside |x| <= 4 (fp32 Horner form, measured against fp64, tools/gelu_poly_fit.py): |Phi error|
// 6.6e-6, |GELU' error| 8.4e-5, |gelu error| 2.7e-5.  Beyond it (round 5) gelu(x) = max(x, t) Phi(t): x Phi(4) on the positive
// side (relative error 3.2e-5) and the CONSTANT gelu(-4) = -1.27e-4 on the negative one -- round 4 multiplied by x there, -3.2e-5 |x|
// without bound (5.2e-4 at x = -20); now |gelu error| <= 1.3e-4 for every x <= 4 and 3.2e-5 RELATIVE above -- and GELU' holds the polynomial's
// boundary values -5.0e-4 | 1 + 5.0e-4, within 5.5e-4 of the true derivative (which approaches 0 | 1 from those values): a bf16 store turns
// 1 + 5e-4 into 1.  All of it below
// the bf16 rounding (2^-9 = 2e-3 relative) the results get anyway.  Two variants were measured and dropped this round: selecting
// step(x) beyond |x| = 4 (four more VALU operations per element: +11 us on fc1's epilogue, 0.2 ms per step), and a fit constrained to
// hit 0 | 1 exactly at a clamp point of 4.5 (no extra instruction, but degree 8 then leaves 3.0e-4 on GELU' INSIDE the range, where
// every element lives: the bias-gradient column sums moved by 3e-4).  One more clamp per element it is (a v_med3 against FLT_MAX, see below).
// Two elements per instruction (v_pk_fma_f32 / v_pk_mul_f32): 8 + 8 packed FMAs per pair against one v_exp, one v_rcp and ~17 scalar
// FMA-class operations per ELEMENT of the exact form.  The exact form above stays for every f32 output (the parity mode) and for the
// fp16-storage adapters (H16 epilogues).
__device__ __forceinline__ void gelu_both_fast2(f32x2 x, f32x2& y, f32x2& dy) {
    f32x2 t;
    t[0] = __builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f);
    t[1] = __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f);
    const f32x2 s = t * t;
    f32x2 q = {8.063375031e-11f, 8.063375031e-11f}, r = {9.796052989e-10f, 9.796052989e-10f};
#define MMAE_PK_STEP(acc, c) acc = __builtin_elementwise_fma(acc, s, (f32x2){c, c})
    MMAE_PK_STEP(q, -7.003438364e-09f); MMAE_PK_STEP(r, -8.218800834e-08f);
    MMAE_PK_STEP(q, 2.716148569e-07f);  MMAE_PK_STEP(r, 3.028347546e-06f);
    MMAE_PK_STEP(q, -6.294988571e-06f); MMAE_PK_STEP(r, -6.495757385e-05f);
    MMAE_PK_STEP(q, 9.890799001e-05f);  MMAE_PK_STEP(r, 9.073266031e-04f);
    MMAE_PK_STEP(q, -1.133921717e-03f); MMAE_PK_STEP(r, -8.716320413e-03f);
    MMAE_PK_STEP(q, 9.877475989e-03f);  MMAE_PK_STEP(r, 5.845610030e-02f);
    MMAE_PK_STEP(q, -6.641059427e-02f); MMAE_PK_STEP(r, -2.648265329e-01f);
    MMAE_PK_STEP(q, 3.989227094e-01f);  MMAE_PK_STEP(r, 7.976095497e-01f);
#undef MMAE_PK_STEP
    const f32x2 half = {0.5f, 0.5f};
    const f32x2 cdf = __builtin_elementwise_fma(t, q, half);
    // x beyond -4 stops growing: gelu(x < -4) = gelu(-4).  max(x, -4) written as ONE v_med3 against FLT_MAX: fmaxf costs a canonicalising
    // v_max per operand beside the v_max itself (three operations per element in the ISA), and a med3 against +inf is folded back into
    // that maxnum.  (+inf comes out as FLT_MAX * Phi(4), which the bf16 / fp16 store rounds to +inf again.)
    f32x2 xm;
    xm[0] = __builtin_amdgcn_fmed3f(x[0], -4.0f, 3.402823466e38f);
    xm[1] = __builtin_amdgcn_fmed3f(x[1], -4.0f, 3.402823466e38f);
    // (a NaN pre-activation must stay a NaN: v_med3 returns the smaller of its two other operands for one, i.e. the finite gelu(-4) -- the term
    // 0 * x restores it, so the non-finite loss / gradient-norm guard of mmae_opt_step still sees a poisoned activation; +-inf comes out as NaN
    // instead of +inf / gelu(-4): non-finite either way.  One packed multiply per pair -- ADVICE r5)
    y = __builtin_elementwise_fma(xm, cdf, x * (f32x2){0.f, 0.f});
    dy = __builtin_elementwise_fma(t, r, half);
}
__device__ __forceinline__ void gelu_both_fast4(f32x4 x, f32x4& y, f32x4& dy) {
    f32x2 ya, da, yb, db;
    gelu_both_fast2((f32x2){x[0], x[1]}, ya, da);
    gelu_both_fast2((f32x2){x[2], x[3]}, yb, db);
    y = (f32x4){ya[0], ya[1], yb[0], yb[1]};
    dy = (f32x4){da[0], da[1], db[0], db[1]};
}
__device__ __forceinline__ f32x4 gelu_grad_fast4(f32x4 x) { f32x4 y, dy; gelu_both_fast4(x, y, dy); return dy; }

// ---- MX-fp8 quantisation helpers (mxfp8.hip, the ..._Q GEMM epilogue flavours, the LayerNorm kernels) ----------------
// shared exponent of a block as a biased E8M0 byte, clamped at 0: floor(log2(amax)) - emax(e4m3 = 8) as OCP MX v1.0 section 6.3 has
// it, PLUS ONE when that scale would push the block's largest element past 448 (amax's mantissa > 1.75): the specification's rule
// saturates such elements -- up to 12.5 % off on exactly the largest value of the block, a systematic shrink that moved the
// initial loss of the cfg1 recipe by 1.2 % -- rounding the scale up costs that block one bit instead (the choice of NVIDIA's
// MX-fp8 pre-training recipe, arXiv 2506.08027 section 3)
__device__ __forceinline__ int mx_shared_exp(float amax) {
    const unsigned b = __float_as_uint(amax);
    const int e = (int)((b >> 23) & 0xffu) - 8 + ((b & 0x7fffffu) > 0x600000u ? 1 : 0);
    return e < 0 ? 0 : e;
}
__device__ __forceinline__ float mx_inv_scale(int e) { return __uint_as_float((unsigned)(254 - e) << 23); }   // 2^(127 - e)
__device__ __forceinline__ float mx_clamp448(float v) { return __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f); }
__device__ __forceinline__ int mx_cvt4_e4m3(float a, float b, float c, float d) {
    int r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(mx_clamp448(a), mx_clamp448(b), r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(mx_clamp448(c), mx_clamp448(d), r, true);
    return r;
}
// byte address of block kb (32 elements) of row r in the packed scale array u32 S[ceil(K/256)][rows][2] (include/mmae.h)
__device__ __forceinline__ long long mx_scale_addr(long long rows, long long r, int kb) {
    return (((long long)(kb >> 3) * rows + r) * 2 + (kb & 1)) * 4 + ((kb >> 1) & 3);
}

// ---- wave / block reductions ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
