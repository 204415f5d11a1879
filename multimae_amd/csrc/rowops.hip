// HBM-bound row kernels: LayerNorm fwd/bwd, row softmax fwd/bwd, column sums, casts.
// One 64-lane wavefront per row, 16-byte vector accesses, shuffle reductions (no LDS on the
// per-row critical path).  Roofline for all of them is HBM bytes / 8 TB/s.
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// LayerNorm forward.  x f32 [R][D] -> y act [R][D], mean/rstd f32 [R].  D % 4 == 0, D <= 1024.
// lane l owns float4 chunks l, l+64, l+128, l+192 of the row (NV chunks).
// ------------------------------------------------------------------------------------------
// MXQ (bf16 y, D % 32 == 0): also write the MX-fp8 quantisation of the bf16 row (what mmae_mx_quant would make of y): a
// 32-element block is the 8 adjacent lanes of one chunk round
template <int NV, typename YT, bool MXQ = false>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, YT* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     long long R, int D, float eps, unsigned char* __restrict__ qy = nullptr,
                                                     unsigned char* __restrict__ sc = nullptr) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* xr = x + row * D;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) { v[i] = ld4(xr + c); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
        else { v[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mu; q += d * d; }
        }
    }
    const float var = wave_sum(q) / (float)D;
    const float rs = 1.0f / sqrtf(var + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    YT* yr = y + row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            const f32x4 g = ld4(gamma + c), b = ld4(beta + c);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
            st4(yr + c, o);
            if (MXQ) {
                float w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = bf16_bits_to_f32(f32_to_bf16_bits(o[j]));
                float am = fmaxf(fmaxf(fabsf(w[0]), fabsf(w[1])), fmaxf(fabsf(w[2]), fabsf(w[3])));
                am = fmaxf(am, __shfl_xor(am, 1, 64)); am = fmaxf(am, __shfl_xor(am, 2, 64)); am = fmaxf(am, __shfl_xor(am, 4, 64));
                const int e = mx_shared_exp(am);
                const float inv = mx_inv_scale(e);
                *reinterpret_cast<int*>(qy + row * D + c) = mx_cvt4_e4m3(w[0] * inv, w[1] * inv, w[2] * inv, w[3] * inv);
                if ((lane & 7) == 0) sc[mx_scale_addr(R, row, c >> 5)] = (unsigned char)e;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward.  Each block walks rows block-stride; each wave keeps dgamma/dbeta
// partial sums for its column chunks in registers; the 4 waves are combined through LDS and
// the block writes part[blk][0][D] (dgamma), part[blk][1][D] (dbeta) and part[blk][2][D] (column sums of
// dx_out = the bias gradient of the Linear whose output this residual-stream gradient is).
// ------------------------------------------------------------------------------------------
template <int NV, typename DT, typename AT>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const DT* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dx_in,
                                                     float* __restrict__ dx_out, AT* __restrict__ dx_act,
                                                     float* __restrict__ part, long long R, int D, int part_rows) {
    // cross-wave combine of the column partials in two rounds through a [2][3][D] buffer (18 KB at D = 768): with the former
    // [4][3][1024] (48 KB) only three workgroups fitted a CU, so the 1 024-workgroup grid ran as 1.33 rounds of the chip
    __shared__ float red[2][3][NV * 256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 g[NV], dg[NV], db[NV], dxs[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        g[i] = (c < D) ? ld4(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        dxs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // Software-pipelined row walk: the loads of the wave's NEXT row are issued before the two wave reductions of the current
    // one, so every wave keeps two rows of HBM traffic in flight (with one row per iteration and ~8 waves per CU the kernel
    // sat at ~60 % of the HBM rate).
    const long long step = (long long)gridDim.x * 4;
    long long row = (long long)blockIdx.x * 4 + w;
    f32x4 nd[NV], nx[NV], na[NV];
    float nmu = 0.f, nrs = 0.f;
    auto fetch = [&](long long r) {
        nmu = mean[r]; nrs = rstd[r];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < D) {
                nd[i] = ld4(dy + r * D + c);
                nx[i] = ld4(x + r * D + c);
                if (dx_in) na[i] = ld4(dx_in + r * D + c);
            }
        }
    };
    if (row < R) fetch(row);
    for (; row < R; row += step) {
        const float mu = nmu, rs = nrs;
        f32x4 d_[NV], xv_[NV], a_[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) { d_[i] = nd[i]; xv_[i] = nx[i]; a_[i] = na[i]; }
        if (row + step < R) fetch(row + step);
        f32x4 xh[NV], dh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < D) {
                const f32x4 d = d_[i], xv = xv_[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[i][j] = (xv[j] - mu) * rs;
                    dh[i][j] = d[j] * g[i][j];
                    s1 += dh[i][j];
                    s2 += dh[i][j] * xh[i][j];
                    dg[i][j] += d[j] * xh[i][j];
                    db[i][j] += d[j];
                }
            } else { xh[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dh[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < D) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rs * (dh[i][j] - c1 - xh[i][j] * c2);
                if (dx_in) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += a_[i][j];
                }
                st4(dx_out + row * D + c, o);
                if (dx_act) st4(dx_act + row * D + c, o);
#pragma unroll
                for (int j = 0; j < 4; ++j) dxs[i][j] += o[j];
            }
        }
    }
    if (w < 2) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { red[w][0][c + j] = dg[i][j]; red[w][1][c + j] = db[i][j]; red[w][2][c + j] = dxs[i][j]; }
            }
        }
    }
    __syncthreads();
    if (w >= 2) {                                            // waves 2, 3 add onto the rows of waves 0, 1 (each lane its own columns)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { red[w - 2][0][c + j] += dg[i][j]; red[w - 2][1][c + j] += db[i][j]; red[w - 2][2][c + j] += dxs[i][j]; }
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * D; c += 256) {
        const int k = c / D, cc = c % D;
        part[((long long)blockIdx.x * 3 + k) * D + cc] = red[0][k][cc] + red[1][k][cc];
    }
    // the caller's partial block has part_rows rows (mmae_layernorm_bwd_nblk); the grid may be smaller (one resident round of
    // the chip): the rows nobody owns are zeros
    for (long long r = (long long)blockIdx.x + gridDim.x; r < part_rows; r += gridDim.x)
        for (int c = threadIdx.x; c < 3 * D; c += 256) part[r * 3 * D + c] = 0.f;
}

// Up to 8 destination segments for a column-sum result: column c goes to dst[c / seg_w][c % seg_w] (a null segment is
// dropped).  Lets ONE reduction feed several parameter gradients (LayerNorm dgamma | dbeta | the bias gradient riding along).
struct ColDst { float* dst[8]; int seg_w; const float* unscale; };      // unscale: mmae_colsum_job.unscale (fp16-storage gradients)

// out[c] (+)= sum_r part[r][c].  Workgroup = 64 columns x 4 row phases (coalesced 256-B rows, 4-way
// unrolled so 16 loads are in flight per lane), fixed summation order (deterministic).
__global__ void __launch_bounds__(256) colsum_partials_kernel(const float* __restrict__ part, const ColDst out, int nrows,
                                                              int ncols, int accumulate) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < ncols) {
        int r = ty;
        for (; r + 12 < nrows; r += 16) {
            s0 += part[(long long)(r + 0) * ncols + c]; s1 += part[(long long)(r + 4) * ncols + c];
            s2 += part[(long long)(r + 8) * ncols + c]; s3 += part[(long long)(r + 12) * ncols + c];
        }
        for (; r < nrows; r += 4) s0 += part[(long long)r * ncols + c];
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && c < ncols) {
        const float s = ((red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx])) * h16_grad_unscale(out.unscale);
        const int seg = c / out.seg_w;
        float* o = out.dst[seg];
        if (o) { o += c - seg * out.seg_w; *o = accumulate ? *o + s : s; }
    }
}

// C[m][n] (+)= sum_s ws[s][m][n]  (split-K combine; ws dense [splits][M][N], C row stride ldc)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N,
                                                            long long ldc, int splits, int accumulate) {
    const long long total4 = ((long long)M * N) >> 2;
    const long long slab = (long long)M * N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long e = i << 2;
        // four slabs per iteration, four loads in flight (the dW products of the K = 256 decoder layers use up to 61 slabs; one
        // dependent load per slab made this reduce 17 us regardless of size).  Fixed association: ((s0+s1)+(s2+s3)) per group.
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 4 <= splits; s += 4) {
            const f32x4 b0 = ld4(ws + (long long)s * slab + e), b1 = ld4(ws + (long long)(s + 1) * slab + e),
                        b2 = ld4(ws + (long long)(s + 2) * slab + e), b3 = ld4(ws + (long long)(s + 3) * slab + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += (b0[j] + b1[j]) + (b2[j] + b3[j]);
        }
        for (; s < splits; ++s) { const f32x4 b = ld4(ws + (long long)s * slab + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += b[j];
        }
        const int m = (int)(e / N), n = (int)(e % N);
        float* c = C + (long long)m * ldc + n;
        if (accumulate) { const f32x4 o = ld4(c);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += o[j];
        }
        st4(c, a);
    }
}

// column sums of dy [M][ld] (N columns, N % 4 == 0): grid (ceil(N/256), NSPLIT); thread owns 4 columns.  The row walk is
// 4-way unrolled (4 independent 16-byte loads in flight per lane): with a single dependent load per iteration and only a
// handful of workgroups this kernel was pure latency -- 66 us to reduce a 512 x 2304 LayerNorm partial block.
template <typename DT>
__global__ void __launch_bounds__(256) colsum_kernel(const DT* __restrict__ dy, long long M, int N, long long ld,
                                                     float* __restrict__ ws) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (c < N) {
        const long long step = (long long)gridDim.y * 4;
        long long r = (long long)blockIdx.y * 4 + w;
        for (; r + 3 * step < M; r += 4 * step) {
            const f32x4 v0 = ld4(dy + r * ld + c), v1 = ld4(dy + (r + step) * ld + c), v2 = ld4(dy + (r + 2 * step) * ld + c),
                        v3 = ld4(dy + (r + 3 * step) * ld + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0[j] += v0[j]; s1[j] += v1[j]; s2[j] += v2[j]; s3[j] += v3[j]; }
        }
        for (; r < M; r += step) {
            const f32x4 v = ld4(dy + r * ld + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) s0[j] += v[j];
        }
    }
    f32x4 s;
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = (s0[j] + s1[j]) + (s2[j] + s3[j]);
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < N) {
        f32x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j]);
        st4(ws + (long long)blockIdx.y * N + c, t);
    }
}

// any N / ld (class counts that are not multiples of 4): one column per lane
template <typename DT>
__global__ void __launch_bounds__(256) colsum_scalar_kernel(const DT* __restrict__ dy, long long M, int N, long long ld,
                                                            float* __restrict__ ws) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < N) for (long long r = (long long)blockIdx.y * 4 + w; r < M; r += (long long)gridDim.y * 4) s += ActT<DT>::ld(dy + r * ld + c);
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < N) ws[(long long)blockIdx.y * N + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ------------------------------------------------------------------------------------------
// Row softmax over materialised scores, n <= 256.  One wave per row; lane l owns l, l+64, ...
// ------------------------------------------------------------------------------------------
template <typename PT>
__global__ void __launch_bounds__(256) softmax_fwd_kernel(const float* __restrict__ S, long long lds_, PT* __restrict__ P,
                                                          long long ldp, long long rows, int n, float scale) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* s = S + row * lds_;
    float v[4];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = (c < n) ? s[c] * scale : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = (lane + 64 * i < n) ? expf(v[i] - mx) : 0.f; sum += v[i]; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    PT* p = P + row * ldp;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < ldp) ActT<PT>::st(p + c, (c < n) ? v[i] * inv : 0.f);
    }
}

template <typename PT>
__global__ void __launch_bounds__(256) softmax_bwd_kernel(const PT* __restrict__ P, long long ldp,
                                                          const float* __restrict__ dP, long long lddp,
                                                          PT* __restrict__ dS, long long ldds, long long rows, int n,
                                                          float scale) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float p[4], d[4], dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        p[i] = (c < n) ? ActT<PT>::ld(P + row * ldp + c) : 0.f;
        d[i] = (c < n) ? dP[row * lddp + c] : 0.f;
        dot += p[i] * d[i];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < ldds) ActT<PT>::st(dS + row * ldds + c, (c < n) ? scale * p[i] * (d[i] - dot) : 0.f);
    }
}

// ------------------------------------------------------------------------------------------
// casts / transposes / axpy
// ------------------------------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ s, uint16_t* __restrict__ d, long long n) {
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long step = (long long)gridDim.x * blockDim.x * 4;
    for (long long i = i4; i < n; i += step) {
        if (i + 4 <= n) st4(d + i, ld4(s + i));
        else for (long long j = i; j < n; ++j) d[j] = f32_to_bf16_bits(s[j]);
    }
}
__global__ void cast_bf16_f32_kernel(const uint16_t* __restrict__ s, float* __restrict__ d, long long n) {
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long step = (long long)gridDim.x * blockDim.x * 4;
    for (long long i = i4; i < n; i += step) {
        if (i + 4 <= n) st4(d + i, ld4(s + i));
        else for (long long j = i; j < n; ++j) d[j] = bf16_bits_to_f32(s[j]);
    }
}
// fp16 storage (MMAE_F16); UP: true f32 -> fp16 * S, false fp16 -> f32 * 1/S (S from the adapter's dy_amax scalar, 1 without)
template <bool UP>
__global__ void cast_f16_kernel(const void* __restrict__ sv, void* __restrict__ dv, long long n, const float* __restrict__ amax) {
    const float sc = UP ? h16_grad_scale(amax) : h16_grad_unscale(amax);
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long step = (long long)gridDim.x * blockDim.x * 4;
    typedef typename std::conditional<UP, float, h16_t>::type ST;
    typedef typename std::conditional<UP, h16_t, float>::type DT;
    const ST* s = (const ST*)sv; DT* d = (DT*)dv;
    for (long long i = i4; i < n; i += step) {
        if (i + 4 <= n) { f32x4 v = ld4(s + i); v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc; st4(d + i, v); }
        else for (long long j = i; j < n; ++j) ActT<DT>::st(d + j, ActT<ST>::ld(s + j) * sc);
    }
}
// dst[c][r] = src[r][c]; 64x64 tiles through LDS (+1 pad), coalesced on both sides.
template <typename DT>
__global__ void __launch_bounds__(256) transpose_cast_kernel(const float* __restrict__ src, DT* __restrict__ dst, int rows,
                                                             int cols) {
    __shared__ float t[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        t[i][tx] = (r < rows && c < cols) ? src[(long long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) ActT<DT>::st(dst + (long long)c * rows + r, t[tx][i]);
    }
}
// out = in[0] + in[1] + ... + in[n - 1] in index order (n <= 8): the sum of the output adapters' encoder-token gradients in ONE pass (round 6; autograd's chain of
// at::native add kernels read and wrote the running sum n - 1 times)
struct AddNArgs { const float* in[8]; int n; };
__global__ void __launch_bounds__(256) add_n_kernel(float* __restrict__ out, const AddNArgs a, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 s = ld4(a.in[0] + i * 4);
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            if (k < a.n) {
                const f32x4 v = ld4(a.in[k] + i * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[j] += v[j];
            }
        }
        st4(out + i * 4, s);
    }
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long long n) {
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long step = (long long)gridDim.x * blockDim.x * 4;
    for (long long i = i4; i < n; i += step) {
        if (i + 4 <= n) {
            f32x4 yv = ld4(y + i); const f32x4 xv = ld4(x + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) yv[j] += a * xv[j];
            st4(y + i, yv);
        } else for (long long j = i; j < n; ++j) y[j] += a * x[j];
    }
}

// stochastic depth on [R][D] activations, N rows per sample: out = resid + s[r / N] * y  /  out = cast(s[r / N] * x)
__global__ void __launch_bounds__(256) rowscale_add_kernel(const float* __restrict__ resid, const float* __restrict__ y, const float* __restrict__ s,
                                                           float* __restrict__ out, long long total4, int d4, int N) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const float sc = s[(i / d4) / N];
        const f32x4 r = ld4(resid + i * 4), v = ld4(y + i * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = r[j] + sc * v[j];
        st4(out + i * 4, o);
    }
}
template <typename DT>
__global__ void __launch_bounds__(256) rowscale_cast_kernel(const float* __restrict__ x, const float* __restrict__ s, DT* __restrict__ out,
                                                            long long total4, int d4, int N) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const float sc = s[(i / d4) / N];
        f32x4 v = ld4(x + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= sc;
        st4(out + i * 4, v);
    }
}

// nn.Dropout as one elementwise pass (mmae_dropout): out = [resid +] keep ? x * scale [* s[i / per]] : 0
template <typename XT, typename OT>
__global__ void __launch_bounds__(256) dropout_kernel(const XT* __restrict__ x, const unsigned char* __restrict__ keep, float scale,
                                                      const float* __restrict__ s, long long per4, const float* __restrict__ resid,
                                                      OT* __restrict__ out, long long total4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const unsigned k4 = *reinterpret_cast<const unsigned*>(keep + i * 4);
        const float sc = s ? scale * s[i / per4] : scale;
        const f32x4 v = ld4(x + i * 4);
        f32x4 o = resid ? ld4(resid + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += ((k4 >> (8 * j)) & 0xffu) ? v[j] * sc : 0.f;
        st4(out + i * 4, o);
    }
}

// y[b][:] = mean_n x[b][n][:]   and its backward  dx[b][n][:] = dy[b][:] / N   (LinearOutputAdapter's mean pooling)
__global__ void __launch_bounds__(256) token_mean_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int D) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= D) return;
    const float* xb = x + (long long)blockIdx.y * N * D + c;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int n = 0;
    for (; n + 1 < N; n += 2) {
        const f32x4 a = ld4(xb + (long long)n * D), b = ld4(xb + (long long)(n + 1) * D);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s0[j] += a[j]; s1[j] += b[j]; }
    }
    if (n < N) { const f32x4 a = ld4(xb + (long long)n * D);
#pragma unroll
        for (int j = 0; j < 4; ++j) s0[j] += a[j];
    }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (s0[j] + s1[j]) / (float)N;
    st4(y + (long long)blockIdx.y * D + c, o);
}
__global__ void __launch_bounds__(256) token_mean_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int D, long long total4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // float4 index over [B][N][D]
    if (i >= total4) return;
    const int d4 = D >> 2;
    const long long b = i / ((long long)N * d4);
    const int c = (int)(i % d4) * 4;
    f32x4 g = ld4(dy + b * D + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] /= (float)N;
    st4(dx + i * 4, g);
}

inline int stream_grid(long long n_vec4) { long long b = (n_vec4 + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

// out[m] (+)= sum_s part[s][m]   (fixed order)
__global__ void __launch_bounds__(256) acs_reduce_kernel(const float* __restrict__ part, int splits, int M, float* __restrict__ out, int accumulate) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float a = accumulate ? out[m] : 0.f;
    for (int s = 0; s < splits; ++s) a += part[(long long)s * M + m];
    out[m] = a;
}

}  // namespace

int mmae_acs_reduce(const float* part, int splits, int M, float* out, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(acs_reduce_kernel, dim3((M + 255) / 256), dim3(256), 0, st, part, splits, M, out, accumulate);
    return mmae_check_launch("acs_reduce");
}

int mmae_splitk_reduce(const float* ws, float* C, int M, int N, long long ldc, int splits, int accumulate, hipStream_t st) {
    MMAE_REQUIRE((N & 3) == 0 && (ldc & 3) == 0 && ((uintptr_t)C % 16) == 0, "gemm: split_k needs N, ldc multiples of 4 and an aligned C");
    long long nb = ((long long)M * N / 4 + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, ws, C, M, N, ldc, splits, accumulate);
    return mmae_check_launch("splitk_reduce");
}

extern "C" {

int mmae_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype, float* mean,
                       float* rstd, int64_t R, int D, float eps, void* stream) {
    MMAE_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
    MMAE_REQUIRE(D % 4 == 0 && D >= 4 && D <= 1024, "layernorm_fwd: D must be a multiple of 4 in [4,1024]");
    MMAE_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0) && ((uintptr_t)y % 8 == 0),
                 "layernorm_fwd: unaligned pointer");
    if (R <= 0) return 0;
    const int nv = (D + 255) / 256;
    dim3 grid((unsigned)cdiv64(R, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define LN_FWD(NV)                                                                                                        \
    if (y_dtype == MMAE_BF16) hipLaunchKernelGGL((ln_fwd_kernel<NV, uint16_t>), grid, block, 0, st, x, gamma, beta,       \
                                                  (uint16_t*)y, mean, rstd, (long long)R, D, eps);                       \
    else if (y_dtype == MMAE_F16) hipLaunchKernelGGL((ln_fwd_kernel<NV, h16_t>), grid, block, 0, st, x, gamma, beta,      \
                                                      (h16_t*)y, mean, rstd, (long long)R, D, eps);                      \
    else hipLaunchKernelGGL((ln_fwd_kernel<NV, float>), grid, block, 0, st, x, gamma, beta, (float*)y, mean, rstd,        \
                            (long long)R, D, eps);
    switch (nv) { case 1: LN_FWD(1) break; case 2: LN_FWD(2) break; case 3: LN_FWD(3) break; default: LN_FWD(4) break; }
#undef LN_FWD
    return mmae_check_launch("layernorm_fwd");
}

int mmae_layernorm_fwd_mx(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean, float* rstd, int64_t R, int D,
                          float eps, void* q, void* scales, void* stream) {
    MMAE_REQUIRE(x && gamma && beta && y_bf16 && mean && rstd && q && scales, "layernorm_fwd_mx: null pointer");
    MMAE_REQUIRE(D % 32 == 0 && D >= 32 && D <= 1024, "layernorm_fwd_mx: D must be a multiple of 32 in [32,1024]");
    if (R <= 0) return 0;
    const int nv = (D + 255) / 256;
    dim3 grid((unsigned)cdiv64(R, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (D % 256) {                                        // exponent bytes of the blocks past the last column
        const int rc = mmae_mx_scale_clear(scales, R, D, stream);
        if (rc) return rc;
    }
#define LN_FWD_MX(NV) hipLaunchKernelGGL((ln_fwd_kernel<NV, uint16_t, true>), grid, block, 0, st, x, gamma, beta, (uint16_t*)y_bf16, mean, rstd, \
                                         (long long)R, D, eps, (unsigned char*)q, (unsigned char*)scales);
    switch (nv) { case 1: LN_FWD_MX(1) break; case 2: LN_FWD_MX(2) break; case 3: LN_FWD_MX(3) break; default: LN_FWD_MX(4) break; }
#undef LN_FWD_MX
    return mmae_check_launch("layernorm_fwd_mx");
}

// workgroups (= rows of the partial block): 4 rows per workgroup pass, capped at 1 024 -- 2 048 for the long, narrow decoder
// activations (50 176 x 256), whose one-float4-per-lane rows need more waves in flight to cover the HBM latency
int mmae_layernorm_bwd_nblk(int64_t R) { const int64_t b = cdiv64(R, 4), cap = R >= 32768 ? 2048 : 1024; return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

int mmae_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean,
                       const float* rstd, const float* dx_in, float* dx_out, void* dx_act, int dx_act_dtype, float* part,
                       int64_t R, int D, void* stream) {
    MMAE_REQUIRE(dy && x && gamma && mean && rstd && dx_out && part, "layernorm_bwd: null pointer");
    MMAE_REQUIRE(D % 4 == 0 && D >= 4 && D <= 1024, "layernorm_bwd: D must be a multiple of 4 in [4,1024]");
    MMAE_REQUIRE(R > 0, "layernorm_bwd: empty");
    const int nv = (D + 255) / 256;
    // The kernel's registers (83 / 127 / 168 / 209 VGPRs for D <= 256 / 512 / 768 / 1024) allow 5 / 4 / 3 / 2 waves per SIMD =
    // workgroups per CU: a 1 024-workgroup grid at D = 768 ran as 1.33 rounds of the chip (768 resident).  Launch what is
    // resident at once -- every workgroup walks rows block-stride, so fewer workgroups just take more rows each.  (Measured: -0.1 ms
    // per cfg3 step, i.e. close to nothing -- the kernel is HBM-bound either way; MMAE_LN_BWD_RESIDENT=0 restores the full grid.)
    const int n_cu = mmae_cu_count();
    static const int env_res = mmae_env_int("MMAE_LN_BWD_RESIDENT", 1);
    const int part_rows = mmae_layernorm_bwd_nblk(R);
    const int per_cu = nv == 1 ? 5 : (nv == 2 ? 4 : (nv == 3 ? 3 : 2));
    const int resident = env_res ? n_cu * per_cu : part_rows;
    dim3 grid(part_rows < resident ? part_rows : resident), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool dyb = dy_dtype == MMAE_BF16, axb = dx_act_dtype == MMAE_BF16;
    const bool dyh = dy_dtype == MMAE_F16, axh = dx_act_dtype == MMAE_F16;       // fp16 storage: the gradient's scale passes through (linear in dy, dx_in)
    MMAE_REQUIRE(!(dyh && axb) && !(dyb && axh), "layernorm_bwd: bf16 and fp16 tensors do not mix");
#define LN_BWD(NV, DT, AT)                                                                                                \
    hipLaunchKernelGGL((ln_bwd_kernel<NV, DT, AT>), grid, block, 0, st, (const DT*)dy, x, gamma, mean, rstd, dx_in, dx_out, \
                       (AT*)dx_act, part, (long long)R, D, part_rows)
#define LN_BWD_T(NV)                                                                                                      \
    if (dyh && axh) LN_BWD(NV, h16_t, h16_t); else if (dyh) LN_BWD(NV, h16_t, float);                                    \
    else if (dyb && axb) LN_BWD(NV, uint16_t, uint16_t); else if (dyb) LN_BWD(NV, uint16_t, float);                      \
    else if (axb) LN_BWD(NV, float, uint16_t); else LN_BWD(NV, float, float);
    switch (nv) { case 1: LN_BWD_T(1) break; case 2: LN_BWD_T(2) break; case 3: LN_BWD_T(3) break; default: LN_BWD_T(4) break; }
#undef LN_BWD_T
#undef LN_BWD
    return mmae_check_launch("layernorm_bwd");
}

static int launch_partials(const float* part, const ColDst& d, int nrows, int ncols, int accumulate, void* stream) {
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((ncols + 63) / 64), dim3(256), 0, (hipStream_t)stream, part, d, nrows, ncols,
                       accumulate);
    return mmae_check_launch("colsum_partials");
}

int mmae_colsum_partials(const float* part, float* out, int nrows, int ncols, int accumulate, void* stream) {
    MMAE_REQUIRE(part && out && nrows > 0 && ncols > 0, "colsum_partials: bad argument");
    ColDst d = {};
    d.dst[0] = out; d.seg_w = ncols;
    return launch_partials(part, d, nrows, ncols, accumulate, stream);
}

// Row-split factor: enough workgroups to cover the load latency even for the short, wide partial blocks the backward pass
// reduces (LayerNorm: 512 x 3D, dGELU: ceil(M/64) x 4D), <= 256 slabs for the long activations.
static int colsum_nsplit(int64_t M) { const int64_t s = cdiv64(M, 16); return (int)(s < 1 ? 1 : (s > 256 ? 256 : s)); }
int64_t mmae_colsum_ws_elems(int64_t M, int N) { return (int64_t)colsum_nsplit(M) * N; }

static int colsum_impl(const void* dy, int dtype, int64_t M, int N, int64_t ld, const ColDst& d, int accumulate, float* ws,
                       void* stream) {
    const int ns = colsum_nsplit(M);
    if (ns == 1 && dtype == MMAE_F32 && ld == N) return launch_partials((const float*)dy, d, (int)M, N, accumulate, stream);
    hipStream_t st = (hipStream_t)stream;
    if (N % 4 != 0 || ld % 4 != 0) {
        dim3 g1((N + 63) / 64, ns);
        if (dtype == MMAE_BF16) hipLaunchKernelGGL((colsum_scalar_kernel<uint16_t>), g1, dim3(256), 0, st, (const uint16_t*)dy, (long long)M, N, (long long)ld, ws);
        else if (dtype == MMAE_F16) hipLaunchKernelGGL((colsum_scalar_kernel<h16_t>), g1, dim3(256), 0, st, (const h16_t*)dy, (long long)M, N, (long long)ld, ws);
        else hipLaunchKernelGGL((colsum_scalar_kernel<float>), g1, dim3(256), 0, st, (const float*)dy, (long long)M, N, (long long)ld, ws);
        int rc1 = mmae_check_launch("colsum");
        if (rc1) return rc1;
        return launch_partials(ws, d, ns, N, accumulate, stream);
    }
    dim3 grid((N + 255) / 256, ns), block(256);
    if (dtype == MMAE_BF16) hipLaunchKernelGGL((colsum_kernel<uint16_t>), grid, block, 0, st, (const uint16_t*)dy, (long long)M, N, (long long)ld, ws);
    else if (dtype == MMAE_F16) hipLaunchKernelGGL((colsum_kernel<h16_t>), grid, block, 0, st, (const h16_t*)dy, (long long)M, N, (long long)ld, ws);
    else hipLaunchKernelGGL((colsum_kernel<float>), grid, block, 0, st, (const float*)dy, (long long)M, N, (long long)ld, ws);
    int rc = mmae_check_launch("colsum");
    if (rc) return rc;
    return launch_partials(ws, d, ns, N, accumulate, stream);
}

int mmae_colsum(const void* dy, int dtype, int64_t M, int N, int64_t ld, float* out, int accumulate, float* ws,
                void* stream) {
    MMAE_REQUIRE(dy && out && ws && M > 0 && N > 0, "colsum: bad argument");
    ColDst d = {};
    d.dst[0] = out; d.seg_w = N;
    return colsum_impl(dy, dtype, M, N, ld, d, accumulate, ws, stream);
}

int mmae_colsum_scatter(const void* dy, int dtype, int64_t M, int N, int64_t ld, int seg_w, const void* dsts, int nseg,
                        int accumulate, float* ws, void* stream) {
    MMAE_REQUIRE(dy && dsts && ws && M > 0 && N > 0, "colsum_scatter: bad argument");
    MMAE_REQUIRE(seg_w > 0 && nseg >= 1 && nseg <= 8 && (int64_t)seg_w * nseg >= N, "colsum_scatter: need 1..8 segments covering N columns");
    ColDst d = {};
    for (int i = 0; i < nseg; ++i) d.dst[i] = ((float* const*)dsts)[i];
    d.seg_w = seg_w;
    return colsum_impl(dy, dtype, M, N, ld, d, accumulate, ws, stream);
}

}  // extern "C"

// ---- batched column sums: several mmae_colsum_scatter jobs in one launch ----------------------------------------------------------
namespace {
constexpr int CB_MAX_GROUPS = 160;                 // 256-column groups per launch (a ViT-L block: 16 + 12 + 12)

struct CbJob { const void* src; long long rows, ld; int dtype, cols, seg_w, ns, group0, ws_off; float* dst[8]; const float* unscale; };
struct CbArgs { int n, accumulate, ngroups; float* ws; CbJob j[MMAE_COLSUM_MAX_JOBS]; };

template <typename DT>
__device__ __forceinline__ f32x4 cb_rows(const DT* src, long long rows, long long ld, int c, int cols, int sp, int ns, int w, bool vec_ok) {
    // rows sp * 4 + w, then every ns * 4 (the walk of colsum_kernel: same partial sums, same order)
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    const long long step = (long long)ns * 4;
    long long r = (long long)sp * 4 + w;
    if (vec_ok && cols - c >= 4) {
        for (; r + 3 * step < rows; r += 4 * step) {
            const f32x4 v0 = ld4(src + r * ld + c), v1 = ld4(src + (r + step) * ld + c), v2 = ld4(src + (r + 2 * step) * ld + c),
                        v3 = ld4(src + (r + 3 * step) * ld + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) { s0[k] += v0[k]; s1[k] += v1[k]; s2[k] += v2[k]; s3[k] += v3[k]; }
        }
        for (; r < rows; r += step) {
            const f32x4 v = ld4(src + r * ld + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) s0[k] += v[k];
        }
    } else {                                       // rows that are not 4-element aligned / a ragged last group (class counts such as 133)
        const int nk = cols - c < 4 ? cols - c : 4;
        for (; r < rows; r += step)
            for (int k = 0; k < nk; ++k) s0[k] += ActT<DT>::ld(src + r * ld + c + k);
    }
    f32x4 s;
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = (s0[k] + s1[k]) + (s2[k] + s3[k]);
    return s;
}

// phase 1: workgroup = (job, 256-column group, row slice) -> one row of the job's slab ws[ns][ncg * 256]
__global__ void __launch_bounds__(256) colsum_batch_rows_kernel(const CbArgs a) {
    __shared__ f32x4 red[4][64];
    int bi = blockIdx.x, ji = 0;
    for (; ji < a.n - 1; ++ji) {
        const int nb = ((a.j[ji].cols + 255) / 256) * a.j[ji].ns;
        if (bi < nb) break;
        bi -= nb;
    }
    const CbJob& J = a.j[ji];
    const int ncg = (J.cols + 255) / 256, cg = bi % ncg, sp = bi / ncg;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = (cg * 64 + lane) * 4;
    const bool vec_ok = (J.ld % 4 == 0) && (J.cols % 4 == 0);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c < J.cols) {
        if (J.dtype == MMAE_BF16) s = cb_rows((const uint16_t*)J.src, J.rows, J.ld, c, J.cols, sp, J.ns, w, vec_ok);
        else if (J.dtype == MMAE_F16) s = cb_rows((const h16_t*)J.src, J.rows, J.ld, c, J.cols, sp, J.ns, w, vec_ok);
        else s = cb_rows((const float*)J.src, J.rows, J.ld, c, J.cols, sp, J.ns, w, vec_ok);
    }
    red[w][lane] = s;
    __syncthreads();
    if (w == 0) {
        f32x4 t;
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = (red[0][lane][k] + red[1][lane][k]) + (red[2][lane][k] + red[3][lane][k]);
        st4(a.ws + J.ws_off + (long long)sp * (ncg * 256) + c, t);
    }
}

// phase 2: workgroup = (job, 256-column group): the group's slices summed in a fixed order, scattered to the destinations
__global__ void __launch_bounds__(256) colsum_batch_reduce_kernel(const CbArgs a) {
    __shared__ f32x4 red[4][64];
    int g = blockIdx.x, ji = 0;
    for (; ji < a.n - 1; ++ji) {
        const int ncg = (a.j[ji].cols + 255) / 256;
        if (g < ncg) break;
        g -= ncg;
    }
    const CbJob& J = a.j[ji];
    const int wcols = ((J.cols + 255) / 256) * 256;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = (g * 64 + lane) * 4;
    const float* slab = a.ws + J.ws_off;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    int r = w;
    for (; r + 4 < J.ns; r += 8) {
        const f32x4 v0 = ld4(slab + (long long)r * wcols + c), v1 = ld4(slab + (long long)(r + 4) * wcols + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) { a0[k] += v0[k]; a1[k] += v1[k]; }
    }
    if (r < J.ns) {
        const f32x4 v = ld4(slab + (long long)r * wcols + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) a0[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) a0[k] += a1[k];
    red[w][lane] = a0;
    __syncthreads();
    if (w == 0) {
        const float us = h16_grad_unscale(J.unscale);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cc = c + k;
            if (cc >= J.cols) break;
            const float v = ((red[0][lane][k] + red[1][lane][k]) + (red[2][lane][k] + red[3][lane][k])) * us;
            const int seg = cc / J.seg_w;
            float* o = J.dst[seg];
            if (o) { o += cc - seg * J.seg_w; *o = a.accumulate ? *o + v : v; }
        }
    }
}

int cb_nsplit(long long rows) { const long long s = (rows + 15) / 16; return (int)(s < 1 ? 1 : (s > 64 ? 64 : s)); }

}  // namespace

extern "C" {

int64_t mmae_colsum_batch_ws_elems(const mmae_colsum_job* jobs, int n) {
    if (!jobs || n < 1 || n > MMAE_COLSUM_MAX_JOBS) return -1;
    int64_t e = 0;
    for (int i = 0; i < n; ++i) e += (int64_t)cb_nsplit(jobs[i].rows) * ((jobs[i].cols + 255) / 256) * 256;
    return e;
}

int mmae_colsum_batch(const mmae_colsum_job* jobs, int n, int accumulate, float* ws, int64_t ws_elems, void* stream) {
    MMAE_REQUIRE(jobs && n >= 1 && n <= MMAE_COLSUM_MAX_JOBS && ws, "colsum_batch: bad argument");
    MMAE_REQUIRE(ws_elems >= mmae_colsum_batch_ws_elems(jobs, n) && ((uintptr_t)ws % 16) == 0, "colsum_batch: workspace too small / unaligned");
    hipStream_t st = (hipStream_t)stream;
    CbArgs a = {};
    a.n = n; a.accumulate = accumulate; a.ws = ws;
    int groups = 0, blocks = 0;
    long long off = 0;
    for (int i = 0; i < n; ++i) {
        const mmae_colsum_job& q = jobs[i];
        MMAE_REQUIRE(q.src && q.rows > 0 && q.cols > 0 && q.seg_w > 0 && q.nseg >= 1 && q.nseg <= 8 && (int64_t)q.seg_w * q.nseg >= q.cols,
                     "colsum_batch: bad job");
        MMAE_REQUIRE(q.dtype == MMAE_F32 || q.dtype == MMAE_BF16 || q.dtype == MMAE_F16, "colsum_batch: bad dtype");
        MMAE_REQUIRE(!(q.ld % 4 == 0 && q.cols % 4 == 0) || ((uintptr_t)q.src % (q.dtype == MMAE_F32 ? 16 : 8)) == 0, "colsum_batch: unaligned source");
        CbJob& J = a.j[i];
        J.src = q.src; J.rows = q.rows; J.ld = q.ld; J.dtype = q.dtype; J.cols = q.cols; J.seg_w = q.seg_w;
        J.ns = cb_nsplit(q.rows); J.group0 = groups; J.ws_off = (int)off; J.unscale = q.unscale;
        for (int k = 0; k < 8; ++k) J.dst[k] = k < q.nseg ? q.dst[k] : nullptr;
        const int ncg = (q.cols + 255) / 256;
        groups += ncg; blocks += ncg * J.ns;
        off += (long long)J.ns * ncg * 256;
        MMAE_REQUIRE(off < 0x7fffffffLL, "colsum_batch: workspace offset overflow");
    }
    a.ngroups = groups;
    if (groups > CB_MAX_GROUPS) {                  // outside the batched form: one scatter per job
        for (int i = 0; i < n; ++i) {
            const mmae_colsum_job& q = jobs[i];
            ColDst d = {};
            for (int k = 0; k < q.nseg; ++k) d.dst[k] = q.dst[k];
            d.seg_w = q.seg_w; d.unscale = q.unscale;
            MMAE_REQUIRE(ws_elems >= mmae_colsum_ws_elems(q.rows, q.cols), "colsum_batch: workspace too small for the per-job form");
            const int rc = colsum_impl(q.src, q.dtype, q.rows, q.cols, q.ld, d, accumulate, ws, stream);
            if (rc) return rc;
        }
        return 0;
    }
    hipLaunchKernelGGL(colsum_batch_rows_kernel, dim3(blocks), dim3(256), 0, st, a);
    int rc = mmae_check_launch("colsum_batch_rows");
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_batch_reduce_kernel, dim3(groups), dim3(256), 0, st, a);
    return mmae_check_launch("colsum_batch_reduce");
}

int mmae_softmax_fwd(const float* S, int64_t lds_, void* P, int p_dtype, int64_t ldp, int64_t rows, int n, float scale,
                     void* stream) {
    MMAE_REQUIRE(S && P && rows > 0, "softmax_fwd: bad argument");
    MMAE_REQUIRE(n >= 1 && n <= 256 && ldp >= n && ldp <= 256 && lds_ >= n, "softmax_fwd: need 1 <= n <= ld <= 256");
    dim3 grid((unsigned)cdiv64(rows, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (p_dtype == MMAE_BF16) hipLaunchKernelGGL((softmax_fwd_kernel<uint16_t>), grid, block, 0, st, S, (long long)lds_, (uint16_t*)P, (long long)ldp, (long long)rows, n, scale);
    else hipLaunchKernelGGL((softmax_fwd_kernel<float>), grid, block, 0, st, S, (long long)lds_, (float*)P, (long long)ldp, (long long)rows, n, scale);
    return mmae_check_launch("softmax_fwd");
}

int mmae_softmax_bwd(const void* P, int p_dtype, int64_t ldp, const float* dP, int64_t lddp, void* dS, int64_t ldds,
                     int64_t rows, int n, float scale, void* stream) {
    MMAE_REQUIRE(P && dP && dS && rows > 0, "softmax_bwd: bad argument");
    MMAE_REQUIRE(n >= 1 && n <= 256 && ldp >= n && ldds >= n && ldds <= 256 && lddp >= n, "softmax_bwd: need 1 <= n <= ld <= 256");
    dim3 grid((unsigned)cdiv64(rows, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (p_dtype == MMAE_BF16) hipLaunchKernelGGL((softmax_bwd_kernel<uint16_t>), grid, block, 0, st, (const uint16_t*)P, (long long)ldp, dP, (long long)lddp, (uint16_t*)dS, (long long)ldds, (long long)rows, n, scale);
    else hipLaunchKernelGGL((softmax_bwd_kernel<float>), grid, block, 0, st, (const float*)P, (long long)ldp, dP, (long long)lddp, (float*)dS, (long long)ldds, (long long)rows, n, scale);
    return mmae_check_launch("softmax_bwd");
}

int mmae_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    MMAE_REQUIRE(src && dst && n >= 0, "cast: bad argument");
    MMAE_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 8 == 0), "cast: unaligned");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, (long long)n);
    return mmae_check_launch("cast_f32_to_bf16");
}
int mmae_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
    MMAE_REQUIRE(src && dst && n >= 0, "cast: bad argument");
    MMAE_REQUIRE(((uintptr_t)src % 8 == 0) && ((uintptr_t)dst % 16 == 0), "cast: unaligned");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, dst, (long long)n);
    return mmae_check_launch("cast_bf16_to_f32");
}
int mmae_cast_f32_to_f16(const float* src, void* dst, int64_t n, const float* scale_amax, void* stream) {
    MMAE_REQUIRE(src && dst && n >= 0, "cast: bad argument");
    MMAE_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 8 == 0), "cast: unaligned");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_f16_kernel<true>, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const void*)src, dst, (long long)n, scale_amax);
    return mmae_check_launch("cast_f32_to_f16");
}
int mmae_cast_f16_to_f32(const void* src, float* dst, int64_t n, const float* scale_amax, void* stream) {
    MMAE_REQUIRE(src && dst && n >= 0, "cast: bad argument");
    MMAE_REQUIRE(((uintptr_t)src % 8 == 0) && ((uintptr_t)dst % 16 == 0), "cast: unaligned");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_f16_kernel<false>, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src, (void*)dst, (long long)n, scale_amax);
    return mmae_check_launch("cast_f16_to_f32");
}
int mmae_transpose_cast(const float* src, void* dst, int dst_dtype, int rows, int cols, void* stream) {
    MMAE_REQUIRE(src && dst && rows > 0 && cols > 0, "transpose_cast: bad argument");
    dim3 grid((cols + 63) / 64, (rows + 63) / 64), block(256);
    if (dst_dtype == MMAE_BF16) hipLaunchKernelGGL((transpose_cast_kernel<uint16_t>), grid, block, 0, (hipStream_t)stream, src, (uint16_t*)dst, rows, cols);
    else hipLaunchKernelGGL((transpose_cast_kernel<float>), grid, block, 0, (hipStream_t)stream, src, (float*)dst, rows, cols);
    return mmae_check_launch("transpose_cast");
}
int mmae_token_mean_fwd(const float* x, float* y, int B, int N, int D, void* stream) {
    MMAE_REQUIRE(x && y && B > 0 && N > 0 && D > 0 && D % 4 == 0, "token_mean_fwd: bad argument");
    MMAE_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), "token_mean_fwd: unaligned");
    hipLaunchKernelGGL(token_mean_fwd_kernel, dim3((D / 4 + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, y, N, D);
    return mmae_check_launch("token_mean_fwd");
}
int mmae_token_mean_bwd(const float* dy, float* dx, int B, int N, int D, void* stream) {
    MMAE_REQUIRE(dy && dx && B > 0 && N > 0 && D > 0 && D % 4 == 0, "token_mean_bwd: bad argument");
    MMAE_REQUIRE(((uintptr_t)dy % 16 == 0) && ((uintptr_t)dx % 16 == 0), "token_mean_bwd: unaligned");
    const long long total4 = (long long)B * N * (D / 4);
    hipLaunchKernelGGL(token_mean_bwd_kernel, dim3((unsigned)cdiv64(total4, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, D, total4);
    return mmae_check_launch("token_mean_bwd");
}
int mmae_rowscale_add(const float* resid, const float* y, const float* s, float* out, int64_t R, int N, int D, void* stream) {
    MMAE_REQUIRE(resid && y && s && out && R > 0 && N > 0 && D > 0 && D % 4 == 0 && R % N == 0, "rowscale_add: bad argument");
    MMAE_REQUIRE(((uintptr_t)resid % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)out % 16 == 0), "rowscale_add: unaligned");
    const long long total4 = (long long)R * (D / 4);
    hipLaunchKernelGGL(rowscale_add_kernel, dim3(stream_grid(total4)), dim3(256), 0, (hipStream_t)stream, resid, y, s, out, total4, D / 4, N);
    return mmae_check_launch("rowscale_add");
}
int mmae_rowscale_cast(const float* x, const float* s, void* out, int out_dtype, int64_t R, int N, int D, void* stream) {
    MMAE_REQUIRE(x && s && out && R > 0 && N > 0 && D > 0 && D % 4 == 0 && R % N == 0, "rowscale_cast: bad argument");
    MMAE_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0), "rowscale_cast: unaligned");
    const long long total4 = (long long)R * (D / 4);
    if (out_dtype == MMAE_BF16) hipLaunchKernelGGL((rowscale_cast_kernel<uint16_t>), dim3(stream_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, s, (uint16_t*)out, total4, D / 4, N);
    else hipLaunchKernelGGL((rowscale_cast_kernel<float>), dim3(stream_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, s, (float*)out, total4, D / 4, N);
    return mmae_check_launch("rowscale_cast");
}
int mmae_dropout(const void* x, int x_dtype, const void* keep, float scale, const float* s, int64_t per, const float* resid, void* out,
                 int out_dtype, int64_t n, void* stream) {
    MMAE_REQUIRE(x && keep && out && n > 0 && n % 4 == 0, "dropout: bad argument (n must be a multiple of 4)");
    MMAE_REQUIRE(!s || (per > 0 && per % 4 == 0 && n % per == 0), "dropout: the per-sample scale needs per % 4 == 0 and n % per == 0");
    MMAE_REQUIRE(!resid || out_dtype == MMAE_F32, "dropout: a residual needs an f32 output");
    // four elements per lane: 16-byte accesses on f32 tensors, 8-byte on 16-bit ones (an f32 view at an odd element offset passed the old 8-byte check)
    MMAE_REQUIRE(((uintptr_t)x % (x_dtype == MMAE_F32 ? 16 : 8) == 0) && ((uintptr_t)out % (out_dtype == MMAE_F32 ? 16 : 8) == 0) && ((uintptr_t)keep % 4 == 0) &&
                 (!resid || (uintptr_t)resid % 16 == 0), "dropout: unaligned (f32 tensors 16 bytes, 16-bit tensors 8 bytes, keep 4 bytes)");
    const long long total4 = n / 4, per4 = s ? per / 4 : 1;
    hipStream_t st = (hipStream_t)stream;
    const unsigned char* k = (const unsigned char*)keep;
#define DROP(XT, OT) hipLaunchKernelGGL((dropout_kernel<XT, OT>), dim3(stream_grid(total4)), dim3(256), 0, st, (const XT*)x, k, scale, s, per4, resid, (OT*)out, total4)
    if (x_dtype == MMAE_F32) {
        if (out_dtype == MMAE_F32) DROP(float, float); else if (out_dtype == MMAE_BF16) DROP(float, uint16_t); else if (out_dtype == MMAE_F16) DROP(float, h16_t);
        else { mmae_set_error("dropout: bad out_dtype"); return MMAE_EINVAL; }
    } else if (x_dtype == MMAE_BF16) {
        if (out_dtype == MMAE_BF16) DROP(uint16_t, uint16_t); else if (out_dtype == MMAE_F32) DROP(uint16_t, float);
        else { mmae_set_error("dropout: bf16 input goes to bf16 or f32"); return MMAE_EINVAL; }
    } else if (x_dtype == MMAE_F16) {
        if (out_dtype == MMAE_F16) DROP(h16_t, h16_t); else if (out_dtype == MMAE_F32) DROP(h16_t, float);
        else { mmae_set_error("dropout: fp16 input goes to fp16 or f32"); return MMAE_EINVAL; }
    } else { mmae_set_error("dropout: bad x_dtype"); return MMAE_EINVAL; }
#undef DROP
    return mmae_check_launch("dropout");
}
int mmae_add_n_f32(float* out, const float* const* in_host, int n_in, int64_t n, void* stream) {
    MMAE_REQUIRE(out && in_host && n_in >= 1 && n_in <= 8 && n >= 0 && n % 4 == 0, "add_n: 1 <= n_in <= 8 inputs, n a multiple of 4");
    AddNArgs a = {};
    a.n = n_in;
    for (int k = 0; k < n_in; ++k) {
        MMAE_REQUIRE(in_host[k] && (uintptr_t)in_host[k] % 16 == 0, "add_n: null / unaligned input");
        a.in[k] = in_host[k];
    }
    MMAE_REQUIRE((uintptr_t)out % 16 == 0, "add_n: unaligned output");
    if (n == 0) return 0;
    hipLaunchKernelGGL(add_n_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, out, a, (long long)(n / 4));
    return mmae_check_launch("add_n");
}
int mmae_axpy_f32(float* y, const float* x, float a, int64_t n, void* stream) {
    MMAE_REQUIRE(y && x && n >= 0, "axpy: bad argument");
    MMAE_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), "axpy: unaligned");
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, x, a, (long long)n);
    return mmae_check_launch("axpy");
}

}  // extern "C"
