// Truncated depth standardisation (run_pretraining_multimae.py:487-492): per sample, mean and unbiased variance of the
// values whose rank lies in [lo, hi) -- the reference sorts all c*h*w = 50 176 values of every depth map per step and slices
// the sorted row -- then (x - mean) / sqrt(var + eps) over the whole map.
//
// No sort: only the two cut VALUES matter.  One workgroup per sample finds the keys of rank lo and hi-1 by a 4-pass 8-bit
// radix select over order-preserving integer keys (both ranks in the same passes, two 256-bin LDS histograms), then sums
// the values strictly between the cuts and adds the right number of copies of the cut values themselves (ties at a cut are
// counted by rank, exactly as the slice of a sorted row would).  The map (196 KB) stays in the workgroup's L2 across the
// passes; HBM sees one read and one write.  Sums are accumulated in fp64 (a few thousand adds per lane, off the critical path)
// so the result does not depend on the summation order the reference's mean()/var() happened to use.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned key_of(float f) {          // monotone: a < b  <=>  key(a) < key(b)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float val_of(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {    // all threads get the total; red: 16 doubles
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}

__global__ void __launch_bounds__(1024) depth_standardize_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int lo, int hi,
                                                                 float eps) {
    __shared__ int hist[2][256];
    __shared__ unsigned sel_prefix[2];
    __shared__ int sel_rank[2], sel_less[2], sel_eq[2];
    __shared__ double red[16];
    const float* xs = x + (long long)blockIdx.x * n;
    float* ys = y + (long long)blockIdx.x * n;
    const int tid = threadIdx.x;
    if (tid < 2) { sel_prefix[tid] = 0u; sel_rank[tid] = tid == 0 ? lo : hi - 1; sel_less[tid] = 0; }
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 512; i += 1024) (&hist[0][0])[i] = 0;
        __syncthreads();
        const unsigned p0 = sel_prefix[0], p1 = sel_prefix[1];
        for (int i = tid; i < n; i += 1024) {
            const unsigned k = key_of(xs[i]);
            const unsigned top = shift == 24 ? 0u : (k >> (shift + 8));
            const int bin = (k >> shift) & 255;
            if (top == p0) atomicAdd(&hist[0][bin], 1);
            if (top == p1) atomicAdd(&hist[1][bin], 1);
        }
        __syncthreads();
        if (tid < 2) {                                             // serial 256-bin scan: negligible next to the histogram pass
            int r = sel_rank[tid], b = 0, cum = 0;
            for (; b < 255; ++b) { if (cum + hist[tid][b] > r) break; cum += hist[tid][b]; }
            sel_rank[tid] = r - cum;
            sel_less[tid] += cum;
            sel_eq[tid] = hist[tid][b];
            sel_prefix[tid] = (sel_prefix[tid] << 8) | (unsigned)b;
        }
        __syncthreads();
    }
    const unsigned k1 = sel_prefix[0], k2 = sel_prefix[1];
    const float v1 = val_of(k1), v2 = val_of(k2);
    // copies of the cut values inside [lo, hi)
    const int cnt = hi - lo;
    const int n1 = (k1 == k2) ? cnt : (sel_less[0] + sel_eq[0] - lo);
    const int n2 = (k1 == k2) ? 0 : (hi - sel_less[1]);
    double s = 0.0;
    for (int i = tid; i < n; i += 1024) {
        const float f = xs[i];
        const unsigned k = key_of(f);
        if (k > k1 && k < k2) s += (double)f;
    }
    s = block_sum_d(s, red) + (double)n1 * (double)v1 + (double)n2 * (double)v2;
    const double mean = s / (double)cnt;
    double q = 0.0;
    for (int i = tid; i < n; i += 1024) {
        const float f = xs[i];
        const unsigned k = key_of(f);
        if (k > k1 && k < k2) { const double d = (double)f - mean; q += d * d; }
    }
    q = block_sum_d(q, red) + (double)n1 * ((double)v1 - mean) * ((double)v1 - mean) + (double)n2 * ((double)v2 - mean) * ((double)v2 - mean);
    const float var = (float)(q / (double)(cnt - 1));            // unbiased, as Tensor.var
    const float mu = (float)mean, rs = 1.0f / sqrtf(var + eps);
    for (int i = tid; i < n; i += 1024) ys[i] = (xs[i] - mu) * rs;
}

}  // namespace

extern "C" int mmae_depth_standardize(const float* x, float* y, int B, int n, int lo, int hi, float eps, void* stream) {
    MMAE_REQUIRE(x && y && B > 0 && n > 1, "depth_standardize: bad argument");
    MMAE_REQUIRE(lo >= 0 && hi <= n && hi - lo >= 2, "depth_standardize: need 0 <= lo, lo + 2 <= hi <= n");
    hipLaunchKernelGGL(depth_standardize_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, y, n, lo, hi, eps);
    return mmae_check_launch("depth_standardize");
}
