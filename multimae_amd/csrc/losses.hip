// Masked reconstruction losses (criterion.py) and the flat-arena optimiser step.
//
// Loss definition reproduced exactly: for every sample,
// sum over masked pixels of the channel-mean error divided by the number of masked pixels;
// the loss is the mean of that over samples that have at least one masked token (nanmean).
#include <math.h>
#include <type_traits>
#include "common.h"

namespace {

constexpr int LSPLIT = 8;   // workgroups per sample

// block reduce of two floats over 256 threads; result valid in thread 0
__device__ __forceinline__ void block_sum2(float& a, float& b) {
    __shared__ float ra[4], rb[4];
    a = wave_sum(a); b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { ra[w] = a; rb[w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) { a = (ra[0] + ra[1]) + (ra[2] + ra[3]); b = (rb[0] + rb[1]) + (rb[2] + rb[3]); }
    __syncthreads();
}

// pixel losses.  grid (LSPLIT, B).  Each wave takes one masked patch at a time.
// stats[b][p] = (mean, rstd) of the target patch (norm_pix).  partial[b][split] = sum of errors.
__global__ void __launch_bounds__(256) pixel_loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                             const long long* __restrict__ mask, int kind, int norm_pix, int C, int H,
                                                             int W, int P, float* __restrict__ stats, float* __restrict__ partial) {
    const int b = blockIdx.y, nh = H / P, nw = W / P, np = nh * nw;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int npix = P * P, nval = npix * C;
    float acc = 0.f;
    for (int p = blockIdx.x * 4 + w; p < np; p += LSPLIT * 4) {
        if (mask[(long long)b * np + p] == 0) continue;
        const int py = p / nw, px = p % nw;
        float mu = 0.f, rs = 1.f;
        if (norm_pix) {
            float s = 0.f;
            for (int e = lane; e < nval; e += 64) {
                const int c = e / npix, ij = e % npix;
                s += target[(((long long)b * C + c) * H + py * P + ij / P) * W + px * P + ij % P];
            }
            mu = wave_sum(s) / (float)nval;
            float q = 0.f;
            for (int e = lane; e < nval; e += 64) {
                const int c = e / npix, ij = e % npix;
                const float d = target[(((long long)b * C + c) * H + py * P + ij / P) * W + px * P + ij % P] - mu;
                q += d * d;
            }
            const float var = wave_sum(q) / (float)(nval - 1);     // unbiased (criterion.py:92)
            rs = 1.0f / sqrtf(var + 1e-6f);
            if (lane == 0) { stats[((long long)b * np + p) * 2] = mu; stats[((long long)b * np + p) * 2 + 1] = rs; }
        }
        float s = 0.f;
        for (int e = lane; e < nval; e += 64) {
            const int c = e / npix, ij = e % npix;
            const long long a = (((long long)b * C + c) * H + py * P + ij / P) * W + px * P + ij % P;
            const float t = (target[a] - mu) * rs;
            const float d = pred[a] - t;
            s += kind == 0 ? d * d : fabsf(d);
        }
        s = wave_sum(s);                 // every lane now holds the patch total
        if (lane == 0) acc += s;
    }
    float dummy = 0.f;
    block_sum2(acc, dummy);
    if (threadIdx.x == 0) partial[(long long)b * LSPLIT + blockIdx.x] = acc / (float)C;
}

// per_sample[b] = (sum, count); loss = mean over samples with count > 0 of sum / count.
// aux[0] = number of such samples (used by backward).  Single workgroup.
__global__ void __launch_bounds__(256) loss_finalize_kernel(const float* __restrict__ partial, const long long* __restrict__ mask, int B,
                                                            int np, int pix_per_patch, float* __restrict__ per_sample,
                                                            float* __restrict__ loss) {
    float tot = 0.f, nvalid = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        float s = 0.f;
        for (int i = 0; i < LSPLIT; ++i) s += partial[(long long)b * LSPLIT + i];
        // one thread walks one sample's mask row: 16-byte loads, four independent ones in flight (a dependent 8-byte load per
        // patch made this single-workgroup kernel 41 us -- 196 serial L2 round trips -- four times per step)
        int cnt = 0;
        const long long* mrow = mask + (long long)b * np;
        int p = 0;
        if ((((uintptr_t)mrow) & 15) == 0) {
            typedef __attribute__((ext_vector_type(2))) long long ll2;
            const ll2* m2 = reinterpret_cast<const ll2*>(mrow);
            const int n2 = np >> 1;
            int q = 0;
            for (; q + 4 <= n2; q += 4) {
                const ll2 a0 = m2[q], a1 = m2[q + 1], a2 = m2[q + 2], a3 = m2[q + 3];
                cnt += (a0[0] != 0) + (a0[1] != 0) + (a1[0] != 0) + (a1[1] != 0) + (a2[0] != 0) + (a2[1] != 0) + (a3[0] != 0) + (a3[1] != 0);
            }
            for (; q < n2; ++q) { const ll2 a = m2[q]; cnt += (a[0] != 0) + (a[1] != 0); }
            p = n2 * 2;
        }
        for (; p < np; ++p) cnt += mrow[p] != 0;
        const float c = (float)cnt * (float)pix_per_patch;
        per_sample[b * 2] = s; per_sample[b * 2 + 1] = c;
        if (cnt > 0) { tot += s / c; nvalid += 1.f; }
    }
    block_sum2(tot, nvalid);
    if (threadIdx.x == 0) { loss[0] = nvalid > 0.f ? tot / nvalid : 0.f; loss[1] = nvalid; }
}

// d_pred.  thread per pixel-run element; grid over B*C*H*W.
__global__ void pixel_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, const long long* __restrict__ mask,
                                      int kind, int norm_pix, int C, int H, int W, int P, const float* __restrict__ stats,
                                      const float* __restrict__ per_sample, const float* __restrict__ loss, const float* __restrict__ upstream,
                                      float* __restrict__ d_pred, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W); long long r = i / W;
    const int y = (int)(r % H); r /= H;
    const long long b = r / C;
    const int nw = W / P, np = (H / P) * nw, p = (y / P) * nw + x / P;
    float g = 0.f;
    if (mask[b * np + p] != 0) {
        float mu = 0.f, rs = 1.f;
        if (norm_pix) { mu = stats[(b * np + p) * 2]; rs = stats[(b * np + p) * 2 + 1]; }
        const float d = pred[i] - (target[i] - mu) * rs;
        const float wgt = upstream[0] / (loss[1] * per_sample[b * 2 + 1] * (float)C);
        g = wgt * (kind == 0 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
    }
    d_pred[i] = g;
}

// cross entropy.  thread per pixel, channel planes strided by H*W (coalesced across pixels).
// label smoothing eps (F.cross_entropy(label_smoothing=eps), criterion.py:47): (1 - eps) * nll(target) + eps * mean_c(-log p_c)
__global__ void __launch_bounds__(256) ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                     const long long* __restrict__ mask, int C, int H, int W, int P, float eps,
                                                     float* __restrict__ lse, float* __restrict__ partial) {
    const int b = blockIdx.y, HW = H * W, nw = W / P, np = (H / P) * nw;
    float acc = 0.f;
    for (int px = blockIdx.x * 256 + threadIdx.x; px < HW; px += LSPLIT * 256) {
        const int y = px / W, x = px % W;
        if (mask[(long long)b * np + (y / P) * nw + x / P] == 0) continue;
        const float* l = logits + (long long)b * C * HW + px;
        // one pass over the C channel planes: running max + rescaled sum (the two-pass form read the logits twice)
        float mx = l[0], s = 1.f, sx = l[0];
        int c = 1;
        for (; c + 4 <= C; c += 4) {
            const float v0 = l[(long long)c * HW], v1 = l[(long long)(c + 1) * HW], v2 = l[(long long)(c + 2) * HW], v3 = l[(long long)(c + 3) * HW];
            const float m4 = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
            if (m4 > mx) { s *= expf(mx - m4); mx = m4; }
            s += (expf(v0 - mx) + expf(v1 - mx)) + (expf(v2 - mx) + expf(v3 - mx));
            sx += (v0 + v1) + (v2 + v3);
        }
        for (; c < C; ++c) {
            const float v = l[(long long)c * HW];
            if (v > mx) { s *= expf(mx - v); mx = v; }
            s += expf(v - mx);
            sx += v;
        }
        const float ls = mx + logf(s);
        lse[(long long)b * HW + px] = ls;
        const float nll = ls - l[target[(long long)b * HW + px] * HW];
        acc += eps == 0.f ? nll : (1.f - eps) * nll + eps * (ls - sx / (float)C);
    }
    float dummy = 0.f;
    block_sum2(acc, dummy);
    if (threadIdx.x == 0) partial[(long long)b * LSPLIT + blockIdx.x] = acc;
}

__global__ void ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target, const long long* __restrict__ mask,
                              int C, int H, int W, int P, float eps, const float* __restrict__ lse, const float* __restrict__ per_sample,
                              const float* __restrict__ loss, const float* __restrict__ upstream, float* __restrict__ d_logits,
                              long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;    // over B*C*H*W
    if (i >= total) return;
    const int HW = H * W;
    const int px = (int)(i % HW); long long r = i / HW;
    const int c = (int)(r % C); const long long b = r / C;
    const int y = px / W, x = px % W, nw = W / P, np = (H / P) * nw;
    float g = 0.f;
    if (mask[b * np + (y / P) * nw + x / P] != 0) {
        const float wgt = upstream[0] / (loss[1] * per_sample[b * 2 + 1]);
        const float sm = expf(logits[i] - lse[b * HW + px]);
        g = wgt * (sm - ((target[b * HW + px] == c ? 1.f - eps : 0.f) + eps / (float)C));
    }
    d_logits[i] = g;
}

// float4 flavours of the two backward kernels (W % 4 == 0 and P % 4 == 0: four consecutive pixels share a row and a patch,
// hence one mask / stats / weight lookup): the scalar ones spend their time in per-element div/mod chains (1.7 TB/s).
__global__ void __launch_bounds__(256) pixel_loss_bwd4_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                              const long long* __restrict__ mask, int kind, int norm_pix, int C, int H, int W, int P,
                                                              const float* __restrict__ stats, const float* __restrict__ per_sample,
                                                              const float* __restrict__ loss, const float* __restrict__ upstream,
                                                              float* __restrict__ d_pred, long long total4) {
    const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= total4) return;
    const long long i = i4 * 4;
    const int x = (int)(i % W); long long r = i / W;
    const int y = (int)(r % H); r /= H;
    const long long b = r / C;
    const int nw = W / P, np = (H / P) * nw, p = (y / P) * nw + x / P;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (mask[b * np + p] != 0) {
        float mu = 0.f, rs = 1.f;
        if (norm_pix) { mu = stats[(b * np + p) * 2]; rs = stats[(b * np + p) * 2 + 1]; }
        const f32x4 pr = ld4(pred + i), tg = ld4(target + i);
        const float wgt = upstream[0] / (loss[1] * per_sample[b * 2 + 1] * (float)C);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = pr[j] - (tg[j] - mu) * rs;
            g[j] = wgt * (kind == 0 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
        }
    }
    st4(d_pred + i, g);
}

__global__ void __launch_bounds__(256) ce_bwd4_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                      const long long* __restrict__ mask, int C, int H, int W, int P, float eps, const float* __restrict__ lse,
                                                      const float* __restrict__ per_sample, const float* __restrict__ loss,
                                                      const float* __restrict__ upstream, float* __restrict__ d_logits, long long total4) {
    const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;          // over B*C*H*W / 4
    if (i4 >= total4) return;
    const long long i = i4 * 4;
    const int HW = H * W;
    const int px = (int)(i % HW); long long r = i / HW;
    const int c = (int)(r % C); const long long b = r / C;
    const int y = px / W, x = px % W, nw = W / P, np = (H / P) * nw;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (mask[b * np + (y / P) * nw + x / P] != 0) {
        const float wgt = upstream[0] / (loss[1] * per_sample[b * 2 + 1]);
        const f32x4 lg = ld4(logits + i), ls = ld4(lse + b * HW + px);
        const long long* t = target + b * HW + px;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = wgt * (expf(lg[j] - ls[j]) - ((t[j] == c ? 1.f - eps : 0.f) + eps / (float)C));
    }
    st4(d_logits + i, g);
}

// ---- the same losses in the PATCH domain ----------------------------------------------------------------------------
// The output adapters produce their prediction as patch rows pat[b * np + p][(c, i, j)] (out_proj, output_adapters.py:274) and
// only rearrange them into an image for the API (:277-280).  These kernels take the loss straight from the rows and write the
// gradient straight back as rows in the adapter's activation dtype: no f32 image-domain gradient, no patchify pass
// (criterion.py does the identical arithmetic per pixel).
__global__ void __launch_bounds__(256) pixel_loss_pat_fwd_kernel(const float* __restrict__ pat, const float* __restrict__ target,
                                                                 const long long* __restrict__ mask, int kind, int norm_pix, int C, int H,
                                                                 int W, int P, float* __restrict__ stats, float* __restrict__ partial) {
    const int b = blockIdx.y, nh = H / P, nw = W / P, np = nh * nw;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int npix = P * P, nval = npix * C;
    float acc = 0.f;
    // fast path (round 4): 4 consecutive (c, i, j) elements per lane -- they share c and i and lie in one target row when P % 4 == 0 --
    // with every load of the patch issued before the first use (up to 1 024 values per patch: 4 x 16 B of prediction and of target
    // per lane), the target kept in registers for the norm_pix statistics AND the loss.  The scalar form below (one dependent 4-byte
    // load per element, three passes over the target under norm_pix) ran at 2 TB/s: 127 us for the rgb loss of a cfg3 step.
    const bool vec = (P & 3) == 0 && (W & 3) == 0 && (nval & 3) == 0 && nval <= 1024 && (((uintptr_t)pat | (uintptr_t)target) & 15) == 0;
    for (int p = blockIdx.x * 4 + w; vec && p < np; p += LSPLIT * 4) {
        if (mask[(long long)b * np + p] == 0) continue;
        const int py = p / nw, px = p % nw;
        const float* prow = pat + ((long long)b * np + p) * nval;
        f32x4 tg[4], pr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = lane * 4 + k * 256;
            if (e < nval) {
                const int c = e / npix, ij = e % npix;
                pr[k] = ld4(prow + e);
                tg[k] = ld4(target + (((long long)b * C + c) * H + py * P + ij / P) * W + px * P + ij % P);
            } else { pr[k] = f32x4{0.f, 0.f, 0.f, 0.f}; tg[k] = pr[k]; }
        }
        float mu = 0.f, rs = 1.f;
        if (norm_pix) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += (tg[k][0] + tg[k][1]) + (tg[k][2] + tg[k][3]);
            mu = wave_sum(s) / (float)nval;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (lane * 4 + k * 256 < nval) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float d = tg[k][j] - mu; q += d * d; }
                }
            }
            const float var = wave_sum(q) / (float)(nval - 1);     // unbiased (criterion.py:92)
            rs = 1.0f / sqrtf(var + 1e-6f);
            if (lane == 0) { stats[((long long)b * np + p) * 2] = mu; stats[((long long)b * np + p) * 2 + 1] = rs; }
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (lane * 4 + k * 256 < nval) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = pr[k][j] - (tg[k][j] - mu) * rs;
                    s += kind == 0 ? d * d : fabsf(d);
                }
            }
        }
        s = wave_sum(s);
        if (lane == 0) acc += s;
    }
    for (int p = blockIdx.x * 4 + w; !vec && p < np; p += LSPLIT * 4) {
        if (mask[(long long)b * np + p] == 0) continue;
        const int py = p / nw, px = p % nw;
        const float* prow = pat + ((long long)b * np + p) * nval;
        float mu = 0.f, rs = 1.f;
        if (norm_pix) {
            float s = 0.f;
            for (int e = lane; e < nval; e += 64) {
                const int c = e / npix, ij = e % npix;
                s += target[(((long long)b * C + c) * H + py * P + ij / P) * W + px * P + ij % P];
            }
            mu = wave_sum(s) / (float)nval;
            float q = 0.f;
            for (int e = lane; e < nval; e += 64) {
                const int c = e / npix, ij = e % npix;
                const float d = target[(((long long)b * C + c) * H + py * P + ij / P) * W + px * P + ij % P] - mu;
                q += d * d;
            }
            const float var = wave_sum(q) / (float)(nval - 1);     // unbiased (criterion.py:92)
            rs = 1.0f / sqrtf(var + 1e-6f);
            if (lane == 0) { stats[((long long)b * np + p) * 2] = mu; stats[((long long)b * np + p) * 2 + 1] = rs; }
        }
        float s = 0.f;
        for (int e = lane; e < nval; e += 64) {
            const int c = e / npix, ij = e % npix;
            const float t = (target[(((long long)b * C + c) * H + py * P + ij / P) * W + px * P + ij % P] - mu) * rs;
            const float d = prow[e] - t;
            s += kind == 0 ? d * d : fabsf(d);
        }
        s = wave_sum(s);
        if (lane == 0) acc += s;
    }
    float dummy = 0.f;
    block_sum2(acc, dummy);
    if (threadIdx.x == 0) partial[(long long)b * LSPLIT + blockIdx.x] = acc / (float)C;
}

// largest |gradient element| of the launch -> *amax (optional): what scales the fp16 gradient operands of an MMAE_F32F16 adapter
// (mmae_gemm_desc.a_amax).  Non-negative floats order like their bit patterns: one atomicMax per wave that beats the value it sees.
__device__ __forceinline__ void note_amax(float m, float* amax) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > *(volatile float*)amax) atomicMax((unsigned*)amax, __float_as_uint(m));
}

// one wave per patch row; 4 consecutive (c, i, j) elements per lane (P % 4 == 0: they share c, i and lie in one target row)
template <typename DT>
__global__ void __launch_bounds__(256) pixel_loss_pat_bwd_kernel(const float* __restrict__ pat, const float* __restrict__ target,
                                                                 const long long* __restrict__ mask, int kind, int norm_pix, int C, int H,
                                                                 int W, int P, const float* __restrict__ stats, const float* __restrict__ per_sample,
                                                                 const float* __restrict__ loss, const float* __restrict__ upstream,
                                                                 DT* __restrict__ d_pat, long long ld, long long n_rows, float* __restrict__ amax) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int nw = W / P, np = (H / P) * nw, npix = P * P, nval = npix * C;
    const long long b = row / np;
    const int p = (int)(row % np), py = p / nw, px = p % nw;
    const bool on = mask[row] != 0;
    float gmax = 0.f;
    float mu = 0.f, rs = 1.f, wgt = 0.f;
    if (on) {
        if (norm_pix) { mu = stats[row * 2]; rs = stats[row * 2 + 1]; }
        wgt = upstream[0] / (loss[1] * per_sample[b * 2 + 1] * (float)C);
    }
    const float* prow = pat + row * nval;
    DT* drow = d_pat + row * ld;
    if ((P & 3) == 0 && (W & 3) == 0 && (ld & 3) == 0) {
        for (int e = lane * 4; e < (int)ld; e += 256) {
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            if (on && e < nval) {
                const int c = e / npix, ij = e % npix;
                const f32x4 pr = ld4(prow + e);
                const f32x4 tg = ld4(target + (((long long)b * C + c) * H + py * P + ij / P) * W + px * P + ij % P);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = pr[j] - (tg[j] - mu) * rs;
                    g[j] = wgt * (kind == 0 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
                    gmax = fmaxf(gmax, fabsf(g[j]));
                }
            }
            st4(drow + e, g);
        }
    } else {
        for (int e = lane; e < (int)ld; e += 64) {
            float g = 0.f;
            if (on && e < nval) {
                const int c = e / npix, ij = e % npix;
                const float d = prow[e] - (target[(((long long)b * C + c) * H + py * P + ij / P) * W + px * P + ij % P] - mu) * rs;
                g = wgt * (kind == 0 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
                gmax = fmaxf(gmax, fabsf(g));
            }
            ActT<DT>::st(drow + e, g);
        }
    }
    if (amax) note_amax(gmax, amax);
}

// cross entropy on patch rows; npix = P * P <= 64, a power of two: lane = (class slot, pixel), 64 / npix class slots per pixel,
// element index of iteration k = k * 64 + lane (perfectly coalesced).  lse_pat f32 [B * np][npix].
__global__ void __launch_bounds__(256) ce_pat_fwd_kernel(const float* __restrict__ pat, const long long* __restrict__ target,
                                                         const long long* __restrict__ mask, int C, int H, int W, int P, float eps,
                                                         float* __restrict__ lse_pat, float* __restrict__ partial) {
    const int b = blockIdx.y, nw = W / P, np = (H / P) * nw, npix = P * P, nval = npix * C;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pix = lane % npix, slot = lane / npix, nslot = 64 / npix;
    float acc = 0.f;
    for (int p = blockIdx.x * 4 + w; p < np; p += LSPLIT * 4) {
        if (mask[(long long)b * np + p] == 0) continue;
        const int py = p / nw, px = p % nw;
        const long long row = (long long)b * np + p;
        const float* prow = pat + row * nval;
        const long long t = target[((long long)b * H + py * P + pix / P) * W + px * P + pix % P];
        float mx = -3.0e38f, s = 0.f, lt = 0.f, sx = 0.f;
        // online softmax over groups of eight values per lane: eight independent loads in flight, one rescale per group (rounds 1-3 loaded one
        // value per iteration: 2.4 TB/s; round 4's first form held the whole row -- 40 values per lane -- in registers and came out at 223 VGPRs,
        // two waves per SIMD: 2.9 TB/s).  ~30 VGPRs: the workgroups of a whole sample row are resident at once.
        for (int e0 = lane; e0 < nval; e0 += 512) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int e = e0 + 64 * k; v[k] = e < nval ? prow[e] : -3.0e38f; }
            float gm = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
            if (gm > mx) { s *= expf(mx - gm); mx = gm; }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + 64 * k;
                if (e < nval) {
                    s += expf(v[k] - mx);
                    sx += v[k];
                    if ((long long)(slot + (e >> 6) * nslot) == t) lt = v[k];
                }
            }
        }
        // combine the class slots of a pixel (lanes pix, pix + npix, ...)
        for (int o = npix; o < 64; o <<= 1) {
            const float om = __shfl_xor(mx, o, 64), os = __shfl_xor(s, o, 64);
            lt += __shfl_xor(lt, o, 64);
            sx += __shfl_xor(sx, o, 64);
            const float nm = fmaxf(mx, om);
            s = s * expf(mx - nm) + os * expf(om - nm);
            mx = nm;
        }
        const float ls = mx + logf(s);
        if (slot == 0) { lse_pat[row * npix + pix] = ls; }
        float contrib = slot == 0 ? (eps == 0.f ? ls - lt : (1.f - eps) * (ls - lt) + eps * (ls - sx / (float)C)) : 0.f;
        contrib = wave_sum(contrib);
        if (lane == 0) acc += contrib;
    }
    float dummy = 0.f;
    block_sum2(acc, dummy);
    if (threadIdx.x == 0) partial[(long long)b * LSPLIT + blockIdx.x] = acc;
}

template <typename DT>
__global__ void __launch_bounds__(256) ce_pat_bwd_kernel(const float* __restrict__ pat, const long long* __restrict__ target,
                                                         const long long* __restrict__ mask, int C, int H, int W, int P, float eps,
                                                         const float* __restrict__ lse_pat, const float* __restrict__ per_sample,
                                                         const float* __restrict__ loss, const float* __restrict__ upstream,
                                                         DT* __restrict__ d_pat, long long ld, long long n_rows, float* __restrict__ amax) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nw = W / P, np = (H / P) * nw, npix = P * P, nval = npix * C;
    const long long b = row / np;
    const int p = (int)(row % np), py = p / nw, px = p % nw;
    const int pix = lane % npix, slot = lane / npix, nslot = 64 / npix;
    // fp16 storage (MMAE_F16): |gradient| <= max_b |weight_b| -- the bound every workgroup derives for itself from the B per-sample
    // counts; it fixes the scale S the adapter's whole backward is stored in (mmae.h) and is what `amax` receives
    float gs = 1.0f;
    if constexpr (std::is_same<DT, h16_t>::value) {
        __shared__ float wred[4];
        const int B = (int)(n_rows / np);
        float wm = 0.f;
        for (int i = threadIdx.x; i < B; i += 256) {
            const float cnt = per_sample[i * 2 + 1];
            if (cnt > 0.f) wm = fmaxf(wm, fabsf(upstream[0] / (loss[1] * cnt)));
        }
        wm = wave_max(wm);
        if (lane == 0) wred[threadIdx.x >> 6] = wm;
        __syncthreads();
        wm = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
        if (blockIdx.x == 0 && threadIdx.x == 0 && amax) *amax = wm;
        gs = h16_grad_scale(&wm);
    }
    if (row >= n_rows) return;
    const bool on = mask[row] != 0;
    DT* drow = d_pat + row * ld;
    if (!on) {
        for (int e = lane; e < (int)ld; e += 64) ActT<DT>::st(drow + e, 0.f);
        return;
    }
    const float wgt = gs * upstream[0] / (loss[1] * per_sample[b * 2 + 1]);
    const float ls = lse_pat[row * npix + pix];
    const long long t = target[((long long)b * H + py * P + pix / P) * W + px * P + pix % P];
    const float* prow = pat + row * nval;
    float gmax = 0.f;
    // four independent loads in flight per lane (one per iteration left the kernel at 22 VGPRs and latency-bound: 3.5 TB/s)
    for (int e0 = lane; e0 < (int)ld; e0 += 256) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int e = e0 + 64 * k; v[k] = e < nval ? prow[e] : 0.f; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = e0 + 64 * k;
            if (e < (int)ld) {
                const int c = slot + (e >> 6) * nslot;
                float g = 0.f;
                if (e < nval) g = wgt * (expf(v[k] - ls) - (((long long)c == t ? 1.f - eps : 0.f) + eps / (float)C));
                gmax = fmaxf(gmax, fabsf(g));
                ActT<DT>::st(drow + e, g);
            }
        }
    }
    if (!std::is_same<DT, h16_t>::value && amax) note_amax(gmax, amax);
}

// ---- optimiser ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sumsq_stage1(const float* __restrict__ x, long long n, float* __restrict__ ws) {
    float s = 0.f, d = 0.f;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
        if (i + 4 <= n) { const f32x4 v = ld4(x + i); s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]); }
        else for (long long j = i; j < n; ++j) s += x[j] * x[j];
    }
    block_sum2(s, d);
    if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) sumsq_stage2(const float* __restrict__ ws, int nb, float* __restrict__ out) {
    float s = 0.f, d = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += ws[i];
    block_sum2(s, d);
    if (threadIdx.x == 0) out[0] = s;
}

// Every decision of the optimiser step on the device (mmae_opt_step): norm, clip / skip, non-finite guards, Adam's step
// counter and bias corrections.  One thread.
__global__ void opt_finalize_kernel(float* __restrict__ state, int* __restrict__ istate, float lr, float wd, const float* __restrict__ lrwd,
                                    float b1, float b2, float clip, float skip_at, float prescale, const float* __restrict__ loss,
                                    const float* __restrict__ found_inf, const float* __restrict__ grad_scale) {
    if (grad_scale) prescale /= grad_scale[0];             // GradScaler's loss scale, still on the gradients (mmae.h)
    const float norm = sqrtf(state[0]) * prescale;
    const bool loss_bad = loss && !isfinite(loss[0]);
    const bool amp_bad = found_inf && found_inf[0] > 0.f;
    // clip and skip are exclusive, clip first: utils/native_scaler.py:24-32 is `if clip_grad ... elif skip_grad`
    const bool skip = !isfinite(norm) || (clip <= 0.f && skip_at > 0.f && norm >= skip_at) || loss_bad || amp_bad;
    float scale = prescale;
    if (clip > 0.f) { const float cc = clip / (norm + 1e-6f); scale *= cc < 1.f ? cc : 1.f; }
    int t = istate[1];
    if (!skip) t += 1;
    istate[0] = skip ? 1 : 0;
    istate[1] = t;
    if (loss_bad) istate[2] += 1;
    if (skip) istate[3] += 1;
    if (amp_bad) istate[4] += 1;
    if (!isfinite(norm)) istate[5] += 1;
    state[1] = norm; state[2] = scale;
    state[3] = lrwd ? lrwd[0] : lr;
    state[4] = lrwd ? lrwd[1] : wd;
    const int te = t > 0 ? t : 1;
    state[5] = (float)(1.0 - pow((double)b1, (double)te));
    state[6] = (float)sqrt(1.0 - pow((double)b2, (double)te));
}

template <typename ST>
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, const float* __restrict__ grad_scale,
                                                    const int* __restrict__ skip, ST* __restrict__ shadow,
                                                    const float* __restrict__ hyper) {
    if (skip && *skip) return;
    if (hyper) { lr = hyper[0]; wd = hyper[1]; bc1 = hyper[2]; bc2_sqrt = hyper[3]; }   // step-dependent scalars from HBM (hipGraph replay)
    const float gs = grad_scale ? *grad_scale : 1.f;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            f32x4 pv = ld4(p + i), mv = ld4(m + i), vv = ld4(v + i);
            const f32x4 gv = ld4(g + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gj = gv[j] * gs;
                pv[j] *= (1.f - lr * wd);
                mv[j] = b1 * mv[j] + (1.f - b1) * gj;
                vv[j] = b2 * vv[j] + (1.f - b2) * gj * gj;
                const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
                pv[j] -= (lr / bc1) * (mv[j] / denom);
            }
            st4(p + i, pv); st4(m + i, mv); st4(v + i, vv);
            if (shadow) st4(shadow + i, pv);
        } else {
            for (long long j = i; j < n; ++j) {
                const float gj = g[j] * gs;
                float pj = p[j] * (1.f - lr * wd);
                const float mj = b1 * m[j] + (1.f - b1) * gj, vj = b2 * v[j] + (1.f - b2) * gj * gj;
                pj -= (lr / bc1) * (mj / (sqrtf(vj) / bc2_sqrt + eps));
                p[j] = pj; m[j] = mj; v[j] = vj;
                if (shadow) ActT<ST>::st(shadow + j, pj);
            }
        }
    }
}

}  // namespace

extern "C" {

int mmae_masked_pixel_loss_fwd(const float* pred, const float* target, const int64_t* mask, int kind, int norm_pix, int B, int C,
                               int H, int W, int patch, float* stats, float* partial, float* per_sample, float* loss, void* stream) {
    MMAE_REQUIRE(pred && target && mask && partial && per_sample && loss, "pixel_loss_fwd: null pointer");
    MMAE_REQUIRE(!norm_pix || stats, "pixel_loss_fwd: norm_pix needs stats");
    MMAE_REQUIRE(patch > 0 && H % patch == 0 && W % patch == 0 && (kind == 0 || kind == 1), "pixel_loss_fwd: bad geometry");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pixel_loss_fwd_kernel, dim3(LSPLIT, B), dim3(256), 0, st, pred, target, (const long long*)mask, kind, norm_pix, C, H,
                       W, patch, stats, partial);
    int rc = mmae_check_launch("pixel_loss_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, partial, (const long long*)mask, B, (H / patch) * (W / patch),
                       patch * patch, per_sample, loss);
    return mmae_check_launch("loss_finalize");
}

int mmae_masked_pixel_loss_bwd(const float* pred, const float* target, const int64_t* mask, int kind, int norm_pix, int B, int C,
                               int H, int W, int patch, const float* stats, const float* per_sample, const float* loss,
                               const float* upstream, float* d_pred, void* stream) {
    MMAE_REQUIRE(pred && target && mask && per_sample && loss && upstream && d_pred, "pixel_loss_bwd: null pointer");
    const long long total = (long long)B * C * H * W;
    if (W % 4 == 0 && patch % 4 == 0 && ((uintptr_t)pred % 16 == 0) && ((uintptr_t)target % 16 == 0) && ((uintptr_t)d_pred % 16 == 0))
        hipLaunchKernelGGL(pixel_loss_bwd4_kernel, dim3((unsigned)cdiv64(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, pred, target,
                           (const long long*)mask, kind, norm_pix, C, H, W, patch, stats, per_sample, loss, upstream, d_pred, total / 4);
    else
    hipLaunchKernelGGL(pixel_loss_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, pred, target,
                       (const long long*)mask, kind, norm_pix, C, H, W, patch, stats, per_sample, loss, upstream, d_pred, total);
    return mmae_check_launch("pixel_loss_bwd");
}

int mmae_masked_ce_fwd(const float* logits, const int64_t* target, const int64_t* mask, int B, int C, int H, int W, int patch,
                       float label_smoothing, float* lse, float* partial, float* per_sample, float* loss, void* stream) {
    MMAE_REQUIRE(logits && target && mask && lse && partial && per_sample && loss, "ce_fwd: null pointer");
    MMAE_REQUIRE(patch > 0 && H % patch == 0 && W % patch == 0, "ce_fwd: bad geometry");
    hipStream_t st = (hipStream_t)stream;
    MMAE_REQUIRE(label_smoothing >= 0.f && label_smoothing <= 1.f, "ce_fwd: label_smoothing outside [0, 1]");
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(LSPLIT, B), dim3(256), 0, st, logits, (const long long*)target, (const long long*)mask, C, H, W,
                       patch, label_smoothing, lse, partial);
    int rc = mmae_check_launch("ce_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, partial, (const long long*)mask, B, (H / patch) * (W / patch),
                       patch * patch, per_sample, loss);
    return mmae_check_launch("loss_finalize");
}

int mmae_masked_ce_bwd(const float* logits, const int64_t* target, const int64_t* mask, int B, int C, int H, int W, int patch,
                       float label_smoothing, const float* lse, const float* per_sample, const float* loss, const float* upstream,
                       float* d_logits, void* stream) {
    MMAE_REQUIRE(logits && target && mask && lse && per_sample && loss && upstream && d_logits, "ce_bwd: null pointer");
    const long long total = (long long)B * C * H * W;
    if (W % 4 == 0 && patch % 4 == 0 && ((uintptr_t)logits % 16 == 0) && ((uintptr_t)lse % 16 == 0) && ((uintptr_t)d_logits % 16 == 0))
        hipLaunchKernelGGL(ce_bwd4_kernel, dim3((unsigned)cdiv64(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, logits,
                           (const long long*)target, (const long long*)mask, C, H, W, patch, label_smoothing, lse, per_sample, loss, upstream, d_logits, total / 4);
    else
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, logits, (const long long*)target,
                       (const long long*)mask, C, H, W, patch, label_smoothing, lse, per_sample, loss, upstream, d_logits, total);
    return mmae_check_launch("ce_bwd");
}

int mmae_masked_pixel_loss_pat_fwd(const float* pat, const float* target, const int64_t* mask, int kind, int norm_pix, int B, int C, int H,
                                   int W, int patch, float* stats, float* partial, float* per_sample, float* loss, void* stream) {
    MMAE_REQUIRE(pat && target && mask && partial && per_sample && loss, "pixel_loss_pat_fwd: null pointer");
    MMAE_REQUIRE(!norm_pix || stats, "pixel_loss_pat_fwd: norm_pix needs stats");
    MMAE_REQUIRE(patch > 0 && H % patch == 0 && W % patch == 0 && (kind == 0 || kind == 1), "pixel_loss_pat_fwd: bad geometry");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pixel_loss_pat_fwd_kernel, dim3(LSPLIT, B), dim3(256), 0, st, pat, target, (const long long*)mask, kind, norm_pix, C, H, W,
                       patch, stats, partial);
    int rc = mmae_check_launch("pixel_loss_pat_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, partial, (const long long*)mask, B, (H / patch) * (W / patch),
                       patch * patch, per_sample, loss);
    return mmae_check_launch("loss_finalize");
}

int mmae_masked_pixel_loss_pat_bwd(const float* pat, const float* target, const int64_t* mask, int kind, int norm_pix, int B, int C, int H,
                                   int W, int patch, const float* stats, const float* per_sample, const float* loss, const float* upstream,
                                   void* d_pat, int d_pat_dtype, int64_t ld_pat, float* amax, void* stream) {
    MMAE_REQUIRE(pat && target && mask && per_sample && loss && upstream && d_pat, "pixel_loss_pat_bwd: null pointer");
    MMAE_REQUIRE(patch > 0 && H % patch == 0 && W % patch == 0 && ld_pat >= (int64_t)C * patch * patch, "pixel_loss_pat_bwd: bad geometry");
    MMAE_REQUIRE(!norm_pix || stats, "pixel_loss_pat_bwd: norm_pix needs stats");
    MMAE_REQUIRE(((uintptr_t)pat % 16 == 0) && ((uintptr_t)target % 16 == 0) && ((uintptr_t)d_pat % 16 == 0), "pixel_loss_pat_bwd: unaligned");
    const long long rows = (long long)B * (H / patch) * (W / patch);
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (d_pat_dtype == MMAE_BF16)
        hipLaunchKernelGGL((pixel_loss_pat_bwd_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, pat, target, (const long long*)mask, kind,
                           norm_pix, C, H, W, patch, stats, per_sample, loss, upstream, (uint16_t*)d_pat, (long long)ld_pat, rows, amax);
    else
        hipLaunchKernelGGL((pixel_loss_pat_bwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, pat, target, (const long long*)mask, kind,
                           norm_pix, C, H, W, patch, stats, per_sample, loss, upstream, (float*)d_pat, (long long)ld_pat, rows, amax);
    return mmae_check_launch("pixel_loss_pat_bwd");
}

int mmae_masked_ce_pat_fwd(const float* pat, const int64_t* target, const int64_t* mask, int B, int C, int H, int W, int patch,
                           float label_smoothing, float* lse_pat, float* partial, float* per_sample, float* loss, void* stream) {
    MMAE_REQUIRE(pat && target && mask && lse_pat && partial && per_sample && loss, "ce_pat_fwd: null pointer");
    MMAE_REQUIRE(patch > 0 && H % patch == 0 && W % patch == 0, "ce_pat_fwd: bad geometry");
    const int npix = patch * patch;
    if (npix > 64 || (npix & (npix - 1))) { mmae_set_error("ce_pat_fwd: patch_size^2 must be a power of two <= 64"); return MMAE_ESUPPORT; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_pat_fwd_kernel, dim3(LSPLIT, B), dim3(256), 0, st, pat, (const long long*)target, (const long long*)mask, C, H, W, patch,
                       label_smoothing, lse_pat, partial);
    int rc = mmae_check_launch("ce_pat_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, partial, (const long long*)mask, B, (H / patch) * (W / patch),
                       patch * patch, per_sample, loss);
    return mmae_check_launch("loss_finalize");
}

int mmae_masked_ce_pat_bwd(const float* pat, const int64_t* target, const int64_t* mask, int B, int C, int H, int W, int patch,
                           float label_smoothing, const float* lse_pat, const float* per_sample, const float* loss, const float* upstream,
                           void* d_pat, int d_pat_dtype, int64_t ld_pat, float* amax, void* stream) {
    MMAE_REQUIRE(pat && target && mask && lse_pat && per_sample && loss && upstream && d_pat, "ce_pat_bwd: null pointer");
    MMAE_REQUIRE(patch > 0 && H % patch == 0 && W % patch == 0 && ld_pat >= (int64_t)C * patch * patch, "ce_pat_bwd: bad geometry");
    const int npix = patch * patch;
    if (npix > 64 || (npix & (npix - 1))) { mmae_set_error("ce_pat_bwd: patch_size^2 must be a power of two <= 64"); return MMAE_ESUPPORT; }
    const long long rows = (long long)B * (H / patch) * (W / patch);
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (d_pat_dtype == MMAE_F16) {
        MMAE_REQUIRE(amax, "ce_pat_bwd: fp16 patch rows need the amax scalar (it carries their scale)");
        hipLaunchKernelGGL((ce_pat_bwd_kernel<h16_t>), grid, dim3(256), 0, (hipStream_t)stream, pat, (const long long*)target, (const long long*)mask,
                           C, H, W, patch, label_smoothing, lse_pat, per_sample, loss, upstream, (h16_t*)d_pat, (long long)ld_pat, rows, amax);
    } else if (d_pat_dtype == MMAE_BF16)
        hipLaunchKernelGGL((ce_pat_bwd_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, pat, (const long long*)target, (const long long*)mask,
                           C, H, W, patch, label_smoothing, lse_pat, per_sample, loss, upstream, (uint16_t*)d_pat, (long long)ld_pat, rows, amax);
    else
        hipLaunchKernelGGL((ce_pat_bwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, pat, (const long long*)target, (const long long*)mask,
                           C, H, W, patch, label_smoothing, lse_pat, per_sample, loss, upstream, (float*)d_pat, (long long)ld_pat, rows, amax);
    return mmae_check_launch("ce_pat_bwd");
}

int mmae_loss_split(void) { return LSPLIT; }

int mmae_sumsq(const float* x, int64_t n, float* out, float* ws, void* stream) {
    MMAE_REQUIRE(x && out && ws && n > 0, "sumsq: bad argument");
    MMAE_REQUIRE((uintptr_t)x % 16 == 0, "sumsq: unaligned");
    long long nb = cdiv64(n, 1024);
    if (nb > 1024) nb = 1024;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_stage1, dim3((unsigned)nb), dim3(256), 0, st, x, (long long)n, ws);
    int rc = mmae_check_launch("sumsq_stage1");
    if (rc) return rc;
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, st, ws, (int)nb, out);
    return mmae_check_launch("sumsq_stage2");
}

int mmae_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int step, const float* grad_scale_dev, const int32_t* skip_flag, void* shadow, int shadow_dtype,
               void* stream) {
    MMAE_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw: bad argument");
    MMAE_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0), "adamw: unaligned");
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    long long nb = cdiv64(n, 1024);
    if (nb > 8192) nb = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (shadow && shadow_dtype == MMAE_BF16)
        hipLaunchKernelGGL((adamw_kernel<uint16_t>), dim3((unsigned)nb), dim3(256), 0, st, p, g, m, v, (long long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale_dev, (const int*)skip_flag, (uint16_t*)shadow, (const float*)nullptr);
    else
        hipLaunchKernelGGL((adamw_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st, p, g, m, v, (long long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale_dev, (const int*)skip_flag, (float*)shadow, (const float*)nullptr);
    return mmae_check_launch("adamw");
}

int mmae_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev, float beta1, float beta2, float eps,
                   const float* grad_scale_dev, const int32_t* skip_flag, void* shadow, int shadow_dtype, void* stream) {
    MMAE_REQUIRE(p && g && m && v && n > 0 && hyper_dev, "adamw_dev: bad argument");
    MMAE_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0), "adamw_dev: unaligned");
    long long nb = cdiv64(n, 1024);
    if (nb > 8192) nb = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (shadow && shadow_dtype == MMAE_BF16)
        hipLaunchKernelGGL((adamw_kernel<uint16_t>), dim3((unsigned)nb), dim3(256), 0, st, p, g, m, v, (long long)n, 0.f, beta1, beta2, eps, 0.f, 1.f, 1.f, grad_scale_dev, (const int*)skip_flag, (uint16_t*)shadow, hyper_dev);
    else
        hipLaunchKernelGGL((adamw_kernel<float>), dim3((unsigned)nb), dim3(256), 0, st, p, g, m, v, (long long)n, 0.f, beta1, beta2, eps, 0.f, 1.f, 1.f, grad_scale_dev, (const int*)skip_flag, (float*)shadow, hyper_dev);
    return mmae_check_launch("adamw_dev");
}

int mmae_opt_step(const mmae_opt_desc* d, void* stream) {
    MMAE_REQUIRE(d && d->p && d->g && d->m && d->v && d->n > 0 && d->state && d->istate && d->ws, "opt_step: bad argument");
    MMAE_REQUIRE(d->grad_prescale > 0.f, "opt_step: grad_prescale must be positive (1 for a single process)");
    int rc = mmae_sumsq(d->g, d->n, d->state, d->ws, stream);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(opt_finalize_kernel, dim3(1), dim3(1), 0, st, d->state, d->istate, d->lr, d->weight_decay, d->lrwd_dev, d->beta1, d->beta2,
                       d->clip_grad, d->skip_grad, d->grad_prescale, d->loss_dev, d->found_inf_dev, d->grad_scale_dev);
    if ((rc = mmae_check_launch("opt_finalize"))) return rc;
    return mmae_adamw_dev(d->p, d->g, d->m, d->v, d->n, d->state + 3, d->beta1, d->beta2, d->eps, d->state + 2, d->istate, d->shadow,
                          d->shadow_dtype, stream);
}

}  // extern "C"
