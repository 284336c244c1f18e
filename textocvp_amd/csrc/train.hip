// Backward / optimiser kernels of the predictor training step (SURVEY.md section 8f rank 2,
// reference 04_train_predictor.py:57-108, lib/loss.py:150-191, lib/setup_model.py:285-332).
// Row-wise HBM-bound kernels: one wave per row, 16-byte accesses, __shfl_xor reductions; column
// reductions (bias / LayerNorm parameter gradients, loss and norm partials) are deterministic
// two-stage sums (per-chunk partials in a caller workspace, then one more pass).
#include "common.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;

// y[r, :] = softmax(scale * x[r, :]) over the first `len` columns (len = key_len[r / rows_per_batch] or
// cols); masked columns get probability 0.  One wave per row, cols <= 4096.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int rows, int cols, float scale,
                                                           const int32_t* __restrict__ key_len,
                                                           int rows_per_batch) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int len = cols;
    if (key_len) {
        len = key_len[row / rows_per_batch];
        len = len < 1 ? 1 : (len > cols ? cols : len);
    }
    const float* xr = x + (size_t)row * cols;
    float* yr = y + (size_t)row * cols;
    float m = NEG_BIG;
    for (int c = lane; c < len; c += 64) m = fmaxf(m, xr[c] * scale);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.f;
    for (int c = lane; c < len; c += 64) s += expf(xr[c] * scale - m);
    s = wave_sum64(s);
    const float inv = 1.0f / s;
    for (int c = lane; c < cols; c += 64) yr[c] = c < len ? expf(xr[c] * scale - m) * inv : 0.f;
}

// ds = scale * p * (dp - sum_j p_j dp_j)   (softmax backward w.r.t. the pre-scale scores)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                          float* __restrict__ ds, int rows, int cols,
                                                          float scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = p + (size_t)row * cols;
    const float* dr = dp + (size_t)row * cols;
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) dot += pr[c] * dr[c];
    dot = wave_sum64(dot);
    float* o = ds + (size_t)row * cols;
    for (int c = lane; c < cols; c += 64) o[c] = scale * pr[c] * (dr[c] - dot);
}

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float v) {
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
    return cdf + v * pdf;
}

// act 1 = ReLU, 2 = exact GELU
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                      int act) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    y[i] = act == 1 ? fmaxf(v, 0.f) : gelu_f(v);
}
// dx = dy * act'(x); for ReLU `x` may be the activation OUTPUT (same sign test)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                      float* __restrict__ dx, long n, int act) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    dx[i] = act == 1 ? (v > 0.f ? dy[i] : 0.f) : dy[i] * gelu_grad(v);
}

// nn.Dropout with a caller-supplied uniform sample r in [0,1): y = r >= p ? x / (1 - p) : 0.  The same r
// applied to a gradient is the backward pass.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                      float* __restrict__ y, long n, float p, float inv_keep) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    y[i] = r[i] >= p ? x[i] * inv_keep : 0.f;
}

// y = a * x + b * y
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                    float a, float b) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    y[i] = a * x[i] + (b == 0.f ? 0.f : b * y[i]);
}

// partial[chunk, c] = sum over the rows of the chunk of x[r, c]; grid (ceil(cols/256), chunks)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                             int rows, int cols, int ld, int rows_per_chunk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += x[(size_t)r * ld + c];
    partial[(size_t)blockIdx.y * cols + c] = s;
}

// LayerNorm backward, one wave per row (D <= 1024, D % 4 == 0):
//   xhat = (x - mean) * rstd;  g = dy * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
// and per-wave partial sums of dgamma = sum dy * xhat, dbeta = sum dy over the rows this wave visits.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy, float* __restrict__ dx,
                                                            float* __restrict__ pgamma, float* __restrict__ pbeta,
                                                            int rows, int D, float eps, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    f32x4 ag[4], ab[4], gm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ag[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = (lane + 64 * i) * 4;
        gm[i] = c < D ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row = wid; row < rows; row += nw) {
        f32x4 v[4], d[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (lane + 64 * i) * 4;
            v[i] = c < D ? *reinterpret_cast<const f32x4*>(x + (size_t)row * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            d[i] = c < D ? *reinterpret_cast<const f32x4*>(dy + (size_t)row * D + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
        const float mean = wave_sum64(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < D)
#pragma unroll
                for (int u = 0; u < 4; ++u) q += (v[i][u] - mean) * (v[i][u] - mean);
        }
        const float rstd = 1.0f / sqrtf(wave_sum64(q) / (float)D + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < D)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float xh = (v[i][u] - mean) * rstd;
                    const float g = d[i][u] * gm[i][u];
                    sg += g;
                    sgx += g * xh;
                    ag[i][u] += d[i][u] * xh;
                    ab[i][u] += d[i][u];
                    v[i][u] = xh;                       // keep xhat for the dx pass
                }
        }
        const float mg = wave_sum64(sg) / (float)D, mgx = wave_sum64(sgx) / (float)D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < D) {
                f32x4 o;
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = rstd * (d[i][u] * gm[i][u] - mg - v[i][u] * mgx);
                f32x4* dst = reinterpret_cast<f32x4*>(dx + (size_t)row * D + c);
                *dst = (accumulate & 2) ? *dst + o : o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            f32x4* pg = reinterpret_cast<f32x4*>(pgamma + (size_t)wid * D + c);
            f32x4* pb = reinterpret_cast<f32x4*>(pbeta + (size_t)wid * D + c);
            *pg = (accumulate & 1) ? *pg + ag[i] : ag[i];
            *pb = (accumulate & 1) ? *pb + ab[i] : ab[i];
        }
    }
}

// dW[ids[i], :] += dy[i, :]  (token embedding gradient; ids < 0 skipped)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dy,
                                                            float* __restrict__ dW, int n, int D) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)n * D) return;
    const int r = (int)(i / D), c = (int)(i % D);
    const long id = ids[r];
    if (id >= 0) atomicAdd(dW + id * D + c, dy[i]);
}

// MSE (mean over all n elements): partial[b] = sum over block b of (p - t)^2;  dp = gscale * (p - t)
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ p, const float* __restrict__ tg,
                                                  float* __restrict__ partial, float* __restrict__ dp, long n,
                                                  float gscale) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = p[i] - tg[i];
        s += d * d;
        if (dp) dp[i] = gscale * d;
    }
    s = wave_sum64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// partial[b] = sum over block b of x^2
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ x, float* __restrict__ partial, long n) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i] * x[i];
    s = wave_sum64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// torch.optim.Adam (no weight decay, no amsgrad).  The step-dependent scalars live in DEVICE memory so
// that a captured HIP graph of the step can be replayed with new values:
//   hyper = {lr, beta1, beta2, eps, 1 - beta1^t, 1 - beta2^t},  *gscale = gradient clipping factor.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   const float* __restrict__ hyper,
                                                   const float* __restrict__ gscale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[4], bc2 = hyper[5];
    const float gi = g[i] * (gscale ? gscale[0] : 1.f);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

// clip_grad_norm_: out[0] = min(1, max_norm / (sqrt(sumsq) + 1e-6)), out[1] = sqrt(sumsq)
__global__ void clip_scale_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ out) {
    const float norm = sqrtf(sumsq[0]);
    out[0] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
    out[1] = norm;
}

// ---- frozen SAVi decoder, backward w.r.t. the slots (image loss of the predictor training step) ----
// Tail (SAVi.py:251-255): img = sum_k rgb_k * m_k, m = softmax_k(alpha).  Given dimg (F,3,H,W):
//   d rgb_k = dimg * m_k;  d m_k = <dimg, rgb_k>;  d alpha_k = m_k * (d m_k - sum_j m_j d m_j)
// written as dy (F*K, H, W, 4) NHWC = gradient of the 3x3 tail conv's output.  Thread = (f, pixel).
__global__ __launch_bounds__(256) void dec_tail_grad_kernel(const float* __restrict__ dimg, const float* __restrict__ recons,
                                                            const float* __restrict__ masks, float* __restrict__ dy,
                                                            int F, int K, int HW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)F * HW) return;
    const int f = (int)(i / HW), p = (int)(i % HW);
    const float d0 = dimg[((size_t)f * 3 + 0) * HW + p], d1 = dimg[((size_t)f * 3 + 1) * HW + p],
                d2 = dimg[((size_t)f * 3 + 2) * HW + p];
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        const float* r = recons + ((size_t)f * K + k) * 3 * HW + p;
        const float m = masks[((size_t)f * K + k) * HW + p];
        s += m * (d0 * r[0] + d1 * r[HW] + d2 * r[2 * HW]);
    }
    for (int k = 0; k < K; ++k) {
        const float* r = recons + ((size_t)f * K + k) * 3 * HW + p;
        const float m = masks[((size_t)f * K + k) * HW + p];
        const float dm = d0 * r[0] + d1 * r[HW] + d2 * r[2 * HW];
        f32x4 o = {d0 * m, d1 * m, d2 * m, m * (dm - s)};
        *reinterpret_cast<f32x4*>(dy + (((size_t)f * K + k) * HW + p) * 4) = o;
    }
}

// dx[n,y,x,ci] = relu'(act[n,y,x,ci]) * sum_{co<4, taps} w[co,ci,ty,tx] * dy[n, y+1-ty, x+1-tx, co]
// (transposed 3x3 conv 4 -> C of the tail, fused with the ReLU mask of the layer that fed it).
// Thread = (pixel, 4 input channels); w (4, C, 3, 3) nn.Conv2d layout, staged once per workgroup in LDS
// as [tap][co][C] so that a thread reads its 4 channels of one (tap, co) with one 16-byte access.
__global__ __launch_bounds__(256) void conv3x3_t4_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                         const float* __restrict__ act, float* __restrict__ dx,
                                                         long npix_total, int H, int W, int C) {
    __shared__ __attribute__((aligned(16))) float ws[9 * 4 * 64];
    for (int i = threadIdx.x; i < 9 * 4 * C; i += 256) {
        const int c = i % C, co = (i / C) & 3, tap = i / (4 * C);
        ws[i] = w[((size_t)co * C + c) * 9 + tap];
    }
    __syncthreads();
    const int cq = C / 4, wq = W / 4;
    // Item = (4 consecutive pixels of a row, 4 channels): one 16-byte LDS read of weights feeds 16 FMAs (four
    // pixels) instead of 4 -- with one pixel per item the kernel was bound by those reads (36 ds_read_b128 per 144
    // FMAs, 2.65 ms per 2040 slot images; grid-stride workgroups that keep their weight image: 2.23 ms).
    const long nitem = npix_total / 4 * cq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nitem; i += (long)gridDim.x * 256) {
        const long quad = i / cq;                                  // pixel quad: pixels 4 * quad .. + 3
        const int c4 = (int)(i % cq) * 4;
        const int x0 = (int)(quad % wq) * 4, y = (int)((quad / wq) % H);
        const long pix0 = quad * 4;
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const int yy = y + 1 - ty;
            if (yy < 0 || yy >= H) continue;
            // dy of the six columns x0 - 1 .. x0 + 4 of row yy (zero outside the image)
            f32x4 g[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int xx = x0 - 1 + j;
                g[j] = (xx >= 0 && xx < W) ? *reinterpret_cast<const f32x4*>(dy + (pix0 + (long)(yy - y) * W + (xx - x0)) * 4)
                                           : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int co = 0; co < 4; ++co) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + ((ty * 3 + tx) * 4 + co) * C + c4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] += wv * g[q + 2 - tx][co];      // pixel x0 + q reads column x0 + q + 1 - tx
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(act + (pix0 + q) * C + c4);
            f32x4 o = acc[q];
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = a[u] > 0.f ? o[u] : 0.f;
            *reinterpret_cast<f32x4*>(dx + (pix0 + q) * C + c4) = o;
        }
    }
}

// Collapsed decoder layer 0: x_in[n,p,:] = relu(cpos[p,:] + S[n, cls(p), :]).
// dS[n, cls, c] = sum over the pixels p of class cls of g[n,p,c] * (x_in > 0).  One workgroup per
// slot image; thread = (16 pixel lanes) x (16 channel quads); the 3600 interior pixels (class 12)
// accumulate in registers, border pixels through LDS atomics.
__device__ __forceinline__ int border_cls(int p, int n) { return p < 2 ? p : (p >= n - 2 ? 4 - (n - 1 - p) : 2); }

__global__ __launch_bounds__(256) void dec_class_reduce_kernel(const float* __restrict__ g, const float* __restrict__ cpos,
                                                               const float* __restrict__ S, float* __restrict__ dS,
                                                               int H, int W) {
    constexpr int C = 64;
    __shared__ float acc_s[25 * C];
    const int n = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < 25 * C; i += 256) acc_s[i] = 0.f;
    __syncthreads();
    const int c4 = (t & 15) * 4, pl = t >> 4;
    const float* Sn = S + (size_t)n * 25 * C;
    const float* gn = g + (size_t)n * H * W * C;
    f32x4 inner = {0.f, 0.f, 0.f, 0.f};
    const f32x4 s_in = *reinterpret_cast<const f32x4*>(Sn + 12 * C + c4);
    for (int p = pl; p < H * W; p += 16) {
        const int y = p / W, x = p % W;
        const int cls = border_cls(y, H) * 5 + border_cls(x, W);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gn + (size_t)p * C + c4);
        const f32x4 cp = *reinterpret_cast<const f32x4*>(cpos + (size_t)p * C + c4);
        if (cls == 12) {
#pragma unroll
            for (int u = 0; u < 4; ++u) inner[u] += (cp[u] + s_in[u]) > 0.f ? gv[u] : 0.f;
        } else {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(Sn + cls * C + c4);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (cp[u] + sv[u] > 0.f) atomicAdd(&acc_s[cls * C + c4 + u], gv[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) atomicAdd(&acc_s[12 * C + c4 + u], inner[u]);
    __syncthreads();
    for (int i = t; i < 25 * C; i += 256) dS[(size_t)n * 25 * C + i] = acc_s[i];
}

inline unsigned blocks256(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int tocvp_softmax_rows_f32(const float* x, float* y, int rows, int cols, float scale,
                                      const int32_t* key_len, int rows_per_batch, void* stream) {
    TOCVP_CHECK_ARG(x && y && rows >= 0 && cols > 0 && (key_len == nullptr || rows_per_batch > 0));
    if (rows == 0) return TOCVP_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, y, rows, cols, scale, key_len, rows_per_batch);
    return tocvp_launch_status();
}

extern "C" int tocvp_softmax_bwd_f32(const float* p, const float* dp, float* ds, int rows, int cols,
                                     float scale, void* stream) {
    TOCVP_CHECK_ARG(p && dp && ds && rows >= 0 && cols > 0);
    if (rows == 0) return TOCVP_OK;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       p, dp, ds, rows, cols, scale);
    return tocvp_launch_status();
}

extern "C" int tocvp_act_f32(const float* x, float* y, long n, int act, void* stream) {
    TOCVP_CHECK_ARG(x && y && n >= 0 && (act == TOCVP_ACT_RELU || act == TOCVP_ACT_GELU));
    if (n == 0) return TOCVP_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks256(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n,
                       act);
    return tocvp_launch_status();
}

extern "C" int tocvp_act_bwd_f32(const float* dy, const float* x, float* dx, long n, int act, void* stream) {
    TOCVP_CHECK_ARG(dy && x && dx && n >= 0 && (act == TOCVP_ACT_RELU || act == TOCVP_ACT_GELU));
    if (n == 0) return TOCVP_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks256(n)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, x, dx,
                       n, act);
    return tocvp_launch_status();
}

extern "C" int tocvp_axpby_f32(const float* x, float* y, long n, float a, float b, void* stream) {
    TOCVP_CHECK_ARG(x && y && n >= 0);
    if (n == 0) return TOCVP_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(blocks256(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n, a, b);
    return tocvp_launch_status();
}

extern "C" int tocvp_colsum_partial_f32(const float* x, float* partial, int rows, int cols, int ld,
                                        int rows_per_chunk, void* stream) {
    TOCVP_CHECK_ARG(x && partial && rows > 0 && cols > 0 && ld >= cols && rows_per_chunk > 0);
    const int chunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
    TOCVP_CHECK_ARG(chunks <= 65535);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 255) / 256, chunks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, partial, rows, cols, ld, rows_per_chunk);
    return tocvp_launch_status();
}

extern "C" int tocvp_layernorm_bwd_f32(const float* x, const float* gamma, const float* dy, float* dx,
                                       float* pgamma, float* pbeta, int nwaves, int rows, int D, float eps,
                                       int accumulate, void* stream) {
    TOCVP_CHECK_ARG(x && gamma && dy && dx && pgamma && pbeta);
    TOCVP_CHECK_ARG(rows > 0 && D > 0 && D <= 1024 && (D & 3) == 0 && nwaves > 0 && (nwaves & 3) == 0);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(dy) || !tocvp_aligned16(dx) || !tocvp_aligned16(gamma) ||
        !tocvp_aligned16(pgamma) || !tocvp_aligned16(pbeta))
        return TOCVP_EALIGN;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nwaves / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       gamma, dy, dx, pgamma, pbeta, rows, D, eps, accumulate & 3);
    return tocvp_launch_status();
}

extern "C" int tocvp_embedding_bwd_f32(const int64_t* ids, const float* dy, float* dW, int n, int D,
                                       void* stream) {
    TOCVP_CHECK_ARG(ids && dy && dW && n >= 0 && D > 0);
    if (n == 0) return TOCVP_OK;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(blocks256((long)n * D)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), ids, dy, dW, n, D);
    return tocvp_launch_status();
}

extern "C" int tocvp_mse_f32(const float* pred, const float* target, float* partial, int nblocks, float* dpred,
                             long n, float gscale, void* stream) {
    TOCVP_CHECK_ARG(pred && target && partial && nblocks > 0 && n > 0);
    hipLaunchKernelGGL(mse_kernel, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), pred, target,
                       partial, dpred, n, gscale);
    return tocvp_launch_status();
}

extern "C" int tocvp_sqnorm_partial_f32(const float* x, float* partial, int nblocks, long n, void* stream) {
    TOCVP_CHECK_ARG(x && partial && nblocks > 0 && n > 0);
    hipLaunchKernelGGL(sqnorm_kernel, dim3(nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, partial, n);
    return tocvp_launch_status();
}

extern "C" int tocvp_adam_f32(float* p, const float* g, float* m, float* v, long n, const float* hyper,
                              const float* gscale, void* stream) {
    TOCVP_CHECK_ARG(p && g && m && v && hyper && n >= 0);
    if (n == 0) return TOCVP_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks256(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v,
                       n, hyper, gscale);
    return tocvp_launch_status();
}

extern "C" int tocvp_clip_scale_f32(const float* sumsq, float max_norm, float* out, void* stream) {
    TOCVP_CHECK_ARG(sumsq && out);
    hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), sumsq, max_norm, out);
    return tocvp_launch_status();
}

extern "C" int tocvp_dec_tail_grad_f32(const float* dimg, const float* recons, const float* masks, float* dy,
                                       int F, int K, int H, int W, void* stream) {
    TOCVP_CHECK_ARG(dimg && recons && masks && dy && F >= 0 && K > 0 && H > 0 && W > 0);
    if (F == 0) return TOCVP_OK;
    hipLaunchKernelGGL(dec_tail_grad_kernel, dim3(blocks256((long)F * H * W)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dimg, recons, masks, dy, F, K, H * W);
    return tocvp_launch_status();
}

extern "C" int tocvp_conv3x3_t4_f32(const float* dy, const float* w, const float* act, float* dx, int nimg,
                                    int H, int W, int C, void* stream) {
    TOCVP_CHECK_ARG(dy && w && act && dx && nimg >= 0 && H > 0 && W > 0 && C > 0 && C <= 64 && (C & 3) == 0);
    if (!tocvp_aligned16(dy) || !tocvp_aligned16(act) || !tocvp_aligned16(dx)) return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    const long npix = (long)nimg * H * W;
    TOCVP_CHECK_ARG((W & 3) == 0);
    const long want = (npix / 4 * (C / 4) + 255) / 256;
    hipLaunchKernelGGL(conv3x3_t4_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dy, w, act, dx, npix, H, W, C);
    return tocvp_launch_status();
}

extern "C" int tocvp_dec_class_reduce_f32(const float* g, const float* cpos, const float* S, float* dS,
                                          int nimg, int H, int W, int C, void* stream) {
    TOCVP_CHECK_ARG(g && cpos && S && dS && nimg >= 0 && H >= 4 && W >= 4 && C == 64);
    if (!tocvp_aligned16(g) || !tocvp_aligned16(cpos) || !tocvp_aligned16(S)) return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    hipLaunchKernelGGL(dec_class_reduce_kernel, dim3(nimg), dim3(256), 0, static_cast<hipStream_t>(stream), g,
                       cpos, S, dS, H, W);
    return tocvp_launch_status();
}

extern "C" int tocvp_dropout_f32(const float* x, const float* r, float* y, long n, float p, void* stream) {
    TOCVP_CHECK_ARG(x && r && y && n >= 0 && p >= 0.f && p < 1.f);
    if (n == 0) return TOCVP_OK;
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks256(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, r, y, n,
                       p, 1.f / (1.f - p));
    return tocvp_launch_status();
}
