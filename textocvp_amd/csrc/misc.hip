// Small HBM-bound row / elementwise kernels: LayerNorm (+fused positional addend), GRU gates,
// SoftPositionEmbed table, text embedding front end, conv weight repacking, decoder tap sums.
// One wave (64 lanes) per row with 16-byte loads and __shfl_xor butterflies; no LDS.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row cached in registers (D <= 1024)
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// NSPLIT = 0: fp32 output y (rows, D); NSPLIT = 2/3: y is (rows, NSPLIT, D) bf16 planes;
// NSPLIT = 22: y is (rows, 2, D) fp16 planes of 2^8 y (f16x3 GEMM operand)
template <int NSPLIT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ add, int add_rows,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        void* __restrict__ yv, int rows, int D,
                                                        float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    const float* ar = add ? add + (size_t)(row % add_rows) * D : nullptr;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c);
            if (ar) v[i] += *reinterpret_cast<const f32x4*>(ar + c);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum64(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            const f32x4 d = v[i] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum64(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 o = (v[i] - mean) * rstd * g + b;
            if (NSPLIT == 0) {
                *reinterpret_cast<f32x4*>(static_cast<float*>(yv) + (size_t)row * D + c) = o;
            } else {
                constexpr int PLANES = NSPLIT == 22 ? 2 : NSPLIT;
                tocvp_store_planes4(yv, (size_t)row * PLANES * D + c, (size_t)D, o, NSPLIT);
            }
        }
    }
}

// LayerNorm of NARROW rows (D = 32 / 64 / 128: the SAVi encoder's per-pixel LayerNorms over 4.2 M rows per chunk):
// D/4 lanes per row (16 bytes each), 256/D rows per wave -- the one-wave-per-row kernel above moves only D*4 bytes per
// wave instruction there (D = 32: 1.33 ms for 1 GB at 1024 images, 0.8 TB/s; this form 0.19 ms).  Same arithmetic
// and the same reduction order per row as layernorm_kernel<0> (whose idle lanes add exact zeros), so the results
// are bit-identical.
template <int D>
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ add, int add_rows,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               float* __restrict__ y, long rows, float eps) {
    constexpr int LPR = D / 4;                       // lanes per row
    const long row = (long)blockIdx.x * (256 / LPR) + (threadIdx.x / LPR);
    const int c = (threadIdx.x % LPR) * 4;
    if (row >= rows) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + row * D + c);
    if (add) v += *reinterpret_cast<const f32x4*>(add + (row % add_rows) * D + c);
    float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    const f32x4 d = v - mean;
    float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)D + eps);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
    *reinterpret_cast<f32x4*>(y + row * D + c) = d * rstd * g + b;
}

// T5LayerNorm: y = x * rsqrt(mean(x^2) + eps) * gamma   (no mean subtraction, no bias)
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ gamma,
                                                      float* __restrict__ y, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    f32x4 v[4];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c);
            q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum64(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D)
            *reinterpret_cast<f32x4*>(y + (size_t)row * D + c) =
                v[i] * rstd * *reinterpret_cast<const f32x4*>(gamma + c);
    }
}

// out[r, :] = table[ids[r], :]   (token embedding lookup, ids clamped into the table)
__global__ __launch_bounds__(256) void embedding_kernel(const int64_t* __restrict__ ids,
                                                        const float* __restrict__ table,
                                                        float* __restrict__ out, int rows, int D,
                                                        int vocab) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    long id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    for (int c = lane * 4; c < D; c += 256)
        *reinterpret_cast<f32x4*>(out + (size_t)row * D + c) =
            *reinterpret_cast<const f32x4*>(table + (size_t)id * D + c);
}

// ------------------------------------------------------------------------------------------
// GRUCell gates
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_gates_kernel(const float* __restrict__ gi,
                                                        const float* __restrict__ gh,
                                                        const float* __restrict__ h,
                                                        float* __restrict__ out, int rows, int D) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * D) return;
    const long row = i / D;
    const int d = (int)(i - row * D);
    const float* gir = gi + row * 3 * D;
    const float* ghr = gh + row * 3 * D;
    const float r = sigmoidf_(gir[d] + ghr[d]);
    const float z = sigmoidf_(gir[D + d] + ghr[D + d]);
    const float n = tanhf(gir[2 * D + d] + r * ghr[2 * D + d]);
    out[i] = (1.0f - z) * n + z * h[i];
}

// ------------------------------------------------------------------------------------------
// SoftPositionEmbed addend table (H, W, C)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pos_embed_kernel(const float* __restrict__ w,
                                                        const float* __restrict__ b,
                                                        float* __restrict__ out, int H, int W, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)H * W * C) return;
    const int c = (int)(i % C);
    const int xx = (int)((i / C) % W);
    const int yy = (int)(i / ((long)C * W));
    // numpy.linspace(-1, 1, n) in float64, cast to fp32; "1 - g" is evaluated in fp32
    const float gy = (H > 1) ? (float)(-1.0 + (double)yy * (2.0 / (double)(H - 1))) : -1.0f;
    const float gx = (W > 1) ? (float)(-1.0 + (double)xx * (2.0 / (double)(W - 1))) : -1.0f;
    const float* wc = w + (size_t)c * 4;
    float v = gy * wc[0];
    v = fmaf(gx, wc[1], v);
    v = fmaf(1.0f - gy, wc[2], v);
    v = fmaf(1.0f - gx, wc[3], v);
    out[i] = v + b[c];
}

// ------------------------------------------------------------------------------------------
// text front end: (tok_emb[id] + pos_emb[j]) -> LayerNorm -> zero padding rows
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_embed_kernel(const int64_t* __restrict__ tokens,
                                                         const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos_emb,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         float* __restrict__ out, int rows, int L,
                                                         int D, int vocab, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    long id = tokens[row];
    const bool pad = (id == 0);
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    const int j = row % L;
    float v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = 0.f;
        if (c < D) {
            v[i] = tok_emb[(size_t)id * D + c] + pos_emb[(size_t)j * D + c];
            s += v[i];
        }
    }
    const float mean = wave_sum64(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < D) q += (v[i] - mean) * (v[i] - mean);
    const float rstd = 1.0f / sqrtf(wave_sum64(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < D) {
            const float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
            out[(size_t)row * D + c] = pad ? 0.f : o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// conv weight repack (Cout, Cin, k, k) -> (k*k, Cout, Cin)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w,
                                                        float* __restrict__ wp, int Cout, int Cin,
                                                        int kk) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)kk * Cout * Cin) return;
    const int ci = (int)(i % Cin);
    const int co = (int)((i / Cin) % Cout);
    const int tap = (int)(i / ((long)Cin * Cout));
    wp[i] = w[((size_t)co * Cin + ci) * kk + tap];
}

// ------------------------------------------------------------------------------------------
// decoder layer-0 tap sums: out[(cy*5+cx), co, ci] = sum of the taps that stay inside the image
// for an output pixel of border class (cy, cx):  class 0/1 = first/second row (or column),
// 2 = interior, 3/4 = second-to-last/last.  Tap dy in [-2,2] is valid iff the input row y+dy
// exists: class 0 -> dy >= 0, class 1 -> dy >= -1, class 3 -> dy <= 1, class 4 -> dy <= 0.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void class_range(int cls, int& lo, int& hi) {
    lo = (cls == 0) ? 2 : (cls == 1) ? 1 : 0;      // tap index = d + 2
    hi = (cls == 4) ? 2 : (cls == 3) ? 3 : 4;
}

__global__ __launch_bounds__(256) void dec_tapsum_kernel(const float* __restrict__ w,
                                                         float* __restrict__ out, int Cout, int Cin) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)25 * Cout * Cin) return;
    const int ci = (int)(i % Cin);
    const int co = (int)((i / Cin) % Cout);
    const int cls = (int)(i / ((long)Cin * Cout));
    int ylo, yhi, xlo, xhi;
    class_range(cls / 5, ylo, yhi);
    class_range(cls % 5, xlo, xhi);
    const float* wk = w + ((size_t)co * Cin + ci) * 25;
    float s = 0.f;
    for (int ty = ylo; ty <= yhi; ++ty)
        for (int tx = xlo; tx <= xhi; ++tx) s += wk[ty * 5 + tx];
    out[i] = s;
}

// LearnedRandom slot initialiser: out[r, d] = mu[d] + sigma[d] * noise[r, d]
__global__ __launch_bounds__(256) void slot_init_kernel(const float* __restrict__ mu,
                                                        const float* __restrict__ sigma,
                                                        const float* __restrict__ noise,
                                                        float* __restrict__ out, long n, int D) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % D);
    out[i] = mu[d] + sigma[d] * noise[i];
}

inline int blocks_for(long n, int per) { return (int)((n + per - 1) / per); }

// max |x| of a tensor as the BIT PATTERN of the (non-negative) float: orders like an unsigned integer, so one
// dst[i0, i1, i2, :L] = src[i0, i1, i2, :L] with independent strides (floats) on both sides and contiguous runs of L floats
// (L % 4 == 0, every stride % 4 == 0, 16-byte aligned bases): the index copies of the hot path -- window slices, stacks,
// the time-major copy of the frames -- without a torch kernel
__global__ __launch_bounds__(256) void copy4d_kernel(const float* __restrict__ src, long ss0, long ss1, long ss2,
                                                     float* __restrict__ dst, long ds0, long ds1, long ds2, int n1, int n2,
                                                     int L4, long total4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int l = (int)(i % L4);
        long r = i / L4;
        const int i2 = (int)(r % n2);
        r /= n2;
        const int i1 = (int)(r % n1);
        const long i0 = r / n1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i0 * ss0 + i1 * ss1 + i2 * ss2 + 4 * l);
        *reinterpret_cast<f32x4*>(dst + i0 * ds0 + i1 * ds1 + i2 * ds2 + 4 * l) = v;
    }
}

// dst[r, :] = clamp(src[r * src_rs + :], 0, 1) for rows of row_len floats (row_len % 4 == 0, 16-byte aligned rows):
// the evaluator's targets = videos[:, ctx : ctx + P].clamp(0, 1) in one pass over a row-strided slice
__global__ __launch_bounds__(256) void clamp01_rows_kernel(const float* __restrict__ src, long src_rs,
                                                           float* __restrict__ dst, long row_len4, long total4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long r = i / row_len4, c = i - r * row_len4;
        f32x4 v = *reinterpret_cast<const f32x4*>(src + r * src_rs + 4 * c);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = tocvp_clamp01(v[u]);
        *reinterpret_cast<f32x4*>(dst + 4 * i) = v;
    }
}

// atomicMax per wave collects it; a NaN anywhere reads as +inf (range checks must not pass on NaN)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        const unsigned b = v != v ? 0x7f800000u : (__float_as_uint(v) & 0x7fffffffu);
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)m, o, 64);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

}  // namespace

extern "C" int tocvp_version(void) { return TOCVP_VERSION; }

extern "C" const char* tocvp_strerror(int code) {
    switch (code) {
        case TOCVP_OK: return "ok";
        case TOCVP_EINVAL: return "invalid argument";
        case TOCVP_ELAUNCH: return "kernel launch failed";
        case TOCVP_EALIGN: return "pointer or leading dimension not 16-byte aligned";
        default: return "unknown error";
    }
}

extern "C" int tocvp_layernorm_f32(const float* x, const float* add, int add_rows,
                                   const float* gamma, const float* beta, float* y, int rows, int D,
                                   float eps, void* stream) {
    TOCVP_CHECK_ARG(x && gamma && beta && y);
    TOCVP_CHECK_ARG(rows >= 0 && D > 0 && D <= 1024 && (D & 3) == 0);
    TOCVP_CHECK_ARG(add == nullptr || add_rows > 0);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(y) || !tocvp_aligned16(gamma) ||
        !tocvp_aligned16(beta) || (add && !tocvp_aligned16(add)))
        return TOCVP_EALIGN;
    if (rows == 0) return TOCVP_OK;
    if (D == 32 || D == 64 || D == 128) {
        const dim3 grid(blocks_for(rows, 1024 / D));
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (D == 32)
            hipLaunchKernelGGL(layernorm_narrow_kernel<32>, grid, dim3(256), 0, st, x, add, add_rows, gamma, beta, y,
                               (long)rows, eps);
        else if (D == 64)
            hipLaunchKernelGGL(layernorm_narrow_kernel<64>, grid, dim3(256), 0, st, x, add, add_rows, gamma, beta, y,
                               (long)rows, eps);
        else
            hipLaunchKernelGGL(layernorm_narrow_kernel<128>, grid, dim3(256), 0, st, x, add, add_rows, gamma, beta, y,
                               (long)rows, eps);
        return tocvp_launch_status();
    }
    hipLaunchKernelGGL(layernorm_kernel<0>, dim3(blocks_for(rows, 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, add, add_rows, gamma, beta,
                       static_cast<void*>(y), rows, D, eps);
    return tocvp_launch_status();
}

extern "C" int tocvp_layernorm_split_bf16(const float* x, const float* add, int add_rows,
                                          const float* gamma, const float* beta, void* ysplit,
                                          int nsplit, int rows, int D, float eps, void* stream) {
    TOCVP_CHECK_ARG(x && gamma && beta && ysplit && (nsplit == 2 || nsplit == 3 || nsplit == 22));
    TOCVP_CHECK_ARG(rows >= 0 && D > 0 && D <= 1024 && (D & 3) == 0);
    TOCVP_CHECK_ARG(add == nullptr || add_rows > 0);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(ysplit) || !tocvp_aligned16(gamma) ||
        !tocvp_aligned16(beta) || (add && !tocvp_aligned16(add)))
        return TOCVP_EALIGN;
    if (rows == 0) return TOCVP_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nsplit == 22)
        hipLaunchKernelGGL(layernorm_kernel<22>, dim3(blocks_for(rows, 4)), dim3(256), 0, s, x, add,
                           add_rows, gamma, beta, ysplit, rows, D, eps);
    else if (nsplit == 2)
        hipLaunchKernelGGL(layernorm_kernel<2>, dim3(blocks_for(rows, 4)), dim3(256), 0, s, x, add,
                           add_rows, gamma, beta, ysplit, rows, D, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel<3>, dim3(blocks_for(rows, 4)), dim3(256), 0, s, x, add,
                           add_rows, gamma, beta, ysplit, rows, D, eps);
    return tocvp_launch_status();
}

extern "C" int tocvp_gru_gates_f32(const float* gi, const float* gh, const float* h, float* out,
                                   int rows, int D, void* stream) {
    TOCVP_CHECK_ARG(gi && gh && h && out && rows >= 0 && D > 0);
    if (rows == 0) return TOCVP_OK;
    hipLaunchKernelGGL(gru_gates_kernel, dim3(blocks_for((long)rows * D, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), gi, gh, h, out, rows, D);
    return tocvp_launch_status();
}

extern "C" int tocvp_pos_embed_f32(const float* w, const float* b, float* out, int H, int W, int C,
                                   void* stream) {
    TOCVP_CHECK_ARG(w && b && out && H > 0 && W > 0 && C > 0);
    hipLaunchKernelGGL(pos_embed_kernel, dim3(blocks_for((long)H * W * C, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, b, out, H, W, C);
    return tocvp_launch_status();
}

extern "C" int tocvp_text_embed_f32(const int64_t* tokens, const float* tok_emb,
                                    const float* pos_emb, const float* gamma, const float* beta,
                                    float* out, int B, int L, int D, int vocab, float eps,
                                    void* stream) {
    TOCVP_CHECK_ARG(tokens && tok_emb && pos_emb && gamma && beta && out);
    TOCVP_CHECK_ARG(B >= 0 && L > 0 && D > 0 && D <= 256 && vocab > 0);
    if (B == 0) return TOCVP_OK;
    hipLaunchKernelGGL(text_embed_kernel, dim3(blocks_for((long)B * L, 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), tokens, tok_emb, pos_emb, gamma, beta, out,
                       B * L, L, D, vocab, eps);
    return tocvp_launch_status();
}

extern "C" int tocvp_pack_conv_weights_f32(const float* w, float* wp, int Cout, int Cin, int ksize,
                                           void* stream) {
    TOCVP_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && ksize > 0);
    const int kk = ksize * ksize;
    hipLaunchKernelGGL(pack_conv_kernel, dim3(blocks_for((long)kk * Cout * Cin, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, wp, Cout, Cin, kk);
    return tocvp_launch_status();
}

extern "C" int tocvp_dec_tapsum_f32(const float* w, float* out, int Cout, int Cin, void* stream) {
    TOCVP_CHECK_ARG(w && out && Cout > 0 && Cin > 0);
    hipLaunchKernelGGL(dec_tapsum_kernel, dim3(blocks_for((long)25 * Cout * Cin, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, out, Cout, Cin);
    return tocvp_launch_status();
}

extern "C" int tocvp_slot_init_f32(const float* mu, const float* sigma, const float* noise,
                                   float* out, int rows, int D, void* stream) {
    TOCVP_CHECK_ARG(mu && sigma && noise && out && rows >= 0 && D > 0);
    if (rows == 0) return TOCVP_OK;
    hipLaunchKernelGGL(slot_init_kernel, dim3(blocks_for((long)rows * D, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), mu, sigma, noise, out, (long)rows * D, D);
    return tocvp_launch_status();
}

extern "C" int tocvp_rmsnorm_f32(const float* x, const float* gamma, float* y, int rows, int D,
                                 float eps, void* stream) {
    TOCVP_CHECK_ARG(x && gamma && y && rows >= 0 && D > 0 && D <= 1024 && (D & 3) == 0);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(y) || !tocvp_aligned16(gamma)) return TOCVP_EALIGN;
    if (rows == 0) return TOCVP_OK;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(blocks_for(rows, 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, gamma, y, rows, D, eps);
    return tocvp_launch_status();
}

extern "C" int tocvp_embedding_f32(const int64_t* ids, const float* table, float* out, int rows,
                                   int D, int vocab, void* stream) {
    TOCVP_CHECK_ARG(ids && table && out && rows >= 0 && D > 0 && (D & 3) == 0 && vocab > 0);
    if (!tocvp_aligned16(table) || !tocvp_aligned16(out)) return TOCVP_EALIGN;
    if (rows == 0) return TOCVP_OK;
    hipLaunchKernelGGL(embedding_kernel, dim3(blocks_for(rows, 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), ids, table, out, rows, D, vocab);
    return tocvp_launch_status();
}

/* out (one 32-bit word) = bit pattern of max |x[i]| over n floats (NaN counts as +inf); the word is zeroed here. */
extern "C" int tocvp_absmax_f32(const float* x, long n, void* out, void* stream) {
    TOCVP_CHECK_ARG(x && out && n >= 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(out, 0, 4, s) != hipSuccess) return TOCVP_ELAUNCH;
    if (n == 0) return TOCVP_OK;
    const long want = (n + 255) / 256;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, s, x, n,
                       static_cast<unsigned*>(out));
    return tocvp_launch_status();
}

/* dst (rows, row_len) contiguous = clamp(src rows of row_len floats, src_row_stride floats apart, 0, 1); NaN stays NaN
 * (torch.clamp).  The evaluator's targets: videos[:, ctx : ctx + P].clamp(0, 1) (05_evaluate_predictor.py:95). */
extern "C" int tocvp_clamp01_rows_f32(const float* src, long src_row_stride, float* dst, long rows, long row_len,
                                      void* stream) {
    TOCVP_CHECK_ARG(src && dst && rows >= 0 && row_len >= 0 && (row_len & 3) == 0 && (src_row_stride & 3) == 0);
    TOCVP_CHECK_ARG(src_row_stride >= row_len);
    if (!tocvp_aligned16(src) || !tocvp_aligned16(dst)) return TOCVP_EALIGN;
    const long total4 = rows * (row_len / 4);
    if (total4 == 0) return TOCVP_OK;
    const long want = (total4 + 255) / 256;
    hipLaunchKernelGGL(clamp01_rows_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, src_row_stride, dst, row_len / 4, total4);
    return tocvp_launch_status();
}

/* dst[i0, i1, i2, 0:L] = src[i0, i1, i2, 0:L]: a strided 4-d copy (strides in floats, independent on both sides; L, every
 * stride and both bases multiples of 4 floats).  Replaces the index copies torch would launch on the hot path
 * (`torch.cat` / `torch.stack` / `.contiguous()` of window slices in predictor_wrapper.py:60-69, text_cond_OCVP.py:96-113,
 * the frame loop of SAVi.py:139-223). */
extern "C" int tocvp_copy4d_f32(const float* src, long ss0, long ss1, long ss2, float* dst, long ds0, long ds1, long ds2,
                                int n0, int n1, int n2, int L, void* stream) {
    TOCVP_CHECK_ARG(src && dst && n0 >= 0 && n1 >= 0 && n2 >= 0 && L >= 0 && (L & 3) == 0);
    TOCVP_CHECK_ARG(((ss0 | ss1 | ss2 | ds0 | ds1 | ds2) & 3) == 0);
    if (!tocvp_aligned16(src) || !tocvp_aligned16(dst)) return TOCVP_EALIGN;
    const long total4 = (long)n0 * n1 * n2 * (L / 4);
    if (total4 == 0) return TOCVP_OK;
    const long want = (total4 + 255) / 256;
    hipLaunchKernelGGL(copy4d_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, ss0, ss1, ss2, dst, ds0, ds1, ds2, n1, n2, L / 4, total4);
    return tocvp_launch_status();
}
