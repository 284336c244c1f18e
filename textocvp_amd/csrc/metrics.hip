// Image-quality metrics of the evaluation step that follows the rollout (reference lib/metrics.py:
// PSNR :181-212, SSIM :216-255, both delegating to piqa==1.2.2), fused with the evaluator's clamp
// (05_evaluate_predictor.py:96-99) so predicted frames are read once.
//
//   PSNR_n = 10 log10(1 / (mean_{c,h,w} (x-y)^2 + 1e-8))                       (piqa.psnr, value_range 1)
//   SSIM_n = mean_{c,h',w'} [(2 mu_x mu_y + c1)(2 s_xy + c2)] / [(mu_x^2 + mu_y^2 + c1)(s_xx + s_yy + c2)]
//            with an 11-tap Gaussian (sigma 1.5) window applied separably per channel, VALID padding,
//            c1 = 0.01^2, c2 = 0.03^2, s_ab = E[ab] - mu_a mu_b                  (piqa.ssim.SSIM defaults)
//
// One workgroup per (image, channel): both planes are clamped into LDS once, every thread evaluates
// the 11x11 window for its output pixels from LDS (tiny problem: 3.2 GFLOP for a 608-frame batch),
// partial sums go to a workspace and a second kernel reduces them in a fixed order (deterministic).
#include "common.h"

namespace {

constexpr int WIN = 11;

__global__ __launch_bounds__(256) void metric_partial_kernel(const float* __restrict__ preds,
                                                             const float* __restrict__ targets,
                                                             float* __restrict__ ws, int C, int H,
                                                             int W, int do_clamp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;
    float* ys = sm + H * W;
    __shared__ float red[2][4];
    const int t = threadIdx.x, c = blockIdx.x, n = blockIdx.y;
    const float* xp = preds + ((size_t)n * C + c) * H * W;
    const float* yp = targets + ((size_t)n * C + c) * H * W;

    float g[WIN];
    {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < WIN; ++i) {
            const float d = (float)i - 0.5f * (WIN - 1);
            g[i] = expf(-d * d / (2.f * 1.5f * 1.5f));
            s += g[i];
        }
#pragma unroll
        for (int i = 0; i < WIN; ++i) g[i] /= s;
    }

    float sq = 0.f;
    for (int i = t; i < H * W; i += 256) {
        float x = xp[i], y = yp[i];
        if (do_clamp) {
            x = fminf(fmaxf(x, 0.f), 1.f);
            y = fminf(fmaxf(y, 0.f), 1.f);
        }
        xs[i] = x;
        ys[i] = y;
        sq += (x - y) * (x - y);
    }
    __syncthreads();

    const int OH = H - WIN + 1, OW = W - WIN + 1;
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    float ss = 0.f;
    for (int o = t; o < OH * OW; o += 256) {
        const int oy = o / OW, ox = o % OW;
        float mx = 0.f, my = 0.f, mxx = 0.f, myy = 0.f, mxy = 0.f;
        for (int i = 0; i < WIN; ++i) {
            float rx = 0.f, ry = 0.f, rxx = 0.f, ryy = 0.f, rxy = 0.f;
            const float* xr = xs + (oy + i) * W + ox;
            const float* yr = ys + (oy + i) * W + ox;
#pragma unroll
            for (int j = 0; j < WIN; ++j) {
                const float x = xr[j], y = yr[j], w = g[j];
                rx = fmaf(w, x, rx);
                ry = fmaf(w, y, ry);
                rxx = fmaf(w, x * x, rxx);
                ryy = fmaf(w, y * y, ryy);
                rxy = fmaf(w, x * y, rxy);
            }
            mx = fmaf(g[i], rx, mx);
            my = fmaf(g[i], ry, my);
            mxx = fmaf(g[i], rxx, mxx);
            myy = fmaf(g[i], ryy, myy);
            mxy = fmaf(g[i], rxy, mxy);
        }
        const float sxx = mxx - mx * mx, syy = myy - my * my, sxy = mxy - mx * my;
        const float cs = (2.f * sxy + c2) / (sxx + syy + c2);
        ss += (2.f * mx * my + c1) / (mx * mx + my * my + c1) * cs;
    }

    sq = wave_sum64(sq);
    ss = wave_sum64(ss);
    if ((t & 63) == 0) {
        red[0][t >> 6] = sq;
        red[1][t >> 6] = ss;
    }
    __syncthreads();
    if (t == 0) {
        ws[((size_t)n * C + c) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        ws[((size_t)n * C + c) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ __launch_bounds__(256) void metric_final_kernel(const float* __restrict__ ws,
                                                           float* __restrict__ psnr,
                                                           float* __restrict__ ssim, int N, int C,
                                                           int H, int W) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float sq = 0.f, ss = 0.f;
    for (int c = 0; c < C; ++c) {
        sq += ws[((size_t)n * C + c) * 2 + 0];
        ss += ws[((size_t)n * C + c) * 2 + 1];
    }
    const float mse = sq / (float)(C * H * W);
    if (psnr) psnr[n] = 10.f * log10f(1.f / (mse + 1e-8f));
    if (ssim) ssim[n] = ss / (float)(C * (H - WIN + 1) * (W - WIN + 1));
}

}  // namespace

extern "C" size_t tocvp_metrics_ws_bytes(int N, int C) {
    return (N > 0 && C > 0) ? (size_t)N * C * 2 * sizeof(float) : 0;
}

extern "C" int tocvp_psnr_ssim_f32(const float* preds, const float* targets, float* psnr,
                                   float* ssim, int N, int C, int H, int W, int clamp01, void* ws,
                                   size_t ws_bytes, void* stream) {
    TOCVP_CHECK_ARG(preds && targets && (psnr || ssim) && ws);
    TOCVP_CHECK_ARG(N >= 0 && N <= 65535 * 256 && C > 0 && C <= 65535 && H >= WIN && W >= WIN);
    TOCVP_CHECK_ARG((size_t)H * W * 2 * sizeof(float) <= 160 * 1024 - 64);
    TOCVP_CHECK_ARG(ws_bytes >= tocvp_metrics_ws_bytes(N, C) && N <= 65535);
    if (N == 0) return TOCVP_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t shm = (size_t)H * W * 2 * sizeof(float);
    if (shm > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(metric_partial_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
            return TOCVP_ELAUNCH;
    }
    hipLaunchKernelGGL(metric_partial_kernel, dim3(C, N), dim3(256), shm, s, preds, targets,
                       static_cast<float*>(ws), C, H, W, clamp01);
    if (hipGetLastError() != hipSuccess) return TOCVP_ELAUNCH;
    hipLaunchKernelGGL(metric_final_kernel, dim3((N + 255) / 256), dim3(256), 0, s,
                       static_cast<const float*>(ws), psnr, ssim, N, C, H, W);
    return tocvp_launch_status();
}
