// Convolutions of the SAVi encoder / spatial-broadcast decoder.
//
//  conv5x5_mfma_kernel : Conv2d(Cin->Cout, k5, p2)(+ReLU) as an implicit GEMM on fp32 MFMA.
//      M = pixels, N = Cout, K = 25 taps x Cin.  NHWC activations.
//      One workgroup (4 waves) = an 8 x 32 pixel tile x all Cout; each wave owns 2 image rows
//      (two 32-pixel MFMA row blocks) x Cout/32 column blocks.
//      The (8+4) x (32+4) x Cin halo tile is staged ONCE in LDS and re-used by all 25 taps
//      (im2col never materialised); per-tap weight slices (Cout x Cin) stream through a
//      double-buffered LDS slot, prefetched into registers under the previous tap's MFMAs.
//      Channel stride is padded to Cin+4 floats so every ds_read_b128 operand fetch
//      (32 neighbouring pixels / 32 output channels per lane group) is bank-conflict-free.
//      in_mode 1 synthesises the tile on the fly from the analytically collapsed decoder layer 0
//      (relu(cpos[y,x,:] + S[n,cls(y,x),:])): the reference's 1.2 GB/sequence broadcast tensor
//      (SAVi.py:264-275) and its 1.68 GFLOP first conv per slot image never exist.
//  conv5x5_in3_kernel  : first encoder layer (3 -> 32), VALU, reads the NCHW video in place.
//  dec_tail_kernel     : Conv2d(64->4,k3,p1) + softmax over slots + alpha compositing, VALU.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// implicit-GEMM 5x5 convolution
// ------------------------------------------------------------------------------------------
constexpr int TH = 8, TW = 32;             // output tile (pixels)
constexpr int IH = TH + 4, IW = TW + 4;    // halo tile

struct ConvArgs {
    const float* x; const float* aux; const float* wp; const float* bias; float* y;
    int nimg, H, W, relu;
};

__device__ __forceinline__ int border_class(int p, int n) {
    return p < 2 ? p : (p >= n - 2 ? 4 - (n - 1 - p) : 2);
}

template <int CIN, int COUT, int MODE>
__global__ __launch_bounds__(256) void conv5x5_mfma_kernel(ConvArgs p) {
    constexpr int CC = CIN < 64 ? CIN : 64;   // input channels resident per pass
    constexpr int NCH = CIN / CC;
    constexpr int CS = CC + 4;                // padded channel stride
    constexpr int NI = COUT / 32;
    constexpr int F4 = CC / 4;
    constexpr int WREG = (COUT * F4) / 256;   // float4 weight prefetch registers per thread
    static_assert((COUT * F4) % 256 == 0, "weight slice must split evenly over 256 threads");

    __shared__ __attribute__((aligned(16))) float lds[IH * IW * CS + 2 * COUT * CS];
    float* in_s = lds;
    float* w_s = lds + IH * IW * CS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_x = p.W / TW, tiles_y = p.H / TH;
    const int tiles = tiles_x * tiles_y;
    const int img = blockIdx.x / tiles;
    const int tile = blockIdx.x % tiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;

    f32x16 acc[2][NI];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 wreg[WREG];
    auto wload = [&](int tap, int ch) {
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int idx = t + 256 * i;
            const int co = idx / F4, c = (idx % F4) * 4;
            wreg[i] = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)tap * COUT + co) * CIN +
                                                      ch * CC + c);
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int idx = t + 256 * i;
            const int co = idx / F4, c = (idx % F4) * 4;
            *reinterpret_cast<f32x4*>(w_s + buf * COUT * CS + co * CS + c) = wreg[i];
        }
    };

    for (int ch = 0; ch < NCH; ++ch) {
        __syncthreads();   // previous pass finished with in_s / w_s
        // ---- halo tile -> LDS (zero padding outside the image)
        for (int i = t; i < IH * IW * F4; i += 256) {
            const int pix = i / F4, c = (i % F4) * 4;
            const int iy = ty0 + pix / IW - 2, ix = tx0 + pix % IW - 2;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                if (MODE == 0) {
                    v = *reinterpret_cast<const f32x4*>(
                        p.x + (((size_t)img * p.H + iy) * p.W + ix) * CIN + ch * CC + c);
                } else {
                    const int cls = border_class(iy, p.H) * 5 + border_class(ix, p.W);
                    const f32x4 a = *reinterpret_cast<const f32x4*>(
                        p.x + ((size_t)iy * p.W + ix) * CIN + ch * CC + c);
                    const f32x4 s = *reinterpret_cast<const f32x4*>(
                        p.aux + ((size_t)img * 25 + cls) * CIN + ch * CC + c);
                    v = a + s;
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
                }
            }
            *reinterpret_cast<f32x4*>(in_s + pix * CS + c) = v;
        }
        wload(0, ch);
        wstore(0);
        __syncthreads();

        for (int tap = 0; tap < 25; ++tap) {
            const int buf = tap & 1;
            if (tap + 1 < 25) wload(tap + 1, ch);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE the MFMAs
            const int dy = tap / 5, dx = tap % 5;
            const float* a_base = in_s + ((2 * wave + dy) * IW + l31 + dx) * CS + 4 * h;
            const float* b_base = w_s + buf * COUT * CS + l31 * CS + 4 * h;
#pragma unroll
            for (int j = 0; j < CC / 8; ++j) {
                f32x4 a[2], b[NI];
                a[0] = *reinterpret_cast<const f32x4*>(a_base + 8 * j);
                a[1] = *reinterpret_cast<const f32x4*>(a_base + IW * CS + 8 * j);
#pragma unroll
                for (int n = 0; n < NI; ++n)
                    b[n] = *reinterpret_cast<const f32x4*>(b_base + n * 32 * CS + 8 * j);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < NI; ++n) acc[m][n] = mfma32(a[m][u], b[n][u], acc[m][n]);
            }
            if (tap + 1 < 25) wstore(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: bias (+ReLU), NHWC store; lanes = 32 consecutive output channels
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const float bv = p.bias[n * 32 + l31];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int oy = ty0 + 2 * wave + m;
            float* yrow = p.y + (((size_t)img * p.H + oy) * p.W + tx0) * COUT + n * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[m][n][r] + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                yrow[(size_t)acc_row(r, h) * COUT] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// first encoder layer: 3 -> 32 channels, input planes (C,H,W), output NHWC
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv5x5_in3_kernel(const float* __restrict__ x,
                                                          long long img_stride,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ y, int H, int W) {
    constexpr int CO = 32;
    __shared__ float in_s[3][20][20];
    __shared__ __attribute__((aligned(16))) float w_s[75 * CO];
    const int t = threadIdx.x;
    const int tiles_x = W / 16;
    const int img = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * 16, tx0 = (blockIdx.x % tiles_x) * 16;
    const float* xi = x + (size_t)img * img_stride;

    for (int i = t; i < 75 * CO; i += 256) {
        const int co = i / 75, k = i % 75;           // w[co][ci][dy][dx], k = ci*25 + dy*5 + dx
        w_s[k * CO + co] = w[i];
    }
    for (int i = t; i < 3 * 400; i += 256) {
        const int ci = i / 400, rem = i % 400;
        const int iy = ty0 + rem / 20 - 2, ix = tx0 + rem % 20 - 2;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = xi[((size_t)ci * H + iy) * W + ix];
        in_s[ci][rem / 20][rem % 20] = v;
    }
    __syncthreads();

    const int py = t >> 4, px = t & 15;
    f32x4 acc[CO / 4];
#pragma unroll
    for (int c = 0; c < CO / 4; ++c) acc[c] = *reinterpret_cast<const f32x4*>(bias + 4 * c);
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                const float v = in_s[ci][py + dy][px + dx];
                const float* wk = w_s + (ci * 25 + dy * 5 + dx) * CO;
#pragma unroll
                for (int c = 0; c < CO / 4; ++c)
                    acc[c] += v * *reinterpret_cast<const f32x4*>(wk + 4 * c);
            }
    float* yo = y + (((size_t)img * H + ty0 + py) * W + tx0 + px) * CO;
#pragma unroll
    for (int c = 0; c < CO / 4; ++c) {
        f32x4 v = acc[c];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
        *reinterpret_cast<f32x4*>(yo + 4 * c) = v;
    }
}

// ------------------------------------------------------------------------------------------
// decoder tail: conv3x3 (64 -> 4) per slot image, softmax over slots, compositing
// one thread = one pixel of an 8 x 16 tile, loops over the K slot images of its frame
// ------------------------------------------------------------------------------------------
// gfx950 notes: 16 x 16 pixel tile per 256-thread workgroup, input channels in two 32-channel
// passes (46.7 KB halo tile -> 2 workgroups / CU).  The 3x3x64x4 weights are read with
// WAVE-UNIFORM indices from a repacked [tap][c][4] table, so hipcc keeps them on the scalar path
// (s_load + v_fmac with an SGPR operand) instead of spending an LDS broadcast read per FMA group;
// all global loads of a halo tile are issued back to back before the first conversion.
#ifndef TOCVP_DT_ABLATE
#define TOCVP_DT_ABLATE 0                 // timing ablations (scripts/probes/dec_tail_ablate.hip): 1 no halo fetch,
#endif                                    // 2 one weight group for every tap / channel, 3 no FMAs
constexpr int DT_ABL = TOCVP_DT_ABLATE;
constexpr int DT_H = 16, DT_W = 16, DT_C = 64, DT_CC = 32, DT_CS = DT_CC + 4;
constexpr int DT_IH = DT_H + 2, DT_IW = DT_W + 2;

__global__ __launch_bounds__(256, 3) void dec_tail_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ wq,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ recons_imgs,
                                                          float* __restrict__ recons,
                                                          float* __restrict__ masks, float* __restrict__ clamped,
                                                          long img_fs, long rec_fs, long mask_fs, int F, int K, int H,
                                                          int W) {
    // frame f lands at recons_imgs + f img_fs, recons + f rec_fs, masks + f mask_fs (floats), as in dec_tail_sum_kernel
    __shared__ __attribute__((aligned(16))) float in_s[DT_IH * DT_IW * DT_CS];
    const int t = threadIdx.x;
    // XCD-aware ids: workgroups are dealt round-robin over the 8 XCDs, so the tiles of ONE frame take ids that are
    // equal mod 8 -- neighbouring tiles walk the same slot images at the same time and their halo rows then hit
    // that XCD's L2 instead of being fetched again from HBM (27 % of the tile's bytes).  Frames padded to 8.
    const int tiles = (H / DT_H) * (W / DT_W);
    const int L = blockIdx.x, j = L >> 3;
    const int f = (j / tiles) * 8 + (L & 7), tile = j % tiles;
    if (f >= F) return;
    const int tiles_x = W / DT_W;
    const int ty0 = (tile / tiles_x) * DT_H, tx0 = (tile % tiles_x) * DT_W;
    const int py = t / DT_W, px = t % DT_W;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)(ty0 + py) * W + tx0 + px;
    const f32x4 bv = {bias[0], bias[1], bias[2], bias[3]};

    constexpr int F4 = DT_CC / 4;                                   // float4 per pixel per pass
    constexpr int NIT = (DT_IH * DT_IW * F4 + 255) / 256;           // 11 loads per thread per pass

    // One step = (slot image k, 32-channel pass).  The halo tile of step s + 1 is fetched into registers BEFORE the
    // FMAs of step s (its ~2 us of memory latency then runs under ~1 us of FMAs here plus the partner
    // workgroup's), instead of in front of its own barrier.
    f32x4 tv[NIT];
    auto fetch = [&](int step) {
        const float* xi = x + ((size_t)f * K + (step >> 1)) * HW * DT_C + (step & 1) * DT_CC;
        if (DT_ABL == 1 && step > 0) return;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {                          // batched, clamped loads
            const int i = min(t + it * 256, DT_IH * DT_IW * F4 - 1);
            const int p = i / F4, c = (i % F4) * 4;
            const int iy = min(max(ty0 + p / DT_IW - 1, 0), H - 1);
            const int ix = min(max(tx0 + p % DT_IW - 1, 0), W - 1);
            tv[it] = *reinterpret_cast<const f32x4*>(xi + ((size_t)iy * W + ix) * DT_C + c);
        }
    };
    fetch(0);
    f32x4 acc = bv;
#pragma unroll 1
    for (int step = 0; step < 2 * K; ++step) {
        const int k = step >> 1, pass = step & 1;
        __syncthreads();                                            // previous step consumed
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = t + it * 256;
            if (i < DT_IH * DT_IW * F4) {
                const int p = i / F4, c = (i % F4) * 4;
                const int iy = ty0 + p / DT_IW - 1, ix = tx0 + p % DT_IW - 1;
                const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
                f32x4 v = tv[it];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = inside ? v[u] : 0.f;
                *reinterpret_cast<f32x4*>(in_s + p * DT_CS + c) = v;
            }
        }
        __syncthreads();
        if (step + 1 < 2 * K) fetch(step + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* ip = in_s + ((py + tap / 3) * DT_IW + px + tap % 3) * DT_CS;
            const float* wt = wq + ((size_t)tap * DT_C + pass * DT_CC) * 4;       // wave-uniform
#pragma unroll
            for (int c4 = 0; c4 < DT_CC / 4; ++c4) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(ip + 4 * c4);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* w4 = DT_ABL == 2 ? wq : wt + (4 * c4 + u) * 4;
                    if (DT_ABL == 3) {
                        acc[u] += xv[u];
                        continue;
                    }
                    acc[0] = fmaf(xv[u], w4[0], acc[0]);
                    acc[1] = fmaf(xv[u], w4[1], acc[1]);
                    acc[2] = fmaf(xv[u], w4[2], acc[2]);
                    acc[3] = fmaf(xv[u], w4[3], acc[3]);
                }
            }
        }
        if (pass == 1) {
            float* ro = recons + (size_t)f * rec_fs + (size_t)k * 3 * HW + pix;
            ro[0] = acc[0];
            ro[HW] = acc[1];
            ro[2 * HW] = acc[2];
            masks[(size_t)f * mask_fs + (size_t)k * HW + pix] = acc[3];   // raw alpha, normalised in place below
            acc = bv;
        }
    }

    // softmax over slots (exact two-pass, as F.softmax) + compositing.  The raw alphas of this pixel sit in its own
    // `masks` words (written by this thread: no other thread or workgroup touches them), not in 32 KB of LDS -- the
    // halo tile alone leaves room for three workgroups per CU.
    float* mp = masks + (size_t)f * mask_fs + pix;
    float m = -1.0e30f;
    for (int k = 0; k < K; ++k) m = fmaxf(m, mp[(size_t)k * HW]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += expf(mp[(size_t)k * HW] - m);
    const float inv = 1.0f / sum;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int k = 0; k < K; ++k) {
        const float mk = expf(mp[(size_t)k * HW] - m) * inv;
        mp[(size_t)k * HW] = mk;
        const float* ro = recons + (size_t)f * rec_fs + (size_t)k * 3 * HW + pix;
        c0 += ro[0] * mk;
        c1 += ro[HW] * mk;
        c2 += ro[2 * HW] * mk;
    }
    float* co = recons_imgs + (size_t)f * img_fs + pix;
    co[0] = c0;
    co[HW] = c1;
    co[2 * HW] = c2;
    if (clamped) {
        float* cc = clamped + (size_t)f * img_fs + pix;
        cc[0] = tocvp_clamp01(c0);
        cc[HW] = tocvp_clamp01(c1);
        cc[2 * HW] = tocvp_clamp01(c2);
    }
}

// (4, C, 3, 3) -> [tap][c][4] so that the 4 output channels of one (tap, c) are one 16-byte group
__global__ __launch_bounds__(256) void dec_tail_pack_kernel(const float* __restrict__ w,
                                                            float* __restrict__ wq, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 9 * C * 4) return;
    const int co = i & 3, c = (i >> 2) % C, tap = i / (4 * C);
    wq[i] = w[((size_t)co * C + c) * 9 + tap];
}

template <int CIN, int COUT>
int launch_conv(const ConvArgs& a, int in_mode, hipStream_t s) {
    const int tiles = (a.H / TH) * (a.W / TW);
    const dim3 grid((unsigned)((size_t)a.nimg * tiles));
    if (in_mode == 0)
        hipLaunchKernelGGL((conv5x5_mfma_kernel<CIN, COUT, 0>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((conv5x5_mfma_kernel<CIN, COUT, 1>), grid, dim3(256), 0, s, a);
    return tocvp_launch_status();
}

}  // namespace

extern "C" int tocvp_conv5x5_f32(const float* x, const float* aux, int in_mode, const float* wp,
                                 const float* bias, float* y, int nimg, int H, int W, int Cin,
                                 int Cout, int relu, void* stream) {
    TOCVP_CHECK_ARG(x && wp && bias && y);
    TOCVP_CHECK_ARG(in_mode == 0 || (in_mode == 1 && aux != nullptr));
    TOCVP_CHECK_ARG(nimg >= 0 && H > 0 && W > 0 && (H % TH) == 0 && (W % TW) == 0);
    TOCVP_CHECK_ARG((size_t)nimg * (H / TH) * (W / TW) < 0x7fffffffu);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(wp) || (aux && !tocvp_aligned16(aux)))
        return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    ConvArgs a{x, aux, wp, bias, y, nimg, H, W, relu};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Cin == 64 && Cout == 64) return launch_conv<64, 64>(a, in_mode, s);
    if (Cin == 32 && Cout == 32) return launch_conv<32, 32>(a, in_mode, s);
    if (Cin == 128 && Cout == 64) return launch_conv<128, 64>(a, in_mode, s);
    if (Cin == 32 && Cout == 64) return launch_conv<32, 64>(a, in_mode, s);
    if (Cin == 64 && Cout == 32) return launch_conv<64, 32>(a, in_mode, s);
    return TOCVP_EINVAL;
}

extern "C" int tocvp_conv5x5_in3_f32(const float* x, long long img_stride, const float* w,
                                     const float* bias, float* y, int nimg, int H, int W, int Cout,
                                     void* stream) {
    TOCVP_CHECK_ARG(x && w && bias && y);
    TOCVP_CHECK_ARG(nimg >= 0 && nimg <= 65535 && Cout == 32 && (H % 16) == 0 && (W % 16) == 0);
    if (!tocvp_aligned16(bias) || !tocvp_aligned16(y)) return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    hipLaunchKernelGGL(conv5x5_in3_kernel, dim3((H / 16) * (W / 16), nimg), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, img_stride, w, bias, y, H, W);
    return tocvp_launch_status();
}

extern "C" int tocvp_dec_tail_placed_f32(const float* x, const float* w, const float* bias, float* recons_imgs,
                                         float* recons, float* masks, float* clamped_imgs, long img_fs, long rec_fs,
                                         long mask_fs, int F, int K, int H, int W, int Cin, void* ws, size_t ws_bytes,
                                         void* stream) {
    TOCVP_CHECK_ARG(x && w && bias && recons_imgs && recons && masks && ws);
    TOCVP_CHECK_ARG(F >= 0 && F <= 65535 && K > 0 && K <= 32 && Cin == DT_C);
    TOCVP_CHECK_ARG((H % DT_H) == 0 && (W % DT_W) == 0);
    TOCVP_CHECK_ARG(ws_bytes >= (size_t)9 * DT_C * 4 * sizeof(float));
    TOCVP_CHECK_ARG(img_fs >= 3L * H * W && rec_fs >= 3L * K * H * W && mask_fs >= (long)K * H * W);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(ws)) return TOCVP_EALIGN;
    if (F == 0) return TOCVP_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* wq = static_cast<float*>(ws);
    hipLaunchKernelGGL(dec_tail_pack_kernel, dim3((9 * DT_C * 4 + 255) / 256), dim3(256), 0, s, w, wq,
                       DT_C);
    if (hipGetLastError() != hipSuccess) return TOCVP_ELAUNCH;
    hipLaunchKernelGGL(dec_tail_kernel, dim3((H / DT_H) * (W / DT_W) * ((F + 7) / 8 * 8)), dim3(256), 0, s, x,
                       static_cast<const float*>(wq), bias, recons_imgs, recons, masks, clamped_imgs, img_fs, rec_fs,
                       mask_fs, F, K, H, W);
    return tocvp_launch_status();
}

extern "C" int tocvp_dec_tail_f32(const float* x, const float* w, const float* bias,
                                  float* recons_imgs, float* recons, float* masks, int F, int K,
                                  int H, int W, int Cin, void* ws, size_t ws_bytes, void* stream) {
    return tocvp_dec_tail_placed_f32(x, w, bias, recons_imgs, recons, masks, nullptr, 3L * H * W, 3L * K * H * W,
                                     (long)K * H * W, F, K, H, W, Cin, ws, ws_bytes, stream);
}
