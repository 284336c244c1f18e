// 5x5 convolution (64 -> 64 channels), HYBRID f16 + fp8 split operands ("f16f8").
//
// Each fp32 operand is scaled by an exact power of two (X = 2^8 x, W = 2^10 w) and written as
//     X = Xh + Xl,  Xh = f16(X)            (11 significant bits),  |Xl| <= 2^-11 |X|
//     W = Wh + Wl,  Wh = f16(W)
// and the product X W = Xh Wh + X Wl + Xl W (+ Xl Wl, dropped, 2^-22) is evaluated as
//     Xh Wh          on the f16 matrix core  (v_mfma_f32_32x32x16_f16, exact products, 32 cycles / K=16)
//     X  Wl + Xl W   on the fp8 matrix core  (v_mfma_scale_f32_32x32x64_f8f6f4 with OCP e4m3
//                    operands, 64 cycles / K=64 = twice the f16 rate; measured,
//                    scripts/probes/mfma_f8_probe.hip)
// The two cross terms are 2^-11 of the result, so 4-bit operands (e4m3) leave ~2^-15 of a product:
// the same error class as the bf16x3 kernel (hi/lo bf16 planes, conv_bf16.hip), measured 2.3x its
// error, for 2/3 of its matrix cycles (256 instead of 384 per 64-deep slice of a 32x32 tile).
// The power-of-two scalings that bring the residuals into the e4m3 range are undone by the
// instruction's E8M0 block scales, so all three products land in ONE fp32 accumulator.
//
// Geometry: 8 x 32 pixel tile x 64 output channels per 4-wave workgroup, two passes of 32 input
// channels (80.6 KB LDS -> 2 workgroups / CU), weights in MFMA-fragment order straight from L1/L2
// (no weight image in LDS, no barrier inside the tap loop).  fp32 NHWC in HBM on both sides.
// LDS image per pixel and pass: [32 f16 Xh | 32 e4m3 x | 32 e4m3 16 Xl | 16 B pad] = 144 B.
// The fp8 instruction is 64 deep: lane half 0 takes the 32 channels of tap 2p, lane half 1 those
// of tap 2p+1 (13 tap pairs; the 26th tap has zero weights).
#include <stdlib.h>
#include <string.h>

#include "common.h"

// timing experiments only (scripts/probes/conv_ablate.hip): 1 = no weight loads in the tap loop,
// 2 = no fp8 MFMAs, 3 = no f16 MFMAs, 4 = no LDS operand reads in the tap loop.  0 in the library.
#ifndef TOCVP_ABLATE
#define TOCVP_ABLATE 0
#endif

namespace {

constexpr int ABL = TOCVP_ABLATE;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32, IH = TH + 4, IW = TW + 4;
constexpr int C = 64, CCH = 32, NPASS = 2, NPAIR = 13;
constexpr int ROWB = 144;
constexpr int OFF_X8 = 64, OFF_L8 = 96;
constexpr float SA = 256.f, SW = 1024.f;      // operand pre-scales (exact)
constexpr float SL = 16.f;                    // residual scale into the e4m3 range
constexpr float SW8 = 8.f;                    // e4m3 copy of the weight holds 8 w = 2^-7 W
// E8M0 exponents (value 2^(e-127)) that map the stored fp8 numbers back to X / W units
constexpr int E_X8 = 127 + 8;                 // x   -> X
constexpr int E_L8 = 127 - 4;                 // 16 Xl -> Xl, 16 Wl -> Wl
constexpr int E_W8 = 127 + 7;                 // 2^-7 W -> W
constexpr float F8MAX = 448.f, F16MAX = 65504.f;

struct Args {
    const float* x; const float* aux; const unsigned char* wf16; const unsigned char* wf8;
    const float* bias; float* y;
    int nimg, H, W, relu;
};

__device__ __forceinline__ int border_class(int p, int n) {
    return p < 2 ? p : (p >= n - 2 ? 4 - (n - 1 - p) : 2);
}

__device__ __forceinline__ float clampf(float v, float m) { return __builtin_amdgcn_fmed3f(v, -m, m); }

// two floats -> two e4m3 bytes in the low / high half of a dword
__device__ __forceinline__ int pack4_e4m3(float a, float b, float c, float d) {
    int r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(clampf(a, F8MAX), clampf(b, F8MAX), r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(clampf(c, F8MAX), clampf(d, F8MAX), r, true);
    return r;
}

// fragment-order weight images (see split kernel below)
constexpr int F16_FRAG = 1024;                                  // 64 lanes x 16 B
constexpr int F8_FRAG = 2048;                                   // 64 lanes x 32 B
constexpr int F16_PER_PAIR = 2 * 2 * 2 * F16_FRAG;              // [tap in pair][ks][nb]
constexpr int F8_PER_PAIR = 2 * 2 * F8_FRAG;                    // [plane][nb]

template <int MODE>
__global__ __launch_bounds__(256, 2) void conv5x5_f16f8_kernel(Args p) {
    constexpr int NT = 256;
    constexpr int SS = C + 4;                                       // padded floats per staged pixel
    constexpr int LDS_BYTES = IH * IW * ROWB > 4 * 64 * SS * 4 ? IH * IW * ROWB : 4 * 64 * SS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    unsigned char* in_s = lds;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_x = p.W / TW, tiles = tiles_x * (p.H / TH);
    const int img = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // B fragments of one tap pair: f16 part [tap in pair][ks][nb], fp8 part [plane][nb]
    f16x8 bf[2][2][2];
    i32x8 b8[2][2];
    auto load_f16 = [&](int q) {                                  // q = pass * NPAIR + pair
        const unsigned char* base = p.wf16 + (size_t)q * F16_PER_PAIR + lane * 16;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    bf[tt][ks][n] = *reinterpret_cast<const f16x8*>(base + ((tt * 2 + ks) * 2 + n) * F16_FRAG);
    };
    auto load_f8 = [&](int q) {
        const unsigned char* base = p.wf8 + (size_t)q * F8_PER_PAIR + lane * 32;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(base + (pl * 2 + n) * F8_FRAG);
                const i32x4 hi = *reinterpret_cast<const i32x4*>(base + (pl * 2 + n) * F8_FRAG + 16);
                b8[pl][n] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
    };

    load_f16(0);
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass > 0) __syncthreads();          // every wave is done reading the previous image
        // ---- halo tile: fp32 -> (Xh f16 | x e4m3 | 16 Xl e4m3) in LDS.  All global loads are issued
        // back to back from clamped (always valid) addresses and only then converted.
        constexpr int NIT = (IH * IW * (CCH / 4) + NT - 1) / NT;
        // MODE 1 carries a second operand per element: two half batches keep it inside 256 VGPRs
        constexpr int NBATCH = MODE == 1 ? 2 : 1, BIT = NIT / NBATCH;
        static_assert(NIT % NBATCH == 0, "batches must divide the tile iterations");
#pragma unroll
        for (int bt = 0; bt < NBATCH; ++bt) {
            f32x4 tv[BIT];
            f32x4 ts[MODE == 1 ? BIT : 1];
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int i = min(t + (bt * BIT + it) * NT, IH * IW * (CCH / 4) - 1);
                const int pix = i / (CCH / 4), c = pass * CCH + (i % (CCH / 4)) * 4;
                const int iy = min(max(ty0 + pix / IW - 2, 0), p.H - 1);
                const int ix = min(max(tx0 + pix % IW - 2, 0), p.W - 1);
                if (MODE == 0) {
                    tv[it] = *reinterpret_cast<const f32x4*>(p.x + (((size_t)img * p.H + iy) * p.W + ix) * C + c);
                } else {
                    const int cls = border_class(iy, p.H) * 5 + border_class(ix, p.W);
                    tv[it] = *reinterpret_cast<const f32x4*>(p.x + ((size_t)iy * p.W + ix) * C + c);
                    ts[it] = *reinterpret_cast<const f32x4*>(p.aux + ((size_t)img * 25 + cls) * C + c);
                }
            }
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int i = t + (bt * BIT + it) * NT;
                if (i < IH * IW * (CCH / 4)) {
                    const int pix = i / (CCH / 4), c = (i % (CCH / 4)) * 4;   // channel inside the pass
                    const int iy = ty0 + pix / IW - 2, ix = tx0 + pix % IW - 2;
                    const bool inside = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    f32x4 v = tv[it];
                    if (MODE == 1) {
                        v += ts[it];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
                    }
                    f16x4 hi;
                    float rl[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        v[u] = inside ? v[u] : 0.f;
                        const float X = clampf(v[u] * SA, F16MAX);
                        hi[u] = (_Float16)X;
                        rl[u] = (X - (float)hi[u]) * SL;
                    }
                    unsigned char* dst = in_s + pix * ROWB;
                    *reinterpret_cast<f16x4*>(dst + c * 2) = hi;
                    *reinterpret_cast<int*>(dst + OFF_X8 + c) = pack4_e4m3(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<int*>(dst + OFF_L8 + c) = pack4_e4m3(rl[0], rl[1], rl[2], rl[3]);
                }
            }
        }
        __syncthreads();

        for (int pr = 0; pr < NPAIR; ++pr) {
            const int q = pass * NPAIR + pr;
            if (ABL != 1) load_f8(q);
            __builtin_amdgcn_sched_barrier(0);
            // ---- main term on the f16 cores: taps 2 pr and 2 pr + 1, two 16-channel k-steps each
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int tap = 2 * pr + tt;
                if (tap < 25) {
                    const int dy = ABL == 4 ? 0 : tap / 5, dx = ABL == 4 ? 0 : tap - 5 * dy;
                    const unsigned char* a_base = in_s + ((2 * wave + dy) * IW + l31 + dx) * ROWB + h * 16;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        f16x8 a[2];
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            a[m] = *reinterpret_cast<const f16x8*>(a_base + m * IW * ROWB + ks * 32);
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int n = 0; n < 2; ++n)
                                if (ABL != 3)
                                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], bf[tt][ks][n],
                                                                                       acc[m][n], 0, 0, 0);
                    }
                }
            }
            if (ABL != 1) load_f16(min(q + 1, NPASS * NPAIR - 1));
            __builtin_amdgcn_sched_barrier(0);
            // ---- cross terms on the fp8 cores: lane half h carries tap 2 pr + h (64 = 2 x 32 deep)
            {
                const int tap = ABL == 4 ? 0 : min(2 * pr + h, 24);  // 26th tap: zero weights, any pixel
                const int dy = (tap * 205) >> 10, dx = tap - 5 * dy;
                const unsigned char* a_base = in_s + ((2 * wave + dy) * IW + l31 + dx) * ROWB;
                i32x8 ax[2], al[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const i32x4 x0 = *reinterpret_cast<const i32x4*>(a_base + m * IW * ROWB + OFF_X8);
                    const i32x4 x1 = *reinterpret_cast<const i32x4*>(a_base + m * IW * ROWB + OFF_X8 + 16);
                    const i32x4 l0 = *reinterpret_cast<const i32x4*>(a_base + m * IW * ROWB + OFF_L8);
                    const i32x4 l1 = *reinterpret_cast<const i32x4*>(a_base + m * IW * ROWB + OFF_L8 + 16);
                    ax[m] = i32x8{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    al[m] = i32x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        if (ABL == 2) continue;
                        // plane 0 = 16 Wl, plane 1 = 2^-7 W
                        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                            ax[m], b8[0][n], acc[m][n], 0, 0, 0, E_X8, 0, E_L8);
                        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                            al[m], b8[1][n], acc[m][n], 0, 0, 0, E_L8, 0, E_W8);
                    }
            }
        }
    }
    __syncthreads();                            // the halo image is dead: reuse it as the store stage

    // Epilogue through LDS (as conv_bf16.hip): each wave stages its 64 pixels x 64 channels and
    // writes 1 KiB of contiguous NHWC output per instruction.
    float* stage = reinterpret_cast<float*>(lds) + wave * (64 * SS);
    constexpr float UNSCALE = 1.f / (SA * SW);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const float bv = p.bias[n * 32 + l31];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[m][n][r] * UNSCALE + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                stage[(m * 32 + acc_row(r, h)) * SS + n * 32 + l31] = v;
            }
    }
    __builtin_amdgcn_wave_barrier();
    float* ybase = p.y + (((size_t)img * p.H + ty0 + 2 * wave) * p.W + tx0) * C;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int idx = lane + 64 * it;
        const int px = idx >> 4, c4 = (idx & 15) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + c4);
        *reinterpret_cast<f32x4*>(ybase + ((size_t)(px >> 5) * p.W + (px & 31)) * C + c4) = v;
    }
}

// (64, 64, 5, 5) fp32 ->
//   wf16: [pass(2)][pair(13)][tap in pair(2)][ks(2)][nb(2)][lane(64)][8 f16]   Wh = f16(2^10 w)
//   wf8 : [pass(2)][pair(13)][plane(2)][nb(2)][lane(64)][32 e4m3]              16 Wl | 2^-7 W
// lane (c = l & 31, hh = l >> 5): output channel nb*32 + c; f16 fragment: input channels
// pass*32 + ks*16 + 8 hh + j of ONE tap; fp8 fragment: input channels pass*32 + j of tap 2 pair + hh.
__global__ __launch_bounds__(256) void split_conv_weights_f16f8_kernel(const float* __restrict__ w,
                                                                       _Float16* __restrict__ wf16,
                                                                       unsigned char* __restrict__ wf8) {
    const int i = blockIdx.x * 256 + threadIdx.x;                  // over 26 taps * 64 * 64
    if (i >= 26 * C * C) return;
    const int ci = i % C, co = (i / C) % C, tap = i / (C * C);
    const float Wv = tap < 25 ? clampf(w[((size_t)co * C + ci) * 25 + tap] * SW, F16MAX) : 0.f;
    const _Float16 hi = (_Float16)Wv;
    const float res = (Wv - (float)hi) * SL;
    const int pass = ci >> 5, cc = ci & 31, pair = tap >> 1, tt = tap & 1;
    const int nb = co >> 5, c = co & 31;
    {
        const int ks = cc >> 4, hh = (cc >> 3) & 1, j = cc & 7;
        const size_t frag = ((((size_t)pass * NPAIR + pair) * 2 + tt) * 2 + ks) * 2 + nb;
        wf16[(frag * 64 + hh * 32 + c) * 8 + j] = hi;
    }
    {
        const size_t frag0 = (((size_t)pass * NPAIR + pair) * 2 + 0) * 2 + nb;     // plane 0: 16 Wl
        const size_t frag1 = (((size_t)pass * NPAIR + pair) * 2 + 1) * 2 + nb;     // plane 1: 2^-7 W
        const int both = __builtin_amdgcn_cvt_pk_fp8_f32(clampf(res, F8MAX), clampf(Wv * (SW8 / SW), F8MAX), 0, false);
        wf8[(frag0 * 64 + tt * 32 + c) * 32 + cc] = (unsigned char)(both & 0xff);
        wf8[(frag1 * 64 + tt * 32 + c) * 32 + cc] = (unsigned char)((both >> 8) & 0xff);
    }
}

}  // namespace

extern "C" size_t tocvp_conv_weights_f16f8_bytes(int which) {
    return which == 0 ? (size_t)NPASS * NPAIR * F16_PER_PAIR : (size_t)NPASS * NPAIR * F8_PER_PAIR;
}

extern "C" int tocvp_split_conv_weights_f16f8(const float* w, void* wf16, void* wf8, int Cout, int Cin,
                                              void* stream) {
    TOCVP_CHECK_ARG(w && wf16 && wf8 && Cout == C && Cin == C);
    hipLaunchKernelGGL(split_conv_weights_f16f8_kernel, dim3((26 * C * C + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, static_cast<_Float16*>(wf16),
                       static_cast<unsigned char*>(wf8));
    return tocvp_launch_status();
}

extern "C" int tocvp_conv5x5_f16f8_f32(const float* x, const float* aux, int in_mode, const void* wf16,
                                       const void* wf8, const float* bias, float* y, int nimg, int H,
                                       int W, int Cin, int Cout, int relu, void* stream) {
    TOCVP_CHECK_ARG(x && wf16 && wf8 && bias && y);
    TOCVP_CHECK_ARG(in_mode == 0 || (in_mode == 1 && aux != nullptr));
    TOCVP_CHECK_ARG(Cin == C && Cout == C);
    TOCVP_CHECK_ARG(nimg >= 0 && H > 0 && W > 0 && (H % TH) == 0 && (W % TW) == 0);
    TOCVP_CHECK_ARG((size_t)nimg * (H / TH) * (W / TW) < 0x7fffffffu);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(wf16) || !tocvp_aligned16(wf8) || !tocvp_aligned16(y) ||
        (aux && !tocvp_aligned16(aux)))
        return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    Args a{x, aux, static_cast<const unsigned char*>(wf16), static_cast<const unsigned char*>(wf8), bias,
           y, nimg, H, W, relu};
    const dim3 grid((unsigned)((size_t)nimg * (H / TH) * (W / TW)));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (in_mode == 0)
        hipLaunchKernelGGL(conv5x5_f16f8_kernel<0>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(conv5x5_f16f8_kernel<1>, grid, dim3(256), 0, s, a);
    return tocvp_launch_status();
}
