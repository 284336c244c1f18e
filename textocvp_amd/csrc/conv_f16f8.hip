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
// Geometry: 8 x 64 pixel tile x 64 output channels per 4-wave workgroup; every wave owns 2 rows x 64
// pixels x 64 channels (4 x 2 accumulator tiles, 128 VGPRs), so a weight fragment fetched from L1/L2
// feeds FOUR pixel blocks: with 2 x 2 tiles the vector-memory path (64 B/clk/CU) was saturated by
// the weight stream before the matrix cores were (measured: -25 % time with the loads removed).
// Four passes of 16 input channels (LDS image 12 x 68 pixels x 80 B = 65 KB -> 2 workgroups / CU),
// weights in MFMA-fragment order straight from L1/L2 (no weight image in LDS, no barrier inside
// the tap loop).  fp32 NHWC in HBM on both sides.
// LDS image per pixel and pass: [16 f16 Xh | 16 e4m3 x | 16 e4m3 16 Xl | 16 B pad] = 80 B.
// The fp8 instruction is 64 deep = 4 taps x 16 channels: lane half h carries taps 4g+2h, 4g+2h+1
// (7 tap groups; taps 25..27 have zero weights).
#include <stdlib.h>
#include <string.h>

#include "common.h"

// timing experiments only (scripts/probes/conv_ablate.hip): 1 = no weight loads in the tap loop,
// 2 = no fp8 MFMAs, 3 = no f16 MFMAs, 4 = no LDS operand reads in the tap loop, 5 = halo tile staged
// for pass 0 only, 6 = no output stores.  0 in the library.
#ifndef TOCVP_ABLATE
#define TOCVP_ABLATE 0
#endif

namespace {

constexpr int ABL = TOCVP_ABLATE;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 64, IH = TH + 4, IW = TW + 4;
constexpr int C = 64, CCH = 16, NPASS = 4, NGRP = 7;
constexpr int ROWB = 80;
constexpr int OFF_X8 = 32, OFF_L8 = 48;
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
    int pm_in, pm_out;      // pass-major activation layout (n, 4, H, W, 16) instead of NHWC (n, H, W, 64)
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
constexpr int F16_PER_GRP = 4 * 2 * F16_FRAG;                   // [tap in group][nb]
constexpr int F8_PER_GRP = 2 * 2 * F8_FRAG;                     // [plane][nb]

__device__ __forceinline__ i32x8 cat8(const i32x4 a, const i32x4 b) {
    return i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void conv5x5_f16f8_kernel(Args p) {
    constexpr int NT = 256;
    constexpr int SS = C + 4;                                       // padded floats per staged pixel
    constexpr int STAGE_BYTES = 4 * 64 * SS * 4;                    // one 64-pixel row per wave
    constexpr int LDS_BYTES = IH * IW * ROWB > STAGE_BYTES ? IH * IW * ROWB : STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    unsigned char* in_s = lds;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_x = p.W / TW, tiles = tiles_x * (p.H / TH);
    const int img = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;

    // accumulator tile m = 2 * (row of the wave's pair) + (32-pixel half of the 64-pixel row)
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const char* const xbase = reinterpret_cast<const char*>(MODE == 0 ? p.x + (size_t)img * p.H * p.W * C : p.x);
    const char* const abase = reinterpret_cast<const char*>(MODE == 1 ? p.aux + (size_t)img * 25 * C : p.x);

    // B fragments of one tap group: f16 part [tap in group][nb], fp8 part [plane][nb]
    f16x8 bf[4][2];
    i32x8 b8[2][2];
    auto load_f16 = [&](int q) {                                  // q = pass * NGRP + group
        const unsigned char* base = p.wf16 + (size_t)q * F16_PER_GRP + lane * 16;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                bf[tt][n] = *reinterpret_cast<const f16x8*>(base + (tt * 2 + n) * F16_FRAG);
    };
    auto load_f8 = [&](int q) {
        const unsigned char* base = p.wf8 + (size_t)q * F8_PER_GRP + lane * 32;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                b8[pl][n] = cat8(*reinterpret_cast<const i32x4*>(base + (pl * 2 + n) * F8_FRAG),
                                 *reinterpret_cast<const i32x4*>(base + (pl * 2 + n) * F8_FRAG + 16));
    };
    // byte offset of the wave's accumulator tile m inside the halo image (tap (0,0), pixel l31)
    auto tile_off = [&](int m) { return ((2 * wave + (m >> 1)) * IW + (m & 1) * 32 + l31) * ROWB; };

    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass > 0) __syncthreads();          // every wave is done reading the previous image
        // ---- halo tile: fp32 -> (Xh f16 | x e4m3 | 16 Xl e4m3) in LDS.  All global loads of a batch
        // are issued back to back from clamped (always valid) addresses and only then converted.
        // The staging addresses depend on the pass only through a constant; hipcc would hoist all of
        // them (and the LDS store addresses) out of the pass loop and spill them next to the 128
        // accumulator registers.  An opaque copy of the thread index keeps them per-pass temporaries.
        int tq = t;
        asm volatile("" : "+v"(tq));
        constexpr int ITEMS = IH * IW * (CCH / 4);
        // every batch exposes one global-memory latency: as few batches as the register file allows
        constexpr int NBATCH = MODE == 1 ? 2 : 1, BIT = (ITEMS + NBATCH * NT - 1) / (NBATCH * NT);
#pragma unroll
        for (int bt = 0; bt < NBATCH; ++bt) {
            if (ABL == 5 && pass > 0) break;
            f32x4 tv[BIT];
            f32x4 ts[MODE == 1 ? BIT : 1];
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int i = min(tq + (bt * BIT + it) * NT, ITEMS - 1);
                const int pix = i / (CCH / 4), c = pass * CCH + (i % (CCH / 4)) * 4;
                const int iy = min(max(ty0 + pix / IW - 2, 0), p.H - 1);
                const int ix = min(max(tx0 + pix % IW - 2, 0), p.W - 1);
                // uniform 64-bit base + 32-bit lane offset: half the address registers of per-lane pointers
                const unsigned off = p.pm_in
                    ? (unsigned)(((pass * p.H + iy) * p.W + ix) * CCH + (c - pass * CCH)) * 4u
                    : (unsigned)((iy * p.W + ix) * C + c) * 4u;
                tv[it] = *reinterpret_cast<const f32x4*>(xbase + off);
                if (MODE == 1) {
                    const int cls = border_class(iy, p.H) * 5 + border_class(ix, p.W);
                    ts[it] = *reinterpret_cast<const f32x4*>(abase + (unsigned)(cls * C + c) * 4u);
                }
            }
#pragma unroll
            for (int it = 0; it < BIT; ++it) {
                const int i = tq + (bt * BIT + it) * NT;
                if (i < ITEMS) {
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
        load_f16(pass * NGRP);                   // first group's weights fly across the barrier
        __syncthreads();

        for (int g = 0; g < NGRP; ++g) {
            const int q = pass * NGRP + g;
            if (ABL != 1) load_f8(q);
            __builtin_amdgcn_sched_barrier(0);
            // ---- main term on the f16 cores: taps 4g .. 4g+3, one 16-channel k-step each
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int tap = 4 * g + tt;
                if (tap < 25) {
                    const int dy = ABL == 4 ? 0 : tap / 5, dx = ABL == 4 ? 0 : tap - 5 * dy;
                    const unsigned char* a_base = in_s + (dy * IW + dx) * ROWB + h * 16;
                    f16x8 a[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        a[m] = *reinterpret_cast<const f16x8*>(a_base + tile_off(m));
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            if (ABL != 3)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], bf[tt][n], acc[m][n],
                                                                                   0, 0, 0);
                }
            }
            if (ABL != 1) load_f16(min(q + 1, pass * NGRP + NGRP - 1));
            __builtin_amdgcn_sched_barrier(0);
            // ---- cross terms on the fp8 cores: lane half h carries taps 4g+2h and 4g+2h+1
            {
                const int ta = ABL == 4 ? 0 : min(4 * g + 2 * h, 24);      // taps > 24: zero weights
                const int tb = ABL == 4 ? 0 : min(4 * g + 2 * h + 1, 24);
                const int dya = (ta * 205) >> 10, dxa = ta - 5 * dya;
                const int dyb = (tb * 205) >> 10, dxb = tb - 5 * dyb;
                const unsigned char* pa = in_s + (dya * IW + dxa) * ROWB;
                const unsigned char* pb = in_s + (dyb * IW + dxb) * ROWB;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    // pl 0: x (e4m3) against 16 Wl;  pl 1: 16 Xl against 2^-7 W
                    const int off = pl == 0 ? OFF_X8 : OFF_L8;
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {               // two accumulator rows at a time
                        i32x8 a[2];
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            a[m] = cat8(*reinterpret_cast<const i32x4*>(pa + tile_off(2 * mh + m) + off),
                                        *reinterpret_cast<const i32x4*>(pb + tile_off(2 * mh + m) + off));
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int n = 0; n < 2; ++n) {
                                if (ABL == 2) continue;
                                acc[2 * mh + m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                                    a[m], b8[pl][n], acc[2 * mh + m][n], 0, 0, 0, pl == 0 ? E_X8 : E_L8, 0,
                                    pl == 0 ? E_L8 : E_W8);
                            }
                    }
                }
            }
        }
    }
    __syncthreads();                            // the halo image is dead: reuse it as the store stage

    // Epilogue through LDS, one 64-pixel output row of the wave at a time: the accumulator layout
    // gives a lane one channel of 16 pixels; staged, every store instruction writes 1 KiB of
    // contiguous NHWC output (4 pixels x 256 B).
    float* stage = reinterpret_cast<float*>(lds) + wave * (64 * SS);
    constexpr float UNSCALE = 1.f / (SA * SW);
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float bv = p.bias[n * 32 + l31];
#pragma unroll
            for (int xh = 0; xh < 2; ++xh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[2 * r2 + xh][n][r] * UNSCALE + bv;
                    if (p.relu) v = fmaxf(v, 0.f);
                    stage[(xh * 32 + acc_row(r, h)) * SS + n * 32 + l31] = v;
                }
        }
        __builtin_amdgcn_wave_barrier();
        const int oy = ty0 + 2 * wave + r2;
        if (p.pm_out) {
            // plane-major: instruction `it` writes 16 pixels x 64 B of channel plane it >> 2 (1 KiB contiguous)
            float* ybase = p.y + (size_t)img * p.H * p.W * C;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int plane = it >> 2, px = (it & 3) * 16 + (lane >> 2), cq = (lane & 3) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + plane * CCH + cq);
                if (ABL != 6 || v[0] == 12345.f)
                    *reinterpret_cast<f32x4*>(ybase + (((size_t)plane * p.H + oy) * p.W + tx0 + px) * CCH + cq) = v;
            }
        } else {
            float* yrow = p.y + (((size_t)img * p.H + oy) * p.W + tx0) * C;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int idx = lane + 64 * it;
                const int px = idx >> 4, c4 = (idx & 15) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + c4);
                if (ABL != 6 || v[0] == 12345.f) *reinterpret_cast<f32x4*>(yrow + (size_t)px * C + c4) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// (64, 64, 5, 5) fp32 ->
//   wf16: [pass(4)][group(7)][tap in group(4)][nb(2)][lane(64)][8 f16]       Wh = f16(2^10 w)
//   wf8 : [pass(4)][group(7)][plane(2)][nb(2)][lane(64)][32 e4m3]             16 Wl | 2^-7 W
// lane (c = l & 31, hh = l >> 5): output channel nb*32 + c; f16 fragment: input channels
// pass*16 + 8 hh + j of ONE tap; fp8 fragment: bytes 0..15 = input channels pass*16 + j of tap
// 4 group + 2 hh, bytes 16..31 = the same channels of tap 4 group + 2 hh + 1.  Taps 25..27 are zero.
__global__ __launch_bounds__(256) void split_conv_weights_f16f8_kernel(const float* __restrict__ w,
                                                                       _Float16* __restrict__ wf16,
                                                                       unsigned char* __restrict__ wf8) {
    const int i = blockIdx.x * 256 + threadIdx.x;                  // over 28 taps * 64 * 64
    if (i >= 4 * NGRP * C * C) return;
    const int ci = i % C, co = (i / C) % C, tap = i / (C * C);
    const float Wv = tap < 25 ? clampf(w[((size_t)co * C + ci) * 25 + tap] * SW, F16MAX) : 0.f;
    const _Float16 hi = (_Float16)Wv;
    const float res = (Wv - (float)hi) * SL;
    const int pass = ci / CCH, cc = ci % CCH, grp = tap >> 2, tt = tap & 3;
    const int nb = co >> 5, c = co & 31;
    {
        const int hh = cc >> 3, j = cc & 7;
        const size_t frag = (((size_t)pass * NGRP + grp) * 4 + tt) * 2 + nb;
        wf16[(frag * 64 + hh * 32 + c) * 8 + j] = hi;
    }
    {
        const int hh = tt >> 1, byte = (tt & 1) * 16 + cc;
        const size_t frag0 = (((size_t)pass * NGRP + grp) * 2 + 0) * 2 + nb;      // plane 0: 16 Wl
        const size_t frag1 = (((size_t)pass * NGRP + grp) * 2 + 1) * 2 + nb;      // plane 1: 2^-7 W
        const int both = __builtin_amdgcn_cvt_pk_fp8_f32(clampf(res, F8MAX), clampf(Wv * (SW8 / SW), F8MAX), 0, false);
        wf8[(frag0 * 64 + hh * 32 + c) * 32 + byte] = (unsigned char)(both & 0xff);
        wf8[(frag1 * 64 + hh * 32 + c) * 32 + byte] = (unsigned char)((both >> 8) & 0xff);
    }
}

}  // namespace

extern "C" size_t tocvp_conv_weights_f16f8_bytes(int which) {
    return which == 0 ? (size_t)NPASS * NGRP * F16_PER_GRP : (size_t)NPASS * NGRP * F8_PER_GRP;
}

extern "C" int tocvp_split_conv_weights_f16f8(const float* w, void* wf16, void* wf8, int Cout, int Cin,
                                              void* stream) {
    TOCVP_CHECK_ARG(w && wf16 && wf8 && Cout == C && Cin == C);
    hipLaunchKernelGGL(split_conv_weights_f16f8_kernel, dim3((4 * NGRP * C * C + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, static_cast<_Float16*>(wf16),
                       static_cast<unsigned char*>(wf8));
    return tocvp_launch_status();
}

extern "C" int tocvp_conv5x5_f16f8_f32(const float* x, const float* aux, int in_mode, const void* wf16,
                                       const void* wf8, const float* bias, float* y, int nimg, int H,
                                       int W, int Cin, int Cout, int relu, int layout, void* stream) {
    TOCVP_CHECK_ARG(layout >= 0 && layout <= 3 && !(in_mode == 1 && (layout & 1)));
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
           y, nimg, H, W, relu, layout & 1, (layout >> 1) & 1};
    const dim3 grid((unsigned)((size_t)nimg * (H / TH) * (W / TW)));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (in_mode == 0)
        hipLaunchKernelGGL(conv5x5_f16f8_kernel<0>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(conv5x5_f16f8_kernel<1>, grid, dim3(256), 0, s, a);
    return tocvp_launch_status();
}
