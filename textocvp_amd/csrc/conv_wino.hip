// Decoder 5x5 convolution (64 -> 64 channels) as a vertical Winograd F(4, 5) x five direct horizontal taps,
// SPLIT-fp16 operands ("f16x3"), fp32-class results.
//
// Replaces nn.Conv2d(64, 64, 5, padding=2) + ReLU of the reference's ConvDecoder
// (models/EncodersDecoders/decoders.py:96-110) for layers 1..3 of the spatial-broadcast decoder -- the same
// operator as conv_f16x3.hip with 2.5 x fewer matrix products:
//
//     y[4t + a, x, o] = sum_xi AT[a][xi] * M_xi[t, x, o]                                  (a = 0..3, xi = 0..7)
//     M_xi[t, x, o]   = sum_dx sum_c V_xi[t, x + dx - 2, c] * U_xi[dx][c, o]              (five taps instead of 25)
//     V_xi[t, x, c]   = sum_i BT[xi][i] * in[4t - 2 + i, x, c]                            (i = 0..7)
//     U_xi[dx][c, o]  = sum_k G[xi][k] * w[o, c, k, dx]                                   (k = 0..4, host side, fp64)
//
// (Cook-Toom over the points 0, +-1, +-2, +-1/2, infinity.)  Eight transform rows and five taps per four output rows =
// 40 products per 4 pixels against 100.  The transforms run in fp32 on the vector ALU; only the products are split:
// V (scaled by 16: |BT| row sums <= 15, so |in| < 255 keeps 16 V inside fp16) and U (scaled per transform row by the
// power of two the HOST picks from the weights at hand, U * s < 2^14) as two fp16 planes each, three
// v_mfma_f32_32x32x16_f16 per product as in conv_f16x3.hip.  scripts/probes/winograd_numerics.py (CPU simulation of
// exactly this arithmetic incl. the matrix core's subnormal flush, fp64 truth): last hidden activation 4.1e-7 of its
// maximum against 3.1e-7 for the direct split-fp16 form and 2.0e-6 for torch's fp32 convolution.
//
// Geometry: 4 output rows (one 4-row group: 8 input rows) x 64 columns x 64 output channels per 256-thread workgroup,
// TWO workgroups per CU.  Wave w owns the transform rows xi = 2 w, 2 w + 1: accumulators [j][32-pixel half][32-channel
// half] = 8 tiles = 128 registers.  Four passes of 16 input channels: every thread loads a column of 8 input rows
// (4 channels) -- one matrix step ahead of the arithmetic: the input is the previous launch's output, HBM latency --
// transforms it, splits the 8 results and writes them into the LDS image [xi 8][x 68][Vh | Vl] (34 KB, 16-byte chunks
// XOR-swizzled by the column so that operand reads are conflict-free at every tap shift without padding); then 10 steps
// (j, dx) of 12 MFMAs per wave: operand fragments in ONE register set (the fragments of a pixel half for step s + 1 are
// read right behind its six MFMAs of step s), weight fragments straight from L2 in fragment order, four register slots,
// three steps ahead.  While one workgroup of the CU transforms, exchanges or stores, the other one's waves keep the
// matrix pipes busy (the structure of conv_f16x3.hip; an 8-wave form that builds this overlap by hand inside one
// workgroup measured the same: scripts/probes/retired/conv_wino_wg8.txt).
// After the last pass the eight M_xi meet through LDS (two rounds of 64 KB, one per pixel half): wave w reads register
// quad w of all eight rows and both channel halves and forms a 4-row x 8-column x 64-channel block with AT scaled by
// the inverse operand scales and the output buffer's scale (kernel arguments) on top of the bias, applies the ReLU and
// stores (fp32 NHWC, fp32 x 16 pass-major for the
// next layer of this kernel, fp16 operand planes for conv_f16x3.hip, or -- last hidden layer -- the 36 tap products of
// the folded decoder tail).
// Where the time goes (scripts/probes/wino_stamp.hip, TOCVP_WINO_ABLATE): profiles/r05_wino.md.
//
// Built without packed-f32 vector instructions (flags line below): next to another wave's MFMAs a v_pk_fma_f32 costs ~30
// issue cycles against 2 x 4 for the two v_fma_f32 it replaces (MI355X_MICROARCH: packed f32 VALU "an anti-lever beside
// MFMAs"), and a workgroup's transform runs beside the other workgroup's multiply on the same SIMDs.  (The host
// half of the compilation does not know the feature and says so; harmless.)
// TOCVP_HIPCC_FLAGS: -Xclang -target-feature -Xclang -packed-fp32-ops
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int TW = 64, IW = TW + 4;
constexpr int NXI = 8;                              // transform rows
constexpr int C = 64, CCH = 16, NPASS = 4, NDX = 5, NSTEP = NPASS * NDX;
constexpr int OFF_LO = 32;
constexpr int FRAG = 1024, STEP_BYTES = 4 * FRAG;   // [plane(h, l)][nb(2)] fragments of one (xi, pass, dx)
constexpr float VS = 16.f;                          // scale of the transformed operand (and of a "x 16" activation buffer)
constexpr float F16MAX = 65504.f;

struct WArgs {
    const float* x; const float* aux; const unsigned char* wf; const float* bias; float* y;
    int nimg, H, W, relu;
    int out_mode;                   // 0 fp32 NHWC; 1 fp32 x 16, pass-major (n, 4, H, W, 16); 2 fp16 operand planes of 2^8 y, pass-major
    const unsigned char* tail_wf;   // TAILP: tap matrix of the folded decoder tail (conv_f16x3.hip: pack_tail_taps_kernel)
    float coef[4 * NXI];            // oscale * AT[a][xi] / (VS * s_xi)
    float oscale;                   // scale of the output buffer: 16 (out_mode 1), 256 (operand planes, folded tail), 1 (NHWC)
    const float* in_amax;           // MODE 2, nullable: device word holding max |x| -- the kernel scales the input by the largest power
                                    // of two that keeps 15 |x| inside fp16 instead of by 16 (data gradients: any magnitude)
    const float* gate;              // out_mode 0, nullable: NHWC array, the output is zeroed where gate <= 0 (ReLU of the layer below)
};

#ifndef TOCVP_WINO_ABLATE
#define TOCVP_WINO_ABLATE 0     // timing experiments only (scripts/probes/wino_stamp.hip): 1 weight fragments always from the same
                                // 4 KB (L1 hits instead of the L2 stream), 2 no exchange / epilogue, 3 no transform, 4 no MFMAs
#endif

// -DTOCVP_WINO_STAMP (scripts/probes/wino_stamp.hip): s_memtime at the phase boundaries of waves 0 and 3, summed per workgroup
#ifdef TOCVP_WINO_STAMP
__device__ unsigned long long tocvp_wino_stamps[16384 * 2 * 16];
#define WINO_T(i)                                                     \
    do {                                                              \
        __builtin_amdgcn_sched_barrier(0);                            \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        stamp_acc[i] += now_ - stamp_last;                            \
        stamp_last = now_;                                            \
        __builtin_amdgcn_sched_barrier(0);                            \
    } while (0)
#else
#define WINO_T(i) do {} while (0)
#endif

__device__ __forceinline__ int border_class(int p, int n) {
    return p < 2 ? p : (p >= n - 2 ? 4 - (n - 1 - p) : 2);
}

__device__ __forceinline__ float clampf(float v, float m) { return __builtin_amdgcn_fmed3f(v, -m, m); }

// Two values -> their fp16 planes (hi = f16(clamp x), lo = f16(x - hi), packed pairs) in five instructions: the compiler's
// form of the same arithmetic takes 8.5 (it converts hi twice and back once).  The mixed-precision FMA subtracts in fp32 and
// rounds once: bit-identical to (_Float16)(X - (float)hi) (checked on 128 values incl. saturating and tiny ones).
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    tocvp_split2_f16(clampf(x0, F16MAX), clampf(x1, F16MAX), hi, lo);
}

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    if (TOCVP_WINO_ABLATE == 4) {               // keeps the operands alive, issues no matrix instruction
        asm volatile("" :: "v"(a), "v"(b));
        return c;
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// Epilogue of one wave's output block: y[n][4 a + e] = row oy0 + a, column ox + 4 h + e, channel 32 n + l31 ("pixel slot"
// s = 8 a + 4 h + e: 4 rows x 8 columns) -- bias / ReLU, then fp32 NHWC, fp32 x 16 pass-major, fp16 operand planes (through a
// wave-private LDS stage: 16-byte stores) or, TAILP, the 36 tap products of the folded decoder tail.
template <bool TAILP>
__device__ __forceinline__ void wino_epilogue(const WArgs& p, f32x16 (&y)[2], unsigned char* lds, int wave, int lane, int img,
                                              int oy0, int ox) {
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int SS = C + 4;
    if constexpr (TAILP) {
        // the decoder tail folded in, as in conv_f16x3.hip (TAILP): 36 tap products per pixel leave the chip
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4* lp4;
        constexpr int TIMG = 64 * 64;
        constexpr int TPS = 36;
        constexpr float SA8 = TOCVP_F16X3_ACT_SCALE, SW10 = TOCVP_F16X3_WEIGHT_SCALE;
        unsigned char* timg = lds + wave * (2 * TIMG + 36 * TPS * 4);
        float* pst = reinterpret_cast<float*>(timg + 2 * TIMG);
        const int i16 = lane & 15, c16 = ((lane >> 4) & 1) * 16;
        const unsigned char* trd = timg + (8 * h + (i16 >> 2)) * 64 + (c16 + 4 * (i16 & 3)) * 2;
        const f16x8* twf = reinterpret_cast<const f16x8*>(p.tail_wf) + lane;
        float* pout = p.y + (size_t)img * 36 * p.H * p.W;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {                           // y carries bias and the planes' 2^8 already
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = p.relu ? fmaxf(y[n][4 * g + e], 0.f) : y[n][4 * g + e];
                unsigned h0, l0, h1, l1;
                split2(v[0], v[1], h0, l0);
                split2(v[2], v[3], h1, l1);
                unsigned char* dd = timg + (n * 32 + l31) * 64 + (8 * g + 4 * h) * 2;
                *reinterpret_cast<u32x2*>(dd) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(dd + TIMG) = u32x2{l0, l1};
            }
        }
        __builtin_amdgcn_wave_barrier();
        f32x16 pacc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[nb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            union { s16x4 s[2]; f16x8 f; } ah, al;
            ah.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(trd + ks * 16 * 64));
            ah.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(trd + ks * 16 * 64 + 4 * 64));
            al.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(trd + TIMG + ks * 16 * 64));
            al.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(trd + TIMG + ks * 16 * 64 + 4 * 64));
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f16x8 bh = twf[((nb * 4 + ks) * 2 + 0) * 64], bl = twf[((nb * 4 + ks) * 2 + 1) * 64];
                pacc[nb] = mfma16(al.f, bh, pacc[nb]);
                pacc[nb] = mfma16(ah.f, bl, pacc[nb]);
                pacc[nb] = mfma16(ah.f, bh, pacc[nb]);
            }
        }
        constexpr float UNS = 1.f / (SA8 * SW10);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int to = nb * 32 + l31;
            if (to < 36) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(pst + to * TPS + 8 * g + 4 * h) =
                        f32x4{pacc[nb][4 * g] * UNS, pacc[nb][4 * g + 1] * UNS,
                              pacc[nb][4 * g + 2] * UNS, pacc[nb][4 * g + 3] * UNS};
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 5; ++it) {                          // 36 rows x 8 float4 = 288 pieces
            const int idx = lane + 64 * it;
            if (idx < 36 * 8) {
                const int to = idx >> 3, c4 = (idx & 7) * 4;
                *reinterpret_cast<f32x4*>(pout + ((size_t)to * p.H + oy0 + (c4 >> 3)) * p.W + ox + (c4 & 7)) =
                    *reinterpret_cast<const f32x4*>(pst + to * TPS + c4);
            }
        }
    } else {
        // through a wave-private LDS stage: 16-byte stores, 32 pixel slots (4 rows x 8 columns) x 64 channels
        float* stage = reinterpret_cast<float*>(lds) + wave * (32 * SS);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)                            // y carries bias and the output buffer's scale already
                stage[acc_row(r, h) * SS + n * 32 + l31] = p.relu ? fmaxf(y[n][r], 0.f) : y[n][r];
        __builtin_amdgcn_wave_barrier();
        if (p.out_mode == 2) {
            unsigned char* ybase = reinterpret_cast<unsigned char*>(p.y + (size_t)img * p.H * p.W * C);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int plane = it >> 1, px = (it & 1) * 16 + (lane >> 2), cq = (lane & 3) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + plane * CCH + cq);
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                unsigned h0, l0, h1, l1;
                split2(v[0], v[1], h0, l0);
                split2(v[2], v[3], h1, l1);
                unsigned char* blk = ybase + (((size_t)plane * p.H + oy0 + (px >> 3)) * p.W + ox + (px & 7)) * 64 + cq * 2;
                *reinterpret_cast<u32x2*>(blk) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(blk + OFF_LO) = u32x2{l0, l1};
            }
        } else if (p.out_mode == 1) {
            float* ybase = p.y + (size_t)img * p.H * p.W * C;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int plane = it >> 1, px = (it & 1) * 16 + (lane >> 2), cq = (lane & 3) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + plane * CCH + cq);
                *reinterpret_cast<f32x4*>(ybase + (((size_t)plane * p.H + oy0 + (px >> 3)) * p.W + ox + (px & 7)) * CCH + cq) = v;
            }
        } else {
            float* ybase = p.y + (size_t)img * p.H * p.W * C;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = lane + 64 * it;
                const int px = idx >> 4, c4 = (idx & 15) * 4;
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + c4);
                const size_t o = (((size_t)(oy0 + (px >> 3))) * p.W + ox + (px & 7)) * C + c4;
                if (p.gate) {
                    const f32x4 gt = *reinterpret_cast<const f32x4*>(p.gate + (size_t)img * p.H * p.W * C + o);
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = gt[u] > 0.f ? v[u] : 0.f;
                }
                *reinterpret_cast<f32x4*>(ybase + o) = v;
            }
        }
    }
}

// MODE 0: fp32 x 16 pass-major input (written by this kernel, out_mode 1); 1: the collapsed first layer
// (relu(cpos[y, x] + aux[img][border class]), conv_f16x3.hip MODE 1); 2: fp32 NHWC input.
constexpr int W4_TH = 4, W4_THREADS = 256;
constexpr int W4_IMG = NXI * IW * 64;                             // 34816
constexpr int W4_XCH = NXI * 2 * 4 * 64 * 16;                     // 65536: [xi][n 2][quad 4][lane 64][4 floats]
constexpr int W4_LDS = W4_XCH > W4_IMG ? W4_XCH : W4_IMG;

template <int MODE, bool TAILP>
__global__ __launch_bounds__(W4_THREADS, 2) void conv5x5_wino_f16x3_kernel(WArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[W4_LDS];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_x = p.W / TW, tiles = tiles_x * (p.H / W4_TH);
    const int img = (blockIdx.x / (8 * tiles)) * 8 + (blockIdx.x & 7), tile = (blockIdx.x >> 3) % tiles;
    if (img >= p.nimg) return;
    const int ty0 = (tile / tiles_x) * W4_TH, tx0 = (tile % tiles_x) * TW;
    const bool wide = p.W != TW;
    // scale of the (MODE 2) input: 16, or the power of two <= 4096 / max |x| (|BT| row sums <= 15: 15 * 4096 < 65504)
    float in_s = VS;
    if (MODE == 2 && p.in_amax) {
        const float r = 4096.f / fmaxf(*p.in_amax, 1.0e-30f);
        in_s = __uint_as_float(__float_as_uint(r) & 0x7f800000u);
    }
    const float csc = VS / in_s;                // the coefficients were built for a x 16 input

    f32x16 acc[2][2][2];                        // [j: xi = 2 wave + j][32-pixel half][32-channel half]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int xh = 0; xh < 2; ++xh)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][xh][n][r] = 0.f;

    const char* const xbase = reinterpret_cast<const char*>(MODE == 1 ? p.x : p.x + (size_t)img * p.H * p.W * C);
    const char* const abase = reinterpret_cast<const char*>(MODE == 1 ? p.aux + (size_t)img * 25 * C : p.x);

    int o_dx[NDX];                              // swizzled operand offsets, as in the 8-wave form
#pragma unroll
    for (int dx = 0; dx < NDX; ++dx) {
        const int x = l31 + dx;
        o_dx[dx] = (x << 6) | (((h ^ (x >> 2)) & 3) << 4);
    }

    // weights: [pair w][pass][j][dx][plane][nb][lane] 16 B -- this wave's 40 steps are one contiguous stream
    const unsigned char* wptr = p.wf + (size_t)wave * (2 * NSTEP) * STEP_BYTES + lane * 16;
    f16x8 w4[4][2][2];
    auto load_w = [&](int slot) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                w4[slot][pl][n] = *reinterpret_cast<const f16x8*>(wptr + (pl * 2 + n) * FRAG);
        if (TOCVP_WINO_ABLATE != 1) wptr += STEP_BYTES;
    };
    load_w(0);
    load_w(1);
    load_w(2);

    constexpr bool PREFETCH = true;             // (MODE 1: the position-table rows; its tap sums are L1 hits, loaded with the arithmetic)
    auto t_load = [&](int pass, int rep, f32x4 (&d)[8], f32x4 (&ts)[MODE == 1 ? 8 : 1]) {
        int tt = t;
        asm volatile("" : "+v"(tt));
        const int cq = tt & 3, col = tt >> 2;
        const int xi = rep ? (col < 2 ? col : 64 + col) : 2 + col;
        const int ixc = min(max(tx0 + xi - 2, 0), p.W - 1);
        const int pc = min(pass, NPASS - 1);
        const int c = pc * CCH + cq * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int iyc = min(max(ty0 - 2 + i, 0), p.H - 1);
            unsigned off;
            if (MODE == 0) off = (unsigned)(((pc * p.H + iyc) * p.W + ixc) * CCH + cq * 4) * 4u;
            else off = (unsigned)((iyc * p.W + ixc) * C + c) * 4u;
            d[i] = *reinterpret_cast<const f32x4*>(xbase + off);
        }
    };
    auto t_store = [&](int pass, int rep, f32x4 (&d)[8], f32x4 (&ts)[MODE == 1 ? 8 : 1]) {
        int tt = t;
        asm volatile("" : "+v"(tt));
        const int cq = tt & 3, col = tt >> 2;
        const int xi = rep ? (col < 2 ? col : 64 + col) : 2 + col;
        const int ix = tx0 + xi - 2;
        const bool xin = ix >= 0 && ix < p.W;
        if (MODE == 1) {
            const int ixc = min(max(ix, 0), p.W - 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int iyc = min(max(ty0 - 2 + i, 0), p.H - 1);
                const int cls = border_class(iyc, p.H) * 5 + border_class(ixc, p.W);
                ts[i] = *reinterpret_cast<const f32x4*>(abase + (unsigned)(cls * C + pass * CCH + cq * 4) * 4u);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 1) {
                d[i] += ts[i];
#pragma unroll
                for (int u = 0; u < 4; ++u) d[i][u] = fmaxf(d[i][u], 0.f) * VS;
            } else if (MODE == 2) {
                d[i] *= in_s;
            }
            // zero padding: only the first and last two rows of the 8 can lie above / below the image; columns beside it
            // exist only for the halo columns of wide images (rep 1)
            if (i < 2 || i >= 6 || rep) {
                const int iy = ty0 - 2 + i;
                const bool inside = (xin || !rep) && iy >= 0 && iy < p.H;
                if (!inside) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        f32x4 v[8];
        v[0] = (d[0] - d[6]) + 5.25f * (d[4] - d[2]);
        v[7] = (d[7] - d[1]) + 5.25f * (d[3] - d[5]);
        {
            const f32x4 e = (d[2] + d[6]) - 4.25f * d[4], o = (d[1] + d[5]) - 4.25f * d[3];
            v[1] = e + o;
            v[2] = e - o;
        }
        {
            const f32x4 e = (0.25f * d[2] + d[6]) - 1.25f * d[4], o = (0.5f * d[1] + 2.f * d[5]) - 2.5f * d[3];
            v[3] = e + o;
            v[4] = e - o;
        }
        {
            const f32x4 e = (4.f * d[2] + d[6]) - 5.f * d[4], o = (2.f * d[1] + 0.5f * d[5]) - 2.5f * d[3];
            v[5] = e + o;
            v[6] = e - o;
        }
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const int off = (xi << 6) + ((((cq >> 1) ^ (xi >> 2)) & 3) << 4) + (cq & 1) * 8;      // Vh; Vl at off ^ 32 (rows are 4352 B apart)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            unsigned h0, l0, h1, l1;
            split2(v[q][0], v[q][1], h0, l0);
            split2(v[q][2], v[q][3], h1, l1);
            *reinterpret_cast<u32x2*>(lds + off + q * IW * 64) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(lds + (off ^ 32) + q * IW * 64) = u32x2{l0, l1};
        }
    };
    f32x4 dpre[8];
    f32x4 tsx[MODE == 1 ? 8 : 1];
    auto transform = [&](int pass, bool loaded) {
        if (TOCVP_WINO_ABLATE == 3) return;
        if (!PREFETCH || !loaded) t_load(pass, 0, dpre, tsx);
        t_store(pass, 0, dpre, tsx);
        if (wide && t < 16) {
            t_load(pass, 1, dpre, tsx);
            t_store(pass, 1, dpre, tsx);
        }
    };

    // ---- 10 steps (j, dx) of 12 MFMAs; FIRST = slot of the pass's first step (0 for even passes, 2 for odd ones)
    auto multiply = [&](auto first_tag, bool last_pass, int prefetch_pass) {
        constexpr int FIRST = decltype(first_tag)::value;
        int wofs = wave * (2 * IW * 64);
        asm volatile("" : "+s"(wofs));
        f16x8 fa[2][2];                                             // [32-pixel half][plane], rolling (one set)
        auto read_half = [&](int xh, int st) {                      // st = 5 j + dx
            const int j = st / NDX, dx = st % NDX;
            const int toff = j * IW * 64 + xh * 32 * 64;
            fa[xh][0] = *reinterpret_cast<const f16x8*>(lds + (o_dx[dx] + wofs) + toff);
            fa[xh][1] = *reinterpret_cast<const f16x8*>(lds + ((o_dx[dx] ^ 32) + wofs) + toff);
        };
        read_half(0, 0);
        read_half(1, 0);
#pragma unroll
        for (int st = 0; st < 2 * NDX; ++st) {
            const int j = st / NDX, sl = (FIRST + st) & 3;
            const bool lastst = st == 2 * NDX - 1;
            // fragments of the step three ahead go into the slot the previous step consumed
            if (!(last_pass && st + 3 >= 2 * NDX)) load_w((sl + 3) & 3);
            if (PREFETCH && lastst && TOCVP_WINO_ABLATE != 3) t_load(prefetch_pass, 0, dpre, tsx);
#pragma unroll
            for (int xh = 0; xh < 2; ++xh) {
                // (the two channel halves alternate: no MFMA reads the accumulator the previous one is still writing)
                acc[j][xh][0] = mfma16(fa[xh][1], w4[sl][0][0], acc[j][xh][0]);      // Vl Uh0
                acc[j][xh][1] = mfma16(fa[xh][1], w4[sl][0][1], acc[j][xh][1]);      // Vl Uh1
                acc[j][xh][0] = mfma16(fa[xh][0], w4[sl][1][0], acc[j][xh][0]);      // Vh Ul0
                acc[j][xh][1] = mfma16(fa[xh][0], w4[sl][1][1], acc[j][xh][1]);      // Vh Ul1
                acc[j][xh][0] = mfma16(fa[xh][0], w4[sl][0][0], acc[j][xh][0]);      // Vh Uh0
                acc[j][xh][1] = mfma16(fa[xh][0], w4[sl][0][1], acc[j][xh][1]);      // Vh Uh1
                if (!lastst) read_half(xh, st + 1);
                __builtin_amdgcn_sched_barrier(0);                  // keeps the reads one step ahead (see the 8-wave form)
            }
        }
    };

#ifdef TOCVP_WINO_STAMP
    unsigned long long stamp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_readcyclecounter();
    const unsigned long long stamp_first = stamp_last;
#endif
    if (!wide && t < 128) {                     // zero padding left and right of a full-width tile
        const int row = t >> 4, cc = (t >> 2) & 3, ch = t & 3;
        *reinterpret_cast<f32x4*>(lds + ((row * IW + (cc < 2 ? cc : 64 + cc)) << 6) + ch * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    transform(0, false);
    WINO_T(0);
    __syncthreads();
    WINO_T(2);
#pragma unroll 1
    for (int pp = 0; pp < NPASS; pp += 2) {
        multiply(std::integral_constant<int, 0>{}, false, pp + 1);
        WINO_T(1);
        __syncthreads();                        // every wave is done reading the image
        WINO_T(2);
        transform(pp + 1, true);
        WINO_T(0);
        __syncthreads();
        WINO_T(2);
        const bool more = pp + 2 < NPASS;
        multiply(std::integral_constant<int, 2>{}, !more, pp + 2);
        WINO_T(1);
        __syncthreads();
        WINO_T(2);
        if (more) {
            transform(pp + 2, true);
            WINO_T(0);
            __syncthreads();
            WINO_T(2);
        }
    }

    // ---- the eight M_xi meet: round r = 32-pixel half.  Wave w writes its 2 x 2 tiles; then reads register quad g = w of
    // all eight xi and both channel halves and forms the four output rows of 8 columns x 64 channels (as the 8-wave form).
    float* const xch = reinterpret_cast<float*>(lds);
    constexpr int SS = C + 4;
    if (TOCVP_WINO_ABLATE == 2) {               // one store per wave keeps the accumulators alive
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int xh = 0; xh < 2; ++xh)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int q = 0; q < 16; ++q) sum += acc[j][xh][n][q];
        if (sum == 12345.f) p.y[t] = sum;
        return;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r > 0) __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& a = acc[j][r][n];
                    *reinterpret_cast<f32x4*>(xch + (((((wave * 2 + j) * 2 + n) * 4 + g) * 64 + lane) << 2)) =
                        f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                }
        WINO_T(3);
        __syncthreads();
        WINO_T(4);
        f32x16 y[2];                            // starts from the bias (x the output buffer's scale, as the coefficients)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float b0 = p.bias[n * 32 + l31] * p.oscale;
#pragma unroll
            for (int q = 0; q < 16; ++q) y[n][q] = b0;
        }
#pragma unroll
        for (int q = 0; q < NXI; ++q)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xch + ((((q * 2 + n) * 4 + wave) * 64 + lane) << 2));
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float cf = p.coef[a * NXI + q] * csc;
                    if (a == 0 ? q == 7 : q == 0) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[n][4 * a + e] = fmaf(cf, v[e], y[n][4 * a + e]);
                }
            }
        WINO_T(5);
        __syncthreads();
        WINO_T(6);

        const int oy0 = ty0, ox = tx0 + r * 32 + wave * 8;
        wino_epilogue<TAILP>(p, y, lds, wave, lane, img, oy0, ox);
        WINO_T(7);
    }
#ifdef TOCVP_WINO_STAMP
    if ((wave == 0 || wave == 3) && lane == 0 && blockIdx.x < 16384) {
        unsigned long long* o = tocvp_wino_stamps + ((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 16;
        for (int i = 0; i < 8; ++i) o[i] = stamp_acc[i];
        o[14] = stamp_first;
        o[15] = stamp_last;
    }
#endif
}


// G of F(4, 5) over the points 0, 1, -1, 2, -2, 1/2, -1/2, infinity (rows normalised so that BT above has its unit entries)
__device__ const double WINO_G[NXI][5] = {
    {1.0, 0.0, 0.0, 0.0, 0.0},
    {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
    {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
    {1.0 / 90, 2.0 / 90, 4.0 / 90, 8.0 / 90, 16.0 / 90},
    {1.0 / 90, -2.0 / 90, 4.0 / 90, -8.0 / 90, 16.0 / 90},
    {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45},
    {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45},
    {0.0, 0.0, 0.0, 0.0, 1.0},
};

// (64, 64, 5, 5) fp32 -> U_xi[dx] = sum_k G[xi][k] w[o][c][k][dx] (fp64), scaled by scales[xi] and split into fp16 planes,
// wf: [xi / 2][pass 4][xi % 2][dx 5][plane(h, l)][nb 2][lane 64][8 f16]; lane (c = l & 31, hh = l >> 5): output channel nb*32 + c,
// input channels pass*16 + 8 hh + j.  absmax != NULL: only max |U_xi| is produced (fp32 bits, atomicMax on non-negative
// floats), for the host's choice of the scales.
struct WScales { float s[NXI]; };
__global__ __launch_bounds__(256) void split_conv_weights_wino_kernel(const float* __restrict__ w, _Float16* __restrict__ wf,
                                                                      WScales sc, unsigned* __restrict__ absmax) {
    const int i = blockIdx.x * 256 + threadIdx.x;                  // over 8 xi * 5 dx * 64 * 64
    if (i >= NXI * NDX * C * C) return;
    const int ci = i % C, co = (i / C) % C, dx = (i / (C * C)) % NDX, xi = i / (C * C * NDX);
    double u = 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) u += WINO_G[xi][k] * (double)w[((size_t)co * C + ci) * 25 + k * 5 + dx];
    if (absmax) {
        atomicMax(absmax + xi, __float_as_uint(fabsf((float)u) * 1.0000002f));
        return;
    }
    const float Uv = clampf((float)(u * (double)sc.s[xi]), F16MAX);
    const _Float16 hi = (_Float16)Uv;
    const _Float16 lo = (_Float16)((float)(u * (double)sc.s[xi] - (double)(float)hi));
    const int pass = ci / CCH, cc = ci % CCH, hh = cc >> 3, j = cc & 7;
    const int nb = co >> 5, c = co & 31;
    const size_t stepbase = ((((size_t)(xi >> 1) * NPASS + pass) * 2 + (xi & 1)) * NDX + dx) * 4;
    wf[((stepbase + 0 * 2 + nb) * 64 + hh * 32 + c) * 8 + j] = hi;
    wf[((stepbase + 1 * 2 + nb) * 64 + hh * 32 + c) * 8 + j] = lo;
}

}  // namespace

extern "C" size_t tocvp_conv_weights_wino_f16x3_bytes(void) { return (size_t)NXI * NSTEP * STEP_BYTES; }

// Step 1 (scales == NULL): absmax_out[8] (device, fp32) <- max |U_xi|.  Step 2: wf <- the fragment-order planes of
// scales[xi] * U_xi (host array of 8 powers of two).
extern "C" int tocvp_split_conv_weights_wino_f16x3(const float* w, void* wf, const float* scales, float* absmax_out,
                                                   int Cout, int Cin, void* stream) {
    TOCVP_CHECK_ARG(w && Cout == C && Cin == C && ((scales && wf) || (!scales && absmax_out)));
    hipStream_t s = static_cast<hipStream_t>(stream);
    WScales sc{};
    if (scales) {
        for (int i = 0; i < NXI; ++i) {
            TOCVP_CHECK_ARG(scales[i] > 0.f);
            sc.s[i] = scales[i];
        }
        if (!tocvp_aligned16(wf)) return TOCVP_EALIGN;
    } else {
        if (hipMemsetAsync(absmax_out, 0, NXI * sizeof(float), s) != hipSuccess) return TOCVP_ELAUNCH;
    }
    hipLaunchKernelGGL(split_conv_weights_wino_kernel, dim3((NXI * NDX * C * C + 255) / 256), dim3(256), 0, s, w,
                       static_cast<_Float16*>(wf), sc, scales ? nullptr : reinterpret_cast<unsigned*>(absmax_out));
    return tocvp_launch_status();
}

// in_mode: 0 fp32 x 16 pass-major (out_mode 1 of this entry), 1 collapsed first layer (x = cpos (H, W, 64), aux = (nimg, 25, 64)),
// 2 fp32 NHWC.  out_mode: 0 fp32 NHWC, 1 fp32 x 16 pass-major, 2 fp16 operand planes (the planes input of
// tocvp_conv5x5_dec_f16x3_f32 / _tail_f32), 3 (tail_taps != NULL) the (nimg, 36, H, W) tap products of the folded decoder tail.
// coef[32] = AT[a][xi] / (16 * scales[xi]) for the scales the weights were split with (host array).
// in_amax (in_mode 2, nullable): DEVICE word with max |x| (tocvp_absmax_f32) -- inputs of any magnitude (data gradients);
// gate (out_mode 0, nullable): NHWC array, outputs are zeroed where gate <= 0.
extern "C" int tocvp_conv5x5_dec_wino_f16x3_f32(const float* x, const float* aux, int in_mode, const void* wf,
                                                const float* coef, const float* bias, const void* tail_taps, float* y,
                                                int nimg, int H, int W, int relu, int out_mode, const float* in_amax,
                                                const float* gate, void* stream) {
    TOCVP_CHECK_ARG((!in_amax || in_mode == 2) && (!gate || out_mode == 0));
    if (gate && !tocvp_aligned16(gate)) return TOCVP_EALIGN;
    TOCVP_CHECK_ARG(x && wf && coef && bias && y);
    TOCVP_CHECK_ARG(in_mode >= 0 && in_mode <= 2 && (in_mode != 1 || aux != nullptr));
    TOCVP_CHECK_ARG(out_mode >= 0 && out_mode <= 3 && ((out_mode == 3) == (tail_taps != nullptr)));
    TOCVP_CHECK_ARG(nimg >= 0 && H > 0 && W > 0 && (H % W4_TH) == 0 && (W % TW) == 0);
    TOCVP_CHECK_ARG((size_t)nimg * (H / W4_TH) * (W / TW) < 0x7fffffffu && (size_t)H * W * C * 4 < 0x7fffffffu);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(wf) || !tocvp_aligned16(y) || (aux && !tocvp_aligned16(aux)) ||
        (tail_taps && !tocvp_aligned16(tail_taps)))
        return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    WArgs a{x, aux, static_cast<const unsigned char*>(wf), bias, y, nimg, H, W, relu, out_mode,
            static_cast<const unsigned char*>(tail_taps), {}, 1.f, in_amax, gate};
    a.oscale = out_mode == 1 ? VS : (out_mode >= 2 ? TOCVP_F16X3_ACT_SCALE : 1.f);      // exact powers of two
    for (int i = 0; i < 4 * NXI; ++i) a.coef[i] = coef[i] * a.oscale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((size_t)((nimg + 7) / 8) * 8 * (H / W4_TH) * (W / TW)));
    const dim3 block(W4_THREADS);
    if (out_mode == 3) {
        if (in_mode == 0) hipLaunchKernelGGL((conv5x5_wino_f16x3_kernel<0, true>), grid, block, 0, s, a);
        else if (in_mode == 1) hipLaunchKernelGGL((conv5x5_wino_f16x3_kernel<1, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv5x5_wino_f16x3_kernel<2, true>), grid, block, 0, s, a);
    } else {
        if (in_mode == 0) hipLaunchKernelGGL((conv5x5_wino_f16x3_kernel<0, false>), grid, block, 0, s, a);
        else if (in_mode == 1) hipLaunchKernelGGL((conv5x5_wino_f16x3_kernel<1, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv5x5_wino_f16x3_kernel<2, false>), grid, block, 0, s, a);
    }
    return tocvp_launch_status();
}
