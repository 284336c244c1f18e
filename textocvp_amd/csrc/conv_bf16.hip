// 5x5 convolution (64 -> 64 channels) with SPLIT-bf16 operands on the bf16 matrix cores.
//
// Every fp32 operand x is split into hi = bf16(x) and lo = bf16(x - hi) (16 significant bits
// together).  A product block is evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation
// ("bf16x3"): three v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16 each) replace eight
// v_mfma_f32_32x32x2_f32 (64 cycles, K = 2 each) -> 5.3x fewer matrix-pipe cycles per FLOP at a
// per-product relative error of ~2^-16 (dropped lo*lo term and the 16-bit operand truncation).
// fp32 in HBM on both sides: the split happens while the halo tile is staged into LDS (each
// element is split once per tile and reused by 25 taps x 64 output channels), weights are
// pre-split once.
//
// Geometry is that of conv5x5_mfma_kernel (conv.hip): 8 x 32 pixel tile x 64 output channels per
// 4-wave workgroup, halo tile staged once, per-tap weight slices double buffered.  LDS image per
// pixel / per output channel: [64 hi bf16 | 64 lo bf16 | 16 B pad] = 272 B, so the 16 rows of
// every ds_read_b128 lane group land on 16 distinct 16-byte slots.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32, IH = TH + 4, IW = TW + 4;
constexpr int C = 64;

struct ConvArgs {
    const float* x; const float* aux; const unsigned char* wsplit; const float* bias; float* y;
    int nimg, H, W, relu;
    const unsigned char* wfrag;   // weights in MFMA-fragment order (WD variants), else unused
};

__device__ __forceinline__ int border_class(int p, int n) {
    return p < 2 ? p : (p >= n - 2 ? 4 - (n - 1 - p) : 2);
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void split4(const f32x4 v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        hi[u] = (__bf16)v[u];
        lo[u] = (__bf16)(v[u] - (float)hi[u]);
    }
}

// NW = waves per workgroup: 4 (one per SIMD, each 2 rows x 64 channels) or 8 (two per SIMD, each
// 2 rows x 32 channels -- the partner wave's MFMAs cover this wave's barrier / LDS latency).
// CCH = input channels resident per pass (64: one pass, 152 KB LDS, 1 workgroup / CU;
//       32: two passes over the taps, 80.6 KB LDS -> 2 workgroups / CU, so one workgroup's tile
//       staging / epilogue / barriers overlap the other's MFMAs).
// WD = weights direct: B fragments come straight from global memory in MFMA-fragment order
//      Wf[tap][pass][ks][nb][plane][lane] (one coalesced 1 KiB load per fragment, L1/L2 resident: the
//      800 KB of split weights are shared by every workgroup), prefetched one tap ahead in registers.
//      No weight image in LDS and NO barrier inside the 25-tap loop: the waves of a workgroup run
//      free between the two channel passes.
template <int MODE, int NW, int CCH, bool WD = false>
__global__ __launch_bounds__(NW * 64, (CCH == 32 ? 2 : 1) * (NW / 4))
void conv5x5_bf16x3_kernel(ConvArgs p) {
    static_assert(!WD || (NW == 4 && CCH == 32), "weights-direct variant is built for 4 waves x 32 ch");
    constexpr int NT = NW * 64;            // threads
    constexpr int NB = 8 / NW;             // 32-channel output blocks per wave (2 or 1)
    constexpr int ROWB = 2 * CCH * 2 + 16; // LDS bytes per pixel / per output channel (hi | lo | pad)
    constexpr int NPASS = C / CCH;
    constexpr int WCHUNKS = C * (CCH / 4); // 16-byte weight chunks per tap per pass
    constexpr int WR = WCHUNKS / NT;       // per thread
    constexpr int CPR = CCH / 4;           // 16-byte chunks per weight row (hi + lo)
    __shared__ __attribute__((aligned(16))) unsigned char lds[IH * IW * ROWB + 2 * C * ROWB];
    unsigned char* in_s = lds;
    unsigned char* w_s = lds + IH * IW * ROWB;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wrow = (NW == 4) ? wave : (wave >> 1);      // row pair of the tile
    const int wcol = (NW == 4) ? 0 : (wave & 1);          // first 32-channel block
    const int tiles_x = p.W / TW, tiles = tiles_x * (p.H / TH);
    const int img = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;

    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-tap weight slice: 64 output channels x 256 B (hi | lo) = 1024 16-byte chunks
    f32x4 wreg[WR];
    auto wload = [&](int tap, int pass) {
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int idx = t + NT * i;
            const int co = idx / CPR, part = idx % CPR;
            const int plane = part / (CPR / 2), sub = part % (CPR / 2);
            wreg[i] = *reinterpret_cast<const f32x4*>(p.wsplit + ((size_t)tap * C + co) * 256 +
                                                      plane * 128 + pass * CCH * 2 + sub * 16);
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int idx = t + NT * i;
            const int co = idx / CPR, part = idx % CPR;
            const int plane = part / (CPR / 2), sub = part % (CPR / 2);
            *reinterpret_cast<f32x4*>(w_s + buf * C * ROWB + co * ROWB + plane * CCH * 2 + sub * 16) =
                wreg[i];
        }
    };

  for (int pass = 0; pass < NPASS; ++pass) {
    if (pass > 0) __syncthreads();          // everyone done with the previous pass's LDS images
    if (!WD) wload(0, pass);
    // ---- halo tile: fp32 -> (hi, lo) bf16 planes in LDS.
    // All global loads of the tile are issued back to back from clamped (always valid) addresses
    // and only then converted: a load-use-per-iteration loop would expose the full memory latency
    // NIT times per tile (measured: ~40 % of the kernel).
    constexpr int NIT = (IH * IW * (CCH / 4) + NT - 1) / NT;
    f32x4 tv[NIT];
    f32x4 ts[MODE == 1 ? NIT : 1];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = min(t + it * NT, IH * IW * (CCH / 4) - 1);
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
    for (int it = 0; it < NIT; ++it) {
        const int i = t + it * NT;
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
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = inside ? v[u] : 0.f;
            bf16x4 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<bf16x4*>(in_s + pix * ROWB + c * 2) = hi;
            *reinterpret_cast<bf16x4*>(in_s + pix * ROWB + CCH * 2 + c * 2) = lo;
        }
    }
    if (!WD) wstore(0);
    __syncthreads();

    if (WD) {
        bf16x8 fb0[2][NB][2], fb1[2][NB][2];                       // [ks][nb][plane]
        auto gload_b = [&](bf16x8 (&fb)[2][NB][2], int tap) {
            const unsigned char* base = p.wfrag + (size_t)(tap * NPASS + pass) * (2 * NB * 2) * 1024 +
                                        lane * 16;               // uniform part + lane offset
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        fb[ks][n][pl] = *reinterpret_cast<const bf16x8*>(
                            base + ((ks * NB + n) * 2 + pl) * 1024);
        };
        auto compute = [&](const bf16x8 (&fb)[2][NB][2], int tap) {
            const int dy = tap / 5, dx = tap % 5;
            const unsigned char* a_base = in_s + ((2 * wrow + dy) * IW + l31 + dx) * ROWB + h * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ah[2], al[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ah[m] = *reinterpret_cast<const bf16x8*>(a_base + m * IW * ROWB + ks * 32);
                    al[m] = *reinterpret_cast<const bf16x8*>(a_base + m * IW * ROWB + ks * 32 + CCH * 2);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        acc[m][n] = mfma_bf16(al[m], fb[ks][n][0], acc[m][n]);
                        acc[m][n] = mfma_bf16(ah[m], fb[ks][n][1], acc[m][n]);
                        acc[m][n] = mfma_bf16(ah[m], fb[ks][n][0], acc[m][n]);
                    }
            }
        };
        gload_b(fb0, 0);
        for (int tap = 0; tap < 25; tap += 2) {                    // prefetches unconditional + clamped
            gload_b(fb1, min(tap + 1, 24));
            __builtin_amdgcn_sched_barrier(0);
            compute(fb0, tap);
            gload_b(fb0, min(tap + 2, 24));
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < 25) compute(fb1, tap + 1);
        }
    } else
    for (int tap = 0; tap < 25; ++tap) {
        const int buf = tap & 1;
        if (tap + 1 < 25) wload(tap + 1, pass);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE the MFMAs (hipcc sinks it)
        const int dy = tap / 5, dx = tap % 5;
        const unsigned char* a_base = in_s + ((2 * wrow + dy) * IW + l31 + dx) * ROWB + h * 16;
        const unsigned char* b_base = w_s + buf * C * ROWB + (wcol * 32 + l31) * ROWB + h * 16;
#pragma unroll
        for (int ks = 0; ks < CCH / 16; ++ks) {
            bf16x8 ah[2], al[2], bh[NB], bl[NB];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8*>(a_base + m * IW * ROWB + ks * 32);
                al[m] = *reinterpret_cast<const bf16x8*>(a_base + m * IW * ROWB + ks * 32 + CCH * 2);
            }
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                bh[n] = *reinterpret_cast<const bf16x8*>(b_base + n * 32 * ROWB + ks * 32);
                bl[n] = *reinterpret_cast<const bf16x8*>(b_base + n * 32 * ROWB + ks * 32 + CCH * 2);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    acc[m][n] = mfma_bf16(al[m], bh[n], acc[m][n]);
                    acc[m][n] = mfma_bf16(ah[m], bl[n], acc[m][n]);
                    acc[m][n] = mfma_bf16(ah[m], bh[n], acc[m][n]);
                }
        }
        if (tap + 1 < 25) wstore(buf ^ 1);
        __syncthreads();
    }
  }  // pass
    if (WD) __syncthreads();   // no barrier in the tap loop: the halo image must be dead before restaging

    if (NW == 4) {
        // Epilogue through LDS: the accumulator layout gives each lane ONE output channel of 16
        // pixels, i.e. 64 four-byte stores per lane in 128-byte segments (measured ~1.5 TB/s).
        // Staging the wave's 64 pixels x 64 channels tile in LDS (the halo / weight images are dead
        // after the last tap's barrier) turns them into 16 dwordx4 stores per lane, each wave
        // instruction writing 1 KiB of contiguous NHWC output (4 pixels x 256 B).
        constexpr int SS = C + 4;                                  // padded floats per pixel
        static_assert(4 * 64 * SS * 4 <= IH * IW * ROWB + 2 * C * ROWB, "staging must fit the LDS image");
        float* stage = reinterpret_cast<float*>(lds) + wave * (64 * SS);
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const float bv = p.bias[n * 32 + l31];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[m][n][r] + bv;
                    if (p.relu == 1) v = fmaxf(v, 0.f);
                    stage[(m * 32 + acc_row(r, h)) * SS + n * 32 + l31] = v;
                }
        }
        __builtin_amdgcn_wave_barrier();
        float* ybase = p.y + (((size_t)img * p.H + ty0 + 2 * wrow) * p.W + tx0) * C;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = lane + 64 * it;
            const int px = idx >> 4, c4 = (idx & 15) * 4;          // pixel 0..63 of the wave, channel
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + c4);
            const size_t off = ((size_t)(px >> 5) * p.W + (px & 31)) * C + c4;
            if (MODE == 0 && p.relu == 2) {        // data gradient: gate by the ReLU of the layer below
                const f32x4 g = *reinterpret_cast<const f32x4*>(p.aux + (ybase - p.y) + off);
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = g[u] > 0.f ? v[u] : 0.f;
            }
            *reinterpret_cast<f32x4*>(ybase + off) = v;
        }
    } else {
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const float bv = p.bias[(wcol + n) * 32 + l31];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int oy = ty0 + 2 * wrow + m;
                float* yrow = p.y + (((size_t)img * p.H + oy) * p.W + tx0) * C + (wcol + n) * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[m][n][r] + bv;
                    if (p.relu == 1) v = fmaxf(v, 0.f);
                    yrow[(size_t)acc_row(r, h) * C] = v;
                }
            }
        }
    }
}

// (Cout, Cin, 5, 5) fp32 -> [tap][cout][hi Cin | lo Cin] bf16
__global__ __launch_bounds__(256) void split_conv_weights_kernel(const float* __restrict__ w,
                                                                 __bf16* __restrict__ out, int Cout,
                                                                 int Cin) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)25 * Cout * Cin) return;
    const int ci = (int)(i % Cin);
    const int co = (int)((i / Cin) % Cout);
    const int tap = (int)(i / ((long)Cin * Cout));
    const float v = w[((size_t)co * Cin + ci) * 25 + tap];
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    __bf16* o = out + ((size_t)tap * Cout + co) * 2 * Cin;
    o[ci] = hi;
    o[Cin + ci] = lo;
}

// (64, 64, 5, 5) fp32 -> fragment order Wf[tap][pass(2)][ks(2)][nb(2)][plane(2)][lane(64)][8] bf16
__global__ __launch_bounds__(256) void split_conv_weights_frag_kernel(const float* __restrict__ w,
                                                                      __bf16* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;                  // over 25 * 64 * 64
    if (i >= 25 * C * C) return;
    const int ci = i % C, co = (i / C) % C, tap = i / (C * C);
    const float v = w[((size_t)co * C + ci) * 25 + tap];
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    const int pass = ci >> 5, ks = (ci >> 4) & 1, hh = (ci >> 3) & 1, j = ci & 7;
    const int nb = co >> 5, c = co & 31;
    const size_t frag = ((((size_t)tap * 2 + pass) * 2 + ks) * 2 + nb) * 2;    // + plane
    out[((frag + 0) * 64 + hh * 32 + c) * 8 + j] = hi;
    out[((frag + 1) * 64 + hh * 32 + c) * 8 + j] = lo;
}

}  // namespace

extern "C" int tocvp_split_conv_weights_frag_bf16(const float* w, void* out, int Cout, int Cin,
                                                  void* stream) {
    TOCVP_CHECK_ARG(w && out && Cout == C && Cin == C);
    hipLaunchKernelGGL(split_conv_weights_frag_kernel, dim3((25 * C * C + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, static_cast<__bf16*>(out));
    return tocvp_launch_status();
}

extern "C" int tocvp_split_conv_weights_bf16(const float* w, void* out, int Cout, int Cin,
                                             void* stream) {
    TOCVP_CHECK_ARG(w && out && Cout > 0 && Cin > 0);
    const long n = (long)25 * Cout * Cin;
    hipLaunchKernelGGL(split_conv_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, static_cast<__bf16*>(out), Cout, Cin);
    return tocvp_launch_status();
}

extern "C" int tocvp_conv5x5_bf16x3_f32(const float* x, const float* aux, int in_mode,
                                        const void* wsplit, const void* wfrag, const float* bias,
                                        float* y, int nimg, int H, int W, int Cin, int Cout, int relu,
                                        void* stream) {
    TOCVP_CHECK_ARG(x && (wsplit || wfrag) && bias && y);
    TOCVP_CHECK_ARG(in_mode == 0 || (in_mode == 1 && aux != nullptr));
    TOCVP_CHECK_ARG(relu >= 0 && relu <= 2 && (relu != 2 || (in_mode == 0 && aux != nullptr && wfrag != nullptr)));
    TOCVP_CHECK_ARG(Cin == C && Cout == C);
    TOCVP_CHECK_ARG(nimg >= 0 && H > 0 && W > 0 && (H % TH) == 0 && (W % TW) == 0);
    TOCVP_CHECK_ARG((size_t)nimg * (H / TH) * (W / TW) < 0x7fffffffu);
    if (!tocvp_aligned16(x) || (wsplit && !tocvp_aligned16(wsplit)) || (wfrag && !tocvp_aligned16(wfrag)) ||
        (aux && !tocvp_aligned16(aux)))
        return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    ConvArgs a{x, aux, static_cast<const unsigned char*>(wsplit), bias, y, nimg, H, W, relu,
               static_cast<const unsigned char*>(wfrag)};
    const dim3 grid((unsigned)((size_t)nimg * (H / TH) * (W / TW)));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // 4 waves, two 32-channel passes, 2 workgroups/CU (the "8x64" / "4x64" / "8x32" forms of round 1 lost their A/B: gone)
    constexpr int variant = 432;
#define TOCVP_LAUNCH_CONV(NW_, CCH_)                                                              \
    do {                                                                                          \
        if (in_mode == 0)                                                                         \
            hipLaunchKernelGGL((conv5x5_bf16x3_kernel<0, NW_, CCH_>), grid, dim3(NW_ * 64), 0, s, a); \
        else                                                                                      \
            hipLaunchKernelGGL((conv5x5_bf16x3_kernel<1, NW_, CCH_>), grid, dim3(NW_ * 64), 0, s, a); \
    } while (0)
    if (wfrag && (variant == 432 || !wsplit || relu == 2)) {      // weights-direct variant (default when given)
        if (in_mode == 0)
            hipLaunchKernelGGL((conv5x5_bf16x3_kernel<0, 4, 32, true>), grid, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((conv5x5_bf16x3_kernel<1, 4, 32, true>), grid, dim3(256), 0, s, a);
        return tocvp_launch_status();
    }
    TOCVP_CHECK_ARG(wsplit != nullptr);
    TOCVP_LAUNCH_CONV(4, 32);
#undef TOCVP_LAUNCH_CONV
    return tocvp_launch_status();
}
