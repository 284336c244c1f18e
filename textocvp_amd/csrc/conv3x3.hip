// Kernels of the ExtendedDINOSAUR decode side (reference models/EncodersDecoders/decoders.py:203-365):
//
//  conv3x3_mfma_kernel   : Conv2d(Cin->Cout, k3, p1) + folded BatchNorm(eval) + ReLU as an implicit
//                          GEMM on fp32 MFMA, NHWC; optional nearest x2 upsampling of the INPUT fused
//                          into the tile loader (model_blocks.py:23-45 `Upsample` between blocks of the
//                          CNN image head never materialises the 4x larger tensor).
//                          Cin % 64 == 0, Cout % 32 == 0, any H % 8 == 0, any W (masked stores).
//                          Same geometry as conv5x5_mfma_kernel: 8 x 32 pixel tile x 64 (or 32) output
//                          channels per workgroup, halo tile (10 x 34 x 64ch) staged per 64-channel
//                          chunk, per-tap weight slices double-buffered in LDS.
//  slot_composite_kernel : alpha-softmax over slots + weighted feature sum of MLPPatchDecoder.forward
//                          (decoders.py:279-283).
//  bilinear_resize_kernel: F.interpolate(mode='bilinear', align_corners=False) of the image head
//                          (decoders.py:291-297), NHWC in -> NCHW out.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2;    // IH / IW: 3x3 fp32 kernel (split kernel: per KS)
constexpr int CC = 64, CS = CC + 4;

struct Conv3Args {
    const float* x; const float* wp; const float* scale; const float* shift; float* y;
    int nimg, H, W, Cin, Cout, relu, upsample;   // H, W = OUTPUT (= conv input after upsampling) size
};

template <int NB>   // 32-channel output blocks per workgroup (2 -> 64 channels, 1 -> 32)
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(Conv3Args p) {
    constexpr int COUTB = NB * 32;
    constexpr int F4 = CC / 4;
    constexpr int WREG = (COUTB * F4) / 256;
    constexpr int NIT = (IH * IW * F4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[IH * IW * CS + 2 * COUTB * CS];
    float* in_s = lds;
    float* w_s = lds + IH * IW * CS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_x = (p.W + TW - 1) / TW, tiles = tiles_x * (p.H / TH);
    const int img = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    const int co0 = blockIdx.y * COUTB;
    const int SH = p.upsample ? p.H / 2 : p.H, SW = p.upsample ? p.W / 2 : p.W;   // source size
    const int sh = p.upsample ? 1 : 0;

    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 wreg[WREG];
    auto wload = [&](int tap, int ch) {
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int idx = t + 256 * i;
            const int co = idx / F4, c = (idx % F4) * 4;
            wreg[i] = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)tap * p.Cout + co0 + co) * p.Cin +
                                                      ch * CC + c);
        }
    };
    auto wstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int idx = t + 256 * i;
            const int co = idx / F4, c = (idx % F4) * 4;
            *reinterpret_cast<f32x4*>(w_s + buf * COUTB * CS + co * CS + c) = wreg[i];
        }
    };

    const int nch = p.Cin / CC;
    for (int ch = 0; ch < nch; ++ch) {
        // halo tile of this channel chunk: batched loads from clamped addresses, zeroed outside
        f32x4 tv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = min(t + it * 256, IH * IW * F4 - 1);
            const int pix = i / F4, c = (i % F4) * 4;
            const int iy = min(max(ty0 + pix / IW - 1, 0), p.H - 1) >> sh;
            const int ix = min(max(tx0 + pix % IW - 1, 0), p.W - 1) >> sh;
            tv[it] = *reinterpret_cast<const f32x4*>(p.x + (((size_t)img * SH + iy) * SW + ix) * p.Cin +
                                                     ch * CC + c);
        }
        wload(0, ch);
        __syncthreads();   // previous chunk fully consumed
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = t + it * 256;
            if (i < IH * IW * F4) {
                const int pix = i / F4, c = (i % F4) * 4;
                const int iy = ty0 + pix / IW - 1, ix = tx0 + pix % IW - 1;
                const bool inside = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                f32x4 v = tv[it];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = inside ? v[u] : 0.f;
                *reinterpret_cast<f32x4*>(in_s + pix * CS + c) = v;
            }
        }
        wstore(0);
        __syncthreads();

        for (int tap = 0; tap < 9; ++tap) {
            const int buf = tap & 1;
            if (tap + 1 < 9) wload(tap + 1, ch);
            __builtin_amdgcn_sched_barrier(0);
            const int dy = tap / 3, dx = tap % 3;
            const float* a_base = in_s + ((2 * wave + dy) * IW + l31 + dx) * CS + 4 * h;
            const float* b_base = w_s + buf * COUTB * CS + l31 * CS + 4 * h;
#pragma unroll
            for (int j = 0; j < CC / 8; ++j) {
                f32x4 a[2], b[NB];
                a[0] = *reinterpret_cast<const f32x4*>(a_base + 8 * j);
                a[1] = *reinterpret_cast<const f32x4*>(a_base + IW * CS + 8 * j);
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    b[n] = *reinterpret_cast<const f32x4*>(b_base + n * 32 * CS + 8 * j);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < NB; ++n) acc[m][n] = mfma32(a[m][u], b[n][u], acc[m][n]);
            }
            if (tap + 1 < 9) wstore(buf ^ 1);
            __syncthreads();
        }
    }

#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int co = co0 + n * 32 + l31;
        const float sc = p.scale ? p.scale[co] : 1.f, sf = p.shift[co];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int oy = ty0 + 2 * wave + m;
            float* yrow = p.y + (((size_t)img * p.H + oy) * p.W) * p.Cout + co;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ox = tx0 + acc_row(r, h);
                if (ox < p.W) {
                    float v = fmaf(acc[m][n][r], sc, sf);
                    if (p.relu) v = fmaxf(v, 0.f);
                    yrow[(size_t)ox * p.Cout] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Same convolution with f16x3 split operands (two fp16 planes of 2^8 x / 2^10 w, three
// v_mfma_f32_32x32x16_f16 products per k-step, fp32-class; arithmetic of gemm_bf16.hip Elem<true>):
// 3/16 of the matrix cycles of the fp32 MFMA form above.  32-channel chunks keep the (hi | lo) halo
// image at 49 KB + 18 KB of weight slices -> two workgroups per CU.  Activations are split while the
// halo tile is staged, the per-tap weight slice while it is copied into LDS.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
constexpr int CH = 32, ROWB = 2 * CH * 2 + 16;

__device__ __forceinline__ void split4_f16(f32x4 v, float scale, unsigned char* dst) {
    h16x4 hi, lo;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float X = __builtin_amdgcn_fmed3f(v[u] * scale, -65504.f, 65504.f);
        hi[u] = (_Float16)X;
        lo[u] = (_Float16)(X - (float)hi[u]);
    }
    *reinterpret_cast<h16x4*>(dst) = hi;
    *reinterpret_cast<h16x4*>(dst + CH * 2) = lo;
}

// KS = 3 (image head of the DINOSAUR decoder) or 5 (SAVi encoder convs 32 -> 32, decoder shapes that the
// dedicated 64 -> 64 kernels do not take); halo KS / 2, same 8 x 32 pixel tile.
//
// KS = 2: one PHASE of "nearest x2 upsampling -> 3x3 conv" (round 4).  Output pixel (2 y + a, 2 x + b) of that pair reads
// the upsampled pixels (2 y + a + dy, 2 x + b + dx), dy, dx in {-1, 0, 1}, i.e. source pixels floor((2 y + a + dy) / 2):
// rows {y - 1, y, y} for a = 0 and {y, y, y + 1} for a = 1 -- a 2x2 conv over the SOURCE image whose four taps are sums of
// the 3x3 taps that fall on the same source pixel (zero padding carries over: an upsampled pixel is outside the image
// exactly when its source pixel is).  Four phases (blockIdx.z = 2 a + b) x four taps = 16 tap products per source pixel
// instead of 36: 2.25x fewer FLOPs than the conv over the upsampled image, same result up to the fp32 rounding of the
// tap sums.  p.H, p.W = source size, weights (4 phases, 4 taps, Cout, Cin), output (nimg, 2 H, 2 W, Cout).
// NARROW: images at most 16 pixels wide (the 16 x 16 patch grid of the image head): a 16 x 16 pixel tile whose 32-pixel MFMA
// blocks hold TWO image rows of 16 pixels instead of one row of 32 (an 8 x 32 tile would compute 16 columns of padding).
template <int NB, int KS, bool NARROW = false>
__global__ __launch_bounds__(256, 2) void convk_f16x3_kernel(Conv3Args p) {
    constexpr bool PH = KS == 2;
    constexpr int TH_ = NARROW ? 16 : TH, TW_ = NARROW ? 16 : TW;
    constexpr int IH = TH_ + KS - 1, IW = TW_ + KS - 1, NTAP = KS * KS;
    const int pa = PH ? (int)blockIdx.z >> 1 : 0, pb = PH ? (int)blockIdx.z & 1 : 0;
    const int oy0 = PH ? pa - 1 : -(KS / 2), ox0 = PH ? pb - 1 : -(KS / 2);     // first tap relative to the output pixel
    constexpr int COUTB = NB * 32;
    constexpr int F4 = CH / 4;
    constexpr int NIT = (IH * IW * F4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[IH * IW * ROWB];
    unsigned char* in_s = lds;
    typedef const __attribute__((address_space(1))) h16x8* gv8;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_x = (p.W + TW_ - 1) / TW_, tiles = tiles_x * (p.H / TH_);
    const int img = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int ty0 = (tile / tiles_x) * TH_, tx0 = (tile % tiles_x) * TW_;
    // pixel l31 of a wave's 32-pixel block m: row (NARROW: 4 wave + 2 m + l31 / 16, else 2 wave + m), column (l31 % 16 / l31)
    const int apy = NARROW ? (l31 >> 4) : 0, apx = NARROW ? (l31 & 15) : l31;
    constexpr int WROWS = NARROW ? 4 : 2, MSTEP = NARROW ? 2 : 1;
    const int co0 = blockIdx.y * COUTB;
    const int SH = p.upsample ? p.H / 2 : p.H, SW = p.upsample ? p.W / 2 : p.W;
    const int sh = p.upsample ? 1 : 0;

    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // weights: fp16 planes of 2^10 w in MFMA-FRAGMENT order, made once per weight version by tocvp_split_weights_frag_f16 on the
    // packed (taps x Cout, Cin) matrix: Wf[(tap Cout + co) / 32][Cin / 16][plane][lane] 16 bytes each -- a B fragment is ONE
    // coalesced 1 KiB load from L1 / L2 straight into the operand registers (round 5; rounds 2-4 split each tap's fp32 slice
    // per workgroup into LDS behind a barrier per tap: 24 MFMAs between barriers, 0.08 of the f16 peak)
    const int KC = p.Cin / 16;
    const unsigned char* wf_base = reinterpret_cast<const unsigned char*>(p.wp) + (size_t)lane * 16;
    const size_t rb0 = (size_t)co0 / 32;                               // row block of this workgroup's first output channel
    const size_t rb_tap = (size_t)p.Cout / 32;                         // row blocks per tap

    const int nch = p.Cin / CH;
    for (int ch = 0; ch < nch; ++ch) {
        f32x4 tv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = min(t + it * 256, IH * IW * F4 - 1);
            const int pix = i / F4, c = (i % F4) * 4;
            const int iy = min(max(ty0 + pix / IW + oy0, 0), p.H - 1) >> sh;
            const int ix = min(max(tx0 + pix % IW + ox0, 0), p.W - 1) >> sh;
            tv[it] = *reinterpret_cast<const f32x4*>(p.x + (((size_t)img * SH + iy) * SW + ix) * p.Cin +
                                                     ch * CH + c);
        }
        __syncthreads();   // previous chunk fully consumed
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = t + it * 256;
            if (i < IH * IW * F4) {
                const int pix = i / F4, c = (i % F4) * 4;
                const int iy = ty0 + pix / IW + oy0, ix = tx0 + pix % IW + ox0;
                const bool inside = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                f32x4 v = tv[it];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = inside ? v[u] : 0.f;
                split4_f16(v, TOCVP_F16X3_ACT_SCALE, in_s + pix * ROWB + c * 2);
            }
        }
        __syncthreads();

        // the taps of a chunk, fully unrolled (LDS offsets are immediates, no barrier inside): (ch, tap, ks) order as before
        const unsigned char* wch = wf_base + ((size_t)(PH ? (int)blockIdx.z * NTAP : 0) * rb_tap + rb0) * KC * 2048 +
                                   (size_t)ch * (CH / 16) * 2048;
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int dy = tap / KS, dx = tap % KS;
            const unsigned char* a_base = in_s + ((WROWS * wave + apy + dy) * IW + apx + dx) * ROWB + h * 16;
            const unsigned char* wtap = wch + (size_t)tap * rb_tap * KC * 2048;
#pragma unroll
            for (int ks = 0; ks < CH / 16; ++ks) {
                h16x8 ah[2], al[2], bh[NB], bl[NB];
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const unsigned char* wp_ = wtap + ((size_t)n * KC + ks) * 2048;
                    bh[n] = *(gv8)(wp_);
                    bl[n] = *(gv8)(wp_ + 1024);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ah[m] = *reinterpret_cast<const h16x8*>(a_base + m * MSTEP * IW * ROWB + ks * 32);
                    al[m] = *reinterpret_cast<const h16x8*>(a_base + m * MSTEP * IW * ROWB + ks * 32 + CH * 2);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
                    }
            }
        }
    }

    constexpr float UNSCALE = 1.f / (TOCVP_F16X3_ACT_SCALE * TOCVP_F16X3_WEIGHT_SCALE);
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int co = co0 + n * 32 + l31;
        const float sc = (p.scale ? p.scale[co] : 1.f) * UNSCALE, sf = p.shift[co];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pp = acc_row(r, h);
                const int oy = ty0 + WROWS * wave + MSTEP * m + (NARROW ? (pp >> 4) : 0);
                const int ox = tx0 + (NARROW ? (pp & 15) : pp);
                if (ox < p.W) {
                    float v = fmaf(acc[m][n][r], sc, sf);
                    if (p.relu) v = fmaxf(v, 0.f);
                    float* yp = PH ? p.y + ((((size_t)img * 2 * p.H + 2 * oy + pa) * 2 * p.W + pb) + 2 * ox) * p.Cout + co
                                   : p.y + (((size_t)img * p.H + oy) * p.W + ox) * p.Cout + co;
                    *yp = v;
                }
            }
        }
    }
}

// decoded (B, K, N, F+1) -> recons (B, N, F), masks (B, K, N):  alpha = softmax_K(decoded[..., F])
__global__ __launch_bounds__(256) void slot_composite_kernel(const float* __restrict__ dec,
                                                             float* __restrict__ recons,
                                                             float* __restrict__ masks, int K, int N,
                                                             int F, int ld) {
    __shared__ float a_s[64];
    const int b = blockIdx.y, n = blockIdx.x, t = threadIdx.x;
    const size_t row = (size_t)ld;                                    // >= F + 1 (padded GEMM output)
    const float* base = dec + ((size_t)b * K * N + n) * row;          // slot k at + k*N*row
    if (t < 64) {
        float a = (t < K) ? base[(size_t)t * N * row + F] : -1.0e30f;
        float m = a;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const float e = (t < K) ? expf(a - m) : 0.f;
        const float s = wave_sum64(e);
        const float w = e / s;
        a_s[t] = w;
        if (t < K) masks[((size_t)b * K + t) * N + n] = w;
    }
    __syncthreads();
    for (int f = t; f < F; f += 256) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += base[(size_t)k * N * row + f] * a_s[k];
        recons[((size_t)b * N + n) * F + f] = acc;
    }
}

// NHWC (n, SH, SW, C) -> NCHW (n, C, OH, OW), bilinear, align_corners=False (PyTorch semantics)
__global__ __launch_bounds__(256) void bilinear_resize_kernel(const float* __restrict__ x,
                                                              float* __restrict__ y, int C, int CSTR,
                                                              int SH, int SW, int OH, int OW,
                                                              long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const int c = (int)((i / ((long)OW * OH)) % C);
    const long n = i / ((long)OW * OH * C);
    const float ry = (float)SH / (float)OH, rx = (float)SW / (float)OW;
    float fy = ((float)oy + 0.5f) * ry - 0.5f, fx = ((float)ox + 0.5f) * rx - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < SH - 1 ? 1 : 0), x1 = x0 + (x0 < SW - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* xp = x + (size_t)n * SH * SW * CSTR + c;
    const float v00 = xp[((size_t)y0 * SW + x0) * CSTR], v01 = xp[((size_t)y0 * SW + x1) * CSTR];
    const float v10 = xp[((size_t)y1 * SW + x0) * CSTR], v11 = xp[((size_t)y1 * SW + x1) * CSTR];
    y[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

}  // namespace

extern "C" int tocvp_conv3x3_f32(const float* x, const float* wp, const float* scale,
                                 const float* shift, float* y, int nimg, int H, int W, int Cin,
                                 int Cout, int relu, int upsample2, void* stream) {
    TOCVP_CHECK_ARG(x && wp && shift && y);
    TOCVP_CHECK_ARG(nimg >= 0 && H > 0 && W > 0 && (H % TH) == 0);
    TOCVP_CHECK_ARG(Cin > 0 && (Cin % CC) == 0 && Cout > 0 && (Cout % 32) == 0);
    TOCVP_CHECK_ARG(!upsample2 || ((H % 2) == 0 && (W % 2) == 0));
    const size_t tiles = (size_t)((W + TW - 1) / TW) * (H / TH);
    TOCVP_CHECK_ARG(nimg * tiles < 0x7fffffffu && Cout / 32 <= 65535);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(wp)) return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    Conv3Args a{x, wp, scale, shift, y, nimg, H, W, Cin, Cout, relu, upsample2 ? 1 : 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Cout % 64 == 0)
        hipLaunchKernelGGL(conv3x3_mfma_kernel<2>, dim3((unsigned)(nimg * tiles), Cout / 64), dim3(256),
                           0, s, a);
    else
        hipLaunchKernelGGL(conv3x3_mfma_kernel<1>, dim3((unsigned)(nimg * tiles), Cout / 32), dim3(256),
                           0, s, a);
    return tocvp_launch_status();
}

// ``wp`` of the three f16x3 entries below: fp16 planes of 2^10 w in MFMA-fragment order (tocvp_split_weights_frag_f16 on the packed
// (taps x Cout, Cin) matrix), passed through the float pointer of Conv3Args
static int launch_convk_f16x3(const float* x, const float* wp, const float* scale, const float* shift,
                              float* y, int nimg, int H, int W, int Cin, int Cout, int relu, int upsample2,
                              int ksize, void* stream) {
    TOCVP_CHECK_ARG(x && wp && shift && y);
    TOCVP_CHECK_ARG(ksize == 3 || ksize == 5);
    TOCVP_CHECK_ARG(nimg >= 0 && H > 0 && W > 0 && (H % TH) == 0);
    TOCVP_CHECK_ARG(Cin > 0 && (Cin % CH) == 0 && Cout > 0 && (Cout % 32) == 0);
    TOCVP_CHECK_ARG(!upsample2 || ((H % 2) == 0 && (W % 2) == 0));
    const size_t tiles = (size_t)((W + TW - 1) / TW) * (H / TH);
    TOCVP_CHECK_ARG(nimg * tiles < 0x7fffffffu && Cout / 32 <= 65535);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(wp)) return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    Conv3Args a{x, wp, scale, shift, y, nimg, H, W, Cin, Cout, relu, upsample2 ? 1 : 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 g64((unsigned)(nimg * tiles), Cout / 64), g32((unsigned)(nimg * tiles), Cout / 32);
    if (ksize == 3 && W <= 16 && (H % 16) == 0) {                  // narrow images: 16 x 16 tiles (two rows per MFMA block)
        const unsigned nt = (unsigned)(nimg * (size_t)(H / 16));
        if (Cout % 64 == 0) hipLaunchKernelGGL((convk_f16x3_kernel<2, 3, true>), dim3(nt, Cout / 64), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((convk_f16x3_kernel<1, 3, true>), dim3(nt, Cout / 32), dim3(256), 0, s, a);
    } else if (ksize == 3) {
        if (Cout % 64 == 0) hipLaunchKernelGGL((convk_f16x3_kernel<2, 3>), g64, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((convk_f16x3_kernel<1, 3>), g32, dim3(256), 0, s, a);
    } else {
        if (Cout % 64 == 0) hipLaunchKernelGGL((convk_f16x3_kernel<2, 5>), g64, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((convk_f16x3_kernel<1, 5>), g32, dim3(256), 0, s, a);
    }
    return tocvp_launch_status();
}

extern "C" int tocvp_conv3x3_f16x3_f32(const float* x, const void* wp, const float* scale,
                                       const float* shift, float* y, int nimg, int H, int W, int Cin,
                                       int Cout, int relu, int upsample2, void* stream) {
    return launch_convk_f16x3(x, static_cast<const float*>(wp), scale, shift, y, nimg, H, W, Cin, Cout, relu, upsample2, 3, stream);
}

extern "C" int tocvp_conv3x3_up2_f16x3_f32(const float* x, const void* wphase, const float* scale, const float* shift,
                                           float* y, int nimg, int SH, int SW, int Cin, int Cout, int relu, void* stream) {
    TOCVP_CHECK_ARG(x && wphase && shift && y);
    TOCVP_CHECK_ARG(nimg >= 0 && SH > 0 && SW > 0 && (SH % TH) == 0);
    TOCVP_CHECK_ARG(Cin > 0 && (Cin % CH) == 0 && Cout > 0 && (Cout % 32) == 0);
    const size_t tiles = (size_t)((SW + TW - 1) / TW) * (SH / TH);
    TOCVP_CHECK_ARG(nimg * tiles < 0x7fffffffu && Cout / 32 <= 65535);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(wphase)) return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    Conv3Args a{x, static_cast<const float*>(wphase), scale, shift, y, nimg, SH, SW, Cin, Cout, relu, 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (SW <= 16 && (SH % 16) == 0) {                              // narrow source images: 16 x 16 tiles
        const unsigned nt = (unsigned)(nimg * (size_t)(SH / 16));
        if (Cout % 64 == 0) hipLaunchKernelGGL((convk_f16x3_kernel<2, 2, true>), dim3(nt, Cout / 64, 4), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((convk_f16x3_kernel<1, 2, true>), dim3(nt, Cout / 32, 4), dim3(256), 0, s, a);
    } else if (Cout % 64 == 0)
        hipLaunchKernelGGL((convk_f16x3_kernel<2, 2>), dim3((unsigned)(nimg * tiles), Cout / 64, 4), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((convk_f16x3_kernel<1, 2>), dim3((unsigned)(nimg * tiles), Cout / 32, 4), dim3(256), 0, s, a);
    return tocvp_launch_status();
}

extern "C" int tocvp_conv5x5_f16x3_f32(const float* x, const void* wp, const float* bias, float* y,
                                       int nimg, int H, int W, int Cin, int Cout, int relu,
                                       void* stream) {
    return launch_convk_f16x3(x, static_cast<const float*>(wp), nullptr, bias, y, nimg, H, W, Cin, Cout, relu, 0, 5, stream);
}

extern "C" int tocvp_slot_composite_f32(const float* decoded, float* recons, float* masks, int B,
                                        int K, int N, int F, int ld, void* stream) {
    TOCVP_CHECK_ARG(decoded && recons && masks);
    TOCVP_CHECK_ARG(B >= 0 && B <= 65535 && K > 0 && K <= 64 && N > 0 && F > 0 && ld >= F + 1);
    if (B == 0) return TOCVP_OK;
    hipLaunchKernelGGL(slot_composite_kernel, dim3(N, B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), decoded, recons, masks, K, N, F, ld);
    return tocvp_launch_status();
}

extern "C" int tocvp_bilinear_resize_f32(const float* x, float* y, int n, int C, int cstride, int SH,
                                         int SW, int OH, int OW, void* stream) {
    TOCVP_CHECK_ARG(x && y && n >= 0 && C > 0 && cstride >= C && SH > 0 && SW > 0 && OH > 0 && OW > 0);
    const long total = (long)n * C * OH * OW;
    if (total == 0) return TOCVP_OK;
    hipLaunchKernelGGL(bilinear_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, y, C, cstride, SH, SW, OH, OW, total);
    return tocvp_launch_status();
}
