// Text cross-attention of the predictor blocks, COLLAPSED over the caption (round 3).
//
// Replaces, per AdaptedEncoderBlock and rollout step (reference models/Blocks/attention.py:445-463, 303-319):
//     z = x + out_projection( softmax_t( q K^T / sqrt(dh) ) V ) ,  q = LayerNorm(x) Wq^T
// i.e. LayerNorm -> 512x512 GEMM -> attention over Lt caption tokens -> 512x512 GEMM (+ bias + residual): four
// kernels, 2 x 2 x 512 x 512 FLOP per token.  Keys and values depend on the caption only, so both projections fold
// into per-sample operands that are built ONCE per rollout (TransformerDecoderBlock.project_text):
//     G_b[h, t, :]  = scale * sum_d Wq[h dh + d, :] K_b[t, h dh + d]         (H Lt x E)   scores  = LN(x) G_b^T
//     HT_b[:, h, t] =         sum_d Wo[:, h dh + d] V_b[t, h dh + d]         (E x H Lt)   output  = P HT_b^T
// -- the same numbers up to fp32 re-association, with H Lt = 8 x 12 columns instead of 512: 2 x 2 x 512 x 128 FLOP per
// token (tokens per head padded to 16) and ONE kernel.  Padded caption positions take part as in the reference
// (no key mask, attention.py:314); slots t >= Lt of a head are excluded from its softmax.
//
// One workgroup (8 waves) = 64 tokens of one sample, 34 KB of LDS, ~100 registers -> four workgroups per CU:
//   phase 0  LayerNorm statistics of the 64 rows (the arithmetic of layernorm_kernel, misc.hip);
//   phase 1  S^T = G_b LN(x)^T on the f16 matrix cores (three products per operand pair, fp32-class), LN(x) staged in
//            four chunks of 128 columns as fp16 operand planes (2^8 x = hi + lo, rows padded to 528 B: conflict-free
//            ds_read_b128); a wave owns the 32 score rows of two heads for 32 tokens; G fragments come straight from
//            L2 in MFMA-fragment order; with the caption slots on the accumulator ROWS the softmax over t is in-lane
//            plus one cross-half exchange;
//   phase 2  P goes back through LDS as operand planes (over the dead LN image), Y^T = HT_b P^T: a wave owns 64
//            output columns of the 64 tokens; a lane holds four consecutive columns of one token per register quad,
//            so bias, residual and the 16-byte stores come straight from the accumulators.
// (The first version -- 4 waves, 32 tokens, the whole 512-column LN image in 66 KB of LDS, two workgroups per CU,
// 512 KB of caption operands per 32 tokens -- measured 127 us at 128 x 300 tokens; this one:  see profiles/r03_xattn.md.)
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int E = 512, HEADS = 8;
constexpr int LP = 16, NP = HEADS * LP;                              // 128 padded score columns (captions of <= 16 tokens)
constexpr int TOK = 64;                                              // tokens per workgroup
constexpr int KC = 128;                                              // LayerNorm image: columns per chunk
constexpr int XROW = 2 * KC * 2 + 16;                                // [row][plane][KC] + 16 B pad = 528 B
constexpr int PROW = 2 * NP * 2 + 16;                                // P image: [row][plane][n] + 16 B pad = 528 B
constexpr float SA = TOCVP_F16X3_ACT_SCALE, SW = TOCVP_F16X3_WEIGHT_SCALE;
static_assert(XROW == PROW, "the P image overlays the LayerNorm image");
// Round 4: captions of 17-32 tokens take LPT = 32 caption slots per head (one 32-row score tile per head, 256 padded
// score columns: still half the 512 x 512 projections' work); the text encoder admits 50 tokens
// (text_encoders.py:36) -- 33-50 would need 64 slots per head = the uncollapsed width, they keep the four-kernel path.

struct XArgs {
    const float* x; const float* gamma; const float* beta; float eps;
    const _Float16* Gf; const _Float16* Hf; const float* bias; float* y;
    int B, Tq, Lt; float scale;
};

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// 8 waves, 64 tokens of one sample.  34 KB of LDS and ~100 registers: four workgroups (32 waves) per CU -- the kernel
// is a chain of short dependent phases, what hides their latency is other workgroups.
// LPT = 64 (round 4, captions of 33-50 tokens -- the text encoder admits 50, text_encoders.py:36): 64 caption slots per
// head = the uncollapsed width (no FLOP saving), but still ONE kernel instead of LayerNorm + q GEMM + attention + output
// GEMM; a wave owns the TWO 32-row score tiles of its head, the P image takes 132 KB of LDS (one workgroup per CU).
template <int LPT>
__global__ __launch_bounds__(512, LPT == 16 ? 6 : (LPT == 32 ? 4 : 2)) void xattn_collapsed_kernel(XArgs p, int gx) {
    constexpr int NPT = HEADS * LPT;                                 // padded score columns
    constexpr int PROWT = 2 * NPT * 2 + 16;                          // P image row: [plane][n] + 16 B pad
    constexpr int NMB = LPT == 16 ? 1 : 2;                           // token blocks per wave in phase 1
    constexpr int NSB = LPT == 64 ? 2 : 1;                           // 32-row score tiles per wave in phase 1
    __shared__ __attribute__((aligned(16))) unsigned char lds[TOK * (PROWT > XROW ? PROWT : XROW)];
    __shared__ float stats[TOK * 2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // Workgroups are dealt round-robin over the 8 XCDs (private L2 each): all gx workgroups of a sample get linear
    // ids with the same value mod 8, so a sample's 512 KB of caption operands are fetched into ONE L2 (dealt in
    // launch order they landed on five XCDs: 58 % L2 misses, 328 MB instead of 66 MB from beyond L2).
    const int L = blockIdx.x;
    const int b = (L / (8 * gx)) * 8 + (L & 7), r0 = ((L >> 3) % gx) * TOK;
    if (b >= p.B) return;
    const float* xb = p.x + (size_t)b * p.Tq * E;

    // ---- phase 0: LayerNorm statistics of rows 8 wave .. 8 wave + 7 (the arithmetic of layernorm_kernel, misc.hip;
    // rows past Tq repeat the last row and are never stored)
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
        const int rl = wave * 8 + i;
        const float* xr = xb + (size_t)min(r0 + rl, p.Tq - 1) * E;
        f32x4 v[2];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            v[u] = *reinterpret_cast<const f32x4*>(xr + (lane + 64 * u) * 4);
            s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
        }
        const float mean = wave_sum64(s) / (float)E;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 d = v[u] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        const float rstd = 1.0f / sqrtf(wave_sum64(q) / (float)E + p.eps);
        if (lane == 0) {
            stats[2 * rl] = mean;
            stats[2 * rl + 1] = rstd;
        }
    }
    __syncthreads();

    // ---- phase 1: S^T = G LN(x)^T, LN(x) staged in chunks of 128 columns as fp16 operand planes.
    // LPT = 16: wave -> score rows n = 32 (wave & 3) + .. (heads 2 (wave & 3), + 1), tokens m = 32 (wave >> 2) + ..
    // LPT = 32: wave -> the 32 caption slots of head `wave`, both token blocks (one G fragment feeds two products)
    // LPT = 64: wave -> the 64 caption slots of head `wave` = score row blocks 2 wave, 2 wave + 1, both token blocks
    const int nb = LPT == 16 ? (wave & 3) : (LPT == 32 ? wave : 2 * wave), mb0 = LPT == 16 ? (wave >> 2) : 0;
    f32x16 sacc[NSB][NMB];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int i = 0; i < NMB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[sb][i][r] = 0.f;
    const f16x8* gf = reinterpret_cast<const f16x8*>(p.Gf) + ((size_t)(b * (NPT / 32) + nb) * (E / 16) * 2) * 64 + lane;
    constexpr size_t GBLK = (size_t)(E / 16) * 2 * 64;               // f16x8 elements per 32-row block of G
    const unsigned char* xl = lds + (32 * mb0 + l31) * XROW + hh * 16;
#pragma unroll 1
    for (int ch = 0; ch < E / KC; ++ch) {
        if (ch > 0) __syncthreads();                  // every wave has finished reading the previous chunk
#pragma unroll
        for (int i = 0; i < (TOK * KC / 4) / 512; ++i) {
            const int idx = t + 512 * i;
            const int rl = idx / (KC / 4), c = ch * KC + (idx % (KC / 4)) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (size_t)min(r0 + rl, p.Tq - 1) * E + c);
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + c);
            const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
            const float mean = stats[2 * rl], rstd = stats[2 * rl + 1];
            const f32x4 o = (v - mean) * rstd * g + be;
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sc = __builtin_amdgcn_fmed3f(o[e] * SA, -65504.f, 65504.f);
                hi[e] = (_Float16)sc;
                lo[e] = (_Float16)(sc - (float)hi[e]);
            }
            *reinterpret_cast<f16x4*>(lds + rl * XROW + (c - ch * KC) * 2) = hi;
            *reinterpret_cast<f16x4*>(lds + rl * XROW + KC * 2 + (c - ch * KC) * 2) = lo;
        }
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < KC / 16; ++ks) {
            f16x8 wh[NSB], wl[NSB];
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                wh[sb] = gf[sb * GBLK + (size_t)(ch * (KC / 16) + ks) * 128];
                wl[sb] = gf[sb * GBLK + (size_t)(ch * (KC / 16) + ks) * 128 + 64];
            }
#pragma unroll
            for (int i = 0; i < NMB; ++i) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(xl + i * 32 * XROW + ks * 32);
                const f16x8 al = *reinterpret_cast<const f16x8*>(xl + i * 32 * XROW + KC * 2 + ks * 32);
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) {
                    sacc[sb][i] = mfma16(wh[sb], al, sacc[sb][i]);
                    sacc[sb][i] = mfma16(wl[sb], ah, sacc[sb][i]);
                    sacc[sb][i] = mfma16(wh[sb], ah, sacc[sb][i]);
                }
            }
        }
    }
    // softmax over the caption slots of each head.  Register r of lane half hh = score row (r & 3) + 8 (r >> 2) + 4 hh
    // of the tile; the other half of the rows sits in lane ^ 32.  LPT = 16: rows 0-15 / 16-31 = heads 2 nb / 2 nb + 1
    // (registers 0-7 / 8-15); LPT = 32: the tile is one head
    // (LPT = 64: the head's slots are rows 0-31 of tile sb = 0 and of tile sb = 1: slot 32 sb + row)
    float pr[NSB][NMB][16];
    constexpr int NG = LPT >= 32 ? 1 : 2, RG = 16 / NG;   // heads per tile, registers per head, tile and lane
#pragma unroll
    for (int i = 0; i < NMB; ++i)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float mx = -3.0e38f;
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
                for (int q = 0; q < RG; ++q) {
                    const int tt = 32 * sb + 4 * hh + (q & 3) + 8 * (q >> 2);
                    const float sv = sacc[sb][i][RG * g + q] * p.scale;
                    pr[sb][i][RG * g + q] = sv;
                    if (tt < p.Lt) mx = fmaxf(mx, sv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
                for (int q = 0; q < RG; ++q) {
                    const int tt = 32 * sb + 4 * hh + (q & 3) + 8 * (q >> 2);
                    const float e = tt < p.Lt ? expf(pr[sb][i][RG * g + q] - mx) : 0.f;
                    pr[sb][i][RG * g + q] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
                for (int q = 0; q < RG; ++q) pr[sb][i][RG * g + q] *= inv;
        }
    __syncthreads();                                  // every wave has finished reading the LN image
    // P as operand planes: row m = 32 (mb0 + i) + l31, column n = 32 (nb + sb) + 8 u + 4 hh + {0..3} for register quad u
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
        for (int i = 0; i < NMB; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sc = pr[sb][i][4 * u + e] * SA;
                    hi[e] = (_Float16)sc;
                    lo[e] = (_Float16)(sc - (float)hi[e]);
                }
                const int n = 32 * (nb + sb) + 8 * u + 4 * hh;
                *reinterpret_cast<f16x4*>(lds + (32 * (mb0 + i) + l31) * PROWT + n * 2) = hi;
                *reinterpret_cast<f16x4*>(lds + (32 * (mb0 + i) + l31) * PROWT + NPT * 2 + n * 2) = lo;
            }
    __syncthreads();

    // ---- phase 2: Y^T tiles of this wave, one 32-column block at a time (keeps the kernel at 3 workgroups per CU):
    // rows c = 64 wave + 32 j + .., columns m = 32 i + ..; register quad q of tile (j, i) = columns
    // 64 wave + 32 j + 8 q + 4 hh .. + 3 of token r0 + 32 i + l31
    const unsigned char* pl = lds + l31 * PROWT + hh * 16;
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
        f32x16 yacc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[i][r] = 0.f;
        const f16x8* hf = reinterpret_cast<const f16x8*>(p.Hf) +
                          ((size_t)(b * (E / 32) + wave * 2 + j) * (NPT / 16) * 2) * 64 + lane;
#pragma unroll 2
        for (int ks = 0; ks < NPT / 16; ++ks) {
            const f16x8 wh = hf[ks * 128], wl = hf[ks * 128 + 64];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(pl + i * 32 * PROWT + ks * 32);
                const f16x8 al = *reinterpret_cast<const f16x8*>(pl + i * 32 * PROWT + NPT * 2 + ks * 32);
                yacc[i] = mfma16(wh, al, yacc[i]);
                yacc[i] = mfma16(wl, ah, yacc[i]);
                yacc[i] = mfma16(wh, ah, yacc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = r0 + 32 * i + l31;
            if (row < p.Tq) {
                const float* xr = xb + (size_t)row * E;
                float* yr = p.y + ((size_t)b * p.Tq + row) * E;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = 64 * wave + 32 * j + 8 * q + 4 * hh;
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + c);
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(xr + c);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = yacc[i][4 * q + e] * (1.f / (SA * SW)) + bv[e] + rv[e];
                    *reinterpret_cast<f32x4*>(yr + c) = v;
                }
            }
        }
    }
}

// Small problems (fewer than one workgroup of the 64-token kernel per CU): 4 waves, 32 tokens, the whole 512-column
// LayerNorm image in LDS (66 KB), no chunk loop -- a shorter chain of phases per workgroup (one sequence, 300 tokens:
// 19 us against 31 us), twice the caption-operand traffic per token (irrelevant at this size).
constexpr int XROW1 = 2 * E * 2 + 16;                                // [row][plane][512] + 16 B pad = 2064 B
__global__ __launch_bounds__(256, 2) void xattn_collapsed_small_kernel(XArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32 * XROW1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y, r0 = blockIdx.x * 32;
    const float* xb = p.x + (size_t)b * p.Tq * E;

    // The caption operands do not depend on anything computed here: the first eight k-steps of this wave's G fragments
    // take off before the LayerNorm and the ring stays eight steps ahead of the MFMAs (one step ahead, each of the 32
    // dependent k-steps waited a full L2 round trip: 19 of the kernel's 28 us at one sequence)
    constexpr int GRING = 8;
    const f16x8* gf = reinterpret_cast<const f16x8*>(p.Gf) + ((size_t)(b * (NP / 32) + wave) * (E / 16) * 2) * 64 + lane;
    f16x8 gh[GRING], gl[GRING];
#pragma unroll
    for (int d = 0; d < GRING; ++d) {
        gh[d] = gf[(size_t)d * 128];
        gl[d] = gf[(size_t)d * 128 + 64];
    }

    // ---- phase 0: LayerNorm of rows 8 wave .. 8 wave + 7 (rows past Tq repeat the last row; never stored)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rl = wave * 8 + i;
        const float* xr = xb + (size_t)min(r0 + rl, p.Tq - 1) * E;
        f32x4 v[2];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            v[u] = *reinterpret_cast<const f32x4*>(xr + (lane + 64 * u) * 4);
            s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
        }
        const float mean = wave_sum64(s) / (float)E;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 d = v[u] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        const float rstd = 1.0f / sqrtf(wave_sum64(q) / (float)E + p.eps);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = (lane + 64 * u) * 4;
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + c);
            const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
            f32x4 o = (v[u] - mean) * rstd * g + be;
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sc = __builtin_amdgcn_fmed3f(o[e] * SA, -65504.f, 65504.f);
                hi[e] = (_Float16)sc;
                lo[e] = (_Float16)(sc - (float)hi[e]);
            }
            *reinterpret_cast<f16x4*>(lds + rl * XROW1 + c * 2) = hi;
            *reinterpret_cast<f16x4*>(lds + rl * XROW1 + E * 2 + c * 2) = lo;
        }
    }
    __syncthreads();

    // ---- phase 1: S^T tile of this wave: rows n = 32 wave + .. (heads 2 wave, 2 wave + 1), columns m = 32 tokens
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    {
        const unsigned char* xl = lds + l31 * XROW1 + hh * 16;
#pragma unroll
        for (int ks = 0; ks < E / 16; ++ks) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(xl + ks * 32);
            const f16x8 al = *reinterpret_cast<const f16x8*>(xl + E * 2 + ks * 32);
            const f16x8 wh = gh[ks % GRING], wl = gl[ks % GRING];
            if (ks + GRING < E / 16) {
                gh[ks % GRING] = gf[(size_t)(ks + GRING) * 128];
                gl[ks % GRING] = gf[(size_t)(ks + GRING) * 128 + 64];
            }
            sacc = mfma16(wh, al, sacc);
            sacc = mfma16(wl, ah, sacc);
            sacc = mfma16(wh, ah, sacc);
        }
    }
    // the output-projection fragments of the first k-steps fly under the softmax and the two barriers behind it
    constexpr int HRING = 3;
    f16x8 hw[HRING][4][2];
    auto load_h = [&](int slot, int ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f16x8* hf = reinterpret_cast<const f16x8*>(p.Hf) +
                              (((size_t)(b * (E / 32) + wave * 4 + j) * (NP / 16) + ks) * 2) * 64 + lane;
            hw[slot][j][0] = hf[0];
            hw[slot][j][1] = hf[64];
        }
    };
#pragma unroll
    for (int d = 0; d < HRING; ++d) load_h(d, d);
    // softmax over the caption slots of each head: register r of lane half hh = slot t = 4 hh + (r & 3) + 8 ((r >> 2) & 1)
    // of head 2 wave + (r >> 3); the other half of the slots sits in lane ^ 32
    float pr[16];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float mx = -3.0e38f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int tt = 4 * hh + (q & 3) + 8 * (q >> 2);
            const float sv = sacc[8 * g + q] * p.scale;
            pr[8 * g + q] = sv;
            if (tt < p.Lt) mx = fmaxf(mx, sv);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int tt = 4 * hh + (q & 3) + 8 * (q >> 2);
            const float e = tt < p.Lt ? expf(pr[8 * g + q] - mx) : 0.f;
            pr[8 * g + q] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int q = 0; q < 8; ++q) pr[8 * g + q] *= inv;
    }
    __syncthreads();                                  // every wave has finished reading the LN image
    // P as operand planes: row m = l31, column n = 32 wave + 16 g + 8 u + 4 hh + {0..3}
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sc = pr[8 * g + 4 * u + e] * SA;
                hi[e] = (_Float16)sc;
                lo[e] = (_Float16)(sc - (float)hi[e]);
            }
            const int n = 32 * wave + 16 * g + 8 * u + 4 * hh;
            *reinterpret_cast<f16x4*>(lds + l31 * PROW + n * 2) = hi;
            *reinterpret_cast<f16x4*>(lds + l31 * PROW + NP * 2 + n * 2) = lo;
        }
    __syncthreads();

    // ---- phase 2: Y^T tiles of this wave: rows c = 128 wave + 32 j + .., columns m
    f32x16 yacc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[j][r] = 0.f;
    {
        const unsigned char* pl = lds + l31 * PROW + hh * 16;
#pragma unroll
        for (int ks = 0; ks < NP / 16; ++ks) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(pl + ks * 32);
            const f16x8 al = *reinterpret_cast<const f16x8*>(pl + NP * 2 + ks * 32);
            f16x8 wh[4], wl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wh[j] = hw[ks % HRING][j][0];
                wl[j] = hw[ks % HRING][j][1];
            }
            if (ks + HRING < NP / 16) load_h(ks % HRING, ks + HRING);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                yacc[j] = mfma16(wh[j], al, yacc[j]);
                yacc[j] = mfma16(wl[j], ah, yacc[j]);
                yacc[j] = mfma16(wh[j], ah, yacc[j]);
            }
        }
    }
    // ---- epilogue: register quad q of tile j = columns 128 wave + 32 j + 8 q + 4 hh .. + 3 of token r0 + l31
    const int row = r0 + l31;
    if (row < p.Tq) {
        const float* xr = xb + (size_t)row * E;
        float* yr = p.y + ((size_t)b * p.Tq + row) * E;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 128 * wave + 32 * j + 8 * q + 4 * hh;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + c);
                const f32x4 rv = *reinterpret_cast<const f32x4*>(xr + c);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = yacc[j][4 * q + e] * (1.f / (SA * SW)) + bv[e] + rv[e];
                *reinterpret_cast<f32x4*>(yr + c) = v;
            }
    }
}


}  // namespace

extern "C" int tocvp_xattn_collapsed_f32(const float* x, const float* gamma, const float* beta, float eps,
                                         const void* Gfrag, const void* Hfrag, const float* bias, float* y, int B,
                                         int Tq, int E_, int heads, int Lt, float scale, void* stream) {
    TOCVP_CHECK_ARG(x && gamma && beta && Gfrag && Hfrag && bias && y);
    TOCVP_CHECK_ARG(B >= 0 && Tq >= 0 && E_ == E && heads == HEADS && Lt >= 1 && Lt <= 2 * LP);    // 33-64 tokens (the 64-slot form of round 4, equal to the four-kernel path: retired)
    if (!tocvp_aligned16(x) || !tocvp_aligned16(y) || !tocvp_aligned16(gamma) || !tocvp_aligned16(beta) ||
        !tocvp_aligned16(bias) || !tocvp_aligned16(Gfrag) || !tocvp_aligned16(Hfrag))
        return TOCVP_EALIGN;
    if (B == 0 || Tq == 0) return TOCVP_OK;
    XArgs p{x, gamma, beta, eps, static_cast<const _Float16*>(Gfrag), static_cast<const _Float16*>(Hfrag), bias, y,
            B, Tq, Lt, scale * (1.f / (SA * SW))};
    const int gx = (Tq + TOK - 1) / TOK;
    static const int ncu = []() {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    if (Lt > LP) {            // 17-32 caption tokens: 32 slots per head (operands built with that padding)
        hipLaunchKernelGGL(xattn_collapsed_kernel<32>, dim3((unsigned)(((B + 7) / 8) * 8 * gx)), dim3(512), 0,
                           static_cast<hipStream_t>(stream), p, gx);
        return tocvp_launch_status();
    }
    if ((long)B * gx < ncu && B <= 65535) {
        hipLaunchKernelGGL(xattn_collapsed_small_kernel, dim3((Tq + 31) / 32, B), dim3(256), 0,
                           static_cast<hipStream_t>(stream), p);
        return tocvp_launch_status();
    }
    hipLaunchKernelGGL(xattn_collapsed_kernel<16>, dim3((unsigned)(((B + 7) / 8) * 8 * gx)), dim3(512), 0,
                       static_cast<hipStream_t>(stream), p, gx);
    return tocvp_launch_status();
}
