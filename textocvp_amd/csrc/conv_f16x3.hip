// Decoder 5x5 convolution (64 -> 64 channels), SPLIT-fp16 operands ("f16x3"), fp32-class results.
//
// Replaces nn.Conv2d(64, 64, 5, padding=2) + ReLU of the reference's ConvDecoder
// (models/EncodersDecoders/decoders.py:96-110) for layers 1..3 of the spatial-broadcast decoder.
//
// Each fp32 operand is scaled by an exact power of two (X = 2^8 x, W = 2^10 w; the scale keeps the
// low plane out of the fp16 subnormals, which the matrix core flushes) and split into two fp16 planes
//     X = Xh + Xl,  Xh = f16(X), Xl = f16(X - Xh)       (22 significant bits together)
//     W = Wh + Wl
// The product is evaluated as  Xl Wh + Xh Wl + Xh Wh  (Xl Wl ~ 2^-22 dropped) by three
// v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator: ~2^-21 per product, i.e. fp32-class, where the
// f16 + e4m3 hybrid of conv_f16f8.hip leaves ~2^-15 (it failed the 1e-4 bar on weights with an O(1)
// RGB head: profiles/r02_parity_by_mode.md).  Valid for |x| < 255, |w| < 63 (saturating beyond).
//
// Geometry (that of conv_f16f8.hip): 8 x 64 pixel tile x 64 output channels per 4-wave workgroup;
// every wave owns 2 rows x 64 pixels x 64 channels = 4 x 2 accumulator tiles (128 VGPRs), so one
// weight fragment from L1/L2 feeds four pixel blocks.  Four passes of 16 input channels:
// LDS image 12 x 68 pixels x 80 B ([16 f16 Xh | 16 f16 Xl | 16 B pad], conflict-free for
// ds_read_b128) = 65 KB -> 2 workgroups per CU, one staging while the other multiplies.
// Weights in MFMA-fragment order straight from L1/L2, one tap (4 KiB) ahead in registers; the 25
// taps of a pass are fully unrolled (LDS offsets are immediates, no barrier inside a pass).
// Per tap and wave: 4 global 16-B loads, 24 MFMAs (768 matrix cycles), and 4.8 ds_read_b128 (operand fragments are
// shared between the wave's two output rows, TOCVP_CONV_ROWREUSE below; 8 without).
// fp32 NHWC (or the pass-major layout (n, 4, H, W, 16) between consecutive layers) in HBM.
#include <stdlib.h>
#include <string.h>

#include "common.h"

#ifndef TOCVP_ABLATE
#define TOCVP_ABLATE 0      // timing experiments only (scripts/probes/conv16_ablate.hip): 1 no weight loads, 4 no LDS
                            // operand reads, 5 no halo staging behind the first, 7 no MFMAs, 8 no output stores
#endif

// 1 (default since round 3): operand fragments one tap ahead, reads / weight loads woven between the MFMAs with
// sched_group_barrier; 4.24 -> 4.18 ms per 2040 slot images on dense random data (A/B in one process, two rounds,
// scripts/probes/conv16_ablate.hip -DLAYOUT=3): +1.4 %, same products in the same order (bit-identical)
// 1 (default since round 3): the tiles of one slot image share an XCD (halo rows become L2 hits): 4.120 -> 4.085 ms per
// 2040 slot images (A/B in one process, two rounds, scripts/probes/conv16_ablate.hip -DLAYOUT=3): +0.9 %
#ifndef TOCVP_CONV_XCD
#define TOCVP_CONV_XCD 1
#endif
#ifndef TOCVP_CONV_WEAVE
#define TOCVP_CONV_WEAVE 1
#endif
// 1 (default, round 3): operand fragments shared by the two output rows of a wave.  Input row q of the wave's 6-row
// window is the operand of output row 0 at tap row dy = q AND of output row 1 at dy = q - 1, so a pass walks (dx, q)
// -- 30 steps, 4 ds_read_b128 each = 120 per pass instead of 25 taps x 8 = 200 -- with the weights of tap (q, dx)
// kept one step longer for row 1 (three rolling register sets).  Same 600 MFMAs per pass; the 25 taps of an
// accumulator are added dx-major instead of dy-major (same products, another fp32 rounding order).  Measured: probe on
// dense random data 3.976 -> 3.924 ms per 2040 slot images (two rounds, scripts/probes/conv16_ablate.hip -DLAYOUT=3);
// in the bench, same box, 3.798 -> 3.692 ms (447.6 -> 460.5 TFLOP/s algorithmic; value_no_overlap 3607 -> 3676).
// 1: the six products of a pixel block are issued so that one operand stays the same between neighbours (Xl Wh0, Xl Wh1,
// Xh Wl1, Xh Wl0, Xh Wh0, Xh Wh1: five operand changes instead of nine; same order per accumulator, bit-identical):
// 4.055 / 4.061 -> 4.045 / 4.036 ms per 2040 slot images in the probe (+0.4 %, two rounds)
#ifndef TOCVP_CONV_MFMA_ORDER
#define TOCVP_CONV_MFMA_ORDER 1
#endif
#ifndef TOCVP_CONV_WDIST
#define TOCVP_CONV_WDIST 2        // steps between a weight fragment's load and its first use (round 4: 2, four register slots)
#endif
#ifndef TOCVP_CONV_READS_FIRST
#define TOCVP_CONV_READS_FIRST 0  // 1: every step fenced, the next step's operand reads behind its first MFMAs (22 instead of
                                  // 4 MFMAs between an LDS read and its use, two fragment sets live) -- measured SLOWER: 3.92 against
                                  // 3.60-3.65 ms per 2040 slot images in the bench (the fence stops the scheduler from overlapping
                                  // neighbouring steps)
#endif
#ifndef TOCVP_CONV_ROWREUSE
#define TOCVP_CONV_ROWREUSE 1
#endif

namespace {

constexpr int ABL = TOCVP_ABLATE;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 64, IH = TH + 4, IW = TW + 4;
constexpr int C = 64, CCH = 16, NPASS = 4, NTAP = 25;
constexpr int ROWB = 80, OFF_LO = 32;
constexpr float SA = TOCVP_F16X3_ACT_SCALE, SW = TOCVP_F16X3_WEIGHT_SCALE;
constexpr float F16MAX = 65504.f;
constexpr int FRAG = 1024;                    // one B fragment: 64 lanes x 16 B
constexpr int TAP_BYTES = 4 * FRAG;           // [plane(h, l)][nb(2)]

struct Args {
    const float* x; const float* aux; const unsigned char* wf; const float* bias; float* y;
    int nimg, H, W, relu;
    int pm_in, pm_out;      // pass-major activation layout (n, 4, H, W, 16) instead of NHWC (n, H, W, 64);
                            // pm_out == 2 (and the PLANES_IN kernel): the 64 bytes of a pixel and pass hold the
                            // operand planes [16 f16 Xh | 16 f16 Xl] of 2^8 x instead of 16 floats
    const unsigned char* tail_wf;   // != NULL: the decoder tail is folded into this layer's epilogue (below): y is then
                                    // the (n, 36, H, W) fp32 array of per-pixel tap products, not the 64-channel output
};

__device__ __forceinline__ int border_class(int p, int n) {
    return p < 2 ? p : (p >= n - 2 ? 4 - (n - 1 - p) : 2);
}

__device__ __forceinline__ float clampf(float v, float m) { return __builtin_amdgcn_fmed3f(v, -m, m); }

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// PLANES_IN (layers 2 and 3 behind a planes-writing layer): the halo tile is NOT converted here.  The producer's
// epilogue wrote, per pixel and 16-channel pass, the 64 bytes [Xh | Xl] this kernel's staging would compute from its
// fp32 output (same arithmetic: bit-identical results); they go global -> LDS by DMA (global_load_lds, 16 bytes per
// lane): the image is 4080 chunks of 16 bytes (816 pixels x [Xh 0-7 | Xh 8-15 | Xl 0-7 | Xl 8-15 | pad]), thread t
// moves chunks t, t + 256, ... with source offsets computed once per tile -- no staging registers, no conversion
// instructions, no ds_write; pixels outside the image are loaded from a clamped address and zeroed afterwards.
template <int MODE, bool PLANES_IN = false, bool TAILP = false>
__global__ __launch_bounds__(256, 2) void conv5x5_dec_f16x3_kernel(Args p) {
    constexpr int NT = 256;
    constexpr int SS = C + 4;                                       // padded floats per staged pixel
    constexpr int STAGE_BYTES = 4 * 64 * SS * 4;                    // one 64-pixel row per wave
    constexpr int NCHUNK = IH * IW * 5, NDMA = (NCHUNK + NT - 1) / NT;   // 4080 chunks, 16 DMA instructions per thread
    constexpr int IMG_BYTES = PLANES_IN ? NDMA * NT * 16 : IH * IW * ROWB;
    constexpr int WORK_BYTES = IMG_BYTES > STAGE_BYTES ? IMG_BYTES : STAGE_BYTES;
    // PLANES_IN: the 16 source offsets of a thread (16-byte units inside a pass plane, < 2^16) wait in LDS between
    // the passes -- in registers they would sit next to the 128 accumulators through every MFMA loop (spills)
    constexpr int LDS_BYTES = WORK_BYTES + (PLANES_IN ? NT * 32 : 0);
    static_assert(NDMA == 16, "two 16-byte records of eight 16-bit offsets per thread");
    static_assert(!(PLANES_IN && MODE != 0), "the collapsed layer synthesises its input");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    unsigned char* in_s = lds;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_x = p.W / TW, tiles = tiles_x * (p.H / TH);
#if TOCVP_CONV_XCD
    // Workgroups are dealt round-robin over the 8 XCDs (private L2 each).  The `tiles` tiles of one slot image share
    // halo rows (12 input rows for 8 output rows): give them linear ids with the same value mod 8, so the halo rows a
    // neighbour already fetched are L2 hits instead of a second trip to HBM.  Ids past the last image idle.
    const int img = (blockIdx.x / (8 * tiles)) * 8 + (blockIdx.x & 7), tile = (blockIdx.x >> 3) % tiles;
    if (img >= p.nimg) return;
#else
    const int img = blockIdx.x / tiles, tile = blockIdx.x % tiles;
#endif
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

    // LDS byte offset of the wave's accumulator tile m (tap (0,0), pixel l31, k-half h)
    int a_off[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) a_off[m] = ((2 * wave + (m >> 1)) * IW + (m & 1) * 32 + l31) * ROWB + h * 16;

    // B fragments of one tap: [plane][nb]; two register sets, tap t in set t & 1
    f16x8 bw[2][2][2];
    auto load_w = [&](int set, int q) {                            // q = pass * 25 + tap
        const unsigned char* base = p.wf + (size_t)q * TAP_BYTES + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                bw[set][pl][n] = *reinterpret_cast<const f16x8*>(base + (pl * 2 + n) * FRAG);
    };

    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    u16x8* const my_offs = reinterpret_cast<u16x8*>(lds + WORK_BYTES) + 2 * t;
    unsigned outside = 0;                       // bit i: chunk i of this thread lies outside the image (zeroed after the DMA)
    if (PLANES_IN) {
        u16x8 o[2];
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int g = i * NT + t, gc = min(g, NCHUNK - 1);
            const int pix = gc / 5, j = gc % 5;
            const int iy = ty0 + pix / IW - 2, ix = tx0 + pix % IW - 2;
            const bool inside = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
            o[i >> 3][i & 7] = (unsigned short)((iyc * p.W + ixc) * 4 + (j & 3));
            if (!inside && j < 4 && g < NCHUNK) outside |= 1u << i;
        }
        my_offs[0] = o[0];
        my_offs[1] = o[1];                       // read back by this thread only: no barrier needed
    }

    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass > 0) __syncthreads();          // every wave is done reading the previous image
        if (PLANES_IN) {
            const char* pbase = xbase + (size_t)pass * p.H * p.W * 64;                  // uniform
            const u16x8 o0 = my_offs[0], o1 = my_offs[1];
#pragma unroll
            for (int i = 0; i < NDMA; ++i)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(pbase + (unsigned)(i < 8 ? o0[i & 7] : o1[i & 7]) * 16u),
                    (__attribute__((address_space(3))) void*)(in_s + (i * 4 + wave) * 1024), 16, 0, 0);
            load_w(0, pass * NTAP);                  // first tap's weights fly across the barrier
        if (TOCVP_CONV_ROWREUSE && TOCVP_CONV_WDIST >= 2) load_w(1, pass * NTAP + 5);    // ... and tap (1, 0)'s
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (outside) {
#pragma unroll
                for (int i = 0; i < NDMA; ++i)
                    if ((outside >> i) & 1u)
                        *reinterpret_cast<f32x4*>(in_s + (i * NT + t) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
        } else {
        // ---- halo tile: fp32 -> (Xh | Xl) fp16 planes in LDS.  All global loads of a batch are issued
        // back to back from clamped (always valid) addresses and only then converted.  An opaque copy of
        // the thread index keeps the staging addresses per-pass temporaries (hoisted out of the pass
        // loop they would be spilled next to the 128 accumulator registers).
        int tq = t;
        asm volatile("" : "+v"(tq));
        constexpr int ITEMS = IH * IW * (CCH / 4);
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
                    f16x4 hi, lo;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float X = inside ? clampf(v[u] * SA, F16MAX) : 0.f;
                        hi[u] = (_Float16)X;
                        lo[u] = (_Float16)(X - (float)hi[u]);
                    }
                    unsigned char* dst = in_s + pix * ROWB + c * 2;
                    *reinterpret_cast<f16x4*>(dst) = hi;
                    *reinterpret_cast<f16x4*>(dst + OFF_LO) = lo;
                }
            }
        }
        load_w(0, pass * NTAP);                  // first tap's weights fly across the barrier
        if (TOCVP_CONV_ROWREUSE && TOCVP_CONV_WDIST >= 2) load_w(1, pass * NTAP + 5);    // ... and tap (1, 0)'s
        __syncthreads();
        }

#if TOCVP_CONV_ROWREUSE
        {
            f16x8 fa[2][2][2];                                      // [set][32-pixel half][plane]
            // rolling weights: tap n = 5 dx + (tap row) lives in slot n % NSLOT from its load, WD steps ahead of its first
            // use, to its second use one step later.  Round 3 loaded ONE step ahead (3 slots): in the compiled loop 16 of a
            // pass's 96 fragment loads were consumed by the very next MFMA and half of them within 10 MFMAs (~320 cycles
            // against an L2 round trip of 600+) -- each such wait stalls the wave, and only the partner wave of the SIMD
            // covers it.  Two steps ahead (4 slots, +16 registers) puts 24-36 MFMAs between load and use.
            constexpr int WD = TOCVP_CONV_WDIST, NSLOT = WD + 2;
            f16x8 w3[NSLOT][2][2];                                  // [slot][plane][n]
            auto read_rows = [&](int set, int q, int dx) {
                const unsigned char* a_base = in_s + (q * IW + dx) * ROWB;
#pragma unroll
                for (int xh = 0; xh < 2; ++xh) {
                    fa[set][xh][0] = *reinterpret_cast<const f16x8*>(a_base + a_off[xh]);
                    fa[set][xh][1] = *reinterpret_cast<const f16x8*>(a_base + a_off[xh] + OFF_LO);
                }
            };
            auto load_w3 = [&](int slot, int dy, int dx) {
                const unsigned char* base = p.wf + (size_t)(pass * NTAP + dy * 5 + dx) * TAP_BYTES + lane * 16;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        w3[slot][pl][n] = *reinterpret_cast<const f16x8*>(base + (pl * 2 + n) * FRAG);
            };
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    w3[0][pl][n] = bw[0][pl][n];                    // tap (0, 0) was fetched across the barrier
                    if (WD >= 2) w3[1][pl][n] = bw[1][pl][n];       // ... and tap (1, 0)
                }
            read_rows(0, 0, 0);
#pragma unroll
            for (int st = 0; st < 30; ++st) {
                const int dx = st / 6, q = st % 6, cur = st & 1;
                int nw = 0, nr = 0;
                {
                    const int s2 = st + WD, dx2 = s2 / 6, q2 = s2 % 6;      // the tap first used WD steps from now
                    if (s2 < 30 && q2 < 5) {
                        load_w3((5 * dx2 + q2) % NSLOT, q2, dx2);
                        nw = 4;
                    }
                }
                const int s0 = (5 * dx + q) % NSLOT, s1 = (5 * dx + q + NSLOT - 1) % NSLOT;   // slots of tap rows q, q - 1
                if (st + 1 < 30) {
                    read_rows(cur ^ 1, (st + 1) % 6, (st + 1) / 6);
                    nr = 4;
                }
                int nm = 0;
                if (q < 5) {                                        // output row 0, tap (q, dx)
#if TOCVP_CONV_MFMA_ORDER
#pragma unroll
                    for (int xh = 0; xh < 2; ++xh) {                // one operand stays put between neighbours
                        acc[xh][0] = mfma16(fa[cur][xh][1], w3[s0][0][0], acc[xh][0]);      // Xl Wh0
                        acc[xh][1] = mfma16(fa[cur][xh][1], w3[s0][0][1], acc[xh][1]);      // Xl Wh1
                        acc[xh][1] = mfma16(fa[cur][xh][0], w3[s0][1][1], acc[xh][1]);      // Xh Wl1
                        acc[xh][0] = mfma16(fa[cur][xh][0], w3[s0][1][0], acc[xh][0]);      // Xh Wl0
                        acc[xh][0] = mfma16(fa[cur][xh][0], w3[s0][0][0], acc[xh][0]);      // Xh Wh0
                        acc[xh][1] = mfma16(fa[cur][xh][0], w3[s0][0][1], acc[xh][1]);      // Xh Wh1
                    }
#else
#pragma unroll
                    for (int xh = 0; xh < 2; ++xh)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            acc[xh][n] = mfma16(fa[cur][xh][1], w3[s0][0][n], acc[xh][n]);      // Xl Wh
                            acc[xh][n] = mfma16(fa[cur][xh][0], w3[s0][1][n], acc[xh][n]);      // Xh Wl
                            acc[xh][n] = mfma16(fa[cur][xh][0], w3[s0][0][n], acc[xh][n]);      // Xh Wh
                        }
#endif
                    nm += 12;
                }
                if (q > 0) {                                        // output row 1, tap (q - 1, dx)
#if TOCVP_CONV_MFMA_ORDER
#pragma unroll
                    for (int xh = 0; xh < 2; ++xh) {
                        acc[2 + xh][0] = mfma16(fa[cur][xh][1], w3[s1][0][0], acc[2 + xh][0]);
                        acc[2 + xh][1] = mfma16(fa[cur][xh][1], w3[s1][0][1], acc[2 + xh][1]);
                        acc[2 + xh][1] = mfma16(fa[cur][xh][0], w3[s1][1][1], acc[2 + xh][1]);
                        acc[2 + xh][0] = mfma16(fa[cur][xh][0], w3[s1][1][0], acc[2 + xh][0]);
                        acc[2 + xh][0] = mfma16(fa[cur][xh][0], w3[s1][0][0], acc[2 + xh][0]);
                        acc[2 + xh][1] = mfma16(fa[cur][xh][0], w3[s1][0][1], acc[2 + xh][1]);
                    }
#else
#pragma unroll
                    for (int xh = 0; xh < 2; ++xh)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            acc[2 + xh][n] = mfma16(fa[cur][xh][1], w3[s1][0][n], acc[2 + xh][n]);
                            acc[2 + xh][n] = mfma16(fa[cur][xh][0], w3[s1][1][n], acc[2 + xh][n]);
                            acc[2 + xh][n] = mfma16(fa[cur][xh][0], w3[s1][0][n], acc[2 + xh][n]);
                        }
#endif
                    nm += 12;
                }
#if TOCVP_CONV_READS_FIRST
                // one memory instruction behind every MFMA: the operand reads of the NEXT step first (round 3 gave them the
                // slots behind the weight loads, and a 12-MFMA step has only six two-MFMA slots: half of its reads were left
                // to the scheduler, which put them right in front of their use -- median 4 MFMAs between an LDS read and the
                // MFMA that consumes it, minimum 0), then the weight fragments used two steps from now
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    if (i >= nm) break;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // MFMA
                    if (i < nr) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);              // operand read
                    else if (i - nr < nw) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // weight fragment load
                }
                // a step's reads and loads stay inside the step: without this fence the groups of step s + 1 capture the
                // reads written in step s, which then land right in front of their use (one register set instead of two)
                __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    if (2 * i >= nm) break;
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                          // MFMA
                    if (i < nw) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              // weight fragment load
                    else if (i - nw < nr) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // operand read
                }
#endif
            }
        }
#elif TOCVP_CONV_WEAVE
        // A fragments one tap ahead in a second register set, reads and weight loads woven between the MFMAs of
        // the current tap (sched_group_barrier): the LDS latency of a tap's eight operand reads no longer sits in
        // front of its first MFMA
        f16x8 ah[2][4], al[2][4];
        auto read_a = [&](int set, int tap) {
            const int dy = ABL == 4 ? 0 : tap / 5, dx = ABL == 4 ? 0 : tap % 5;
            const unsigned char* a_base = in_s + (dy * IW + dx) * ROWB;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                ah[set][m] = *reinterpret_cast<const f16x8*>(a_base + a_off[m]);
                al[set][m] = *reinterpret_cast<const f16x8*>(a_base + a_off[m] + OFF_LO);
            }
        };
        read_a(0, 0);
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int cur = tap & 1;
            if (ABL != 1) load_w(cur ^ 1, pass * NTAP + (tap + 1 < NTAP ? tap + 1 : tap));
            if (tap + 1 < NTAP) read_a(cur ^ 1, tap + 1);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[m][n] = mfma16(al[cur][m], bw[cur][0][n], acc[m][n]);      // Xl Wh
                    acc[m][n] = mfma16(ah[cur][m], bw[cur][1][n], acc[m][n]);      // Xh Wl
                    acc[m][n] = mfma16(ah[cur][m], bw[cur][0][n], acc[m][n]);      // Xh Wh
                }
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                              // MFMA
                if (i < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                   // weight fragment load
                else if (tap + 1 < NTAP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // operand read
            }
        }
#else
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int cur = tap & 1;
            // next tap's fragments (clamped at the last tap of the pass: a harmless re-load)
            if (ABL != 1) load_w(cur ^ 1, pass * NTAP + (tap + 1 < NTAP ? tap + 1 : tap));
            const int dy = ABL == 4 ? 0 : tap / 5, dx = ABL == 4 ? 0 : tap % 5;
            const unsigned char* a_base = in_s + (dy * IW + dx) * ROWB;
            f16x8 ah[4], al[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                ah[m] = *reinterpret_cast<const f16x8*>(a_base + a_off[m]);
                al[m] = *reinterpret_cast<const f16x8*>(a_base + a_off[m] + OFF_LO);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[m][n] = mfma16(al[m], bw[cur][0][n], acc[m][n]);      // Xl Wh
                    acc[m][n] = mfma16(ah[m], bw[cur][1][n], acc[m][n]);      // Xh Wl
                    acc[m][n] = mfma16(ah[m], bw[cur][0][n], acc[m][n]);      // Xh Wh
                }
        }
#endif
    }
    __syncthreads();                            // the halo image is dead: reuse it as the store stage

    // Epilogue through LDS, one 64-pixel output row of the wave at a time: the accumulator layout
    // gives a lane one channel of 16 pixels; staged, every store instruction writes 1 KiB of
    // contiguous output (4 pixels x 256 B NHWC, or 16 pixels x 64 B of one pass-major plane).
    float* stage = reinterpret_cast<float*>(lds) + wave * (64 * SS);
    constexpr float UNSCALE = 1.f / (SA * SW);
    if constexpr (TAILP) {
        // ---- decoder tail folded in (the last hidden layer).  The tail's 3 x 3 x 64 -> 4 convolution is linear: each
        // pixel's 64 outputs are multiplied HERE with the 36 x 64 tap matrix (36 = 9 taps x 4 outputs) and only the 36
        // products P[tap][out](pixel) leave the chip; the tail kernel then adds nine shifted planes.  Per 32-pixel
        // block of the wave: the outputs (bias + ReLU applied) go to a wave-private LDS image as fp16 planes
        // [channel][pixel] -- a lane holds one channel of four consecutive pixels per register quad: 8-byte stores --
        // and come back as the A operand through the hardware transpose (the product sums over the channel, the
        // LANE index of the accumulator); the tap matrix arrives in fragment order from L2; 24 MFMAs per block.
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4* lp4;
        constexpr int TIMG = 64 * 64;                                 // one plane: 64 channel rows x 32 pixels x 2 B
        constexpr int TPS = 36;                                       // floats per staged product row (32 pixels + pad)
        unsigned char* timg = lds + wave * (2 * TIMG + 36 * TPS * 4);
        float* pst = reinterpret_cast<float*>(timg + 2 * TIMG);
        const int i16 = lane & 15, c16 = ((lane >> 4) & 1) * 16;
        const unsigned char* trd = timg + (8 * h + (i16 >> 2)) * 64 + (c16 + 4 * (i16 & 3)) * 2;
        const f16x8* twf = reinterpret_cast<const f16x8*>(p.tail_wf) + lane;       // [nb 2][ks 4][plane 2][lane]
        f16x8 tb[2][4][2];                                            // the whole tap matrix of this lane: loaded once
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) tb[nb][ks][pl] = twf[((nb * 4 + ks) * 2 + pl) * 64];
        float* pout = p.y + (size_t)img * 36 * p.H * p.W;
#pragma unroll
        for (int m = 0; m < 4; ++m) {                                 // m = 2 * (row of the pair) + 32-pixel half
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float bv = p.bias[n * 32 + l31];
#pragma unroll
                for (int g = 0; g < 4; ++g) {                         // registers 4g .. 4g + 3 = pixels 8g + 4h .. + 3
                    f16x4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[m][n][4 * g + e] * UNSCALE + bv;
                        if (p.relu) v = fmaxf(v, 0.f);
                        const float X = clampf(v * SA, F16MAX);
                        hi[e] = (_Float16)X;
                        lo[e] = (_Float16)(X - (float)hi[e]);
                    }
                    unsigned char* d = timg + (n * 32 + l31) * 64 + (8 * g + 4 * h) * 2;
                    *reinterpret_cast<f16x4*>(d) = hi;
                    *reinterpret_cast<f16x4*>(d + TIMG) = lo;
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
                    const f16x8 bh = tb[nb][ks][0], bl = tb[nb][ks][1];
                    pacc[nb] = mfma16(al.f, bh, pacc[nb]);
                    pacc[nb] = mfma16(ah.f, bl, pacc[nb]);
                    pacc[nb] = mfma16(ah.f, bh, pacc[nb]);
                }
            }
            // products: lane = column (tap, out) = 32 nb + l31 (36 used), register quad g = pixels 8g + 4h .. + 3
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int to = nb * 32 + l31;
                if (to < 36) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(pst + to * TPS + 8 * g + 4 * h) =
                            f32x4{pacc[nb][4 * g] * UNSCALE, pacc[nb][4 * g + 1] * UNSCALE,
                                  pacc[nb][4 * g + 2] * UNSCALE, pacc[nb][4 * g + 3] * UNSCALE};
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int oy = ty0 + 2 * wave + (m >> 1), ox = tx0 + (m & 1) * 32;
#pragma unroll
            for (int it = 0; it < 5; ++it) {                          // 36 rows x 8 float4 = 288 pieces
                const int idx = lane + 64 * it;
                if (idx < 36 * 8) {
                    const int to = idx >> 3, c4 = (idx & 7) * 4;
                    *reinterpret_cast<f32x4*>(pout + ((size_t)to * p.H + oy) * p.W + ox + c4) =
                        *reinterpret_cast<const f32x4*>(pst + to * TPS + c4);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
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
        if (p.pm_out == 2) {
            // operand planes for the next layer: exactly the split its own staging would make of these values
            unsigned char* ybase = reinterpret_cast<unsigned char*>(p.y + (size_t)img * p.H * p.W * C);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int plane = it >> 2, px = (it & 3) * 16 + (lane >> 2), cq = (lane & 3) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + plane * CCH + cq);
                f16x4 hi, lo;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float X = clampf(v[u] * SA, F16MAX);
                    hi[u] = (_Float16)X;
                    lo[u] = (_Float16)(X - (float)hi[u]);
                }
                unsigned char* blk = ybase + (((size_t)plane * p.H + oy) * p.W + tx0 + px) * 64 + cq * 2;
                *reinterpret_cast<f16x4*>(blk) = hi;
                *reinterpret_cast<f16x4*>(blk + OFF_LO) = lo;
            }
        } else if (p.pm_out) {
            float* ybase = p.y + (size_t)img * p.H * p.W * C;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int plane = it >> 2, px = (it & 3) * 16 + (lane >> 2), cq = (lane & 3) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + plane * CCH + cq);
                *reinterpret_cast<f32x4*>(ybase + (((size_t)plane * p.H + oy) * p.W + tx0 + px) * CCH + cq) = v;
            }
        } else {
            float* yrow = p.y + (((size_t)img * p.H + oy) * p.W + tx0) * C;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int idx = lane + 64 * it;
                const int px = idx >> 4, c4 = (idx & 15) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + c4);
                *reinterpret_cast<f32x4*>(yrow + (size_t)px * C + c4) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// (64, 64, 5, 5) fp32 -> wf: [pass(4)][tap(25)][plane(Wh, Wl)][nb(2)][lane(64)][8 f16]
// lane (c = l & 31, hh = l >> 5): output channel nb*32 + c, input channels pass*16 + 8 hh + j of the tap.
__global__ __launch_bounds__(256) void split_conv_weights_dec_f16x3_kernel(const float* __restrict__ w,
                                                                           _Float16* __restrict__ wf) {
    const int i = blockIdx.x * 256 + threadIdx.x;                  // over 25 taps * 64 * 64
    if (i >= NTAP * C * C) return;
    const int ci = i % C, co = (i / C) % C, tap = i / (C * C);
    const float Wv = clampf(w[((size_t)co * C + ci) * NTAP + tap] * SW, F16MAX);
    const _Float16 hi = (_Float16)Wv;
    const _Float16 lo = (_Float16)(Wv - (float)hi);
    const int pass = ci / CCH, cc = ci % CCH, hh = cc >> 3, j = cc & 7;
    const int nb = co >> 5, c = co & 31;
    const size_t tapbase = ((size_t)pass * NTAP + tap) * 4;        // fragments of this (pass, tap)
    wf[((tapbase + 0 * 2 + nb) * 64 + hh * 32 + c) * 8 + j] = hi;
    wf[((tapbase + 1 * 2 + nb) * 64 + hh * 32 + c) * 8 + j] = lo;
}


// Tap matrix of the folded decoder tail: Conv2d(64 -> 4, k = 3) weights w (4, 64, 3, 3) as the B operand of the epilogue
// product, rows (tap, out) = 4 t + o (36 of 64 used), fp16 planes of 2^10 w in fragment order [nb 2][ks 4][plane 2][lane]:
// lane (column c = l & 31 -> row 32 nb + c, h = l >> 5) holds channels 16 ks + 8 h .. + 7.
__global__ __launch_bounds__(256) void pack_tail_taps_kernel(const float* __restrict__ w, _Float16* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;                   // one fp16 element of one plane pair
    if (i >= 2 * 4 * 64 * 8) return;
    const int e = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) & 3, nb = i >> 11;
    const int to = nb * 32 + (lane & 31), ch = ks * 16 + 8 * (lane >> 5) + e;
    float v = 0.f;
    if (to < 36) v = w[((size_t)(to & 3) * 64 + ch) * 9 + (to >> 2)];
    const float X = clampf(v * SW, F16MAX);
    const _Float16 hi = (_Float16)X, lo = (_Float16)(X - (float)hi);
    _Float16* base = out + (size_t)((nb * 4 + ks) * 2) * 512 + lane * 8 + e;
    base[0] = hi;
    base[512] = lo;
}

// Decoder tail over the tap products P (n = F K slot images, 36, H, W): rgba(p) = bias + sum_t P[4 t + o](p + off_t)
// (zero outside the image), softmax of alpha over the K slots, compositing -- the three outputs of dec_tail_kernel.
// One thread per pixel, 4 rows x 64 pixels per workgroup; every P value is read once, coalesced along x.
__global__ __launch_bounds__(256) void dec_tail_sum_kernel(const float* __restrict__ P, const float* __restrict__ bias,
                                                           float* __restrict__ recons_imgs, float* __restrict__ recons,
                                                           float* __restrict__ masks, float* __restrict__ clamped,
                                                           long img_fs, long rec_fs, long mask_fs, int K, int H, int W) {
    // frame f of the launch lands at recons_imgs + f img_fs, recons + f rec_fs, masks + f mask_fs (floats): the
    // evaluator's per-step decodes write straight into the (B, P, ...) result tensors at their step offset
    const int f = blockIdx.y;
    const int pix_lin = blockIdx.x * 256 + threadIdx.x;
    const size_t HW = (size_t)H * W;
    if (pix_lin >= H * W) return;
    const int y = pix_lin / W, x = pix_lin % W;
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2], b3 = bias[3];
    int off[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        ok[t] = yy >= 0 && yy < H && xx >= 0 && xx < W;
        off[t] = ok[t] ? yy * W + xx : pix_lin;
    }
    float* mp = masks + (size_t)f * mask_fs + pix_lin;
    float* rf = recons + (size_t)f * rec_fs + pix_lin;
    for (int k = 0; k < K; ++k) {
        const float* Pk = P + ((size_t)f * K + k) * 36 * HW;
        float v[4] = {b0, b1, b2, b3};
        float q[36];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int o = 0; o < 4; ++o) q[4 * t + o] = Pk[(size_t)(4 * t + o) * HW + off[t]];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int o = 0; o < 4; ++o) v[o] += ok[t] ? q[4 * t + o] : 0.f;
        float* ro = rf + (size_t)k * 3 * HW;
        ro[0] = v[0];
        ro[HW] = v[1];
        ro[2 * HW] = v[2];
        mp[(size_t)k * HW] = v[3];                                   // raw alpha, normalised in place below
    }
    float m = -1.0e30f;
    for (int k = 0; k < K; ++k) m = fmaxf(m, mp[(size_t)k * HW]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += expf(mp[(size_t)k * HW] - m);
    const float inv = 1.0f / sum;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int k = 0; k < K; ++k) {
        const float mk = expf(mp[(size_t)k * HW] - m) * inv;
        mp[(size_t)k * HW] = mk;
        const float* ro = rf + (size_t)k * 3 * HW;
        c0 += ro[0] * mk;
        c1 += ro[HW] * mk;
        c2 += ro[2 * HW] * mk;
    }
    float* co = recons_imgs + (size_t)f * img_fs + pix_lin;
    co[0] = c0;
    co[HW] = c1;
    co[2 * HW] = c2;
    if (clamped) {                                    // the evaluator's .clamp(0, 1) (05_evaluate_predictor.py:93-96)
        float* cc = clamped + (size_t)f * img_fs + pix_lin;
        cc[0] = tocvp_clamp01(c0);
        cc[HW] = tocvp_clamp01(c1);
        cc[2 * HW] = tocvp_clamp01(c2);
    }
}

}  // namespace

extern "C" size_t tocvp_conv_weights_dec_f16x3_bytes(void) { return (size_t)NPASS * NTAP * TAP_BYTES; }

extern "C" int tocvp_split_conv_weights_dec_f16x3(const float* w, void* wf, int Cout, int Cin, void* stream) {
    TOCVP_CHECK_ARG(w && wf && Cout == C && Cin == C);
    hipLaunchKernelGGL(split_conv_weights_dec_f16x3_kernel, dim3((NTAP * C * C + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, static_cast<_Float16*>(wf));
    return tocvp_launch_status();
}

extern "C" int tocvp_conv5x5_dec_f16x3_f32(const float* x, const float* aux, int in_mode, const void* wf,
                                           const float* bias, float* y, int nimg, int H, int W, int Cin,
                                           int Cout, int relu, int layout, void* stream) {
    // layout: bit 0 pass-major input, bit 1 pass-major output, bit 3 the pass-major buffers hold operand planes (written by /
    // read from a neighbouring layer of this kernel); bit 2 (the persistent form of rounds 2-4, retired in round 5: it
    // measured the same as this kernel) is refused
    TOCVP_CHECK_ARG(layout >= 0 && layout <= 15 && !(layout & 4) && !(in_mode == 1 && (layout & 1)));
    TOCVP_CHECK_ARG(!(layout & 8) || (layout & 3) != 0);
    // planes input: the DMA source offsets are kept as 16-bit counts of 16-byte pieces inside one pass plane
    TOCVP_CHECK_ARG((layout & 9) != 9 || (long)H * W <= 16384);
    TOCVP_CHECK_ARG(x && wf && bias && y);
    TOCVP_CHECK_ARG(in_mode == 0 || (in_mode == 1 && aux != nullptr));
    TOCVP_CHECK_ARG(Cin == C && Cout == C);
    TOCVP_CHECK_ARG(nimg >= 0 && H > 0 && W > 0 && (H % TH) == 0 && (W % TW) == 0);
    TOCVP_CHECK_ARG((size_t)nimg * (H / TH) * (W / TW) < 0x7fffffffu);
    // the staging addresses are 32-bit byte offsets from the image base
    TOCVP_CHECK_ARG((size_t)H * W * C * 4 < 0x7fffffffu);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(wf) || !tocvp_aligned16(y) || (aux && !tocvp_aligned16(aux)))
        return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    Args a{x, aux, static_cast<const unsigned char*>(wf), bias, y, nimg, H, W, relu, layout & 1,
           (layout & 2) ? ((layout & 8) ? 2 : 1) : 0};
    const int ntiles = (int)((size_t)nimg * (H / TH) * (W / TW));
    hipStream_t s = static_cast<hipStream_t>(stream);
#if TOCVP_CONV_XCD
    const dim3 grid((unsigned)((size_t)((nimg + 7) / 8) * 8 * (H / TH) * (W / TW)));
#else
    const dim3 grid((unsigned)ntiles);
#endif
    if (in_mode == 0 && (layout & 9) == 9)
        hipLaunchKernelGGL((conv5x5_dec_f16x3_kernel<0, true>), grid, dim3(256), 0, s, a);
    else if (in_mode == 0)
        hipLaunchKernelGGL((conv5x5_dec_f16x3_kernel<0, false>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((conv5x5_dec_f16x3_kernel<1, false>), grid, dim3(256), 0, s, a);
    return tocvp_launch_status();
}

/* ---- decoder tail folded into the last hidden layer -------------------------------------------------------------- */
extern "C" size_t tocvp_tail_taps_f16x3_bytes(void) { return (size_t)2 * 4 * 2 * 64 * 16; }

extern "C" int tocvp_pack_tail_taps_f16x3(const float* w, void* out, void* stream) {
    TOCVP_CHECK_ARG(w && out);
    if (!tocvp_aligned16(out)) return TOCVP_EALIGN;
    hipLaunchKernelGGL(pack_tail_taps_kernel, dim3((2 * 4 * 64 * 8 + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, static_cast<_Float16*>(out));
    return tocvp_launch_status();
}

extern "C" int tocvp_conv5x5_dec_f16x3_tail_f32(const float* x, const void* wf, const float* bias, const void* tail_taps,
                                                float* products, int nimg, int H, int W, int relu, int layout,
                                                void* stream) {
    // layout: bit 0 pass-major input, bit 3 operand planes in it (as tocvp_conv5x5_dec_f16x3_f32); the output is always
    // the (nimg, 36, H, W) array of tap products
    TOCVP_CHECK_ARG(x && wf && bias && tail_taps && products);
    TOCVP_CHECK_ARG((layout & ~9) == 0 && ((layout & 8) == 0 || (layout & 1)));
    TOCVP_CHECK_ARG(nimg >= 0 && H > 0 && W > 0 && (H % TH) == 0 && (W % TW) == 0);
    TOCVP_CHECK_ARG((size_t)nimg * (H / TH) * (W / TW) < 0x7fffffffu && (size_t)H * W * C * 4 < 0x7fffffffu);
    TOCVP_CHECK_ARG((layout & 9) != 9 || (long)H * W <= 16384);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(wf) || !tocvp_aligned16(products) || !tocvp_aligned16(tail_taps))
        return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    Args a{x, nullptr, static_cast<const unsigned char*>(wf), bias, products, nimg, H, W, relu, layout & 1, 0,
           static_cast<const unsigned char*>(tail_taps)};
    hipStream_t s = static_cast<hipStream_t>(stream);
#if TOCVP_CONV_XCD
    const dim3 grid((unsigned)((size_t)((nimg + 7) / 8) * 8 * (H / TH) * (W / TW)));
#else
    const dim3 grid((unsigned)((size_t)nimg * (H / TH) * (W / TW)));
#endif
    if ((layout & 9) == 9)
        hipLaunchKernelGGL((conv5x5_dec_f16x3_kernel<0, true, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((conv5x5_dec_f16x3_kernel<0, false, true>), grid, dim3(256), 0, s, a);
    return tocvp_launch_status();
}

extern "C" int tocvp_dec_tail_sum_placed_f32(const float* products, const float* bias, float* recons_imgs, float* recons,
                                             float* masks, float* clamped_imgs, long img_fs, long rec_fs, long mask_fs,
                                             int F, int K, int H, int W, void* stream) {
    TOCVP_CHECK_ARG(products && bias && recons_imgs && recons && masks);
    TOCVP_CHECK_ARG(F >= 0 && F <= 65535 && K > 0 && H > 0 && W > 0 && (long)H * W < (1L << 30));
    // frames must not overlap: strides at least one frame of each output
    TOCVP_CHECK_ARG(img_fs >= 3L * H * W && rec_fs >= 3L * K * H * W && mask_fs >= (long)K * H * W);
    if (F == 0) return TOCVP_OK;
    hipLaunchKernelGGL(dec_tail_sum_kernel, dim3((unsigned)((H * W + 255) / 256), F), dim3(256), 0,
                       static_cast<hipStream_t>(stream), products, bias, recons_imgs, recons, masks, clamped_imgs, img_fs,
                       rec_fs, mask_fs, K, H, W);
    return tocvp_launch_status();
}

extern "C" int tocvp_dec_tail_sum_f32(const float* products, const float* bias, float* recons_imgs, float* recons,
                                      float* masks, int F, int K, int H, int W, void* stream) {
    return tocvp_dec_tail_sum_placed_f32(products, bias, recons_imgs, recons, masks, nullptr, 3L * H * W, 3L * K * H * W,
                                         (long)K * H * W, F, K, H, W, stream);
}
