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
// Geometry: 8 x 64 output pixels x 64 output channels per 8-wave workgroup (one per CU: 512 registers per SIMD = two
// waves of 256).  WAVE xi OWNS TRANSFORM ROW xi: its accumulators are M_xi for both 4-row groups (t = 0, 1), both
// 32-pixel halves and both 32-channel halves = 8 tiles = 128 registers -- the per-wave shape of conv_f16x3.hip, so a
// weight fragment from L2 feeds four pixel blocks and an operand fragment from LDS two channel blocks.  Four passes of
// 16 input channels: every thread loads a column of 8 input rows (4 channels), transforms it, splits the 8 results and
// writes them into the LDS image [t][xi][x 68][Vh | Vl] (68 KB, XOR-swizzled 16-byte chunks, two images: the transform
// of pass p + 1 runs on one wave of every SIMD while the other wave multiplies pass p); 5 steps (dx) of 24 MFMAs per
// wave and pass.  After the last pass the eight M_xi meet through LDS (two rounds of 128 KB, one per t): wave w combines
// a 4-row x 8-column block of both channel halves with AT scaled by the inverse operand scales (kernel arguments), adds
// bias / ReLU and stores (fp32 NHWC, fp32 x 16 pass-major for the next layer of this kernel, fp16 operand planes for
// conv_f16x3.hip, or -- last hidden layer -- the 36 tap products of the folded decoder tail).
//
// Built without packed-f32 vector instructions (flags line below): next to another wave's MFMAs a v_pk_fma_f32 costs ~30
// issue cycles against 2 x 4 for the two v_fma_f32 it replaces (MI355X_MICROARCH: packed f32 VALU "an anti-lever beside
// MFMAs"), and the transform of one pass runs beside the partner wave's multiply of the previous one.  (The host
// half of the compilation does not know the feature and says so; harmless.)
// TOCVP_HIPCC_FLAGS: -Xclang -target-feature -Xclang -packed-fp32-ops
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 64, IW = TW + 4;
constexpr int NXI = 8, NT_ROWS = 2;                 // transform rows, 4-row groups per tile
constexpr int C = 64, CCH = 16, NPASS = 4, NDX = 5, NSTEP = NPASS * NDX;
constexpr int OFF_LO = 32;
constexpr int FRAG = 1024, STEP_BYTES = 4 * FRAG;   // [plane(h, l)][nb(2)] fragments of one (xi, pass, dx)
constexpr float VS = 16.f;                          // scale of the transformed operand (and of a "x 16" activation buffer)
constexpr float F16MAX = 65504.f;
constexpr int IMG_BYTES = NT_ROWS * NXI * IW * 64;              // 69632: one pass, unpadded (swizzled chunks)
constexpr int XCH_BYTES = NXI * 4 * 4 * 64 * 16;                // 131072: [xi][tile 4][quad 4][lane 64][4 floats]
constexpr int LDS_BYTES = XCH_BYTES > 2 * IMG_BYTES ? XCH_BYTES : 2 * IMG_BYTES;
constexpr int NTHREADS = 512;

struct WArgs {
    const float* x; const float* aux; const unsigned char* wf; const float* bias; float* y;
    int nimg, H, W, relu;
    int out_mode;                   // 0 fp32 NHWC; 1 fp32 x 16, pass-major (n, 4, H, W, 16); 2 fp16 operand planes of 2^8 y, pass-major
    const unsigned char* tail_wf;   // TAILP: tap matrix of the folded decoder tail (conv_f16x3.hip: pack_tail_taps_kernel)
    float coef[4 * NXI];            // AT[a][xi] / (VS * s_xi)
};

// -DTOCVP_WINO_STAMP (scripts/probes/wino_stamp.hip): s_memtime at the phase boundaries of waves 0 and 4, summed per workgroup
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

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// MODE 0: fp32 x 16 pass-major input (written by this kernel, out_mode 1); 1: the collapsed first layer
// (relu(cpos[y, x] + aux[img][border class]), conv_f16x3.hip MODE 1); 2: fp32 NHWC input.
template <int MODE, bool TAILP>
__global__ __launch_bounds__(NTHREADS, 2) void conv5x5_wino_f16x3_kernel(WArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_x = p.W / TW, tiles = tiles_x * (p.H / TH);
    // the tiles of one slot image share an XCD (halo rows are L2 hits), as in conv_f16x3.hip
    const int img = (blockIdx.x / (8 * tiles)) * 8 + (blockIdx.x & 7), tile = (blockIdx.x >> 3) % tiles;
    if (img >= p.nimg) return;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;

    // accumulator tile m = 2 * (4-row group) + (32-pixel half): M_xi of this wave's transform row
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const char* const xbase = reinterpret_cast<const char*>(MODE == 1 ? p.x : p.x + (size_t)img * p.H * p.W * C);
    const char* const abase = reinterpret_cast<const char*>(MODE == 1 ? p.aux + (size_t)img * 25 * C : p.x);

    // LDS image of one pass: [t 2][xi 8][x 68] pixels of 64 B = four 16-byte chunks (Vh ch 0-7, Vh 8-15, Vl 0-7, Vl 8-15),
    // chunk c of column x stored at position c ^ ((x >> 2) & 3): a ds_read_b128 of 16 lanes (consecutive columns, one
    // chunk) then touches every bank once at any tap shift -- no padding, so TWO images fit (139 KB) and the transform of
    // pass p + 1 runs while pass p multiplies.  Byte offset of this lane's operand (column l31 + dx, k-half h, plane Vh);
    // plane Vl is at offset ^ 32.
    int o_dx[NDX];
#pragma unroll
    for (int dx = 0; dx < NDX; ++dx) {
        const int x = l31 + dx;
        o_dx[dx] = (x << 6) | (((h ^ (x >> 2)) & 3) << 4);
    }

    // weights of this wave's transform row: [xi][pass][dx][plane][nb][lane] 16 B, two register slots: step s lives in slot
    // s % 2 and is loaded while step s - 1 multiplies (24 MFMAs = 768 matrix cycles ahead of its use, across the passes'
    // barriers).  The pass loop below runs two passes (10 steps) per iteration, so the slot of a step is a compile-time
    // choice without copying registers.
    const unsigned char* wptr = p.wf + (size_t)wave * NSTEP * STEP_BYTES + lane * 16;     // fragments of the next step to load
    f16x8 w2[2][2][2];
    auto load_w = [&](int slot) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                w2[slot][pl][n] = *reinterpret_cast<const f16x8*>(wptr + (pl * 2 + n) * FRAG);
        wptr += STEP_BYTES;
    };
    load_w(0);

    const bool wide = p.W != TW;
#ifdef TOCVP_WINO_STAMP
    unsigned long long stamp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_readcyclecounter();
    const unsigned long long stamp_first = stamp_last;
#endif
    // ---- input columns -> V planes.  Thread (4-row group tq, column xi, channel quad cq): 8 rows x 4 channels.
    // The LOADS of a transform (t_load) are issued one matrix step ahead of its arithmetic (t_store): a tile's input is the
    // previous launch's output, i.e. HBM latency, and a transform that waits for it is as long as a multiply phase.
    // rep 0: the 64 columns of the tile (image columns 2..65); rep 1: the four halo columns 0, 1, 66, 67 by 16 threads per
    // row group -- only when the tile has neighbours (W > 64, never prefetched): at W == 64 they are zero padding,
    // written once below.
    constexpr bool PREFETCH = MODE != 1;        // the collapsed layer reads two small L2-resident tables
    auto t_load = [&](int pass, int rep, f32x4 (&d)[8], f32x4 (&ts)[MODE == 1 ? 8 : 1]) {
        int tt = t;
        asm volatile("" : "+v"(tt));             // keeps the staging addresses temporaries
        const int tq = tt >> 8, rem = tt & 255, cq = rem & 3;
        const int xi = rep ? ((rem >> 2) < 2 ? (rem >> 2) : 64 + (rem >> 2)) : 2 + (rem >> 2);
        const int ixc = min(max(tx0 + xi - 2, 0), p.W - 1);
        const int pc = min(pass, NPASS - 1);     // a load past the last pass is harmless and unused
        const int c = pc * CCH + cq * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int iyc = min(max(ty0 + 4 * tq - 2 + i, 0), p.H - 1);
            unsigned off;
            if (MODE == 0) off = (unsigned)(((pc * p.H + iyc) * p.W + ixc) * CCH + cq * 4) * 4u;
            else off = (unsigned)((iyc * p.W + ixc) * C + c) * 4u;
            d[i] = *reinterpret_cast<const f32x4*>(xbase + off);
            if (MODE == 1) {
                const int cls = border_class(iyc, p.H) * 5 + border_class(ixc, p.W);
                ts[i] = *reinterpret_cast<const f32x4*>(abase + (unsigned)(cls * C + c) * 4u);
            }
        }
    };
    auto t_store = [&](int rep, f32x4 (&d)[8], f32x4 (&ts)[MODE == 1 ? 8 : 1], unsigned char* img_s) {
        int tt = t;
        asm volatile("" : "+v"(tt));
        const int tq = tt >> 8, rem = tt & 255, cq = rem & 3;
        const int xi = rep ? ((rem >> 2) < 2 ? (rem >> 2) : 64 + (rem >> 2)) : 2 + (rem >> 2);
        const int ix = tx0 + xi - 2;
        const bool xin = ix >= 0 && ix < p.W;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int iy = ty0 + 4 * tq - 2 + i;
            const bool inside = xin && iy >= 0 && iy < p.H;
            if (MODE == 1) {
                d[i] += ts[i];
#pragma unroll
                for (int u = 0; u < 4; ++u) d[i][u] = fmaxf(d[i][u], 0.f) * VS;
            } else if (MODE == 2) {
                d[i] *= VS;
            }
            if (!inside) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // BT (points 0, 1, -1, 2, -2, 1/2, -1/2, inf), even / odd parts shared by the +- pairs
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
        unsigned char* dst = img_s + (((tq * NXI) * IW + xi) << 6) + ((((cq >> 1) ^ (xi >> 2)) & 3) << 4) + (cq & 1) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f16x4 hi, lo;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float X = clampf(v[q][u], F16MAX);
                hi[u] = (_Float16)X;
                lo[u] = (_Float16)(X - (float)hi[u]);
            }
            *reinterpret_cast<f16x4*>(dst + q * IW * 64) = hi;
            *reinterpret_cast<f16x4*>((unsigned char*)((size_t)(dst + q * IW * 64) ^ 32)) = lo;
        }
    };
    f32x4 dpre[8];                               // the prefetched columns of the next transform (PREFETCH)
    f32x4 tsx[MODE == 1 ? 8 : 1];
    // the arithmetic + LDS stores of the transform of `pass` (its loads too where they were not issued ahead)
    auto transform = [&](int pass, unsigned char* img_s, bool loaded) {
        if (!PREFETCH || !loaded) t_load(pass, 0, dpre, tsx);
#ifdef TOCVP_WINO_STAMP
        WINO_T(8);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WINO_T(9);
#endif
        t_store(0, dpre, tsx, img_s);
#ifdef TOCVP_WINO_STAMP
        WINO_T(10);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WINO_T(11);
#endif
        if (wide && (t & 255) < 16) {
            t_load(pass, 1, dpre, tsx);
            t_store(1, dpre, tsx, img_s);
        }
    };

    // ---- 5 steps (dx) of 24 MFMAs on the image of one pass; PAR = parity of the pass (slot of its first step, LDS image)
    auto multiply = [&](auto par_tag, bool last_pass, int prefetch_pass) {
        constexpr int PAR = decltype(par_tag)::value;
        // LDS offset of this wave's rows in image PAR (an opaque copy per pass: the per-step address arithmetic stays
        // two vector instructions instead of values kept, and spilled, across the passes)
        int wofs = PAR * IMG_BYTES + wave * (IW * 64);
        asm volatile("" : "+s"(wofs));
        // operand fragments [tile m][plane]: ONE set -- the fragments of tile m for step dx + 1 are read into the same
        // registers right behind tile m's six MFMAs of step dx, 18 MFMAs ahead of their use
        f16x8 fa[4][2];
        auto read_tile = [&](int m, int dx) {
            const int toff = (m >> 1) * NXI * IW * 64 + (m & 1) * 32 * 64;
            fa[m][0] = *reinterpret_cast<const f16x8*>(lds + (o_dx[dx] + wofs) + toff);
            fa[m][1] = *reinterpret_cast<const f16x8*>(lds + ((o_dx[dx] ^ 32) + wofs) + toff);
        };
        // the multiplying wave outranks its partner's transform at the SIMD's issue port (an MFMA holds it for 8 of its 32
        // cycles; the transform's vector instructions take what is left)
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int m = 0; m < 4; ++m) read_tile(m, 0);
#pragma unroll
        for (int dx = 0; dx < NDX; ++dx) {
            const int sl = (PAR + dx) & 1;
            const bool more_w = !(dx == NDX - 1) || !last_pass;     // a step follows (run-time only for the last dx)
            if (dx + 1 < NDX) load_w(sl ^ 1);
            else if (more_w) load_w(sl ^ 1);
            // the last step has no operand reads to issue: the next transform's columns start their trip from HBM here
            if (PREFETCH && dx == NDX - 1) t_load(prefetch_pass, 0, dpre, tsx);
#pragma unroll
            for (int m = 0; m < 4; ++m) {                           // one operand stays put between neighbours
                acc[m][0] = mfma16(fa[m][1], w2[sl][0][0], acc[m][0]);      // Vl Uh0
                acc[m][1] = mfma16(fa[m][1], w2[sl][0][1], acc[m][1]);      // Vl Uh1
                acc[m][1] = mfma16(fa[m][0], w2[sl][1][1], acc[m][1]);      // Vh Ul1
                acc[m][0] = mfma16(fa[m][0], w2[sl][1][0], acc[m][0]);      // Vh Ul0
                acc[m][0] = mfma16(fa[m][0], w2[sl][0][0], acc[m][0]);      // Vh Uh0
                acc[m][1] = mfma16(fa[m][0], w2[sl][0][1], acc[m][1]);      // Vh Uh1
                if (dx + 1 < NDX) read_tile(m, dx + 1);
                // fence: left alone, the scheduler sinks these reads to just in front of their use one step later (one
                // register pair for all four tiles, LDS latency exposed every six MFMAs)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // Waves w and w + 4 share a SIMD.  Waves 0-3 transform pass p + 1 BEFORE they multiply pass p, waves 4-7 AFTER: on
    // every SIMD one wave's vector-ALU / load phase runs under the other wave's matrix phase.
    const bool early = wave < 4;
    if (!wide) {                                // zero padding left and right of a full-width tile, both images
        const int im = t >> 8, row = (t >> 4) & 15, cc = (t >> 2) & 3, ch = t & 3;
        *reinterpret_cast<f32x4*>(lds + im * IMG_BYTES + ((row * IW + (cc < 2 ? cc : 64 + cc)) << 6) + ch * 16) =
            f32x4{0.f, 0.f, 0.f, 0.f};
    }
    transform(0, lds, false);
    if (PREFETCH && early) t_load(1, 0, dpre, tsx);     // the early waves hold the columns of pass m + 1 when pass m multiplies
    WINO_T(0);
    __syncthreads();
    WINO_T(2);
#pragma unroll 1
    for (int pp = 0; pp < NPASS; pp += 2) {
        // pass pp on image 0 (image 1 receives pass pp + 1), then pass pp + 1 on image 1 (image 0 receives pass pp + 2).
        // The last step of a multiply issues the loads of the transform the wave runs next: pass m + 2 on the early
        // waves (m = the pass being multiplied), m + 1 on the late ones.
        if (early) transform(pp + 1, lds + IMG_BYTES, true);
        WINO_T(0);
        multiply(std::integral_constant<int, 0>{}, false, early ? pp + 2 : pp + 1);
        WINO_T(1);
        if (!early) transform(pp + 1, lds + IMG_BYTES, true);
        WINO_T(0);
        __syncthreads();                        // image 1 complete, image 0 free
        WINO_T(2);
        const bool more = pp + 2 < NPASS;
        if (early && more) transform(pp + 2, lds, true);
        WINO_T(0);
        multiply(std::integral_constant<int, 1>{}, !more, early ? pp + 3 : pp + 2);
        WINO_T(1);
        if (!early && more) transform(pp + 2, lds, true);
        WINO_T(0);
        __syncthreads();
        WINO_T(2);
    }

    // ---- the eight M_xi meet: round tq = 4-row group.  Writer layout = reader layout (lane-preserving): wave xi writes
    // its four tiles (pixel half, channel half) as quads of registers; wave w then reads, for pixel half w / 4 and register
    // quad g = w % 4, the same quad of all eight xi and both channel halves (16 reads) and forms ALL FOUR output rows:
    // y[n][4 a + e] = row 4 tq + a, column 32 xh + 8 g + 4 h + e, channel 32 n + l31 -- "pixel slot" 8 a + 4 h + e.
    float* const xch = reinterpret_cast<float*>(lds);
    const int oxh = wave >> 2, og = wave & 3;
    constexpr int SS = C + 4;

#pragma unroll
    for (int tq = 0; tq < NT_ROWS; ++tq) {
        if (tq > 0) __syncthreads();            // the stages of the previous round are dead
#pragma unroll
        for (int xh = 0; xh < 2; ++xh)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& a = tq == 0 ? acc[xh][n] : acc[2 + xh][n];
                    *reinterpret_cast<f32x4*>(xch + ((((wave * 4 + xh * 2 + n) * 4 + g) * 64 + lane) << 2)) =
                        f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                }
        WINO_T(3);
        __syncthreads();
        WINO_T(4);
        f32x16 y[2];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[n][r] = 0.f;
#pragma unroll
        for (int q = 0; q < NXI; ++q)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xch + ((((q * 4 + oxh * 2 + n) * 4 + og) * 64 + lane) << 2));
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float cf = p.coef[a * NXI + q];
                    if (a == 0 ? q == 7 : q == 0) continue;             // AT[0][7] = AT[1..3][0] = 0
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[n][4 * a + e] = fmaf(cf, v[e], y[n][4 * a + e]);
                }
            }
        WINO_T(5);
        __syncthreads();                        // everyone has read: the area becomes eight wave-private stages
        WINO_T(6);

        // pixel slot s = 8 a + c (c = 4 h + e): image row 4 tq + (s >> 3), column 32 xh + 8 g + (s & 7)
        const int oy0 = ty0 + 4 * tq, ox = tx0 + oxh * 32 + og * 8;
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
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float bv = p.bias[n * 32 + l31];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = y[n][4 * g + e] + bv;
                        if (p.relu) v = fmaxf(v, 0.f);
                        const float X = clampf(v * SA8, F16MAX);
                        hi[e] = (_Float16)X;
                        lo[e] = (_Float16)(X - (float)hi[e]);
                    }
                    unsigned char* dd = timg + (n * 32 + l31) * 64 + (8 * g + 4 * h) * 2;
                    *reinterpret_cast<f16x4*>(dd) = hi;
                    *reinterpret_cast<f16x4*>(dd + TIMG) = lo;
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
            const float oscale = p.out_mode == 1 ? VS : 1.f;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float bv = p.bias[n * 32 + l31];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = y[n][r] + bv;
                    if (p.relu) v = fmaxf(v, 0.f);
                    stage[acc_row(r, h) * SS + n * 32 + l31] = v * oscale;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (p.out_mode == 2) {
                unsigned char* ybase = reinterpret_cast<unsigned char*>(p.y + (size_t)img * p.H * p.W * C);
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int plane = it >> 1, px = (it & 1) * 16 + (lane >> 2), cq = (lane & 3) * 4;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + plane * CCH + cq);
                    f16x4 hi, lo;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float X = clampf(v[u] * TOCVP_F16X3_ACT_SCALE, F16MAX);
                        hi[u] = (_Float16)X;
                        lo[u] = (_Float16)(X - (float)hi[u]);
                    }
                    unsigned char* blk = ybase + (((size_t)plane * p.H + oy0 + (px >> 3)) * p.W + ox + (px & 7)) * 64 + cq * 2;
                    *reinterpret_cast<f16x4*>(blk) = hi;
                    *reinterpret_cast<f16x4*>(blk + OFF_LO) = lo;
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
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + px * SS + c4);
                    *reinterpret_cast<f32x4*>(ybase + (((size_t)(oy0 + (px >> 3))) * p.W + ox + (px & 7)) * C + c4) = v;
                }
            }
        }
        WINO_T(7);
    }
#ifdef TOCVP_WINO_STAMP
    if ((wave == 0 || wave == 4) && lane == 0 && blockIdx.x < 16384) {
        unsigned long long* o = tocvp_wino_stamps + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 16;
        for (int i = 0; i < 12; ++i) o[i] = stamp_acc[i];
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
// wf: [xi 8][pass 4][dx 5][plane(h, l)][nb 2][lane 64][8 f16]; lane (c = l & 31, hh = l >> 5): output channel nb*32 + c,
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
    const size_t stepbase = (((size_t)xi * NPASS + pass) * NDX + dx) * 4;
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
extern "C" int tocvp_conv5x5_dec_wino_f16x3_f32(const float* x, const float* aux, int in_mode, const void* wf,
                                                const float* coef, const float* bias, const void* tail_taps, float* y,
                                                int nimg, int H, int W, int relu, int out_mode, void* stream) {
    TOCVP_CHECK_ARG(x && wf && coef && bias && y);
    TOCVP_CHECK_ARG(in_mode >= 0 && in_mode <= 2 && (in_mode != 1 || aux != nullptr));
    TOCVP_CHECK_ARG(out_mode >= 0 && out_mode <= 3 && ((out_mode == 3) == (tail_taps != nullptr)));
    TOCVP_CHECK_ARG(nimg >= 0 && H > 0 && W > 0 && (H % TH) == 0 && (W % TW) == 0);
    TOCVP_CHECK_ARG((size_t)nimg * (H / TH) * (W / TW) < 0x7fffffffu && (size_t)H * W * C * 4 < 0x7fffffffu);
    if (!tocvp_aligned16(x) || !tocvp_aligned16(wf) || !tocvp_aligned16(y) || (aux && !tocvp_aligned16(aux)) ||
        (tail_taps && !tocvp_aligned16(tail_taps)))
        return TOCVP_EALIGN;
    if (nimg == 0) return TOCVP_OK;
    WArgs a{x, aux, static_cast<const unsigned char*>(wf), bias, y, nimg, H, W, relu, out_mode,
            static_cast<const unsigned char*>(tail_taps), {}};
    for (int i = 0; i < 4 * NXI; ++i) a.coef[i] = coef[i];
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((size_t)((nimg + 7) / 8) * 8 * (H / TH) * (W / TW)));
    const dim3 block(NTHREADS);
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
