// Fused transformer MLP of the slot predictor on split-fp16 operands ("f16x3", fp32-class):
//
//     Y = relu(X W1^T + b1) W2^T + b2 + R            X (M, 512), W1 (Hd, 512), W2 (512, Hd), Hd = 2048
//
// replaces the nn.Linear -> ReLU -> nn.Linear pairs of the reference's predictor blocks
// (models/Blocks/attention.py:355-359 TransformerBlock.mlp and :428-432 TransformerDecoderBlock.mlp, applied at
// :395 / :461-463 / :521-523) -- 80 % of the predictor's GEMM FLOPs.  As two GEMMs the 2048-wide hidden activation
// (0.3 GB of operand planes per MLP at 38400 tokens) is written to HBM by the first and read back by the second, the
// first pays a 2048-wide store epilogue per 512-deep product, and the second's 512-wide output quantises badly over
// the CUs.  Here a workgroup owns 128 tokens and walks the hidden dimension in chunks of 128:
//
//     for chunk c:   H_c^T (128 hidden x 128 tokens)  = W1[c] X^T          K = 512   (acc1: 64 registers per lane)
//                    h_c = planes(relu(H_c + b1[c]))  -> LDS (fp16 hi | lo, 64 KB)
//                    Y^T (512 x 128 tokens)          += W2[:, c] h_c^T      K = 128   (acc2: 256 registers per lane)
//
// so the hidden activation never leaves the CU and the only epilogue is the 128 x 512 output tile.
//   * 4 waves, ONE per SIMD (512 registers): 320 of them are accumulators.  Wave w owns hidden columns
//     [32 w, 32 w + 32) of the chunk in the first product and output columns [128 w, 128 w + 128) in the second, all
//     128 tokens in both: every weight fragment is fetched by exactly one wave (no duplicated bytes on the CU's
//     vector-memory path) straight from L2 in MFMA-fragment order, the token operand (X k-tiles by LDS-DMA with a
//     source-side chunk swizzle; h from the first product's epilogue) is shared through LDS.
//   * MFMA operands are swapped (D^T = W act^T): a lane holds 4 consecutive hidden / output columns of ONE token per
//     register quad, so h is written with 8-byte LDS stores and Y with 16-byte global stores, no transposes.
//   * Arithmetic is bit-identical to the two-GEMM path (gemm_bf16.hip: same operand planes, same k order, products
//     act_hi w_lo + act_lo w_hi + act_hi w_hi per 16-deep step into one accumulator, same epilogue expressions).
//   * Operands: X as fp16 planes (M, 2, 512) of 2^8 x (LayerNorm's plane output), weights as fragment-order planes of
//     2^10 w (tocvp_split_weights_frag_f16).  Valid for |activation| < 255, |w| < 63 (saturating beyond; the checked
//     pass of the Python layer runs the two-GEMM path, which verifies both activations).
#include <stdlib.h>

#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int ME = 512;                      // model width: K of the first product, N of the second
constexpr int HC = 128;                      // hidden columns per chunk
constexpr int BM = 128;                      // tokens per workgroup
constexpr int BK = 64;                       // k-tile of the first product
constexpr int XROW = 256;                    // bytes per token in an X stage: [plane 0: 64 k | plane 1: 64 k]
constexpr int XSTAGE = BM * XROW;            // 32 KB
constexpr int HROW = 512;                    // bytes per token in the h image: [plane 0: 128 hidden | plane 1]
constexpr int HBYTES = BM * HROW;            // 64 KB
constexpr int NXS = 3;                       // X stages: the LDS-DMA runs two k-tiles ahead of the products
constexpr int KS1 = ME / 16;                 // 16-deep steps of the first product
constexpr float SA = TOCVP_F16X3_ACT_SCALE, SW = TOCVP_F16X3_WEIGHT_SCALE;

struct MlpArgs {
    const unsigned char* X;                  // (M, 2, 512) fp16 planes of 2^8 x
    const unsigned char* W1f;                // fragment-order planes of 2^10 W1 (Hd, 512)
    const float* b1;
    const unsigned char* W2f;                // fragment-order planes of 2^10 W2 (512, Hd)
    const float* b2;
    const float* R; int ldr;                 // residual (M, 512) or nullptr
    float* Y; int ldy;
    int M, Hd;
    // work plan: workgroups [0, n_full) take whole tiles; the remaining tiles are cut into S slices of the hidden dimension
    // (workgroup n_full + S j + s = slice s of tile n_full + j), their partial sums meet in the workspace
    int n_full, S;
    float* ws_part;                          // [split tile][slice] raw accumulators, REC floats each
    unsigned* ws_ctr;                        // [split tile] arrival counters, zero between launches
    // zig-zag: odd hidden chunks walk the X k-tiles (and W1's) from the last to the first -- the 32 X images of an XCD (8 MB)
    // cycle through its 4 MB L2 once per chunk, front to back every time = no hits under LRU; turning round at the end of a
    // chunk finds the most recent half still there.  Other accumulation order in those chunks: not the default (TOCVP_MLP_ZIGZAG)
};
constexpr int REC = BM * ME;                 // floats per parked accumulator record (256 KB)
constexpr int WS_CTR_BYTES = 4096;           // 1024 counters
constexpr int WS_RECORDS = 256;              // one per CU at most

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#ifdef TOCVP_MLP_STAMP
__device__ unsigned long long tocvp_mlp_stamps[1024 * 8];   // per workgroup: start, end, cycles in product 1 / epilogue 1 / product 2
#define MLP_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define MLP_STAMP(v)
#endif
#ifndef TOCVP_MLP_ABLATE
#define TOCVP_MLP_ABLATE 0      // timing experiments (scripts/probes/mlp_fused_check.hip): 1 no X DMA, 2 no W loads in the loop,
#endif                          // 3 no h conversion (stores only), 4 no MFMAs, 5 no barrier / wait per k-tile
constexpr int MABL = TOCVP_MLP_ABLATE;
#ifndef TOCVP_MLP_DSPLIT
#define TOCVP_MLP_DSPLIT 8
#endif
#ifndef TOCVP_MLP_DEPHASE
#define TOCVP_MLP_DEPHASE 1
#endif
#ifndef TOCVP_MLP_SB1
#define TOCVP_MLP_SB1 1
#endif
constexpr bool SB1 = TOCVP_MLP_SB1 != 0;      // first product's bias through the scalar cache
constexpr int DSPLIT = TOCVP_MLP_DSPLIT;     // LDS-DMA instructions of a k-tile issued in the step behind the barrier; the rest one step later
constexpr bool DEPHASE = TOCVP_MLP_DEPHASE != 0;
// vmcnt in front of a k-tile's barrier = vector-memory instructions younger than the last DMA instruction of k-tile kt + 1:
//   all 8 behind the barrier of k-tile kt - 2: weight halves (8) + DMA of kt + 2 (8) of k-tile kt - 1, two weight halves (8)
//   split: the last ones in the first step of k-tile kt - 1 (behind its weight half): 4 + DSPLIT + 4 + (8 - DSPLIT) + 4
constexpr int VMW = TOCVP_MLP_DSPLIT == 8 ? 24 : 20;

// weave: order the block's NM MFMAs with its NDS LDS reads and NVM vector-memory instructions (weight-fragment loads /
// LDS-DMA) so that the memory instructions issue in the shadow of the MFMAs, one wave per SIMD (sched_group_barrier
// pipeline: LDS reads first -- they feed the NEXT block -- then the vector-memory instructions)
// weave2: NM MFMAs; an LDS read behind each of the first NDS of them, a vector-memory instruction behind every SECOND one
template <int NM, int NDS, int NVM>
__device__ __forceinline__ void weave2() {
    int vm = 0;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if ((i & 1) == 0 && vm < NVM) {
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            ++vm;
        }
    }
}

template <int NM, int NDS, int NVM>
__device__ __forceinline__ void weave() {
    constexpr int NMEM = NDS + NVM;
    constexpr int SLOTS = NMEM < NM ? NMEM : NM;                     // memory slots (one or two instructions each)
    constexpr int PER = SLOTS > 0 ? NM / SLOTS : NM;
    int mem = 0;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        const int take = (NMEM - mem + (SLOTS - i) - 1) / (SLOTS - i);   // 1 or 2 memory instructions in this slot
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < take) {
                if (mem < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                ++mem;
            }
    }
    if (NM - PER * SLOTS > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * SLOTS, 0);
}

template <bool HASR>
__global__ __launch_bounds__(256, 1) void mlp_f16x3_fused_kernel(MlpArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NXS * XSTAGE + HBYTES];   // 160 KB: the whole LDS of the CU
    unsigned char* const xs = lds;
    unsigned char* const hs = lds + NXS * XSTAGE;
    typedef const __attribute__((address_space(1))) unsigned char* gptr;
    typedef const __attribute__((address_space(1))) f16x8* gv8;

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int KS2 = p.Hd / 16, nchunk = p.Hd / HC;
    int tile = blockIdx.x, c_begin = 0, c_end = nchunk, slice = 0;
    const bool cut = p.S > 1 && (int)blockIdx.x >= p.n_full;      // this workgroup holds ONE slice of a tile's hidden range
    if (cut) {
        const int j = (int)blockIdx.x - p.n_full;
        tile = p.n_full + j / p.S;
        slice = j % p.S;
        c_begin = nchunk * slice / p.S;
        c_end = nchunk * (slice + 1) / p.S;
    }
    const int m0 = tile * BM;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned x15 = (unsigned)(l31 & 15);

    // ---- X k-tile by LDS-DMA: 32 instructions of 1 KiB (4 tokens x 256 B) per k-tile, 8 per wave.  The LDS image is
    // lane-linear; the conflict-free order comes from the SOURCE side: physical 16-byte chunk c of token r holds
    // logical chunk c ^ (r & 15), logical chunk = plane * 8 + k / 8
    unsigned voff_x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * (w * 8 + i) + (lane >> 4);
        const int lc = (lane & 15) ^ (row & 15);
        const int grow = min(m0 + row, p.M - 1);                     // rows past M re-read the last row (never stored)
        voff_x[i] = (unsigned)((((size_t)grow * 2 + (lc >> 3)) * ME + (lc & 7) * 8) * 2);
    }
    // The DMA is issued as inline assembly ON PURPOSE: the compiler's wait-count pass knows that the builtin writes LDS and,
    // unable to tell which bytes, puts `s_waitcnt vmcnt(0)` in front of the next LDS read -- which drained the weight
    // prefetch ring and the DMA itself at EVERY 16-deep step of the first product (59 % of the matrix rate there against
    // 92 % in the second product, whose steps follow no DMA).  The landing is guarded by hand: the counted vmcnt in front of
    // each k-tile's barrier below.  (The hardware still counts these instructions in vmcnt, so the compiler's own counted
    // waits for the weight fragments only ever wait longer than it thinks, never shorter.)
    const unsigned xs_lds = (unsigned)(size_t)xs;                    // LDS byte address of the stage ring
    auto kphys = [&](int cc, int kt) { return kt; };                 // physical k-tile of chunk cc (a zig-zag order was tried: retired)
    auto dma_x = [&](unsigned stage_off, int kt, int i0 = 0, int i1 = 8) {
        const unsigned char* base = p.X + (size_t)kt * (BK * 2);     // uniform
        asm volatile("" : "+s"(base));          // an SGPR base per call: nothing per-lane and 64-bit is hoisted out of the loop
        const gptr g = (gptr)base;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const gptr src = g + voff_x[i];
            const unsigned dst = xs_lds + stage_off + (unsigned)((w * 8 + i) * 1024);
            // (s_nop 0: the wait state the ISA asks for between a write of M0 and an LDS-DMA that reads it -- the compiler puts the
            // same nop behind its own M0 writes)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(dst), "v"(src) : "memory", "m0");
        }
    };

    // ---- weight fragments from L2: Wf[n / 32][k / 16][plane][lane] 16 bytes each (1 KiB per wave-instruction), in HALVES
    // of 24 MFMAs: half q of a chunk = 16-deep steps 2 q, 2 q + 1 of the first product (q < 16: b[2 sl + plane]) or output
    // tiles 2 (q & 1), + 1 of step (q - 16) / 2 of the second (b[2 jl + plane]).  A ring of four halves, three ahead.
    auto load_w_half = [&](f16x8 (&b)[4], int c, int q) {
        if (q >= 32) {
            q -= 32;
            c = min(c + 1, nchunk - 1);                              // past the end: a harmless re-load, never used
        }
        if (q < 16) {
            const int qp = 2 * kphys(c, q >> 1) + (q & 1);           // the steps inside a k-tile keep their order
            const unsigned char* base = p.W1f + ((size_t)((c * 4 + w) * KS1 + 2 * qp) * 2) * 1024;
            asm volatile("" : "+s"(base));
            const gptr g = (gptr)base + lane16;
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(gv8)(g + j * 1024);
        } else {
            const int s = (q - 16) >> 1, jt0 = ((q - 16) & 1) * 2;
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) {
                const unsigned char* base = p.W2f + ((size_t)((4 * w + jt0 + jl) * KS2 + c * 8 + s) * 2) * 1024;
                asm volatile("" : "+s"(base));
                const gptr g = (gptr)base + lane16;
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) b[2 * jl + pl] = *(gv8)(g + pl * 1024);
            }
        }
    };

    f32x16 acc2[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    f32x16 acc1[4];

    struct Frag { f16x8 v[4][2]; };                                  // token operand of one 16-deep step: [token block][plane]
    auto read_x = [&](Frag& f, unsigned stage_off, int s) {
        const unsigned char* xb = xs + stage_off + l31 * XROW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                f.v[i][pl] = *reinterpret_cast<const f16x8*>(xb + i * 32 * XROW + (((unsigned)(pl * 8 + 2 * s + h) ^ x15) << 4));
    };
    auto read_h = [&](Frag& f, int s) {
        const unsigned char* hb = hs + l31 * HROW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                f.v[i][pl] = *reinterpret_cast<const f16x8*>(hb + i * 32 * HROW + (((unsigned)(pl * 16 + 2 * s + h) ^ x15) << 4));
    };
    // first product, one 16-deep step: acc1[i] (32 hidden x 32 tokens of token block i) += W1 frag x X frag
    auto mfma1 = [&](const Frag& f, const f16x8 (&b)[4], int sl) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc1[i] = mfma16(b[2 * sl + 1], f.v[i][0], acc1[i]);     // act hi x w lo
            acc1[i] = mfma16(b[2 * sl + 0], f.v[i][1], acc1[i]);     // act lo x w hi
            acc1[i] = mfma16(b[2 * sl + 0], f.v[i][0], acc1[i]);     // act hi x w hi
        }
    };
    // second product, one 16-deep step, output tiles jt0, jt0 + 1 of this wave: acc2[i][jt] += W2 frag x h frag
    auto mfma2 = [&](const Frag& f, const f16x8 (&b)[4], int jt0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) {
                acc2[i][jt0 + jl] = mfma16(b[2 * jl + 1], f.v[i][0], acc2[i][jt0 + jl]);
                acc2[i][jt0 + jl] = mfma16(b[2 * jl + 0], f.v[i][1], acc2[i][jt0 + jl]);
                acc2[i][jt0 + jl] = mfma16(b[2 * jl + 0], f.v[i][0], acc2[i][jt0 + jl]);
            }
    };
    // epilogue of the first product: bias + ReLU + fp16 planes of 2^8 h (the expressions of the GEMM epilogue and of
    // tocvp_store_planes4) -> h image.  Register quad g of acc1[i] = hidden 32 w + 8 g + 4 h .. + 3 of token 32 i + l31
    // (the chunk's 32 bias values of this wave are wave-uniform: they arrive through the SCALAR cache, requested at the top
    // of the chunk -- as vector loads inside this epilogue they cost three exposed L2 round trips per chunk, each behind a
    // vmcnt(0) that also drained the weight ring)
    typedef const __attribute__((address_space(4))) float* cfptr;
    float sb1[32];
    auto load_b1 = [&](int c) {
        const cfptr b = (cfptr)(p.b1 + c * HC + 32 * w);
#pragma unroll
        for (int j = 0; j < 32; ++j) sb1[j] = b[j];
    };
    auto epilogue1 = [&](int c) {
        const float* b1c = p.b1 + c * HC + 32 * w + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 bq;
            if (SB1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) bq[u] = (h ? sb1[8 * g + 4 + u] : sb1[8 * g + u]) * SA;      // 2^8 b (exact)
            } else {
                bq = *reinterpret_cast<const f32x4*>(b1c + 8 * g) * SA;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f16x4 hi, lo;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    // v = relu(acc 2^-18 + b); planes of 2^8 v (tocvp_store_planes4: clamp to the fp16 range, hi, residual).
                    // 2^8 (acc 2^-18 + b) == acc 2^-10 + 2^8 b in fp32 (a power of two commutes with the rounding), and the
                    // clamp to [0, 65504] is the ReLU and the saturation at once: same bits, half the instructions
                    float v = __builtin_amdgcn_fmed3f(__builtin_fmaf(acc1[i][4 * g + u], SA / (SA * SW), bq[u]), 0.f, 65504.f);
                    if (MABL == 3) {
                        hi[u] = (_Float16)0.f;
                        lo[u] = (_Float16)(v > 1e30f ? 1.f : 0.f);
                        continue;
                    }
                    hi[u] = (_Float16)v;
                    lo[u] = (_Float16)__builtin_fmaf((float)hi[u], -1.0f, v);     // v - hi, one rounding (v_fma_mix_f32)
                }
                unsigned char* row = hs + (32 * i + l31) * HROW + h * 8;
                *reinterpret_cast<f16x4*>(row + (((unsigned)(4 * w + g) ^ x15) << 4)) = hi;
                *reinterpret_cast<f16x4*>(row + (((unsigned)(16 + 4 * w + g) ^ x15) << 4)) = lo;
            }
        }
    };

    f16x8 wr[4][4];                                                  // ring of weight-fragment halves, slot = half & 3
    Frag F0, F1;
#ifdef TOCVP_MLP_STAMP
    unsigned long long st_g1 = 0, st_e1 = 0, st_g2 = 0;
#endif
    MLP_STAMP(st_start);
    // X stage ring (byte offsets): `st_rd` is read by the current k-tile g, `st_nx` holds g + 1, `st_fr` holds g + 2
    // (landed or landing); behind the barrier of k-tile g its stage takes k-tile g + 3
    unsigned st_rd = 0, st_nx = XSTAGE, st_fr = 2 * XSTAGE;
    dma_x(st_rd, kphys(c_begin, 0));
    dma_x(st_nx, kphys(c_begin, 1));
    dma_x(st_fr, kphys(c_begin, 2));
    load_w_half(wr[0], c_begin, 0);
    load_w_half(wr[1], c_begin, 1);
    load_w_half(wr[2], c_begin, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_x(F0, st_rd, 0);
#pragma unroll 1
    for (int c = c_begin; c < c_end; ++c) {
        const bool last = c + 1 == c_end;                            // no look-ahead past the end of this workgroup's range
        MLP_STAMP(s0);
        if (SB1) load_b1(c);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
        // ---- first product: 8 k-tiles of four 16-deep steps.  Entering k-tile kt: F0 = its step-0 token fragments,
        // X k-tile kt + 1 on its way into (or landed in) st_nx, weight halves 2 kt .. 2 kt + 2 loaded or in flight.
#pragma unroll
        for (int kt = 0; kt < ME / BK; ++kt) {
            const int q = 2 * kt;
            if (MABL != 2) load_w_half(wr[(q + 3) & 3], c, q + 3);
            // the last DMA instructions of k-tile kt + 2 (begun behind the previous k-tile's barrier, into the stage it read)
            if (MABL != 1 && DSPLIT < 8 && (kt + 2 < ME / BK || !last)) dma_x(st_fr, kphys(c + (kt + 2 >= ME / BK), (kt + 2) & 7), DSPLIT, 8);
            read_x(F1, st_rd, 1);
            if (MABL != 4) mfma1(F0, wr[q & 3], 0);
            weave2<12, 8, 4 + (8 - DSPLIT)>();
            __builtin_amdgcn_sched_barrier(0);
            read_x(F0, st_rd, 2);
            if (MABL != 4) mfma1(F1, wr[q & 3], 1);
            weave<12, 8, 0>();
            __builtin_amdgcn_sched_barrier(0);
            if (MABL != 2) load_w_half(wr[(q + 4) & 3], c, q + 4);
            read_x(F1, st_rd, 3);
            if (MABL != 4) mfma1(F0, wr[(q + 1) & 3], 0);
            weave<12, 8, 4>();
            __builtin_amdgcn_sched_barrier(0);
            // every wave has read all of this stage (F1 landed) and its share of k-tile kt + 1 has landed: its last DMA
            // instruction was issued in k-tile kt - 1 (behind that k-tile's barrier, or in its first step when the issue is
            // split); younger are weight halves and the DMA instructions of k-tile kt + 2 (see VMW)
            if (MABL != 5) {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VMW) : "memory");
                __syncthreads();
                // de-phase the four waves by 16 cycles each: the CU's LDS-DMA address path is shared, an instruction that
                // finds it busy blocks its wave (and the wave's MFMA issue); issued every second MFMA (64 cycles apart)
                // the four waves' instructions then interleave instead of colliding
                if (DEPHASE)
                    for (int i = 0; i < w; ++i) asm volatile("s_nop 15");
            }
            // k-tile kt + 3 (of the next chunk past the end) into the stage this k-tile has just finished reading
            if (MABL != 1 && (kt + 3 < ME / BK || !last)) dma_x(st_rd, kphys(c + (kt + 3 >= ME / BK), (kt + 3) & 7), 0, DSPLIT);
            if (kt + 1 < ME / BK) read_x(F0, st_nx, 0);
            if (MABL != 4) mfma1(F1, wr[(q + 1) & 3], 1);
            if (kt + 1 < ME / BK) weave2<12, 8, DSPLIT>();
            else weave2<12, 0, DSPLIT>();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned t_ = st_rd;
            st_rd = st_nx;
            st_nx = st_fr;
            st_fr = t_;
        }
        MLP_STAMP(s1);
        epilogue1(c);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                             // h complete
        MLP_STAMP(s2);
        // ---- second product: 8 steps of 16 hidden columns, two halves (output tile pairs) each
        read_h(F0, 0);
#pragma unroll
        for (int s = 0; s < HC / 16; ++s) {
            const int q = 16 + 2 * s;
            Frag& cur = (s & 1) ? F1 : F0;
            Frag& nxt = (s & 1) ? F0 : F1;
            if (MABL != 2 && (q + 3 < 32 || !last)) load_w_half(wr[(q + 3) & 3], c, q + 3);
            if (s + 1 < HC / 16) read_h(nxt, s + 1);
            else read_x(nxt, st_rd, 0);                              // step 0 of the next chunk's first k-tile (landed long ago)
            if (MABL != 4) mfma2(cur, wr[q & 3], 0);
            weave<24, 8, 4>();
            __builtin_amdgcn_sched_barrier(0);
            if (MABL != 2 && (q + 4 < 32 || !last)) load_w_half(wr[(q + 4) & 3], c, q + 4);
            if (MABL != 4) mfma2(cur, wr[(q + 1) & 3], 2);
            weave<24, 0, 4>();
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef TOCVP_MLP_STAMP
        const unsigned long long s3 = __builtin_amdgcn_s_memtime();
        st_g1 += s1 - s0;
        st_e1 += s2 - s1;
        st_g2 += s3 - s2;
#endif
    }
    MLP_STAMP(st_loop_end);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the look-ahead DMA / fragment loads past the end

    // ---- a cut tile: every slice parks its raw accumulators (1 KiB per wave-instruction), counts itself in, and the LAST
    // arriver adds the records in slice order -- deterministic, one launch.  Hand-off as cdna_hip_programming.md Guideline
    // 16: plain stores -> every storing wave drains -> barrier -> one lane: agent-scope release, counter; the last arriver:
    // agent-scope acquire -> barrier -> plain loads; it re-arms the counter.
    if (cut) {
        const int sj = tile - p.n_full;
        float* mine = p.ws_part + ((size_t)sj * p.S + slice) * REC + ((size_t)w * 64 * 64 + lane) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(mine + ((i * 4 + jt) * 4 + g) * 256) =
                        f32x4{acc2[i][jt][4 * g], acc2[i][jt][4 * g + 1], acc2[i][jt][4 * g + 2], acc2[i][jt][4 * g + 3]};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* arrived = reinterpret_cast<unsigned*>(lds);        // the stages are dead
        if (t == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned old = __hip_atomic_fetch_add(p.ws_ctr + sj, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)(p.S - 1)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(p.ws_ctr + sj, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            }
            *arrived = old;
        }
        __syncthreads();
        if (*arrived != (unsigned)(p.S - 1)) return;
        const float* rec0 = p.ws_part + (size_t)sj * p.S * REC + ((size_t)w * 64 * 64 + lane) * 4;
        // a quarter of the record (token block i) at a time, two slices in flight: one memory round trip per pair
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][jt][r] = 0.f;
#pragma unroll 1
            for (int s0 = 0; s0 < p.S; s0 += 2) {
                const int s1 = min(s0 + 1, p.S - 1);
                f32x4 va[16], vb[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    va[e] = *reinterpret_cast<const f32x4*>(rec0 + (size_t)s0 * REC + (i * 16 + e) * 256);
                    vb[e] = *reinterpret_cast<const f32x4*>(rec0 + (size_t)s1 * REC + (i * 16 + e) * 256);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc2[i][e >> 2][4 * (e & 3) + u] += va[e][u];
                if (s0 + 1 < p.S) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc2[i][e >> 2][4 * (e & 3) + u] += vb[e][u];
                }
            }
        }
    }

    // ---- epilogue.  Register quad g of acc2[i][jt] = output columns 128 w + 32 jt + 8 g + 4 h .. + 3 of token 32 i + l31:
    // stored as it stands, a wave-instruction would touch 32 rows x 32 bytes.  The tile goes through LDS instead (all of
    // it is free now), 64 tokens at a time as fp32 rows of 512 + 4 floats, and leaves as whole rows: every load of the
    // residual and every store covers 1 KiB of contiguous memory.
    // The rows are staged as RAW accumulators; scale and bias are applied on the way out, where a lane writes the same
    // four output columns in every row: ONE bias quad per lane (sixteen bias loads inside the staging loop compiled into
    // sixteen serial L2 round trips per half tile).
    constexpr int OS = ME + 4;                                       // floats per staged row (16 B pad: conflict-free stores)
    float* const ost = reinterpret_cast<float*>(lds);
    const f32x4 bq2 = *reinterpret_cast<const f32x4*>(p.b2 + (t & 127) * 4);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                             // LDS free: the products / the previous half are done
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = 128 * w + 32 * jt + 8 * g + 4 * h;
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    // (static accumulator indices: both halves are spelled out)
                    f32x4 v;
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = half == 0 ? acc2[ii][jt][4 * g + u] : acc2[2 + ii][jt][4 * g + u];
                    *reinterpret_cast<f32x4*>(ost + (32 * ii + l31) * OS + col) = v;
                }
            }
        __syncthreads();
        // 64 rows x 128 quads; a wave-instruction = half a row (64 lanes x 16 B), 8 residual quads in flight per lane
#pragma unroll 1
        for (int it0 = 0; it0 < 32; it0 += 8) {
            f32x4 rq[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = t + 256 * (it0 + k);
                const int row = m0 + 64 * half + (idx >> 7), c4 = (idx & 127) * 4;
                if (HASR) rq[k] = *reinterpret_cast<const f32x4*>(p.R + (size_t)min(row, p.M - 1) * p.ldr + c4);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = t + 256 * (it0 + k);
                const int rl = idx >> 7, c4 = (idx & 127) * 4;
                const int row = m0 + 64 * half + rl;
                f32x4 v = *reinterpret_cast<const f32x4*>(ost + rl * OS + c4);
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = v[u] * (1.f / (SA * SW)) + bq2[u];
                if (HASR) v += rq[k];
                if (row < p.M) *reinterpret_cast<f32x4*>(p.Y + (size_t)row * p.ldy + c4) = v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the look-ahead DMA / fragment loads past the end
#ifdef TOCVP_MLP_STAMP
    if (t == 0 && blockIdx.x < 1024) {
        unsigned long long* st = tocvp_mlp_stamps + blockIdx.x * 8;
        st[0] = st_start; st[1] = __builtin_amdgcn_s_memtime(); st[2] = st_g1; st[3] = st_e1; st[4] = st_g2;
        st[5] = st_loop_end;
    }
#endif
}

}  // namespace

// plan of a launch: tiles of 128 rows; whole rounds of one tile per CU run uncut, the tiles of a last, partly filled round
// are cut into S slices of the hidden dimension so that the round fills the CUs (300 tiles on 256 CUs: 256 whole tiles,
// then 44 tiles x 5 slices instead of 44 whole tiles on 44 CUs)
static int mlp_cus() {
    static const int v = []() {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    return v;
}

extern "C" size_t tocvp_mlp_f16x3_fused_ws_bytes(void) { return (size_t)WS_CTR_BYTES + (size_t)WS_RECORDS * REC * sizeof(float); }

extern "C" int tocvp_mlp_f16x3_fused_f32(const void* x_planes, const void* w1_frag, const float* b1, const void* w2_frag,
                                         const float* b2, const float* R, int ldr, float* Y, int ldy, int M, int E,
                                         int Hd, void* ws, size_t ws_bytes, void* stream) {
    TOCVP_CHECK_ARG(x_planes && w1_frag && b1 && w2_frag && b2 && Y);
    TOCVP_CHECK_ARG(E == ME && Hd > 0 && (Hd % HC) == 0 && M >= 0);
    TOCVP_CHECK_ARG(ldy >= E && (ldy & 3) == 0 && (R == nullptr || (ldr >= E && (ldr & 3) == 0)));
    TOCVP_CHECK_ARG((size_t)M * 2 * ME * 2 < 0xffffffffull);        // 32-bit DMA source offsets
    TOCVP_CHECK_ARG(ws == nullptr || (ws_bytes >= tocvp_mlp_f16x3_fused_ws_bytes() && tocvp_aligned16(ws)));
    if (!tocvp_aligned16(x_planes) || !tocvp_aligned16(w1_frag) || !tocvp_aligned16(w2_frag) || !tocvp_aligned16(b1) ||
        !tocvp_aligned16(b2) || !tocvp_aligned16(Y) || (R && !tocvp_aligned16(R)))
        return TOCVP_EALIGN;
    if (M == 0) return TOCVP_OK;
    const int tiles = (M + BM - 1) / BM, cus = mlp_cus(), nchunk = Hd / HC;
    int n_full = tiles, S = 1;
    const int left = tiles % cus;
    if (ws && left > 0) {
        // 2 to 4 slices: the last arriver alone adds the records of a tile, 256 KB each at one CU's share of the L2
        // bandwidth (~4 us per record; 16 slices measured ~65 us of reduction behind 25 us of products)
        const int s = cus / left >= 4 ? 4 : cus / left;          // 1 .. 4
        if (s >= 2 && s <= nchunk && left * s <= WS_RECORDS && left <= WS_CTR_BYTES / 4) {
            S = s;
            n_full = tiles - left;
        }
    }
    MlpArgs p{static_cast<const unsigned char*>(x_planes), static_cast<const unsigned char*>(w1_frag), b1,
              static_cast<const unsigned char*>(w2_frag), b2, R, ldr, Y, ldy, M, Hd, n_full, S,
              ws ? reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + WS_CTR_BYTES) : nullptr,
              static_cast<unsigned*>(ws)};
    const dim3 grid((unsigned)(n_full + (tiles - n_full) * S));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (R) hipLaunchKernelGGL((mlp_f16x3_fused_kernel<true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((mlp_f16x3_fused_kernel<false>), grid, dim3(256), 0, s, p);
    return tocvp_launch_status();
}
