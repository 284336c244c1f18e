// Split-fp16 ("f16x3", fp32-class) GEMM with the ACTIVATION CHUNK RESIDENT IN LDS and the weights streamed from L2 in
// MFMA-fragment order -- the loop of the fused MLP's second product (mlp_fused.hip) as a GEMM of its own:
//
//     C (M, N) = act(A W^T + bias) (+ R)        A as fp16 operand planes (M, 2, K) of 2^8 a, N % 512 == 0 or N % 384 == 0,
//                                               K % 128 == 0
//
// replaces nn.Linear where both dimensions are wide (reference decoders.py:264-307 MLPPatchDecoder layers 1024 -> 1024,
// the DINOv2 ViT projections behind timm_encoders.py:59-70, Blocks/attention.py:167-175).  The in-loop-split kernel
// (gemm_bf16.hip) and the all-DMA planes kernel (gemm_f16p.hip) stage BOTH operands per 32 / 64-deep k-tile and reach
// 30-40 % of the matrix rate; the fused MLP's second product reaches 92 % with this structure:
//   * a workgroup owns 128 tokens x 512 outputs, 4 waves, ONE per SIMD (512 registers, 256 accumulators); wave w owns
//     outputs [128 w, 128 w + 128) and all 128 tokens, so every weight fragment is fetched by exactly one wave, straight
//     from L2 into the MFMA operand registers (fragment order Wf[n / 32][k / 16][plane][lane], a ring of four 24-MFMA
//     halves three ahead) -- no LDS traffic and no barrier for W;
//   * the token operand is walked in chunks of 128 k: a 64 KB image [token][plane 0: 128 k | plane 1: 128 k] with a
//     source-side 16-byte-chunk swizzle, double buffered, filled by LDS-DMA (16 instructions of 1 KiB per wave and
//     chunk, against 64 per wave for the same 384 MFMAs in the fused MLP's first product); ONE barrier per chunk;
//   * MFMA operands swapped (D^T = W A^T): a lane holds 4 consecutive outputs of one token per register quad; the tile
//     leaves through LDS as whole rows (1 KiB per wave-instruction), as fp32 or as the fp16 operand planes of the next
//     split GEMM;
//   * workgroup ids: the column tiles of one row tile take consecutive slots of ONE XCD (the A chunk is fetched from HBM
//     once and served to the other column tiles by that XCD's L2).
// Arithmetic is bit-identical to tocvp_gemm_bf16wfrag_f32 with f16x3 planes (same planes, same k order, products
// a_hi w_lo + a_lo w_hi + a_hi w_hi per 16-deep step into one accumulator, same epilogue expressions).
#include <stdlib.h>

#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128;                      // tokens per workgroup
constexpr int CK = 128;                      // k per chunk
constexpr int AROW = 512;                    // bytes per token in a chunk image: [plane 0: 128 k | plane 1: 128 k]
constexpr int ABYTES = BM * AROW;            // 64 KB
constexpr float SA = TOCVP_F16X3_ACT_SCALE, SW = TOCVP_F16X3_WEIGHT_SCALE;

struct ChunkArgs {
    const unsigned char* A;                  // (M, 2, K) fp16 planes of 2^8 a
    const unsigned char* Wf;                 // fragment-order planes of 2^10 W (N, K)
    const float* bias;
    const float* R; int ldr;                 // residual (M, N) or nullptr
    void* C; int ldc; int c_split;           // fp32 (M, N) row stride ldc, or fp16 planes (M, 2, N)
    int M, N, K, act;
    int row_tiles, col_tiles;
};

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#ifndef TOCVP_GC_ABLATE
#define TOCVP_GC_ABLATE 0       // timing experiments: 1 no A DMA in the loop, 2 no W loads in the loop, 4 no MFMAs, 5 no epilogue,
#endif                          // 6 epilogue without its global stores, 7 epilogue without the LDS staging
constexpr int GABL = TOCVP_GC_ABLATE;
#ifdef TOCVP_GC_STAMP
__device__ unsigned long long tocvp_gc_stamps[4096 * 4];     // per workgroup: start, loop start, loop end, end (s_memtime)
#define GC_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define GC_STAMP(v)
#endif

// NM MFMAs with NDS LDS reads and NVM vector-memory instructions issued in their shadow (see mlp_fused.hip)
template <int NM, int NDS, int NVM>
__device__ __forceinline__ void weave() {
    constexpr int NMEM = NDS + NVM;
    constexpr int SLOTS = NMEM < NM ? NMEM : NM;
    constexpr int PER = SLOTS > 0 ? NM / SLOTS : NM;
    int mem = 0;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        const int take = (NMEM - mem + (SLOTS - i) - 1) / (SLOTS - i);
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

__device__ __forceinline__ float act_of(float v, int act) {
    if (act == TOCVP_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == TOCVP_ACT_GELU) return tocvp_gelu(v);
    return v;
}

// ACT / CSPLIT / HASR are compile-time: with run-time switches the write-out loop compiled into a branch ladder that
// waits for every LDS read on its own.
//
// PERSISTENT: the grid is one workgroup per CU (a multiple of 8).  Workgroup j runs on XCD j & 7 as that XCD's slot j >> 3
// and walks the XCD's tiles -- row tiles r = xcd (mod 8), all column tiles of a row tile consecutive -- with the stride of
// the slot count, so the slots of an XCD work on neighbouring column tiles of the same row tiles at the same time (the A
// chunk comes from HBM once per XCD) and every CU gets the same number of tiles by construction (dispatched one tile per
// workgroup, 1536 equal tiles on 256 CUs ran 534 us with 6 x 151 k cycles per CU = 1.7 GHz-equivalent: some CUs took a
// seventh tile).  The chunks of consecutive tiles form ONE stream: chunk 0 of the next tile is fetched during the last
// chunk of this one and the weight ring runs on into the next tile's fragments, so a tile has no prologue; its epilogue
// goes through the chunk buffer that is free at that point.
// NT = 32-output blocks per wave: 4 (512 outputs per workgroup) or 3 (384: widths like 768 / 2304 of the ViT); a step's
// weight fragments come in two halves, output blocks {0, 1} and {2 .. NT - 1}.
template <int NT, int ACT, bool CSPLIT, bool HASR>
__global__ __launch_bounds__(256, 1) void gemm_f16x3_chunk_kernel(ChunkArgs p) {
    constexpr int BN = 128 * NT;                                     // outputs per workgroup (32 NT per wave)
    constexpr int NJ1 = NT - 2;                                      // output blocks of a step's second half
    constexpr int OROW = BN * 4;                                     // bytes per staged output row
    constexpr int QPR = BN / 4;                                      // output quads per row
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * ABYTES];
    typedef const __attribute__((address_space(1))) unsigned char* gptr;
    typedef const __attribute__((address_space(1))) f16x8* gv8;

    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, nslot = (int)gridDim.x >> 3;
    const int nrows_x = p.row_tiles > xcd ? (p.row_tiles - xcd + 7) >> 3 : 0;     // row tiles of this XCD
    const int nitems = nrows_x * p.col_tiles;
    int item = slot;
    if (item >= nitems) return;

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int KS = p.K / 16, nchunk = p.K / CK;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned x15 = (unsigned)(l31 & 15);

    // ---- A chunk by LDS-DMA: 64 instructions of 1 KiB (2 tokens x 512 B) per chunk, 16 per wave.  The LDS image is
    // lane-linear; the conflict-free order comes from the SOURCE side: physical 16-byte chunk c of token r holds logical
    // chunk c ^ (r & 15), logical chunk = plane * 16 + k / 8
    unsigned voff_a[16];
    auto set_voff = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 2 * (w * 16 + i) + (lane >> 5);
            const int lc = (lane & 31) ^ (row & 15);
            const int grow = min(m0 + row, p.M - 1);                 // rows past M re-read the last row (never stored)
            voff_a[i] = (unsigned)((((size_t)grow * 2 + (lc >> 4)) * p.K + (lc & 15) * 8) * 2);
        }
    };
    // (inline assembly on purpose: see mlp_fused.hip -- the compiler's wait-count pass would put vmcnt(0) in front of
    // every LDS read behind a global_load_lds builtin; the landing is guarded by the counted wait in front of the
    // chunk barrier)
    const unsigned a_lds = (unsigned)(size_t)lds;
    auto dma_a = [&](unsigned buf_off, int c, int i0, int i1) {
        const unsigned char* base = p.A + (size_t)c * (CK * 2);      // uniform
        asm volatile("" : "+s"(base));
        const gptr g = (gptr)base;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const gptr src = g + voff_a[i];
            const unsigned dst = a_lds + buf_off + (unsigned)((w * 16 + i) * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(dst), "v"(src) : "memory", "m0");
        }
    };

    // ---- weight fragments from L2 in HALVES of 24 MFMAs: half q of a tile = output tiles 2 (q & 1), + 1 of 16-deep step
    // q / 2; halves past the end of the tile are the first halves of the NEXT tile (the ring never drains)
    const int nq = 2 * KS;
    const unsigned char* wcur;
    const unsigned char* wnxt;
    auto load_w_half = [&](f16x8 (&b)[4], int q) {
        const unsigned char* wb = wcur;
        if (q >= nq) {
            q -= nq;
            wb = wnxt;
        }
        const int s = q >> 1, jt0 = (q & 1) * 2;
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) {
            if (jl >= NJ1 && (q & 1)) break;                         // NT = 3: the second half holds ONE output block
            const unsigned char* base = wb + ((size_t)(jt0 + jl) * KS + s) * 2048;
            asm volatile("" : "+s"(base));
            const gptr g = (gptr)base + lane16;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) b[2 * jl + pl] = *(gv8)(g + pl * 1024);
        }
    };

    f32x16 acc[4][NT];
    struct Frag { f16x8 v[4][2]; };                                  // token operand of one 16-deep step: [token block][plane]
    auto read_a = [&](Frag& f, unsigned buf_off, int s) {
        const unsigned char* ab = lds + buf_off + l31 * AROW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                f.v[i][pl] = *reinterpret_cast<const f16x8*>(ab + i * 32 * AROW + (((unsigned)(pl * 16 + 2 * s + h) ^ x15) << 4));
    };
    auto mfma2 = [&](const Frag& f, const f16x8 (&b)[4], int jt0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jl = 0; jl < (jt0 == 0 ? 2 : NJ1); ++jl) {
                acc[i][jt0 + jl] = mfma16(b[2 * jl + 1], f.v[i][0], acc[i][jt0 + jl]);     // act hi x w lo
                acc[i][jt0 + jl] = mfma16(b[2 * jl + 0], f.v[i][1], acc[i][jt0 + jl]);     // act lo x w hi
                acc[i][jt0 + jl] = mfma16(b[2 * jl + 0], f.v[i][0], acc[i][jt0 + jl]);     // act hi x w hi
            }
    };

    // tile of an item of this XCD: row tile 8 (item / col_tiles) + xcd, column tile item % col_tiles
    int m0 = ((item / p.col_tiles) * 8 + xcd) * BM, n0 = (item % p.col_tiles) * BN;
    wcur = p.Wf + (size_t)(n0 / 32 + NT * w) * KS * 2048;
    wnxt = wcur;

    f16x8 wr[4][4];                                                  // ring of weight-fragment halves, slot = half & 3
    Frag F0, F1;
    GC_STAMP(st_start);
    set_voff(m0);
    dma_a(0, 0, 0, 16);
    {
        const int nx = item + nslot < nitems ? item + nslot : item;
        wnxt = p.Wf + (size_t)((nx % p.col_tiles) * (BN / 32) + NT * w) * KS * 2048;
    }
    load_w_half(wr[0], 0);
    load_w_half(wr[1], 1);
    load_w_half(wr[2], 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_a(F0, 0, 0);
    GC_STAMP(st_loop);
#ifdef TOCVP_GC_STAMP
    unsigned long long st_k = 0, st_e = 0;
#endif
    unsigned cur = 0, oth = ABYTES;
#pragma unroll 1
    for (;;) {
        const int nitem = item + nslot;
        const bool has_next = nitem < nitems;
        const int nx = has_next ? nitem : item;
        const int m0n = ((nx / p.col_tiles) * 8 + xcd) * BM, n0n = (nx % p.col_tiles) * BN;
        wnxt = p.Wf + (size_t)(n0n / 32 + NT * w) * KS * 2048;
        GC_STAMP(s0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
        for (int c = 0; c < nchunk; ++c) {
            // entering chunk c: its image is complete in `cur`, F0 = its step-0 fragments, every wave is done with `oth`
            // (barrier in step 7 of the previous chunk): the next chunk of the stream goes there -- chunk c + 1 of this
            // tile or chunk 0 of the next -- two DMA instructions per half in steps 0 .. 3
            const bool lastc = c + 1 == nchunk;
            if (lastc) set_voff(m0n);                                // this tile's rows have all been requested
            const bool more = !lastc || has_next;
            const int cn = lastc ? 0 : c + 1;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int q = 16 * c + 2 * s;
                Frag& fc = (s & 1) ? F1 : F0;
                Frag& fn = (s & 1) ? F0 : F1;
                if (GABL != 2) load_w_half(wr[(q + 3) & 3], q + 3);
                if (GABL != 1 && s < 4 && more) dma_a(oth, cn, 4 * s, 4 * s + 2);
                if (s < 7) {
                    read_a(fn, cur, s + 1);
                } else {
                    // every wave holds its step-7 fragments (the last reads of `cur`) and its share of the next chunk has
                    // landed: those DMA instructions were issued in steps 0 .. 3, at least 12 weight-fragment loads ago
                    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (!lastc) read_a(fn, oth, 0);
                }
                if (GABL != 4) mfma2(fc, wr[q & 3], 0);
                if (s < 4) weave<24, 8, 6>();
                else weave<24, 8, 4>();
                __builtin_amdgcn_sched_barrier(0);
                if (GABL != 2) load_w_half(wr[(q + 4) & 3], q + 4);
                if (GABL != 1 && s < 4 && more) dma_a(oth, cn, 4 * s + 2, 4 * s + 4);
                if (GABL != 4) mfma2(fc, wr[(q + 1) & 3], 2);
                if (s < 4) weave<12 * NJ1, 0, 2 * NJ1 + 2>();
                else weave<12 * NJ1, 0, 2 * NJ1>();
                __builtin_amdgcn_sched_barrier(0);
            }
            const unsigned t_ = cur;
            cur = oth;
            oth = t_;
        }
        GC_STAMP(s1);
        // here: `cur` holds chunk 0 of the next tile (landed, if there is one), `oth` -- the last chunk's image -- is free
        // (barrier of its step 7), the ring holds halves 0 .. 2 of the next tile

        // ---- epilogue.  Register quad g of acc[i][jt] = outputs n0 + 128 w + 32 jt + 8 g + 4 h .. + 3 of token 32 i + l31.
        // Token block i (32 tokens x 512 outputs = 64 KB of raw accumulators) goes through the free chunk buffer with a
        // 16-byte-chunk swizzle (physical chunk = logical ^ (token & 15): conflict-free both ways) and leaves as whole rows:
        // every load of the residual and every store covers 1 KiB (fp32) / 512 B per plane of contiguous memory.  A lane
        // writes the same four output columns in every row, so the bias is ONE quad per lane.
        if (GABL != 5) {
            unsigned char* const ost = lds + oth;
            // lane -> (row parity, output quad): QPR quads per row, two rows per pass of the workgroup (NT = 3: 192 of
            // the 256 threads write out)
            const int lcq = t % QPR, rsel = t / QPR;
            const bool wout = rsel < 2;
            f32x4 bq = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bq = *reinterpret_cast<const f32x4*>(p.bias + n0 + 4 * lcq);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i > 0) __syncthreads();                          // the previous block has been read out
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const unsigned lc = (unsigned)(8 * NT * w + 8 * jt + 2 * g + h);
                        const f32x4 v = {acc[i][jt][4 * g], acc[i][jt][4 * g + 1], acc[i][jt][4 * g + 2], acc[i][jt][4 * g + 3]};
                        if (GABL != 7) *reinterpret_cast<f32x4*>(ost + l31 * OROW + ((lc ^ x15) << 4)) = v;
                        else if (v[0] == 123.456f) ost[0] = 1;
                    }
                __syncthreads();
                // 32 rows x QPR quads; a wave-instruction covers 64 lanes x 16 B of one or two rows, 8 quads in flight per lane
                if (wout) {
#pragma unroll 1
                    for (int it0 = 0; it0 < 16; it0 += 8) {
                        f32x4 rq[8], vq[8];
                        if (HASR) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const int row = m0 + 32 * i + (rsel + 2 * (it0 + k));
                                rq[k] = *reinterpret_cast<const f32x4*>(p.R + (size_t)min(row, p.M - 1) * p.ldr + n0 + 4 * lcq);
                            }
                        }
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int rl = rsel + 2 * (it0 + k);
                            vq[k] = f32x4{1.f, 2.f, 3.f, 4.f};
                            if (GABL != 7)
                                vq[k] = *reinterpret_cast<const f32x4*>(ost + rl * OROW + (((unsigned)lcq ^ (unsigned)(rl & 15)) << 4));
                        }
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int rl = rsel + 2 * (it0 + k);
                            const int row = m0 + 32 * i + rl;
                            f32x4 v;
#pragma unroll
                            for (int u = 0; u < 4; ++u) v[u] = act_of(vq[k][u] * (1.f / (SA * SW)) + bq[u], ACT);
                            if (HASR) v += rq[k];
                            if (GABL == 6) {
                                if (v[0] == 123.456f) static_cast<float*>(p.C)[0] = 1.f;
                            } else if (row < p.M) {
                                if (CSPLIT) tocvp_store_planes4(p.C, (size_t)row * 2 * p.N + n0 + 4 * lcq, (size_t)p.N, v, 22);
                                else *reinterpret_cast<f32x4*>(static_cast<float*>(p.C) + (size_t)row * p.ldc + n0 + 4 * lcq) = v;
                            }
                        }
                    }
                }
            }
        } else if (acc[0][0][0] + acc[1][1][1] + acc[2][2][2] + acc[3][NT - 1][3] == 123.456f) {
            static_cast<float*>(p.C)[0] = 1.f;
        }
        GC_STAMP(s2);
#ifdef TOCVP_GC_STAMP
        st_k += s1 - s0;
        st_e += s2 - s1;
#endif
        if (!has_next) break;
        item = nitem;
        m0 = m0n;
        n0 = n0n;
        wcur = wnxt;
        __syncthreads();                                             // `oth` has been read out: the stream may fill it again
        read_a(F0, cur, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the fragment loads past the end
#ifdef TOCVP_GC_STAMP
    if (t == 0 && blockIdx.x < 4096) {
        unsigned long long* st = tocvp_gc_stamps + blockIdx.x * 4;
        st[0] = st_loop - st_start; st[1] = st_k; st[2] = st_e; st[3] = __builtin_amdgcn_s_memtime() - st_start;
    }
#endif
}


// ------------------------------------------------------------------------------------------------------------------
// The same structure for MID-SIZE row counts (small evaluation batches: 600 .. 10 000 rows of the predictor's products,
// reference models/Blocks/attention.py:167-175, 355-359, 428-432).  The two-operand kernels (gemm_bf16.hip, 64 x 64 tiles)
// walk K in 64-deep tiles of load -> split -> LDS store -> barrier -> 12 MFMAs per wave: ~1200 cycles per k-tile, 41 us for
// 2400 x 512 x 2048 whose products take 6 us of the matrix pipe.  Here:
//   * a workgroup owns 64 rows x 256 outputs (4 waves, each 64 rows x 64 outputs: 2 x 2 accumulator tiles), 64 KB of LDS
//     (two 32 KB chunk images) and 256 registers -> TWO workgroups per CU, one computing while the other waits;
//   * A in 128-deep chunks by LDS-DMA (8 instructions per wave and chunk, one barrier per chunk), weights from L2 in
//     fragment order through a ring of four 16-deep steps (three ahead);
//   * SPLIT-K over chunk ranges when the tiles do not fill the chip: every slice parks its raw accumulators (lane order,
//     64 KB) in a per-stream workspace, counts itself in, and the LAST arriver of a tile adds the slices in slice order
//     (deterministic), then runs the epilogue (hand-off as cdna_hip_programming.md Guideline 16);
//   * epilogue through the 64 KB of LDS in one pass (raw accumulators, 16-byte-chunk swizzle), scale + ONE bias quad per
//     lane, activation, residual, fp32 rows or fp16 planes on the way out.
// Unsplit (S = 1) it is bit-identical to tocvp_gemm_bf16wfrag_f32; split, the k order inside a slice is kept and the slices
// are added in order (differs from the unsplit sum in the last bits, every run the same).
// ------------------------------------------------------------------------------------------------------------------
constexpr int MBM = 64;                      // rows per workgroup
constexpr int MBN = 256;                     // outputs per workgroup
constexpr int MABYTES = MBM * AROW;          // 32 KB chunk image

struct MidArgs {
    const unsigned char* A; const unsigned char* Wf; const float* bias; const float* R; int ldr;
    void* C; int ldc; int c_split;
    int M, N, K, col_tiles;
    int row_tiles, xcd_rows;
};

template <int ACT, bool CSPLIT, bool HASR>
__global__ __launch_bounds__(256, 2) void gemm_f16x3_mid_kernel(MidArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * MABYTES];
    typedef const __attribute__((address_space(1))) unsigned char* gptr;
    typedef const __attribute__((address_space(1))) f16x8* gv8;

    // Workgroup ids are dealt round-robin over the 8 XCDs.  Unsplit launches (xcd_rows): the id's low three bits pick the row
    // tile inside a group of eight and the column tiles of a row tile follow each other on that XCD, so a row tile's A chunks
    // come from HBM once (dealt in launch order the eight column tiles of a 2048-wide product landed on eight XCDs and A was
    // fetched eight times: 173 MB from beyond L2 for 24 MB of operands, PMC of 9600 x 2048 x 512)
    int slice, tile, rt, ct;
    if (p.xcd_rows) {
        const int per_group = 8 * p.col_tiles;
        const int grp = (int)blockIdx.x / per_group, rem = (int)blockIdx.x % per_group;
        rt = grp * 8 + (rem & 7);
        ct = rem >> 3;
        if (rt >= p.row_tiles) return;
        slice = 0;
        tile = rt * p.col_tiles + ct;
    } else {
        slice = 0;
        tile = (int)blockIdx.x;
        rt = tile / p.col_tiles;
        ct = tile % p.col_tiles;
    }
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int KS = p.K / 16, nchunk = p.K / CK;
    const int c0 = 0, c1 = nchunk;                                   // (split-K over chunk ranges was built, measured slower and retired)
    (void)slice;
    const int m0 = rt * MBM, n0 = ct * MBN;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned x15 = (unsigned)(l31 & 15);

    // A chunk: 32 instructions of 1 KiB (2 rows x 512 B), 8 per wave; source-side chunk swizzle as above
    unsigned voff_a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 2 * (w * 8 + i) + (lane >> 5);
        const int lc = (lane & 31) ^ (row & 15);
        const int grow = min(m0 + row, p.M - 1);
        voff_a[i] = (unsigned)((((size_t)grow * 2 + (lc >> 4)) * p.K + (lc & 15) * 8) * 2);
    }
    const unsigned a_lds = (unsigned)(size_t)lds;
    auto dma_a = [&](unsigned buf_off, int c, int i0, int i1) {
        const unsigned char* base = p.A + (size_t)c * (CK * 2);
        asm volatile("" : "+s"(base));
        const gptr g = (gptr)base;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const gptr src = g + voff_a[i];
            const unsigned dst = a_lds + buf_off + (unsigned)((w * 8 + i) * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(dst), "v"(src) : "memory", "m0");
        }
    };
    // weight fragments of one 16-deep step: output blocks 2 w, 2 w + 1 of this column tile, both planes
    const unsigned char* const wbase = p.Wf + (size_t)(n0 / 32 + 2 * w) * KS * 2048;
    const int gmax = 8 * c1 - 1;
    auto load_w = [&](f16x8 (&b)[4], int g) {
        g = min(g, gmax);                                            // past the end of the slice: a harmless re-load
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) {
            const unsigned char* base = wbase + ((size_t)jl * KS + g) * 2048;
            asm volatile("" : "+s"(base));
            const gptr gp = (gptr)base + lane16;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) b[2 * jl + pl] = *(gv8)(gp + pl * 1024);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    struct Frag { f16x8 v[2][2]; };
    auto read_a = [&](Frag& f, unsigned buf_off, int s) {
        const unsigned char* ab = lds + buf_off + l31 * AROW;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                f.v[i][pl] = *reinterpret_cast<const f16x8*>(ab + i * 32 * AROW + (((unsigned)(pl * 16 + 2 * s + h) ^ x15) << 4));
    };
    auto mfma_step = [&](const Frag& f, const f16x8 (&b)[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) {
                acc[i][jl] = mfma16(b[2 * jl + 1], f.v[i][0], acc[i][jl]);     // act hi x w lo
                acc[i][jl] = mfma16(b[2 * jl + 0], f.v[i][1], acc[i][jl]);     // act lo x w hi
                acc[i][jl] = mfma16(b[2 * jl + 0], f.v[i][0], acc[i][jl]);     // act hi x w hi
            }
    };

    f16x8 wr[4][4];                                                  // ring of steps, slot = step & 3
    Frag F0, F1;
    dma_a(0, c0, 0, 8);
    load_w(wr[0], 8 * c0);
    load_w(wr[1], 8 * c0 + 1);
    load_w(wr[2], 8 * c0 + 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_a(F0, 0, 0);
    unsigned cur = 0, oth = MABYTES;
#pragma unroll 1
    for (int c = c0; c < c1; ++c) {
        const bool more = c + 1 < c1;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int g = 8 * c + s;
            Frag& fc = (s & 1) ? F1 : F0;
            Frag& fn = (s & 1) ? F0 : F1;
            load_w(wr[(s + 3) & 3], g + 3);
            if (s < 4 && more) dma_a(oth, c + 1, 2 * s, 2 * s + 2);
            if (s < 7) {
                read_a(fn, cur, s + 1);
            } else {
                // the DMA instructions of the next chunk were issued in steps 0 .. 3, 16 weight-fragment loads ago
                asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
                __syncthreads();
                if (more) read_a(fn, oth, 0);
            }
            mfma_step(fc, wr[s & 3]);
            if (s < 4) weave<12, 4, 6>();
            else weave<12, 4, 4>();
            __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned t_ = cur;
        cur = oth;
        oth = t_;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the fragment loads past the end

    // ---- epilogue: register quad g of acc[i][j] = outputs n0 + 64 w + 32 j + 8 g + 4 h .. + 3 of row 32 i + l31; 64 rows x
    // 256 floats through LDS (row = 64 chunks of 16 B, physical chunk = logical ^ (row & 15)); a lane writes out the same
    // four columns of rows (t >> 6) + 4 k
    __syncthreads();
    {
        unsigned char* const ost = lds;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned lc = (unsigned)(16 * w + 8 * j + 2 * g + h);
                    const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(ost + (32 * i + l31) * 1024 + ((lc ^ x15) << 4)) = v;
                }
        __syncthreads();
        const int lcq = t & 63, rsel = t >> 6;
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bq = *reinterpret_cast<const f32x4*>(p.bias + n0 + 4 * lcq);
#pragma unroll 1
        for (int it0 = 0; it0 < 16; it0 += 8) {
            f32x4 rq[8], vq[8];
            if (HASR) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = m0 + rsel + 4 * (it0 + k);
                    rq[k] = *reinterpret_cast<const f32x4*>(p.R + (size_t)min(row, p.M - 1) * p.ldr + n0 + 4 * lcq);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int rl = rsel + 4 * (it0 + k);
                vq[k] = *reinterpret_cast<const f32x4*>(ost + rl * 1024 + (((unsigned)lcq ^ (unsigned)(rl & 15)) << 4));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = m0 + rsel + 4 * (it0 + k);
                f32x4 v;
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = act_of(vq[k][u] * (1.f / (SA * SW)) + bq[u], ACT);
                if (HASR) v += rq[k];
                if (row < p.M) {
                    if (CSPLIT) tocvp_store_planes4(p.C, (size_t)row * 2 * p.N + n0 + 4 * lcq, (size_t)p.N, v, 22);
                    else *reinterpret_cast<f32x4*>(static_cast<float*>(p.C) + (size_t)row * p.ldc + n0 + 4 * lcq) = v;
                }
            }
        }
    }
}

}  // namespace

#ifdef TOCVP_GC_STAMP
extern "C" int tocvp_gc_read_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(tocvp_gc_stamps), (size_t)n * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

static int gc_cus() {
    static const int v = []() {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            n = 256;
        return n >= 8 ? n / 8 * 8 : 8;
    }();
    return v;
}

extern "C" int tocvp_gemm_f16chunk_f32(const void* A_planes, const void* W_frag, const float* bias, const float* R, int ldr,
                                       void* C, int c_split, int ldc, int M, int N, int K, int act, void* stream) {
    TOCVP_CHECK_ARG(A_planes && W_frag && C && M >= 0 && N > 0 && K > 0);
    TOCVP_CHECK_ARG(((N % 512) == 0 || (N % 384) == 0) && (K % CK) == 0);
    TOCVP_CHECK_ARG(act >= TOCVP_ACT_NONE && act <= TOCVP_ACT_GELU);
    TOCVP_CHECK_ARG(c_split || (ldc >= N && (ldc & 3) == 0));
    TOCVP_CHECK_ARG(R == nullptr || (ldr >= N && (ldr & 3) == 0));
    TOCVP_CHECK_ARG((size_t)M * 2 * K * 2 < 0xffffffffull);        // 32-bit DMA source offsets
    if (!tocvp_aligned16(A_planes) || !tocvp_aligned16(W_frag) || !tocvp_aligned16(C) || (bias && !tocvp_aligned16(bias)) ||
        (R && !tocvp_aligned16(R)))
        return TOCVP_EALIGN;
    if (M == 0) return TOCVP_OK;
    const int bn = (N % 512) == 0 ? 512 : 384;
    const int row_tiles = (M + BM - 1) / BM, col_tiles = N / bn;
    ChunkArgs p{static_cast<const unsigned char*>(A_planes), static_cast<const unsigned char*>(W_frag), bias, R, ldr, C, ldc,
                c_split, M, N, K, act, row_tiles, col_tiles};
    // one workgroup per CU (a multiple of 8: the kernel takes id & 7 as its XCD), fewer when there are fewer tiles
    const int per_xcd = ((row_tiles + 7) / 8) * col_tiles;              // most tiles any XCD holds
    const int slots = per_xcd < gc_cus() / 8 ? per_xcd : gc_cus() / 8;
    const dim3 grid((unsigned)(8 * slots));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GC_LAUNCH(A_, S_, R_)                                                                                   \
    do {                                                                                                        \
        if (bn == 512) hipLaunchKernelGGL((gemm_f16x3_chunk_kernel<4, A_, S_, R_>), grid, dim3(256), 0, st, p);  \
        else hipLaunchKernelGGL((gemm_f16x3_chunk_kernel<3, A_, S_, R_>), grid, dim3(256), 0, st, p);           \
    } while (0)
#define GC_LAUNCH_SR(A_)                                  \
    do {                                                  \
        if (c_split) {                                    \
            if (R) GC_LAUNCH(A_, true, true);             \
            else GC_LAUNCH(A_, true, false);              \
        } else {                                          \
            if (R) GC_LAUNCH(A_, false, true);            \
            else GC_LAUNCH(A_, false, false);             \
        }                                                 \
    } while (0)
    if (act == TOCVP_ACT_RELU) GC_LAUNCH_SR(TOCVP_ACT_RELU);
    else if (act == TOCVP_ACT_GELU) GC_LAUNCH_SR(TOCVP_ACT_GELU);
    else GC_LAUNCH_SR(TOCVP_ACT_NONE);
#undef GC_LAUNCH_SR
#undef GC_LAUNCH
    return tocvp_launch_status();
}

extern "C" int tocvp_gemm_f16mid_f32(const void* A_planes, const void* W_frag, const float* bias, const float* R, int ldr,
                                     void* C, int c_split, int ldc, int M, int N, int K, int act, void* stream) {
    TOCVP_CHECK_ARG(A_planes && W_frag && C && M >= 0 && N > 0 && K > 0);
    TOCVP_CHECK_ARG((N % MBN) == 0 && (K % CK) == 0);
    TOCVP_CHECK_ARG(act >= TOCVP_ACT_NONE && act <= TOCVP_ACT_GELU);
    TOCVP_CHECK_ARG(c_split || (ldc >= N && (ldc & 3) == 0));
    TOCVP_CHECK_ARG(R == nullptr || (ldr >= N && (ldr & 3) == 0));
    TOCVP_CHECK_ARG((size_t)M * 2 * K * 2 < 0xffffffffull);
    if (!tocvp_aligned16(A_planes) || !tocvp_aligned16(W_frag) || !tocvp_aligned16(C) || (bias && !tocvp_aligned16(bias)) ||
        (R && !tocvp_aligned16(R)))
        return TOCVP_EALIGN;
    if (M == 0) return TOCVP_OK;
    const int row_tiles = (M + MBM - 1) / MBM, col_tiles = N / MBN, tiles = row_tiles * col_tiles, nchunk = K / CK;
    MidArgs p{static_cast<const unsigned char*>(A_planes), static_cast<const unsigned char*>(W_frag), bias, R, ldr, C, ldc,
              c_split, M, N, K, col_tiles, row_tiles, 0};
    unsigned nwg = (unsigned)tiles;
    if (col_tiles > 1) {
        p.xcd_rows = 1;
        nwg = (unsigned)(((row_tiles + 7) / 8) * 8 * col_tiles);
    }
    const dim3 grid(nwg);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GM_LAUNCH(A_, S_, R_) hipLaunchKernelGGL((gemm_f16x3_mid_kernel<A_, S_, R_>), grid, dim3(256), 0, st, p)
#define GM_LAUNCH_SR(A_)                                  \
    do {                                                  \
        if (c_split) {                                    \
            if (R) GM_LAUNCH(A_, true, true);             \
            else GM_LAUNCH(A_, true, false);              \
        } else {                                          \
            if (R) GM_LAUNCH(A_, false, true);            \
            else GM_LAUNCH(A_, false, false);             \
        }                                                 \
    } while (0)
    if (act == TOCVP_ACT_RELU) GM_LAUNCH_SR(TOCVP_ACT_RELU);
    else if (act == TOCVP_ACT_GELU) GM_LAUNCH_SR(TOCVP_ACT_GELU);
    else GM_LAUNCH_SR(TOCVP_ACT_NONE);
#undef GM_LAUNCH_SR
#undef GM_LAUNCH
    return tocvp_launch_status();
}
