// Weight gradient of nn.Linear in the predictor training step (reference
// 04_train_predictor.py:96-104 lets torch.autograd do this): the "TN" product
//
//     dW (N x K) = G^T X        G (M x N) = dL/dY, X (M x K) = the layer input, both row-major fp32,
//     db (N)     = column sums of G                                             (fused, optional)
//
// reduced over the M token rows.  Both operands are read the way they lie in memory (the reduction index is
// the ROW index of both), so nothing is transposed or re-split on the way: the exact fp32 MFMA
// (v_mfma_f32_32x32x2_f32, A[i][k] / B[k][j] one float per lane) takes lane-consecutive floats of one row,
// which is a conflict-free ds_read_b32 on the natural row-major tile.
//
//   * 128 x 128 output tile per 4-wave workgroup (64 x 64 per wave), 16 token rows per stage, 3 stages of
//     16 KiB (4 measured the same, 2 % slower on the MLP shapes) filled by LDS-DMA (global_load_lds, 16 B per
//     lane) two stages ahead of their use; one barrier per stage.  Rows are stored in PAIRS: [32-column block][row parity][32 floats], so the two k-slices of
//     an MFMA operand (rows 2s and 2s+1, 32 columns) are 64 consecutive floats = one read per lane, all
//     64 banks.  The DMA does that interleave on the source side (the LDS side of a DMA is lane-linear).
//   * split-K over blockIdx.z: split z owns token rows [z * chunk, (z + 1) * chunk) and its own (N x K)
//     slice of `c_part` (+ (N) of `bias_part`), written or accumulated in place -- the caller keeps one
//     partial buffer per weight for the whole backward pass (all rollout steps add into it) and reduces the
//     splits once at the end, in a fixed order: deterministic, no atomics.
#include <stdlib.h>

#include "common.h"

namespace {

struct TnArgs {
    const float* G; const float* X; float* C; float* bias;
    int ldg, ldx, M, N, K, chunk, accumulate;
};

// Several (G, X) row segments reduced by ONE launch (round 5): back-propagation through time uses every weight once per
// rollout step, dW = sum_t G_t^T X_t is one product over the concatenated rows.  The pointers travel as kernel arguments (no
// device-side table: a captured HIP graph holds them by value); rend[s] = rows of segments 0 .. s.
constexpr int TN_MAXSEG = 20;
struct TnSegs {
    const float* G[TN_MAXSEG]; const float* X[TN_MAXSEG];
    int rend[TN_MAXSEG];
    int nseg;
};

#ifndef TOCVP_TN_STAGES
#define TOCVP_TN_STAGES 3
#endif
constexpr int TN_T = 128, TN_ROWS = 16, TN_STAGES = TOCVP_TN_STAGES;
constexpr int TN_OPER = TN_ROWS * TN_T;            // floats per operand tile (8 KiB)
constexpr int TN_STAGE = 2 * TN_OPER;              // floats per stage

__global__ __launch_bounds__(256, TN_STAGES == 3 ? 3 : 2) void gemm_tn_f32_kernel(TnArgs p) {
    __shared__ __attribute__((aligned(1024))) float lds[TN_STAGES * TN_STAGE];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wy = wave >> 1, wx = wave & 1;
    const int k0 = blockIdx.x * TN_T, n0 = blockIdx.y * TN_T, z = blockIdx.z;
    const int r_lo = min(p.M, z * p.chunk), r_hi = min(p.M, r_lo + p.chunk);
    const int nit = (r_hi - r_lo) / TN_ROWS;
    if (nit == 0 && p.accumulate) return;          // nothing to add (workgroup-uniform)

    // DMA source offsets (floats) of this lane inside a 2-row block: 128-byte segment s = lane >> 3 holds
    // row parity s & 1, columns (s >> 1) * 32 + (lane & 7) * 4 ..
    const int seg = lane >> 3;
    const int d_row = seg & 1, d_col = (seg >> 1) * 32 + (lane & 7) * 4;
    const size_t goff = (size_t)d_row * p.ldg + n0 + d_col, xoff = (size_t)d_row * p.ldx + k0 + d_col;
    auto dma = [&](int it) {
        float* st = lds + (it % TN_STAGES) * TN_STAGE;
        const int row = r_lo + it * TN_ROWS + wave * 4;           // this wave: row pairs 2 * wave, 2 * wave + 1
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* gs = p.G + (size_t)(row + 2 * i) * p.ldg + goff;
            const float* xs = p.X + (size_t)(row + 2 * i) * p.ldx + xoff;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gs,
                                             (__attribute__((address_space(3))) void*)(st + (wave * 2 + i) * 256),
                                             16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xs,
                                             (__attribute__((address_space(3))) void*)(st + TN_OPER + (wave * 2 + i) * 256),
                                             16, 0, 0);
        }
    };

    // The accumulators START from the partial sums already in c_part (accumulate) -- the loads are issued
    // here and land behind the DMA prologue, so the epilogue is stores only.
    float* C = p.C + (size_t)z * p.N * p.K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wy * 64 + i * 32 + acc_row(r, h), k = k0 + wx * 64 + j * 32 + l31;
                acc[i][j][r] = p.accumulate ? C[(size_t)n * p.K + k] : 0.f;
            }
    const bool want_bias = p.bias != nullptr && blockIdx.x == 0;        // workgroup-uniform
    float bsum = 0.f;
    const int b_off = (t >> 7) * 4 * 256 + ((t & 127) >> 5) * 64 + (t & 31);   // half of the row pairs, one column

    for (int it = 0; it < TN_STAGES - 1 && it < nit; ++it) dma(it);
    for (int it = 0; it < nit; ++it) {
        // LDS-DMA completion is tracked by vmcnt only; 4 DMA instructions per wave and stage (the older
        // loads of the initial accumulators return in order before them)
        if (TN_STAGES == 4 && it + 2 < nit) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (it + 1 < nit) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();              // stage `it` has landed for every wave; stage it - 1 is no longer read
        if (it + TN_STAGES - 1 < nit) dma(it + TN_STAGES - 1);
        const float* gs = lds + (it % TN_STAGES) * TN_STAGE;
        const float* xs = gs + TN_OPER;
#pragma unroll
        for (int s = 0; s < TN_ROWS / 2; ++s) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = gs[s * 256 + (wy * 2 + i) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = xs[s * 256 + (wx * 2 + j) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
        }
        if (want_bias) {
#pragma unroll
            for (int s = 0; s < 4; ++s) bsum += gs[b_off + s * 256] + gs[b_off + s * 256 + 32];
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wy * 64 + i * 32 + acc_row(r, h), k = k0 + wx * 64 + j * 32 + l31;
                C[(size_t)n * p.K + k] = acc[i][j][r];
            }
    if (want_bias) {
        __syncthreads();
        lds[t] = bsum;
        __syncthreads();
        if (t < 128) {
            float* b = p.bias + (size_t)z * p.N + n0 + t;
            const float v = lds[t] + lds[t + 128];
            *b = p.accumulate ? *b + v : v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix cores with SPLIT operands (round 3): every fp32 value is two bf16 planes
// (x = hi + lo, 16 significant bits, the fp32 exponent range -- gradients need no scale), the product
// hi*lo + lo*hi + hi*hi by three v_mfma_f32_32x32x16_bf16 into one fp32 accumulator (~2^-17 per product, the
// arithmetic the data-gradient GEMMs use).  Both operands are reduced over their ROW index, i.e. the MFMA wants 8
// consecutive token rows of ONE column per lane: the tiles stay row-major in LDS ([32 token rows][128 columns] per
// plane, split while staged through registers) and both fragments are hardware-transposed reads (ds_read_b64_tr_b16;
// rows 320 bytes apart so that the four rows of a read tile the 64 banks).  128 x 128 output tile per 4-wave
// workgroup, 32 token rows per stage, two stages (80 KB: two workgroups per CU), the next stage's global loads in
// flight under the current stage's 24 MFMAs per wave.  Same split-K / accumulate / bias contract as the fp32 kernel;
// the bias sums are taken from the fp32 values in the staging registers (exact).
// ------------------------------------------------------------------------------------------------
typedef __bf16 tn_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tn_bf16x4 __attribute__((ext_vector_type(4)));
typedef short tn_s16x4 __attribute__((ext_vector_type(4)));
constexpr int TB_ROWS = 32;

template <int RS>
__device__ __forceinline__ tn_bf16x8 tn_tr_frag(const unsigned char* addr) {
    typedef __attribute__((address_space(3))) tn_s16x4* lp;
    union { tn_s16x4 s[2]; tn_bf16x8 b; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(addr));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(addr + 4 * RS));
    return u.b;
}

// MI = 2: 128 (n) x 128 (k) output tile, a wave owns 64 x 64, two LDS stages (80 KB), one barrier per stage.
// MI = 4: 256 (n) x 128 (k), a wave owns 128 x 64 = 4 x 2 accumulator tiles (the kernel is LDS- and VALU-bound: per
//         stage and wave 32 transposed reads + 16 plane stores + 32 elements to split for 24 MFMAs at MI = 2; 48 + 24
//         + 48 for 48 MFMAs at MI = 4); ONE LDS stage (56 KB, two workgroups per CU) filled from registers behind a
//         second barrier.  Measured slower in the training step (see the launch function): off by default.
template <int MI, bool MULTI>
__global__ __launch_bounds__(256, 2) void gemm_tn_bf16x3_kernel(TnArgs p, TnSegs sg) {
    constexpr int GN = 64 * MI;                                   // columns of the G (n) tile
    constexpr int RSG = GN * 2 + 64, RSX = 320;                   // row pitches: 64 bytes past a multiple of 256
    constexpr int GPLANE = TB_ROWS * RSG, XPLANE = TB_ROWS * RSX;
    constexpr int STAGE = 2 * GPLANE + 2 * XPLANE, NSTAGE = MI == 2 ? 2 : 1;
    constexpr int GPT = GN / 32;                                  // float4 of G per thread and stage (4 or 8)
    constexpr int GTR = GN / 4, GRS = 256 / GTR;                  // threads per G row, rows per staging pass
    __shared__ __attribute__((aligned(16))) unsigned char lds[NSTAGE * STAGE];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wy = wave >> 1, wx = wave & 1;
    const int k0 = blockIdx.x * TN_T, n0 = blockIdx.y * GN, z = blockIdx.z;
    const int r_lo = min(p.M, z * p.chunk), r_hi = min(p.M, r_lo + p.chunk);
    const int nit = (r_hi - r_lo) / TB_ROWS;
    if (nit == 0 && p.accumulate) return;          // nothing to add (workgroup-uniform)

    // staging: G rows gr + GRS * i, columns gc .. gc + 3;  X rows xr + 8 * i, columns xc .. xc + 3
    const int gr = t / GTR, gc = (t % GTR) * 4;
    const int xr = t >> 5, xc = (t & 31) * 4;
    const bool want_bias = p.bias != nullptr && blockIdx.x == 0;        // workgroup-uniform
    f32x4 gq[GPT], xq[4];
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};
    int cs = 0;                                                   // MULTI: segment of the tile being loaded (tiles never straddle)
    auto gload = [&](int it) {
        size_t row = (size_t)(r_lo + it * TB_ROWS);
        const float* G = p.G;
        const float* X = p.X;
        if (MULTI) {
            const int R = r_lo + it * TB_ROWS;
            while (cs + 1 < sg.nseg && R >= sg.rend[cs]) ++cs;     // workgroup-uniform, monotonic in ``it``
            G = sg.G[cs];
            X = sg.X[cs];
            row = (size_t)(R - (cs ? sg.rend[cs - 1] : 0));
        }
#pragma unroll
        for (int i = 0; i < GPT; ++i)
            gq[i] = *reinterpret_cast<const f32x4*>(G + (row + gr + GRS * i) * p.ldg + n0 + gc);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            xq[i] = *reinterpret_cast<const f32x4*>(X + (row + xr + 8 * i) * p.ldx + k0 + xc);
    };
    auto split_store = [&](unsigned char* dst, int plane, const f32x4 v) {
        tn_bf16x4 hi, lo;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            hi[u] = (__bf16)v[u];
            lo[u] = (__bf16)(v[u] - (float)hi[u]);
        }
        *reinterpret_cast<tn_bf16x4*>(dst) = hi;
        *reinterpret_cast<tn_bf16x4*>(dst + plane) = lo;
    };
    auto lstore = [&](int buf) {
        unsigned char* st = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            split_store(st + (gr + GRS * i) * RSG + gc * 2, GPLANE, gq[i]);
            if (want_bias) bs += gq[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) split_store(st + 2 * GPLANE + (xr + 8 * i) * RSX + xc * 2, XPLANE, xq[i]);
    };

    float* C = p.C + (size_t)z * p.N * p.K;
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wy * 32 * MI + i * 32 + acc_row(r, h), k = k0 + wx * 64 + j * 32 + l31;
                acc[i][j][r] = p.accumulate ? C[(size_t)n * p.K + k] : 0.f;
            }

    // transposed-read address of this lane inside a 16-row k-step: the 16-lane group (lane >> 4) & 1 takes columns
    // 16 .. 31 of the fragment, lane 4q + p of a group supplies row 8h + q, columns 4p .. 4p + 3
    const int i16 = lane & 15, c16 = ((lane >> 4) & 1) * 16;
    const int g_off = (8 * h + (i16 >> 2)) * RSG + (c16 + 4 * (i16 & 3)) * 2 + wy * 64 * MI;
    const int x_off = (8 * h + (i16 >> 2)) * RSX + (c16 + 4 * (i16 & 3)) * 2 + wx * 128;
    auto compute = [&](int buf) {
        const unsigned char* gs = lds + buf * STAGE + g_off;
        const unsigned char* xs = lds + buf * STAGE + 2 * GPLANE + x_off;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            tn_bf16x8 ah[MI], al[MI], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = tn_tr_frag<RSG>(gs + kk * 16 * RSG + i * 64);
                al[i] = tn_tr_frag<RSG>(gs + kk * 16 * RSG + i * 64 + GPLANE);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = tn_tr_frag<RSX>(xs + kk * 16 * RSX + j * 64);
                bl[j] = tn_tr_frag<RSX>(xs + kk * 16 * RSX + j * 64 + XPLANE);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    };

    if (nit > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
        const bool more = it + 1 < nit;               // workgroup-uniform
        if (more) gload(it + 1);
        if (NSTAGE == 2) {
            compute(it & 1);
            if (more) lstore((it + 1) & 1);
            __syncthreads();
        } else {
            compute(0);
            __syncthreads();                          // every wave has read the stage
            if (more) lstore(0);
            __syncthreads();
        }
    }

#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wy * 32 * MI + i * 32 + acc_row(r, h), k = k0 + wx * 64 + j * 32 + l31;
                C[(size_t)n * p.K + k] = acc[i][j][r];
            }
    if (want_bias) {                                  // GRS threads (gr = 0 .. GRS - 1) share the columns gc .. gc + 3
        float* red = reinterpret_cast<float*>(lds);
        *reinterpret_cast<f32x4*>(red + gr * GN + gc) = bs;
        __syncthreads();
        if (t < GN) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < GRS; ++q) v += red[q * GN + t];
            float* b = p.bias + (size_t)z * p.N + n0 + t;
            *b = p.accumulate ? *b + v : v;
        }
    }
}

}  // namespace

// `splits` may be smaller than the number of slices of the caller's buffer when it accumulates (few token
// rows: fewer, longer reductions); slices >= splits are then left as they are.
extern "C" int tocvp_gemm_tn_f32(const float* G, int ldg, const float* X, int ldx, float* c_part,
                                 float* bias_part, int M, int N, int K, int splits, int accumulate,
                                 void* stream) {
    TOCVP_CHECK_ARG(G && X && c_part);
    TOCVP_CHECK_ARG(M > 0 && M % TN_ROWS == 0 && N > 0 && N % TN_T == 0 && K > 0 && K % TN_T == 0);
    TOCVP_CHECK_ARG(ldg >= N && ldx >= K && ldg % 4 == 0 && ldx % 4 == 0);
    TOCVP_CHECK_ARG(tocvp_aligned16(G) && tocvp_aligned16(X));
    TOCVP_CHECK_ARG(splits >= 1 && splits <= 65535);
    int chunk = (M + splits - 1) / splits;
    chunk = (chunk + TN_ROWS - 1) / TN_ROWS * TN_ROWS;
    TnArgs a{G, X, c_part, bias_part, ldg, ldx, M, N, K, chunk, accumulate ? 1 : 0};
    const dim3 grid(K / TN_T, N / TN_T, splits);
    hipLaunchKernelGGL(gemm_tn_f32_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return tocvp_launch_status();
}

// Same contract on split bf16 operands (two planes each, three products: ~2^-17 per product); M % 32 == 0, and the
// chunk of a split is rounded up to 32 rows.
extern "C" int tocvp_gemm_tn_bf16x3_f32(const float* G, int ldg, const float* X, int ldx, float* c_part,
                                        float* bias_part, int M, int N, int K, int splits, int accumulate,
                                        void* stream) {
    TOCVP_CHECK_ARG(G && X && c_part);
    TOCVP_CHECK_ARG(M > 0 && M % TB_ROWS == 0 && N > 0 && N % TN_T == 0 && K > 0 && K % TN_T == 0);
    TOCVP_CHECK_ARG(ldg >= N && ldx >= K && ldg % 4 == 0 && ldx % 4 == 0);
    TOCVP_CHECK_ARG(tocvp_aligned16(G) && tocvp_aligned16(X));
    TOCVP_CHECK_ARG(splits >= 1 && splits <= 65535);
    int chunk = (M + splits - 1) / splits;
    chunk = (chunk + TB_ROWS - 1) / TB_ROWS * TB_ROWS;
    TnArgs a{G, X, c_part, bias_part, ldg, ldx, M, N, K, chunk, accumulate ? 1 : 0};
    // (256-row tiles -- gemm_tn_bf16x3_kernel<4> -- were built in round 3 and measured slower in the step, 520.3 vs 511.8 ms: gone)
    const dim3 grid(K / TN_T, N / TN_T, splits);
    hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<2, false>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), a, TnSegs{});
    return tocvp_launch_status();
}

// dW (+ db) over SEVERAL row segments in one launch: G[s] (rows[s] x N, row stride ldg), X[s] (rows[s] x K, row stride ldx),
// s < nseg <= 20, every rows[s] a multiple of 32.  ``G`` / ``X`` / ``rows`` are HOST arrays: their entries are copied into
// the kernel's arguments.  Same partial buffers, split and accumulate semantics as tocvp_gemm_tn_bf16x3_f32 on the
// concatenated rows; the sum over rows is associated per split of the CONCATENATED row range.
extern "C" int tocvp_gemm_tn_bf16x3_multi_f32(const float* const* G, const float* const* X, const int* rows, int nseg, int ldg,
                                              int ldx, float* c_part, float* bias_part, int N, int K, int splits,
                                              int accumulate, void* stream) {
    TOCVP_CHECK_ARG(G && X && rows && c_part && nseg >= 1 && nseg <= TN_MAXSEG);
    TOCVP_CHECK_ARG(N > 0 && N % TN_T == 0 && K > 0 && K % TN_T == 0);
    TOCVP_CHECK_ARG(ldg >= N && ldx >= K && ldg % 4 == 0 && ldx % 4 == 0);
    TOCVP_CHECK_ARG(splits >= 1 && splits <= 65535);
    TnSegs sg{};
    long total = 0;
    for (int i = 0; i < nseg; ++i) {
        TOCVP_CHECK_ARG(G[i] && X[i] && rows[i] > 0 && rows[i] % TB_ROWS == 0);
        TOCVP_CHECK_ARG(tocvp_aligned16(G[i]) && tocvp_aligned16(X[i]));
        total += rows[i];
        TOCVP_CHECK_ARG(total < 0x7fffffffL);
        sg.G[i] = G[i];
        sg.X[i] = X[i];
        sg.rend[i] = (int)total;
    }
    sg.nseg = nseg;
    const int M = (int)total;
    int chunk = (M + splits - 1) / splits;
    chunk = (chunk + TB_ROWS - 1) / TB_ROWS * TB_ROWS;
    TnArgs a{G[0], X[0], c_part, bias_part, ldg, ldx, M, N, K, chunk, accumulate ? 1 : 0};
    const dim3 grid(K / TN_T, N / TN_T, splits);
    hipLaunchKernelGGL((gemm_tn_bf16x3_kernel<2, true>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), a, sg);
    return tocvp_launch_status();
}
