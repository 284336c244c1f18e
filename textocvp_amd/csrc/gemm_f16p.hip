// f16x3 GEMM on fp16 OPERAND PLANES of both operands:  C = act(A W^T + bias) + R
//
// Replaces nn.Linear of the predictor blocks (reference models/Blocks/attention.py:167-175, 355-359,
// Predictors/text_cond_OCVP.py:49-50) and of the DINOv2 ViT blocks when the activation arrives as planes.
//   A: (M, 2, K) fp16 planes of 2^8 x   (hi | lo, written by the producer: LayerNorm, attention, GEMM epilogue)
//   W: (N, 2, K) fp16 planes of 2^10 w  (tocvp_split_weights_planes_f16, once per weight version)
//   product = Al Wh + Ah Wl + Ah Wh on v_mfma_f32_32x32x16_f16, fp32 accumulate: fp32-class (~2^-21).
//
// The round-1 kernels fed the weights from L2 as per-wave register fragments and split A in the k-loop:
// 48 KB crossed the CU's 64 B/clk vector-memory path per 128x128x32 tile (as long as the MFMAs took) and
// ~6 vector instructions per MFMA went into the split.  Here (geometry of the 256x256 bf16 template of
// cdna_hip_programming.md section 5, with two fp16 planes of 32 k in the place of 64 bf16 k):
//   * (64 MI) x 256 tile, 8 waves as 2 (M) x 4 (N), a wave owns (32 MI) x 64 = MI x 2 accumulator tiles;
//     MI = 4 (256 x 256: 21 B/clk of operand traffic per CU) or MI = 2 (128 x 256, more workgroups for N = 512);
//   * BOTH operands go global -> LDS by DMA (global_load_lds, 16 B per lane, no staging registers, no
//     conversion instructions); rows are 128 B = [hi k0..31 | lo k0..31]; the image is lane-linear, the
//     conflict-free order is made on the SOURCE side: physical 16-byte chunk c of row r holds logical chunk
//     c ^ ((r >> 1) & 7), so the 16 rows of every ds_read_b128 lane group hit 16 distinct slots;
//   * two LDS stages (2 x 64 KB at MI = 4), one barrier per k-tile, two waves per SIMD; the fragments of BOTH
//     k-steps of a k-tile are in registers before its barrier, so the stage is refilled right behind the barrier
//     (DMA 1.5 k-tiles ahead with two stages) and every fragment read is issued one MFMA block ahead of its use.
//
// Measured (MI355X, 38400 rows, dense random operands, scripts/gemm_shapes.py; round-1 kernel with the in-loop
// split in brackets): 2048 x 512  311 us = 259 TFLOP/s [383], 2048 x 2048  925 us = 348 [1190], 1536 x 512  227 [231],
// 512 x 2048  331 [284], 512 x 512  106 [77].  The k-loop runs at ~420 TFLOP/s; a workgroup owns its CU alone and pays
// ~24 us of prologue + epilogue without cover (its 256 KB of output leave at the ~14 B/clk/CU store-issue rate), so
// it wins only on wide or deep products.  Ablations (scripts/probes/gemm16p_ablate.hip, 2048 x 512): no DMA -14 %,
// no stores -12 %, MFMAs removed 0.63 x.  A two-workgroups-per-CU form (256 x 128 tiles, three 16-deep stages, a
// barrier per 24 MFMAs) measured 389 us on the same product, with or without a start offset that de-phases the two
// workgroups of a CU: slower, not kept.  Used for plane inputs when
// TOCVP_PRESPLIT selects them (off by default: neutral in the rollout, models/Blocks/attention.py).
#include <stdlib.h>

#include "common.h"

// timing experiments only (scripts/probes/gemm16p_ablate.hip): 1 = no DMA in the k-loop, 2 = no MFMAs,
// 3 = no LDS fragment reads in the k-loop, 4 = no output stores
#ifndef TOCVP_GEMM_P2_ABLATE
#define TOCVP_GEMM_P2_ABLATE 0
#endif

namespace {

constexpr int PABL = TOCVP_GEMM_P2_ABLATE;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr float SA = TOCVP_F16X3_ACT_SCALE, SW = TOCVP_F16X3_WEIGHT_SCALE;
constexpr int BK = 32, ROWB = 128;                  // k per stage, bytes per LDS row (2 planes x 32 k x 2 B)

struct PArgs {
    const unsigned char* A; const unsigned char* W;
    const float* bias; const float* R; int ldr;
    float* C; int ldc; int c_split;
    int M, N, K, act;
};

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == TOCVP_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == TOCVP_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

template <int MI>
__global__ __launch_bounds__(512, 2) void gemm_f16_planes2_kernel(PArgs p) {
    constexpr int NSTAGE = 2;
    constexpr int NI = 2, BM = 64 * MI, BN = 256;
    constexpr int A_STAGE = BM * ROWB, B_STAGE = BN * ROWB, STAGE = A_STAGE + B_STAGE;
    constexpr int A_DMA = A_STAGE / (8 * 1024), B_DMA = B_STAGE / (8 * 1024);    // 1 KiB instructions per wave
    static_assert(NSTAGE * STAGE <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTAGE * STAGE];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntn = p.N / BN;
    int bid = blockIdx.x;
    {   // XCD-contiguous tile order (bijective for any grid size): neighbours in the grid share A / W panels in one L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    const int nk = p.K / BK;

    // ---- DMA source offsets: instruction i of this wave writes LDS rows (wave * CNT + i) * 8 .. + 7
    unsigned voff_a[A_DMA], voff_b[B_DMA];
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) {
        const int row = (wave * A_DMA + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);                 // logical chunk landing in this lane's slot
        const int grow = min(m0 + row, p.M - 1);
        voff_a[i] = (unsigned)((((size_t)grow * 2 + (c >> 2)) * p.K + (c & 3) * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < B_DMA; ++i) {
        const int row = (wave * B_DMA + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voff_b[i] = (unsigned)((((size_t)(n0 + row) * 2 + (c >> 2)) * p.K + (c & 3) * 8) * 2);
    }
    auto dma = [&](int stage, int kt) {
        const unsigned char* abase = p.A + (size_t)kt * (BK * 2);    // uniform
        const unsigned char* wbase = p.W + (size_t)kt * (BK * 2);
        unsigned char* la = lds + stage * STAGE + (wave * A_DMA) * 1024;
        unsigned char* lb = lds + stage * STAGE + A_STAGE + (wave * B_DMA) * 1024;
#pragma unroll
        for (int i = 0; i < A_DMA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase + voff_a[i]),
                                             (__attribute__((address_space(3))) void*)(la + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_DMA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + voff_b[i]),
                                             (__attribute__((address_space(3))) void*)(lb + i * 1024), 16, 0, 0);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: row = base + l31 (base a multiple of 32), chunk (plane * 4 + ks * 2 + h) ^ ((l31 >> 1) & 7)
    const int x16 = ((l31 >> 1) & 7) << 4;
    const int a_row = (wm * (32 * MI) + l31) * ROWB, b_row = A_STAGE + (wn * 64 + l31) * ROWB;
    struct Frags { f16x8 a[MI][2], b[NI][2]; };
    auto read_frags = [&](Frags& f, int stage, int ks) {
        const unsigned char* sb = lds + stage * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int coff = (((s * 4 + (PABL == 3 ? 0 : ks) * 2 + h) << 4) ^ x16);
#pragma unroll
            for (int i = 0; i < MI; ++i) f.a[i][s] = *reinterpret_cast<const f16x8*>(sb + a_row + i * 32 * ROWB + coff);
#pragma unroll
            for (int j = 0; j < NI; ++j) f.b[j][s] = *reinterpret_cast<const f16x8*>(sb + b_row + j * 32 * ROWB + coff);
        }
    };
    auto mfma = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (PABL == 2) {
                    asm volatile("" :: "v"(f.a[i][0]), "v"(f.a[i][1]), "v"(f.b[j][0]), "v"(f.b[j][1]));
                    continue;
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][1], f.b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][0], f.b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][0], f.b[j][0], acc[i][j], 0, 0, 0);
            }
    };

    // Software pipeline.  Both k-steps of k-tile kt sit in REGISTERS (f0, f1) before the barrier of iteration kt, so
    // its stage is refilled right behind that barrier with k-tile kt + 2 (two stages, DMA 1.5 k-tiles ahead of its
    // use), and the fragment reads of k-tile kt + 1 are issued a whole MFMA block (24 or 48 MFMAs) before their use.
    Frags f0, f1;
    // LDS-DMA completion is tracked by vmcnt only: the compiler does not wait for it at a barrier, so every wave
    // waits for its own share (asm, invisible to the waitcnt pass) and the barrier then covers everybody's.
    dma(0, 0);
    dma(1, nk > 1 ? 1 : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_frags(f0, 0, 0);
    read_frags(f1, 0, 1);
    for (int kt = 0; kt + 1 < nk; ++kt) {
        mfma(f0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of k-tile kt + 1 (issued one iteration ago)
        __syncthreads();                               // every wave holds k-tile kt in registers; k-tile kt + 1 has landed
        if (PABL != 1) dma(kt & 1, kt + 2 < nk ? kt + 2 : nk - 1);   // past the end: a harmless re-load, never read
        read_frags(f0, (kt + 1) & 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma(f1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(f1, (kt + 1) & 1, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    mfma(f0);
    mfma(f1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-load of the last k-tile
    __syncthreads();                                   // the epilogue reuses the stages

    // ---- epilogue: 32-row x 64-column blocks staged through LDS, written back as 16-byte rows
    constexpr int SS = 64 + 4;
    float* stage_f = reinterpret_cast<float*>(lds) + wave * (32 * SS);
    constexpr int F4R = 64 / 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage_f[acc_row(r, h) * SS + j * 32 + l31] = act_fn(acc[i][j][r] * (1.f / (SA * SW)) + bv, p.act);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < (32 * F4R) / 64; ++it) {
            const int idx = lane + 64 * it;
            const int rr = idx / F4R, c4 = (idx % F4R) * 4;
            const int row = m0 + wm * (32 * MI) + i * 32 + rr, col = n0 + wn * 64 + c4;
            if (row < p.M) {
                f32x4 v = *reinterpret_cast<const f32x4*>(stage_f + rr * SS + c4);
                if (p.R) v += *reinterpret_cast<const f32x4*>(p.R + (size_t)row * p.ldr + col);
                if (PABL == 4 && v[0] != 12345.f) continue;
                if (p.c_split)
                    tocvp_store_planes4(p.C, (size_t)row * 2 * p.N + col, (size_t)p.N, v, 22);
                else
                    *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + col) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// W (N, K) fp32 -> (N, 2, K) fp16 planes of 2^10 w
__global__ __launch_bounds__(256) void split_weights_planes_f16_kernel(const float* __restrict__ w,
                                                                       _Float16* __restrict__ out, long n, int K) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long row = i / K;
    const int k = (int)(i - row * K);
    const float X = fminf(fmaxf(w[i] * SW, -65504.f), 65504.f);
    const _Float16 hi = (_Float16)X;
    out[(row * 2 + 0) * K + k] = hi;
    out[(row * 2 + 1) * K + k] = (_Float16)(X - (float)hi);
}

}  // namespace

extern "C" int tocvp_split_weights_planes_f16(const float* w, void* out, int N, int K, void* stream) {
    TOCVP_CHECK_ARG(w && out && N > 0 && K > 0);
    const long n = (long)N * K;
    hipLaunchKernelGGL(split_weights_planes_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, static_cast<_Float16*>(out), n, K);
    return tocvp_launch_status();
}

extern "C" int tocvp_gemm_f16planes_f32(const void* A_planes, const void* W_planes, const float* bias,
                                        const float* R, int ldr, void* C, int c_split, int ldc, int M, int N,
                                        int K, int act, void* stream) {
    TOCVP_CHECK_ARG(A_planes && W_planes && C);
    TOCVP_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % 32) == 0 && (N % 256) == 0);
    TOCVP_CHECK_ARG(c_split || (ldc >= N && (ldc & 3) == 0 && tocvp_aligned16(C)));
    TOCVP_CHECK_ARG(R == nullptr || (ldr >= N && (ldr & 3) == 0 && tocvp_aligned16(R)));
    TOCVP_CHECK_ARG(act >= TOCVP_ACT_NONE && act <= TOCVP_ACT_GELU);
    TOCVP_CHECK_ARG((size_t)M * 2 * K * 2 < 0xffffffffull && (size_t)N * 2 * K * 2 < 0xffffffffull);   // 32-bit DMA offsets
    if (!tocvp_aligned16(A_planes) || !tocvp_aligned16(W_planes)) return TOCVP_EALIGN;
    if (M == 0) return TOCVP_OK;
    PArgs p{static_cast<const unsigned char*>(A_planes), static_cast<const unsigned char*>(W_planes), bias, R, ldr,
            static_cast<float*>(C), ldc, c_split ? 1 : 0, M, N, K, act};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ntn = N / 256;
    // 256-row tiles when they still fill the chip about twice over, else 128-row tiles
    const long big = (long)((M + 255) / 256) * ntn;
    static const int force = []() { const char* e = getenv("TOCVP_GEMM_P2_MI"); return e ? atoi(e) : 0; }();
    const int mi = force ? force : (big >= 448 ? 4 : 2);
    if (mi == 4)
        hipLaunchKernelGGL(gemm_f16_planes2_kernel<4>, dim3((unsigned)(((M + 255) / 256) * ntn)), dim3(512), 0, s, p);
    else
        hipLaunchKernelGGL(gemm_f16_planes2_kernel<2>, dim3((unsigned)(((M + 127) / 128) * ntn)), dim3(512), 0, s, p);
    return tocvp_launch_status();
}
