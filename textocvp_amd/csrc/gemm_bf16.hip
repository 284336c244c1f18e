// GEMM with SPLIT-bf16 operands on the bf16 matrix cores, fp32 in/out, same fused epilogue as
// gemm.hip:   C = act(A W^T + bias + rowvec) + R
//
//   NS = 2 ("bf16x3"): x = hi + lo,        products hh + hl + lh          (~2^-16 per product)
//   NS = 3 ("bf16x6"): x = hi + mid + lo,  products hh + hm + mh + hl + mm + lh  (~2^-23: fp32-class)
// v_mfma_f32_32x32x16_bf16 does 16 k per 32 cycles versus 2 k per 64 cycles for the fp32 MFMA, so
// bf16x3 / bf16x6 need 3/16 / 6/16 of the matrix-pipe cycles of the exact-fp32 kernel.
//
// A (activations, fp32) is split while its tile is staged into LDS; W is pre-split once on the host
// side of the ABI (tocvp_split_weights_bf16: (N, NS, K) bf16 planes) because it is reused by every
// row block and every call.  128x128x32 tiles, 4 waves x (64x64), double-buffered LDS; LDS rows are
// [NS planes of 32 bf16 | 16 B pad] -> 144 B / 208 B strides, conflict-free for ds_read_b128.
#include <stdlib.h>

#include "common.h"
#include <type_traits>

// timing experiments only (scripts/probes/gemm_ablate.hip): 1 = no weight-fragment loads in the k-loop,
// 2 = no A loads, 3 = no A conversion / LDS stores, 4 = no MFMAs (and no LDS reads), 5 = no epilogue.
#ifndef TOCVP_GEMM_ABLATE
#define TOCVP_GEMM_ABLATE 0
#endif

namespace {
constexpr int GABL = TOCVP_GEMM_ABLATE;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr size_t WFRAG_WS_CTR_BYTES = 4096;                 // split-K workspace: 1024 tile counters ...
constexpr int WFRAG_WS_RECORDS = 1024;                      // ... and 1024 accumulator records of 16 KB

struct GemmArgs {
    const float* A; int lda;
    const __bf16* W;             // (N, NS, K)
    const float* bias;
    const float* R; int ldr;
    const float* rowvec; int rv_div, rv_mod, rv_flip;
    float* C; int ldc;
    int M, N, K, act;
    int a_split;                 // A is (M, NS, K) bf16 planes (producer already split it)
    int c_split;                 // write C as (M, NS, N) bf16 planes instead of fp32
    int ksplit;                  // split-K (wfrag 64 x 64 kernel, SK = true): grid.y slices of K, > 1 needs the workspace
    float* ws_part;              // [tile][slice] raw accumulator records (lane order), 16 KB each
    unsigned* ws_ctr;            // [tile] arrival counters, zero between launches (the last arriver re-arms its own)
    int sk_fence;                // diagnostics (TOCVP_GEMM_KSPLIT_FENCE=1): agent release in front of the arrival count
};

// R operand of the epilogue: residual (added) or, for TOCVP_ACT_GATE, the tensor whose sign gates the output
__device__ __forceinline__ float with_r(float v, float r, int act) {
    return act == TOCVP_ACT_GATE ? (r > 0.f ? v : 0.f) : v + r;
}
__device__ __forceinline__ f32x4 with_r4(f32x4 v, f32x4 r, int act) {
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = with_r(v[u], r[u], act);
    return v;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == TOCVP_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == TOCVP_ACT_GELU) return tocvp_gelu(v);
    return v;
}

// compile-time forms for the staged epilogues: with the run-time `act` inside their unrolled loops every staged element
// carried a ladder of scalar branches (and the inlined erff of the GELU it did not take)
template <int ACT> __device__ __forceinline__ float apply_act_c(float v) {
    if constexpr (ACT == TOCVP_ACT_RELU) return fmaxf(v, 0.0f);
    else if constexpr (ACT == TOCVP_ACT_GELU) return tocvp_gelu(v);
    else return v;
}
template <int ACT> __device__ __forceinline__ f32x4 with_r4_c(f32x4 v, f32x4 r) {
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = ACT == TOCVP_ACT_GATE ? (r[u] > 0.f ? v[u] : 0.f) : v[u] + r[u];
    return v;
}
// run `f(std::integral_constant<int, act>)` for the run-time activation code
template <class F> __device__ __forceinline__ void dispatch_act(int act, F&& f) {
    if (act == TOCVP_ACT_RELU) f(std::integral_constant<int, TOCVP_ACT_RELU>{});
    else if (act == TOCVP_ACT_GELU) f(std::integral_constant<int, TOCVP_ACT_GELU>{});
    else if (act == TOCVP_ACT_GATE) f(std::integral_constant<int, TOCVP_ACT_GATE>{});
    else f(std::integral_constant<int, TOCVP_ACT_NONE>{});
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// 16-bit operand element of the split kernels.  bf16: 8 significant bits per plane (x3 -> ~2^-16,
// x6 -> ~2^-23).  fp16: 11 significant bits per plane, so TWO planes already carry 22 bits and the
// three products hh + hl + lh are fp32-class (~2^-22) at half the MFMA count of bf16x6 ("f16x3");
// valid while |x| < 255 after the operand pre-scaling described at Elem<true>.
template <bool F16> struct Elem;
template <> struct Elem<false> {
    using T = __bf16; using V8 = bf16x8; using V4 = bf16x4;
    static constexpr float SA = 1.f, SW = 1.f;          // bf16 has the fp32 exponent range
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Elem<true> {
    using T = _Float16; using V8 = f16x8; using V4 = f16x4;
    // The matrix core flushes fp16 SUBNORMAL inputs, i.e. a lo plane below 6.1e-5 would vanish
    // (measured: 2e-4 drift of the rollout).  Operands are therefore pre-scaled by exact powers of
    // two (undone in the epilogue): activations x 2^8 (fp32-class for |x| < 255, 11-bit up to 511,
    // saturating beyond; the flush threshold drops to 2.4e-7 absolute), weights x 2^10 (|w| < 63).
    static constexpr float SA = 256.f, SW = 1024.f;
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// NS planes, BM x BN x BK tile, WM x WN per wave, NW waves (4: one per SIMD, 8: two per SIMD so the
// partner's MFMAs cover this wave's split / LDS-store / barrier phases), MINW = waves per SIMD the
// register budget must allow (occupancy across workgroups).
template <int NS, int BM, int BN, int WM, int WN, int BK, int NW, int MINW>
__global__ __launch_bounds__(NW * 64, MINW) void gemm_bf16_split_kernel(GemmArgs p) {
    constexpr int NT = NW * 64;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == NW, "wave grid must cover the tile");
    constexpr int ROWB = NS * BK * 2 + 16;           // bytes per LDS row (planes + pad)
    constexpr int AF4 = BK / 4;                      // float4 per A row per k-tile
    constexpr int RA = (BM * AF4) / NT;              // float4 A loads per thread per tile
    static_assert((BM * AF4) % NT == 0, "A tile must split evenly");
    constexpr int PPP = BK / 8;                      // 16-byte parts per plane per row
    constexpr int WCH = BN * NS * PPP;               // 16-byte W chunks per tile
    constexpr int RW = (WCH + NT - 1) / NT;

    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (BM + BN) * ROWB];
    unsigned char* As = lds;
    unsigned char* Bs = lds + 2 * BM * ROWB;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntn = (p.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const int lr = t / AF4, lc = (t % AF4) * 4;
    constexpr int RSTEP = NT / AF4;                  // rows covered per load pass

    f32x4 ra[RA], rw[RW];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            // unconditional loads from clamped rows (see gemm_bf16_wfrag_kernel)
            const int row = min(m0 + lr + RSTEP * i, p.M - 1);
            ra[i] = *reinterpret_cast<const f32x4*>(p.A + (size_t)row * p.lda + k0 + lc);
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int idx = min(t + NT * i, WCH - 1);    // chunk id: row-major (row, plane, part)
            const int row = min(n0 + idx / (NS * PPP), p.N - 1), rem = idx % (NS * PPP);
            const int plane = rem / PPP, part = rem % PPP;
            rw[i] = *reinterpret_cast<const f32x4*>(
                p.W + ((size_t)row * NS + plane) * p.K + k0 + part * 8);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            unsigned char* dst = As + buf * BM * ROWB + (lr + RSTEP * i) * ROWB + lc * 2;
            f32x4 rem = ra[i];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                bf16x4 piece;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    piece[u] = (__bf16)rem[u];
                    rem[u] -= (float)piece[u];
                }
                *reinterpret_cast<bf16x4*>(dst + s * BK * 2) = piece;
            }
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int idx = t + NT * i;
            if (idx < WCH) {
                const int row = idx / (NS * PPP), rem = idx % (NS * PPP);
                *reinterpret_cast<f32x4*>(Bs + buf * BN * ROWB + row * ROWB + (rem / PPP) * BK * 2 +
                                          (rem % PPP) * 16) = rw[i];
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload((kt + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE the MFMAs (hipcc sinks it)
        const unsigned char* a_base = As + buf * BM * ROWB + (wm * WM + l31) * ROWB + h * 16;
        const unsigned char* b_base = Bs + buf * BN * ROWB + (wn * WN + l31) * ROWB + h * 16;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[MI][NS], b[NI][NS];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    a[i][s] = *reinterpret_cast<const bf16x8*>(a_base + i * 32 * ROWB + s * BK * 2 +
                                                               ks * 32);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    b[j][s] = *reinterpret_cast<const bf16x8*>(b_base + j * 32 * ROWB + s * BK * 2 +
                                                               ks * 32);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    // smallest terms first; all cross terms with (plane_a + plane_b) < NS
#pragma unroll
                    for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
                        for (int sa = 0; sa <= sum; ++sa)
                            acc[i][j] = mfma_bf16(a[i][sa], b[j][sum - sa], acc[i][j]);
                }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * WN + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + acc_row(r, h);
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.rowvec) {
                    int idx = (row / p.rv_div) % p.rv_mod;
                    if (p.rv_flip) idx = p.rv_mod - 1 - idx;
                    v += p.rowvec[(size_t)idx * p.N + col];
                }
                v = apply_act(v, p.act);
                if (p.R) v = with_r(v, p.R[(size_t)row * p.ldr + col], p.act);
                p.C[(size_t)row * p.ldc + col] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v3: W operand bypasses LDS.  The (static) weight is stored once in MFMA-FRAGMENT ORDER
//     Wf[nb][ks][plane][lane] (16 B each): lane (c = l & 31, h = l >> 5) of block nb, k-step ks holds
//     W[nb*32 + c][ks*16 + 8h .. +7] of that plane, so a wave fetches a whole B fragment with ONE
//     fully coalesced 1 KiB global_load_dwordx4 (L2 / L1 resident: weights are a few MB) directly
//     into the registers the MFMA reads.  That removes half of the LDS write+read traffic (the
//     ds_write pipe was the limiter of the LDS-staged split kernels) and halves the LDS footprint.
//     B fragments are prefetched one whole k-tile (2 k-steps) ahead in registers.
// ------------------------------------------------------------------------------------------------
//
// SK = true (64 x 64 tiles, fp32 A, f16x3): split-K for the skinny GEMMs of small batches (M <= 300 rows x N = 512
// columns is 40 workgroups on 256 CUs, each walking K = 2048 as 64 dependent k-tiles of one memory round trip each:
// 47 us however few rows there are).  blockIdx.y owns K / ksplit; every workgroup parks its raw accumulators in the
// workspace (lane order, 16-byte stores), counts itself in, and the LAST arriver of a tile adds the ksplit records
// in slice order 0 .. ksplit-1 (its own included, re-read: the result does not depend on who arrives last) and runs
// the normal epilogue.  Hand-off per cdna_hip_programming.md Guideline 16 (drain, barrier, agent release, counter;
// agent acquire, barrier, plain loads; the records are stored write-through, recipe R1, instead of a release).
//
// BK_ = 64 (fp32 A only): a deeper k-tile for the 64 x 64 kernel of the small batches -- half the barriers and load
// waits per K; the MFMA sequence per output element is the same, so results are bit-identical across BK.
template <int NS, int BM, int BN, int WM, int WN, int NW, int MINW, bool ASPLIT, bool F16, bool SK = false,
          int BK_ = 32>
__global__ __launch_bounds__(NW * 64, MINW) void gemm_bf16_wfrag_kernel(GemmArgs p) {
    using E = Elem<F16>;
    using ET = typename E::T;
    using EV8 = typename E::V8;
    using EV4 = typename E::V4;
    constexpr int BK = BK_;
    constexpr int KSTEPS = BK / 16;                  // 16-deep MFMA k-steps per k-tile
    static_assert(BK == 32 || !ASPLIT, "pre-split A tiles are staged 32 deep");
    constexpr int ACH = BM * NS * 4;                 // 16-byte chunks of a pre-split A tile
    constexpr int RAS = (ACH + NW * 64 - 1) / (NW * 64);
    constexpr int NT = NW * 64;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == NW, "wave grid must cover the tile");
    constexpr int ROWB = NS * BK * 2 + 16;
    constexpr int AF4 = BK / 4;
    constexpr int RA = (BM * AF4) / NT;
    static_assert((BM * AF4) % NT == 0, "A tile must split evenly");
    constexpr int RSTEP = NT / AF4;

    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BM * ROWB];
    unsigned char* As = lds;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntn = (p.N + BN - 1) / BN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so
    // give every XCD a CONTIGUOUS range of tiles (row-block major): the column blocks that share an
    // A row panel then hit that XCD's L2 instead of refetching it eight times.  Bijective remap.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    // A staging rows: a ds_write_b64 is serviced in groups of 16 lanes on 32 banks (MI355X_MICROARCH.md, LDS); at BK = 32 a
    // group holds TWO rows of 8 lanes, and neighbouring rows (36 words apart = 4 mod 32) put 12 of their 16 words on the same
    // banks (PMC: 33 % of this kernel's LDS cycles were conflicts).  Rows r and r + 4 of an 8-row block are 16 banks apart:
    // lane pairs (2 m, 2 m + 1) take rows (m, m + 4).  Which thread splits which row changes, nothing else.
    const int lr0 = t / AF4, lc = (t % AF4) * 4;
    const int lr = AF4 == 8 ? ((lr0 & ~7) | ((lr0 & 7) >> 1) | ((lr0 & 1) << 2)) : lr0;
    const int KS = p.K / 16;                                // k-steps in total
    const int NBLK = p.N / 32;

    // All global addresses in the k-loop are  UNIFORM base (SGPR pair, advanced per k-tile on the
    // scalar unit) + per-lane 32-bit byte offset computed once.  Per-lane 64-bit pointers cost two
    // VGPRs + a v_lshl_add_u64 per load and pushed the kernel into scratch spills; a spilled
    // address reload inside the loop is a VMEM op whose s_waitcnt vmcnt(0) drains the whole
    // prefetch (measured: the k-loop ran 3x slower than its MFMA time).
    const int nk = SK ? (p.K / BK) / p.ksplit : p.K / BK;   // k-tiles of this workgroup (even, host-checked)
    const int kt0 = SK ? (int)blockIdx.y * nk : 0;
    const char* const a_bytes = reinterpret_cast<const char*>(p.A) + (size_t)kt0 * BK * (ASPLIT ? 2 : 4);
    const char* const w_bytes = reinterpret_cast<const char*>(p.W) + (size_t)kt0 * (KSTEPS * NS * 64 * 16);
    constexpr int NRA = ASPLIT ? RAS : RA;
    unsigned voff_a[NRA];
#pragma unroll
    for (int i = 0; i < NRA; ++i) {
        if (ASPLIT) {
            const int idc = min(t + NT * i, ACH - 1);      // (row, plane, part) row-major, clamped
            const int rowc = min(m0 + idc / (NS * 4), p.M - 1), remc = idc % (NS * 4);
            voff_a[i] = (unsigned)(((size_t)rowc * NS + (remc >> 2)) * p.K + (remc & 3) * 8) * 2u;
        } else {
            const int rowc = min(m0 + lr + RSTEP * i, p.M - 1);
            voff_a[i] = (unsigned)((size_t)rowc * p.lda + lc) * 4u;
        }
    }
    unsigned voff_b[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        int nb = (n0 + wn * WN) / 32 + j;
        nb = nb < NBLK ? nb : NBLK - 1;                    // clamped: never stored if out of range
        voff_b[j] = (unsigned)(((size_t)nb * KS * NS) * 64 + lane) * 16u;
    }

    f32x4 ra0[NRA];
    auto gload_a = [&](f32x4 (&ra)[NRA], int k0) {
        const char* base = a_bytes + (size_t)k0 * (ASPLIT ? 2 : 4);      // uniform
#pragma unroll
        for (int i = 0; i < NRA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(base + voff_a[i]);
    };
    auto lstore_a = [&](const f32x4 (&ra)[NRA], int buf) {
        if (ASPLIT) {
#pragma unroll
            for (int i = 0; i < RAS; ++i) {
                const int idx = t + NT * i;
                if (idx < ACH) {
                    const int row = idx / (NS * 4), rem = idx % (NS * 4);
                    *reinterpret_cast<f32x4*>(As + buf * BM * ROWB + row * ROWB + (rem >> 2) * BK * 2 +
                                              (rem & 3) * 16) = ra[i];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                unsigned char* dst = As + buf * BM * ROWB + (lr + RSTEP * i) * ROWB + lc * 2;
                f32x4 rem = ra[i];
                if (F16) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        rem[u] = rem[u] * E::SA;
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    EV4 piece;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        // fp16: every plane saturates instead of overflowing to inf, so |x| up to
                        // 2 x 255 degrades gracefully (hi pinned at 65504, the rest in lo)
                        piece[u] = (ET)(F16 ? __builtin_amdgcn_fmed3f(rem[u], -65504.f, 65504.f) : rem[u]);
                        rem[u] -= (float)piece[u];
                    }
                    *reinterpret_cast<EV4*>(dst + s * BK * 2) = piece;
                }
            }
        }
    };
    // B fragments of one k-tile: [ks in tile][column block][plane]
    auto gload_b = [&](EV8 (&b)[KSTEPS][NI][NS], int kt) {
        const char* base = w_bytes + (size_t)kt * (KSTEPS * NS * 64 * 16);    // uniform
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    b[ks][j][s] = *reinterpret_cast<const EV8*>(base + voff_b[j] +
                                                                   (ks * NS + s) * 64 * 16);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](const EV8 (&b)[KSTEPS][NI][NS], int buf) {
        const unsigned char* a_base = As + buf * BM * ROWB + (wm * WM + l31) * ROWB + h * 16;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            EV8 a[MI][NS];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    a[i][s] = *reinterpret_cast<const EV8*>(a_base + i * 32 * ROWB + s * BK * 2 +
                                                               ks * 32);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
                        for (int sa = 0; sa <= sum; ++sa)
                            acc[i][j] = E::mfma(a[i][sa], b[ks][j][sum - sa], acc[i][j]);
        }
    };

    EV8 b0[KSTEPS][NI][NS], b1[KSTEPS][NI][NS];
    // Branch-free software pipeline (nk even, host-checked; prefetch indices clamped so the tail
    // re-fetches tile nk-1): A one k-tile ahead (registers -> LDS), weight fragments one k-tile
    // ahead in registers, two k-tiles per iteration so the register sets have static names.
    gload_a(ra0, 0);
    gload_b(b0, 0);
    lstore_a(ra0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        const int k1 = min(kt + 1, nk - 1), k2 = min(kt + 2, nk - 1);
        if (GABL != 2) gload_a(ra0, k1 * BK);
        if (GABL != 1) gload_b(b1, k1);
        __builtin_amdgcn_sched_barrier(0);
        if (GABL != 4) compute(b0, 0);
        if (GABL != 3) lstore_a(ra0, 1);
        __syncthreads();
        if (GABL != 2) gload_a(ra0, k2 * BK);
        if (GABL != 1) gload_b(b0, k2);
        __builtin_amdgcn_sched_barrier(0);
        if (GABL != 4) compute(b1, 1);
        if (GABL != 3) lstore_a(ra0, 0);
        __syncthreads();
    }
    if (GABL == 5) return;

    if (SK && p.ksplit > 1) {
        constexpr int REC = NT * MI * NI * 16;                         // floats per record
        const int S = p.ksplit;
        float* rec = p.ws_part + ((size_t)bid * S) * REC + t * 4;
        {
            // write-through (sc1) 16-byte stores: the records leave the XCD's L2 as they are written, so no release
            // fence is needed (its L2 write-back costs ~6.5 us with 16 KB freshly dirtied per workgroup)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(p.ws_part, 0, WFRAG_WS_RECORDS * REC * 4, 0x00020000);
            const unsigned off0 = (unsigned)((((size_t)bid * S + blockIdx.y) * REC + t * 4) * sizeof(float));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc,
                                                               off0 + ((i * NI + j) * 4 + q) * (NT * 16), 0, 16);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // every storing wave drains its stores
        __syncthreads();
        unsigned* arrived = reinterpret_cast<unsigned*>(lds);          // the A images are dead
        if (t == 0) {
            if (p.sk_fence) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const unsigned old = __hip_atomic_fetch_add(p.ws_ctr + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)(S - 1)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(p.ws_ctr + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
            }
            *arrived = old;
        }
        __syncthreads();
        if (*arrived != (unsigned)(S - 1)) return;
        __syncthreads();                                               // the staging epilogue overwrites lds[0..]
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // records in groups of four (two when S = 2): all loads of a group in flight together -- one memory round
        // trip per group, not per slice -- then added in slice order
        constexpr int NV = MI * NI * 4;
        auto add_group = [&](auto gc, int s0) __attribute__((always_inline)) {
            constexpr int G = decltype(gc)::value;
            f32x4 v[G][NV];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int e = 0; e < NV; ++e)
                    v[g][e] = *reinterpret_cast<const f32x4*>(rec + (size_t)(s0 + g) * REC + e * (NT * 4));
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int e = 0; e < NV; ++e)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[e / (NI * 4)][(e / 4) % NI][4 * (e % 4) + u] += v[g][e][u];
        };
        if (S == 2) {
            add_group(std::integral_constant<int, 2>{}, 0);
        } else {
            for (int s0 = 0; s0 < S; s0 += 4) add_group(std::integral_constant<int, 4>{}, s0);
        }
    }

    // ---- fast epilogue (no row-vector, fp32 output): stage each 32-row block of the wave's tile in
    // LDS (the A images are dead after the loop's last barrier) and write it back as dwordx4 rows:
    // 8 stores per lane per block, each wave instruction covering 4 rows x WN*4 contiguous bytes,
    // instead of 32 four-byte stores in 128-byte segments; the residual is read the same way.
    constexpr int SS = WN + 4;
    constexpr bool STAGE_FITS = (NW * 32 * SS * 4) <= (2 * BM * ROWB);
    if (STAGE_FITS && !p.rowvec) {
        float* stage = reinterpret_cast<float*>(lds) + wave * (32 * SS);
        constexpr int F4R = WN / 4;                                   // float4 per staged row
        float bvj[NI];                                                // one load per column block (not per row block)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * WN + j * 32 + l31;
            bvj[j] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
        }
        dispatch_act(p.act, [&](auto AC) {
            constexpr int ACT = decltype(AC)::value;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stage[acc_row(r, h) * SS + j * 32 + l31] =
                            apply_act_c<ACT>(acc[i][j][r] * (1.f / (E::SA * E::SW)) + bvj[j]);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < (32 * F4R) / 64; ++it) {
                    const int idx = lane + 64 * it;
                    const int rr = idx / F4R, c4 = (idx % F4R) * 4;
                    const int row = m0 + wm * WM + i * 32 + rr, col = n0 + wn * WN + c4;
                    if (row < p.M && col < p.N) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(stage + rr * SS + c4);
                        if (p.R) v = with_r4_c<ACT>(v, *reinterpret_cast<const f32x4*>(p.R + (size_t)row * p.ldr + col));
                        if (p.c_split)      // operand planes for the next split GEMM: (M, NS, N)
                            tocvp_store_planes4(p.C, (size_t)row * NS * p.N + col, (size_t)p.N, v, F16 ? 22 : NS);
                        else
                            *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + col) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        });
        return;
    }

#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * WN + j * 32 + l31;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + acc_row(r, h);
                if (row >= p.M) continue;
                float v = acc[i][j][r] * (1.f / (E::SA * E::SW)) + bv;
                if (p.rowvec) {
                    int idx = (row / p.rv_div) % p.rv_mod;
                    if (p.rv_flip) idx = p.rv_mod - 1 - idx;
                    v += p.rowvec[(size_t)idx * p.N + col];
                }
                v = apply_act(v, p.act);
                if (p.R) v = with_r(v, p.R[(size_t)row * p.ldr + col], p.act);
                if (p.c_split) {
                    if (F16) v = __builtin_amdgcn_fmed3f(v * E::SA, -65504.f, 65504.f);
                    ET* cs = reinterpret_cast<ET*>(p.C) + (size_t)row * NS * p.N + col;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const ET piece = (ET)v;
                        cs[(size_t)s * p.N] = piece;
                        v -= (float)piece;
                    }
                } else {
                    p.C[(size_t)row * p.ldc + col] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v4 ("planes"): f16x3 GEMM whose A operand arrives as fp16 operand planes (M, 2, K) written by its
//     producer (LayerNorm / attention / GEMM epilogue).  Measured on v3 (scripts/probes/gemm_ablate.hip,
//     38400 x 2048 x 512): MFMA-only 142 us, everything-else-only 166 us, together 377 us -- the
//     64 B/clk vector-memory path of the CU carries A (16 KB) + B (4 waves x 8 KB) per 128x128x32
//     tile and is as busy as the matrix cores.  Here
//       * a wave owns 128 x 64 of a 256 x 128 tile (4 x 2 accumulator tiles): a B fragment from
//         L1/L2 feeds four row blocks -> 2/3 of the bytes per MFMA;
//       * A planes go global -> LDS by DMA (global_load_lds, 16 B per lane): no staging registers
//         (they pay for the 128 accumulator VGPRs) and no conversion instructions in the k-loop;
//         the LDS image is lane-linear (8 lanes = one 128-byte row [plane 0 k0..31 | plane 1]), the
//         bank-conflict-free order is obtained by permuting the 16-byte chunks on the SOURCE side:
//         physical chunk c of row r holds logical chunk c ^ (r & 7);
//       * two LDS stages, one barrier per k-tile, 2 workgroups / CU.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_f16_planes_kernel(GemmArgs p) {
    constexpr int BM = 256, BN = 128, BK = 32, MI = 4, NI = 2, NS = 2;
    constexpr int STAGE = BM * 128;                                  // bytes per A stage
    using E = Elem<true>;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = p.N / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
    const int KS = p.K / 16, nk = p.K / BK;

    // ---- A: 8 DMA instructions per wave and k-tile, each 8 rows x 128 B = 1 KiB of LDS
    const char* const a_bytes = reinterpret_cast<const char*>(p.A);
    unsigned voff_a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave * 8 + i) * 8 + (lane >> 3);            // row of the tile
        const int j = (lane & 7) ^ (row & 7);                        // logical chunk landing here
        const int grow = min(m0 + row, p.M - 1);
        voff_a[i] = (unsigned)((((size_t)grow * NS + (j >> 2)) * p.K + (j & 3) * 8) * 2);
    }
    auto dma_a = [&](int stage, int kt) {
        const char* base = a_bytes + (size_t)kt * (BK * 2);          // uniform
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(base + voff_a[i]),
                (__attribute__((address_space(3))) void*)(lds + stage * STAGE + (wave * 8 + i) * 1024),
                16, 0, 0);
    };

    // ---- B fragments straight from L1/L2 (fragment order), one k-tile ahead in registers
    const char* const w_bytes = reinterpret_cast<const char*>(p.W);
    unsigned voff_b[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j)
        voff_b[j] = (unsigned)(((size_t)((n0 + wn * 64) / 32 + j) * KS * NS) * 64 + lane) * 16u;
    auto gload_b = [&](f16x8 (&b)[2][NI][NS], int kt) {
        const char* base = w_bytes + (size_t)kt * (2 * NS * 64 * 16);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    b[ks][j][s] = *reinterpret_cast<const f16x8*>(base + voff_b[j] + (ks * NS + s) * 64 * 16);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int x7 = (l31 & 7) << 4;                                   // lane's chunk permutation
    auto compute = [&](const f16x8 (&b)[2][NI][NS], int stage) {
        const unsigned char* a_base = lds + stage * STAGE + (wm * 128 + l31) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 a[MI][NS];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    a[i][s] = *reinterpret_cast<const f16x8*>(a_base + i * 32 * 128 +
                                                              (((s * 4 + ks * 2 + h) << 4) ^ x7));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    acc[i][j] = E::mfma(a[i][0], b[ks][j][1], acc[i][j]);
                    acc[i][j] = E::mfma(a[i][1], b[ks][j][0], acc[i][j]);
                    acc[i][j] = E::mfma(a[i][0], b[ks][j][0], acc[i][j]);
                }
        }
    };

    f16x8 b0[2][NI][NS], b1[2][NI][NS];
    dma_a(0, 0);
    gload_b(b0, 0);
    __builtin_amdgcn_s_waitcnt(0);                                   // vmcnt(0) lgkmcnt(0) expcnt(0)
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {                             // nk even (host-checked)
        const int k1 = min(kt + 1, nk - 1), k2 = min(kt + 2, nk - 1);
        if (GABL != 2) dma_a(1, k1);
        if (GABL != 1) gload_b(b1, k1);
        __builtin_amdgcn_sched_barrier(0);
        if (GABL != 4) compute(b0, 0);
        __syncthreads();                                             // DMA of stage 1 landed, stage 0 free
        if (GABL != 2) dma_a(0, k2);
        if (GABL != 1) gload_b(b0, k2);
        __builtin_amdgcn_sched_barrier(0);
        if (GABL != 4) compute(b1, 1);
        __syncthreads();
    }

    // ---- epilogue: 32-row blocks staged through LDS, written back as dwordx4 rows (see v3)
    constexpr int SS = 64 + 4;
    float* stage_f = reinterpret_cast<float*>(lds) + wave * (32 * SS);
    constexpr int F4R = 64 / 4;
    float bvj[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) bvj[j] = p.bias ? p.bias[n0 + wn * 64 + j * 32 + l31] : 0.f;
    dispatch_act(p.act, [&](auto AC) {
        constexpr int ACT = decltype(AC)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage_f[acc_row(r, h) * SS + j * 32 + l31] =
                        apply_act_c<ACT>(acc[i][j][r] * (1.f / (E::SA * E::SW)) + bvj[j]);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < (32 * F4R) / 64; ++it) {
                const int idx = lane + 64 * it;
                const int rr = idx / F4R, c4 = (idx % F4R) * 4;
                const int row = m0 + wm * 128 + i * 32 + rr, col = n0 + wn * 64 + c4;
                if (row < p.M) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(stage_f + rr * SS + c4);
                    if (p.R) v = with_r4_c<ACT>(v, *reinterpret_cast<const f32x4*>(p.R + (size_t)row * p.ldr + col));
                    if (p.c_split)
                        tocvp_store_planes4(p.C, (size_t)row * NS * p.N + col, (size_t)p.N, v, 22);
                    else
                        *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + col) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    });
}

// W (N, K) fp32 -> fragment-order bf16 planes Wf[nb][ks][plane][lane][8]
template <bool F16>
__global__ __launch_bounds__(256) void split_weights_frag_kernel(const float* __restrict__ w,
                                                                 typename Elem<F16>::T* __restrict__ out,
                                                                 long n, int K, int NS) {
    using ET = typename Elem<F16>::T;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long row = i / K;
    const int k = (int)(i - row * K);
    const int nb = (int)(row >> 5), c = (int)(row & 31);
    const int ks = k >> 4, hh = (k >> 3) & 1, j = k & 7;
    const int KS = K / 16;
    float rem = w[i] * Elem<F16>::SW;
    if (F16) rem = fminf(fmaxf(rem, -65504.f), 65504.f);
    for (int s = 0; s < NS; ++s) {
        const ET piece = (ET)rem;
        out[((((size_t)nb * KS + ks) * NS + s) * 64 + (hh * 32 + c)) * 8 + j] = piece;
        rem -= (float)piece;
    }
}

// W (N, K) fp32 -> (N, NS, K) bf16 planes: plane s holds bf16 of the residual after planes < s
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w,
                                                            __bf16* __restrict__ out, long n, int K,
                                                            int NS) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long row = i / K;
    const int k = (int)(i - row * K);
    float rem = w[i];
    for (int s = 0; s < NS; ++s) {
        const __bf16 piece = (__bf16)rem;
        out[(row * NS + s) * K + k] = piece;
        rem -= (float)piece;
    }
}

template <int NS, int BM, int BN, int WM, int WN, int BK, int NW, int MINW>
int launch(const GemmArgs& p, hipStream_t s) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_bf16_split_kernel<NS, BM, BN, WM, WN, BK, NW, MINW>), dim3(ntm * ntn),
                       dim3(NW * 64), 0, s, p);
    return tocvp_launch_status();
}

template <int NS, int BM, int BN, int WM, int WN, int NW, int MINW, bool F16 = false>
int launch_wfrag(const GemmArgs& p, hipStream_t s) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    if (p.a_split)
        hipLaunchKernelGGL((gemm_bf16_wfrag_kernel<NS, BM, BN, WM, WN, NW, MINW, true, F16>),
                           dim3(ntm * ntn), dim3(NW * 64), 0, s, p);
    else
        hipLaunchKernelGGL((gemm_bf16_wfrag_kernel<NS, BM, BN, WM, WN, NW, MINW, false, F16>),
                           dim3(ntm * ntn), dim3(NW * 64), 0, s, p);
    return tocvp_launch_status();
}

// split-K plan of the 64 x 64 kernel: slices double while the grid stays within ~2 workgroups per CU, every slice
// keeps >= 4 k-tiles (an even count: the loop takes two per iteration) and the records fit the workspace
static int ksplit_wgs() { return 256; }                               // (tuned on one MI355X: 512 loses, scripts/ksplit_bench.py)
static int pick_ksplit(const GemmArgs& p, int bk) {
    const int tiles = ((p.M + 63) / 64) * ((p.N + 63) / 64), nk = p.K / bk, least = bk == 32 ? 4 : 2;
    int S = 1;
    while (S < 16 && tiles * S * 2 <= ksplit_wgs() && tiles * S * 2 <= WFRAG_WS_RECORDS && nk % (S * 4) == 0 &&
           nk / (S * 2) >= least)
        S *= 2;                                                        // 1, 2, 4, 8, 16
    return tiles <= 1024 ? S : 1;
}
// k-tile depth of the 64 x 64 kernel: 64 where it leaves an even number of k-tiles (the loop takes two per iteration),
// else 32; TOCVP_GEMM_SMALL_BK=32 pins the shallow form.  Measured (scripts/ksplit_bench.py, graph replay, us):
// 300x512x2048 split-K 15.1 / 12.7 / 13.8 at BK 32 / 64 / 128, 2400x512x2048 41.4 / 36.4 / 37.7, K = 512 shapes equal at
// 32 and 64 and 10-15 % slower at 128 (216 VGPRs, 66 KB LDS) -- the 128-deep form is not built.
static int pick_small_bk(int K) { return K % 128 == 0 ? 64 : 32; }

template <int BK>
int launch_small_f16(const GemmArgs& p, hipStream_t s) {
    GemmArgs q = p;
    q.ksplit = p.ws_part ? pick_ksplit(p, BK) : 1;
    q.sk_fence = 0;                                                    // (diagnostic release fence of round 3: off)
    const int ntm = (p.M + 63) / 64, ntn = (p.N + 63) / 64;
    hipLaunchKernelGGL((gemm_bf16_wfrag_kernel<2, 64, 64, 32, 32, 4, 1, false, true, true, BK>),
                       dim3(ntm * ntn, q.ksplit), dim3(256), 0, s, q);
    return tocvp_launch_status();
}

// 128 x 128 tiles only when there are enough of them: below this count the 64 x 64 kernel fills the CUs better
static long small_below() { return 192L; }

int dispatch_wfrag_f16(const GemmArgs& p, hipStream_t s) {
    const long big_tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (p.a_split && !p.rowvec && (p.N % 128) == 0 && (p.K % 64) == 0 && big_tiles >= 512 &&
        (size_t)p.M * 2 * p.K * 2 < 0xffffffffull) {
        const int ntm = (p.M + 255) / 256, ntn = p.N / 128;
        hipLaunchKernelGGL(gemm_f16_planes_kernel, dim3(ntm * ntn), dim3(256), 0, s, p);
        return tocvp_launch_status();
    }
    if (big_tiles < small_below()) {
        if (!p.a_split) {
            return pick_small_bk(p.K) == 64 ? launch_small_f16<64>(p, s) : launch_small_f16<32>(p, s);
        }
        return launch_wfrag<2, 64, 64, 32, 32, 4, 1, true>(p, s);
    }
    // (256 x 128 / 8-wave and 128 x 128 / 8-wave forms were instantiated in rounds 1-4 behind TOCVP_GEMM_VARIANT and never won: gone)
    // (64-deep k-tiles in THIS kernel need 2 x 64 fragment registers: 256 VGPRs + 50 spilled, not built)
    // Also measured at the end of round 3 on 38400 x 2048 x 512 / 38400 x 512 x 2048 (scripts/probes/gemm_ablate.hip,
    // two rounds) and dropped: the four waves side by side along N (128 x 32 per wave: half the weight-fragment bytes on
    // the vector-memory path, twice the LDS fragment reads) 391.8 / 348.5 -> 395.6 / 346.4 us; the split of the next A
    // tile woven between the MFMAs of the current one with sched_group_barrier 380.6 / 341.2 -> 381.3-388.2 / 344.5-351.7.
    return launch_wfrag<2, 128, 128, 64, 64, 4, 2, true>(p, s);
}

template <int NS>
int dispatch_wfrag(const GemmArgs& p, hipStream_t s) {
    const long big_tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (big_tiles < small_below()) return launch_wfrag<NS, 64, 64, 32, 32, 4, 1>(p, s);
    return launch_wfrag<NS, 128, 128, 64, 64, 4, 2>(p, s);
}

template <int NS>
int dispatch(const GemmArgs& p, hipStream_t s) {
    const long big_tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (big_tiles < 192) return launch<NS, 64, 64, 32, 32, 32, 4, 1>(p, s);
    return launch<NS, 128, 128, 64, 64, 32, 4, 1>(p, s);
}

}  // namespace

extern "C" int tocvp_split_weights_bf16(const float* w, void* out, int N, int K, int nsplit,
                                        void* stream) {
    TOCVP_CHECK_ARG(w && out && N > 0 && K > 0 && (nsplit == 2 || nsplit == 3));
    const long n = (long)N * K;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, static_cast<__bf16*>(out), n, K, nsplit);
    return tocvp_launch_status();
}

extern "C" int tocvp_gemm_bf16split_f32(const float* A, int lda, const void* Wsplit, int nsplit,
                                        const float* bias, const float* R, int ldr,
                                        const float* rowvec, int rv_div, int rv_mod, int rv_flip,
                                        float* C, int ldc, int M, int N, int K, int act,
                                        void* stream) {
    TOCVP_CHECK_ARG(A && Wsplit && C);
    TOCVP_CHECK_ARG(nsplit == 2 || nsplit == 3);
    TOCVP_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % 32) == 0);
    TOCVP_CHECK_ARG(lda >= K && ldc >= N);
    // A is addressed with 32-bit byte offsets: at most 2^32 bytes of activations per call (the caller cuts rows)
    TOCVP_CHECK_ARG((size_t)M * (size_t)lda * sizeof(float) < 0x100000000ull);
    TOCVP_CHECK_ARG(R == nullptr || ldr >= N);
    TOCVP_CHECK_ARG(rowvec == nullptr || (rv_div > 0 && rv_mod > 0));
    TOCVP_CHECK_ARG(act >= TOCVP_ACT_NONE && act <= TOCVP_ACT_GELU);
    if ((lda & 3) || !tocvp_aligned16(A) || !tocvp_aligned16(Wsplit)) return TOCVP_EALIGN;
    if (M == 0) return TOCVP_OK;
    GemmArgs p{A, lda, static_cast<const __bf16*>(Wsplit), bias, R, ldr, rowvec, rv_div, rv_mod,
               rv_flip, C, ldc, M, N, K, act, 0, 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    return nsplit == 2 ? dispatch<2>(p, s) : dispatch<3>(p, s);
}

extern "C" int tocvp_split_weights_frag_bf16(const float* w, void* out, int N, int K, int nsplit,
                                             void* stream) {
    TOCVP_CHECK_ARG(w && out && N > 0 && K > 0 && (nsplit == 2 || nsplit == 3));
    TOCVP_CHECK_ARG((N % 32) == 0 && (K % 32) == 0);
    const long n = (long)N * K;
    hipLaunchKernelGGL(split_weights_frag_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256),
                       0, static_cast<hipStream_t>(stream), w, static_cast<__bf16*>(out), n, K, nsplit);
    return tocvp_launch_status();
}

extern "C" int tocvp_gemm_bf16wfrag_f32(const void* A, int a_split, int lda, const void* Wfrag,
                                        int nsplit, const float* bias, const float* R, int ldr,
                                        const float* rowvec, int rv_div, int rv_mod, int rv_flip,
                                        void* C, int c_split, int ldc, int M, int N, int K, int act,
                                        void* stream) {
    TOCVP_CHECK_ARG(A && Wfrag && C);
    TOCVP_CHECK_ARG(nsplit == 2 || nsplit == 3 || nsplit == 22);
    TOCVP_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % 64) == 0 && (N % 32) == 0);
    TOCVP_CHECK_ARG(a_split || lda >= K);
    // operand planes are addressed with 32-bit byte offsets: at most 2^32 bytes of planes per call (the caller cuts rows)
    TOCVP_CHECK_ARG(!a_split || (size_t)M * (nsplit == 22 ? 2 : nsplit) * K * 2 < 0x100000000ull);
    TOCVP_CHECK_ARG(a_split || (size_t)M * (size_t)lda * sizeof(float) < 0x100000000ull);     // the same for fp32 rows
    TOCVP_CHECK_ARG(c_split || ldc >= N);
    TOCVP_CHECK_ARG(R == nullptr || (ldr >= N && (ldr & 3) == 0 && tocvp_aligned16(R)));
    TOCVP_CHECK_ARG(rowvec == nullptr || (rv_div > 0 && rv_mod > 0));
    TOCVP_CHECK_ARG((act >= TOCVP_ACT_NONE && act <= TOCVP_ACT_GELU) || (act == TOCVP_ACT_GATE && R != nullptr));
    TOCVP_CHECK_ARG(c_split || ((ldc & 3) == 0 && tocvp_aligned16(C)));
    if ((!a_split && (lda & 3)) || !tocvp_aligned16(A) || !tocvp_aligned16(Wfrag)) return TOCVP_EALIGN;
    if (M == 0) return TOCVP_OK;
    GemmArgs p{static_cast<const float*>(A), lda, static_cast<const __bf16*>(Wfrag), bias, R, ldr,
               rowvec, rv_div, rv_mod, rv_flip, static_cast<float*>(C), ldc, M, N, K, act,
               a_split ? 1 : 0, c_split ? 1 : 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nsplit == 22) return dispatch_wfrag_f16(p, s);         // fp16 planes ("f16x3")
    return nsplit == 2 ? dispatch_wfrag<2>(p, s) : dispatch_wfrag<3>(p, s);
}

extern "C" int tocvp_split_weights_frag_f16(const float* w, void* out, int N, int K, void* stream) {
    TOCVP_CHECK_ARG(w && out && N > 0 && K > 0);
    TOCVP_CHECK_ARG((N % 32) == 0 && (K % 32) == 0);
    const long n = (long)N * K;
    hipLaunchKernelGGL(split_weights_frag_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, static_cast<_Float16*>(out), n, K, 2);
    return tocvp_launch_status();
}

extern "C" size_t tocvp_gemm_wfrag_ws_bytes(void) {
    return WFRAG_WS_CTR_BYTES + (size_t)WFRAG_WS_RECORDS * 256 * 16 * sizeof(float);
}

extern "C" int tocvp_gemm_f16wfrag_ws_f32(const void* A, int lda, const void* Wfrag, const float* bias,
                                          const float* R, int ldr, const float* rowvec, int rv_div,
                                          int rv_mod, int rv_flip, void* C, int c_split, int ldc, int M, int N,
                                          int K, int act, void* ws, size_t ws_bytes, void* stream) {
    TOCVP_CHECK_ARG(A && Wfrag && C);
    TOCVP_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % 64) == 0 && (N % 32) == 0 && lda >= K);
    TOCVP_CHECK_ARG((size_t)M * (size_t)lda * sizeof(float) < 0x100000000ull);               // 32-bit byte offsets into A
    TOCVP_CHECK_ARG(c_split || ldc >= N);
    TOCVP_CHECK_ARG(R == nullptr || (ldr >= N && (ldr & 3) == 0 && tocvp_aligned16(R)));
    TOCVP_CHECK_ARG(rowvec == nullptr || (rv_div > 0 && rv_mod > 0));
    TOCVP_CHECK_ARG((act >= TOCVP_ACT_NONE && act <= TOCVP_ACT_GELU) || (act == TOCVP_ACT_GATE && R != nullptr));
    TOCVP_CHECK_ARG(c_split || ((ldc & 3) == 0 && tocvp_aligned16(C)));
    TOCVP_CHECK_ARG(ws == nullptr || (ws_bytes >= tocvp_gemm_wfrag_ws_bytes() && tocvp_aligned16(ws)));
    if ((lda & 3) || !tocvp_aligned16(A) || !tocvp_aligned16(Wfrag)) return TOCVP_EALIGN;
    if (M == 0) return TOCVP_OK;
    GemmArgs p{static_cast<const float*>(A), lda, static_cast<const __bf16*>(Wfrag), bias, R, ldr,
               rowvec, rv_div, rv_mod, rv_flip, static_cast<float*>(C), ldc, M, N, K, act, 0, c_split ? 1 : 0,
               1, nullptr, nullptr};
    if (ws) {
        p.ws_ctr = static_cast<unsigned*>(ws);
        p.ws_part = reinterpret_cast<float*>(static_cast<char*>(ws) + WFRAG_WS_CTR_BYTES);
    }
    return dispatch_wfrag_f16(p, static_cast<hipStream_t>(stream));
}

extern "C" int tocvp_gemm_f16wfrag_f32(const void* A, int lda, const void* Wfrag, const float* bias,
                                       const float* R, int ldr, const float* rowvec, int rv_div,
                                       int rv_mod, int rv_flip, void* C, int ldc, int M, int N, int K,
                                       int act, void* stream) {
    return tocvp_gemm_bf16wfrag_f32(A, 0, lda, Wfrag, 22, bias, R, ldr, rowvec, rv_div, rv_mod, rv_flip,
                                    C, 0, ldc, M, N, K, act, stream);
}
