// fp32 MFMA GEMM with fused epilogue:  C = act(A W^T + bias + rowvec) + R
//
// gfx950 design notes
//  * v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain (needed for the 1e-4 parity bar) at the
//    fp32 matrix peak (157 TFLOP/s).  Each MFMA consumes ONE fp32 of A and B per lane and takes
//    64 cycles, so operand bandwidth is tiny: LDS tiles are read with ds_read_b128, one 16-byte
//    read feeds four MFMAs.  The k index inside a group of 8 is permuted (lane half h owns
//    k = 4h..4h+3) identically for A and B, which leaves the dot product unchanged.
//  * LDS rows are padded to 36 floats: 36*r mod 64 visits 16 distinct 16-byte slots for the 16
//    rows of every ds_read_b128 lane group -> conflict-free.
//  * global -> register -> LDS double buffering, one barrier per 32-deep k-tile.
#include "common.h"

namespace {

struct GemmArgs {
    const float* A; int lda;
    const float* W;
    const float* bias;
    const float* R; int ldr;
    const float* rowvec; int rv_div, rv_mod, rv_flip;
    float* C; int ldc;
    int M, N, K, act;
};

constexpr int BK = 32;
constexpr int LS = BK + 4;  // padded LDS row stride (floats)

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == TOCVP_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == TOCVP_ACT_GELU) return tocvp_gelu(v);
    return v;
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    constexpr int RA = BM / 32, RB = BN / 32;  // float4 loads per thread per tile

    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LS];
    float* As = lds;
    float* Bs = lds + 2 * BM * LS;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int ntn = (p.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / ntn) * BM;
    const int n0 = (blockIdx.x % ntn) * BN;

    const int lr = t >> 3;        // 0..31
    const int lc = (t & 7) * 4;   // 0,4,..,28

    f32x4 ra[RA], rb[RB];
    float kmask = 1.f;            // zeroes the k-tail of the tile fetched last (applied in lstore)
    auto gload = [&](int k0) {
        // UNCONDITIONAL loads from clamped (always valid) addresses: rows >= M / N produce values
        // that are never stored; the k-tail (K % 32 != 0) is zeroed by a multiply.  A predicated
        // load sits behind a branch, which makes hipcc's s_waitcnt insertion under-count the loads
        // in flight and stall the MFMAs on the prefetch that was just issued.
        const int kc = min(k0 + lc, p.K - 4);
        kmask = (k0 + lc < p.K) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int row = min(m0 + lr + 32 * i, p.M - 1);
            ra[i] = *reinterpret_cast<const f32x4*>(p.A + (size_t)row * p.lda + kc);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = min(n0 + lr + 32 * i, p.N - 1);
            rb[i] = *reinterpret_cast<const f32x4*>(p.W + (size_t)row * p.K + kc);
        }
    };
    const bool ktail = (p.K % BK) != 0;   // kernel-uniform
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i)
            *reinterpret_cast<f32x4*>(As + buf * BM * LS + (lr + 32 * i) * LS + lc) =
                ktail ? ra[i] * kmask : ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i)
            *reinterpret_cast<f32x4*>(Bs + buf * BN * LS + (lr + 32 * i) * LS + lc) =
                ktail ? rb[i] * kmask : rb[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload((kt + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE the MFMAs (hipcc sinks it)
        const float* a_base = As + buf * BM * LS + (wm * WM + l31) * LS + 4 * h;
        const float* b_base = Bs + buf * BN * LS + (wn * WN + l31) * LS + 4 * h;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LS + kk * 8);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                b[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * LS + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma32(a[i][s], b[j][s], acc[i][j]);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // fast epilogue (no row-vector): 32-row blocks of the wave tile staged in LDS (tiles are dead
    // after the last barrier) and written / residual-read as dwordx4 rows (see gemm_bf16.hip)
    constexpr int SS = WN + 4;
    if (!p.rowvec && (p.ldc & 3) == 0 && (!p.R || (p.ldr & 3) == 0)) {
        float* stage = lds + wave * (32 * SS);
        constexpr int F4R = WN / 4;
        float bvj[NI];                                                // one load per column block
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * WN + j * 32 + l31;
            bvj[j] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
        }
        // (the activation switch hoisted out of the unrolled loops: inside them every staged element carried a ladder of
        // scalar branches and the inlined erff of the GELU it did not take)
        const int act_sel = p.act == TOCVP_ACT_RELU ? 1 : (p.act == TOCVP_ACT_GELU ? 2 : 0);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (act_sel == 1) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) stage[acc_row(r, h) * SS + j * 32 + l31] = fmaxf(acc[i][j][r] + bvj[j], 0.0f);
            } else if (act_sel == 2) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stage[acc_row(r, h) * SS + j * 32 + l31] = apply_act(acc[i][j][r] + bvj[j], TOCVP_ACT_GELU);
            } else {
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) stage[acc_row(r, h) * SS + j * 32 + l31] = acc[i][j][r] + bvj[j];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < (32 * F4R) / 64; ++it) {
                const int idx = lane + 64 * it;
                const int rr = idx / F4R, c4 = (idx % F4R) * 4;
                const int row = m0 + wm * WM + i * 32 + rr, col = n0 + wn * WN + c4;
                if (row < p.M && col + 3 < p.N) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(stage + rr * SS + c4);
                    if (p.R) v += *reinterpret_cast<const f32x4*>(p.R + (size_t)row * p.ldr + col);
                    *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + col) = v;
                } else if (row < p.M) {
                    for (int u = 0; u < 4; ++u)
                        if (col + u < p.N) {
                            float v = stage[rr * SS + c4 + u];
                            if (p.R) v += p.R[(size_t)row * p.ldr + col + u];
                            p.C[(size_t)row * p.ldc + col + u] = v;
                        }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }

    // epilogue: lane owns column (l31) of each 32x32 block, 16 rows per block
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
                if (p.R) v += p.R[(size_t)row * p.ldr + col];
                p.C[(size_t)row * p.ldc + col] = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch(const GemmArgs& p, hipStream_t s) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN>), dim3(ntm * ntn), dim3(256), 0, s, p);
    return tocvp_launch_status();
}

}  // namespace

extern "C" int tocvp_gemm_f32(const float* A, int lda, const float* W, const float* bias,
                              const float* R, int ldr, const float* rowvec, int rv_div, int rv_mod,
                              int rv_flip, float* C, int ldc, int M, int N, int K, int act,
                              void* stream) {
    TOCVP_CHECK_ARG(A && W && C);
    TOCVP_CHECK_ARG(M >= 0 && N > 0 && K > 0);
    TOCVP_CHECK_ARG(lda >= K && ldc >= N);
    TOCVP_CHECK_ARG(R == nullptr || ldr >= N);
    TOCVP_CHECK_ARG(rowvec == nullptr || (rv_div > 0 && rv_mod > 0));
    TOCVP_CHECK_ARG(act >= TOCVP_ACT_NONE && act <= TOCVP_ACT_GELU);
    if ((K & 3) || (lda & 3) || !tocvp_aligned16(A) || !tocvp_aligned16(W)) return TOCVP_EALIGN;
    if (M == 0) return TOCVP_OK;
    GemmArgs p{A, lda, W, bias, R, ldr, rowvec, rv_div, rv_mod, rv_flip, C, ldc, M, N, K, act};
    hipStream_t s = static_cast<hipStream_t>(stream);
    // big tiles once the grid fills the 256 CUs; small tiles keep more CUs busy on short M
    const long big_tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (big_tiles >= 192) return launch<128, 128, 64, 64>(p, s);
    return launch<64, 64, 32, 32>(p, s);
}
