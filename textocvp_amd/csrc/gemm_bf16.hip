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
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* A; int lda;
    const __bf16* W;             // (N, NS, K)
    const float* bias;
    const float* R; int ldr;
    const float* rowvec; int rv_div, rv_mod, rv_flip;
    float* C; int ldc;
    int M, N, K, act;
};

constexpr int BK = 32;

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == TOCVP_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == TOCVP_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int NS, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_bf16_split_kernel(GemmArgs p) {
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    constexpr int ROWB = NS * BK * 2 + 16;           // bytes per LDS row
    constexpr int RA = BM / 32;                      // float4 A loads per thread per tile
    constexpr int WCH = BN * NS * 4;                 // 16-byte W chunks per tile
    constexpr int RW = (WCH + 255) / 256;

    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (BM + BN) * ROWB];
    unsigned char* As = lds;
    unsigned char* Bs = lds + 2 * BM * ROWB;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ntn = (p.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const int lr = t >> 3, lc = (t & 7) * 4;

    f32x4 ra[RA], rw[RW];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int row = m0 + lr + 32 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < p.M) v = *reinterpret_cast<const f32x4*>(p.A + (size_t)row * p.lda + k0 + lc);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            const int idx = t + 256 * i;                 // chunk id: row-major (row, plane, part)
            const int row = idx / (NS * 4), rem = idx % (NS * 4);
            const int plane = rem >> 2, part = rem & 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < WCH && n0 + row < p.N)
                v = *reinterpret_cast<const f32x4*>(
                    p.W + ((size_t)(n0 + row) * NS + plane) * p.K + k0 + part * 8);
            rw[i] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            unsigned char* dst = As + buf * BM * ROWB + (lr + 32 * i) * ROWB + lc * 2;
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
            const int idx = t + 256 * i;
            if (idx < WCH) {
                const int row = idx / (NS * 4), rem = idx % (NS * 4);
                *reinterpret_cast<f32x4*>(Bs + buf * BN * ROWB + row * ROWB + (rem >> 2) * BK * 2 +
                                          (rem & 3) * 16) = rw[i];
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
                if (p.R) v += p.R[(size_t)row * p.ldr + col];
                p.C[(size_t)row * p.ldc + col] = v;
            }
        }
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

template <int NS, int BM, int BN, int WM, int WN>
int launch(const GemmArgs& p, hipStream_t s) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_bf16_split_kernel<NS, BM, BN, WM, WN>), dim3(ntm * ntn), dim3(256), 0, s,
                       p);
    return tocvp_launch_status();
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
    TOCVP_CHECK_ARG(M >= 0 && N > 0 && K > 0 && (K % BK) == 0);
    TOCVP_CHECK_ARG(lda >= K && ldc >= N);
    TOCVP_CHECK_ARG(R == nullptr || ldr >= N);
    TOCVP_CHECK_ARG(rowvec == nullptr || (rv_div > 0 && rv_mod > 0));
    TOCVP_CHECK_ARG(act >= TOCVP_ACT_NONE && act <= TOCVP_ACT_GELU);
    if ((lda & 3) || !tocvp_aligned16(A) || !tocvp_aligned16(Wsplit)) return TOCVP_EALIGN;
    if (M == 0) return TOCVP_OK;
    GemmArgs p{A, lda, static_cast<const __bf16*>(Wsplit), bias, R, ldr, rowvec, rv_div, rv_mod,
               rv_flip, C, ldc, M, N, K, act};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long big_tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (nsplit == 2) {
        if (big_tiles >= 192) return launch<2, 128, 128, 64, 64>(p, s);
        return launch<2, 64, 64, 32, 32>(p, s);
    }
    if (big_tiles >= 192) return launch<3, 128, 128, 64, 64>(p, s);
    return launch<3, 64, 64, 32, 32>(p, s);
}
