// Generic strided / batched fp32 GEMM on the exact fp32 MFMA, for the BACKWARD pass of the predictor
// training step: weight and data gradients of nn.Linear
// (dW = dY^T X, dX = dY W), and the per-head products of attention backward
// (S = Q K^T, dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q).
//
//   C[b1,b2] (M x N) = alpha * op(A[b1,b2]) (M x K) * op(B[b1,b2]) (K x N)  (+ C[b1,b2] if accumulate)
//   op(X) = X or X^T; every operand has a leading dimension and two batch strides (elements), so the
//   (batch, head) slices of a (B, T, H*dh) tensor are addressed in place.
//
// 64 x 64 tile per 4-wave workgroup (32 x 32 per wave), 64-deep (16 for short reductions) k-stages staged
// through LDS as [row][k] images for BOTH operands (rows padded by 4 words: conflict-free ds_read_b128), the
// next stage's global loads in flight during the MFMAs, any M / N / K (zero fill on load, masked stores).
// Loads are 16 bytes per lane along the operand's contiguous axis when the addresses allow it, element-wise
// otherwise; built for generality, not for the roofline.
#include "common.h"

namespace {

struct BmmArgs {
    const float* A; const float* B; float* C;
    long sA1, sA2, sB1, sB2, sC1, sC2;
    int lda, ldb, ldc;
    int transA, transB, nb2;
    int M, N, K;
    float alpha; int accumulate;
    int vec;                 // every base pointer and batch stride is a multiple of 16 bytes
};

constexpr int BT = 64;

// KT = k-depth of a stage (16: short reductions, 64: everything with K >= 64).  The global loads of stage s + 1
// are issued into registers before the MFMAs of stage s (one stage of prefetch), so a workgroup pays the
// memory latency once, not once per stage.
// MODE -1: operand orientations and load widths decided at run time (any alignment); MODE 0..3: 16-byte loads
// with the orientations fixed at compile time (bit 0: A row-major along k, bit 1: B^T row-major along k) --
// one code path per operand keeps the 64-deep variant at 4 waves per SIMD.
template <int KT, int MODE>
__global__ __launch_bounds__(256) void bmm_f32_kernel(BmmArgs p) {
    constexpr int KS = KT + 4;                 // padded row: conflict-free ds_read_b128 (KS % 16 == 4)
    constexpr int NV = KT / 16;                // float4 per thread, operand and stage
    __shared__ __attribute__((aligned(16))) float As[BT * KS];
    __shared__ __attribute__((aligned(16))) float Bs[BT * KS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int b1 = blockIdx.z / p.nb2, b2 = blockIdx.z % p.nb2;
    const float* A = p.A + b1 * p.sA1 + b2 * p.sA2;
    const float* B = p.B + b1 * p.sB1 + b2 * p.sB2;
    float* C = p.C + b1 * p.sC1 + b2 * p.sC2;
    const int m0 = blockIdx.y * BT, n0 = blockIdx.x * BT;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // 16-byte loads along each operand's contiguous axis when every address involved is 16-byte aligned
    // (leading dimensions, batch strides and base pointers multiples of 4 floats); element-wise otherwise
    const bool vecA = MODE >= 0 || (p.vec && (p.lda & 3) == 0), vecB = MODE >= 0 || (p.vec && (p.ldb & 3) == 0);

    // One operand tile: T[r][k] = X[r0 + r][k0 + k] when `rows_contig_k` (row-major along k), else
    // X[k0 + k][r0 + r]; R = valid rows (M or N), ld = leading dimension.
    auto fetch = [&](const float* X, int ld, bool rows_contig_k, bool vec, int r0, int R, int k0, f32x4* v) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (vec) {
                if (rows_contig_k) {
                    const int r = t >> 2, k4 = (t & 3) * 4 + 16 * i;
                    if (r0 + r < R) {
                        const float* src = X + (size_t)(r0 + r) * ld + k0 + k4;
                        if (k0 + k4 + 3 < p.K) v[i] = *reinterpret_cast<const f32x4*>(src);
                        else
                            for (int u = 0; u < 4; ++u) if (k0 + k4 + u < p.K) v[i][u] = src[u];
                    }
                } else {
                    const int k = (t >> 4) + 16 * i, r4 = (t & 15) * 4;
                    if (k0 + k < p.K) {
                        const float* src = X + (size_t)(k0 + k) * ld + r0 + r4;
                        if (r0 + r4 + 3 < R) v[i] = *reinterpret_cast<const f32x4*>(src);
                        else
                            for (int u = 0; u < 4; ++u) if (r0 + r4 + u < R) v[i][u] = src[u];
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = t + 256 * (4 * i + u);
                    const int r = rows_contig_k ? idx / KT : idx % BT, k = rows_contig_k ? idx % KT : idx / BT;
                    if (r0 + r < R && k0 + k < p.K)
                        v[i][u] = rows_contig_k ? X[(size_t)(r0 + r) * ld + k0 + k] : X[(size_t)(k0 + k) * ld + r0 + r];
                }
            }
        }
    };
    auto stash = [&](float* T, bool rows_contig_k, bool vec, const f32x4* v) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (vec) {
                if (rows_contig_k) {
                    const int r = t >> 2, k4 = (t & 3) * 4 + 16 * i;
                    *reinterpret_cast<f32x4*>(T + r * KS + k4) = v[i];
                } else {
                    const int k = (t >> 4) + 16 * i, r4 = (t & 15) * 4;
#pragma unroll
                    for (int u = 0; u < 4; ++u) T[(r4 + u) * KS + k] = v[i][u];
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = t + 256 * (4 * i + u);
                    const int r = rows_contig_k ? idx / KT : idx % BT, k = rows_contig_k ? idx % KT : idx / BT;
                    T[r * KS + k] = v[i][u];
                }
            }
        }
    };
    // op(A)[m][k]: row-major along k unless transposed; op(B)[k][n] viewed as T[n][k]: along k iff transposed
    const bool a_k = MODE >= 0 ? (MODE & 1) != 0 : !p.transA, b_k = MODE >= 0 ? (MODE & 2) != 0 : p.transB != 0;
    f32x4 va[NV], vb[NV];
    fetch(A, p.lda, a_k, vecA, m0, p.M, 0, va);
    fetch(B, p.ldb, b_k, vecB, n0, p.N, 0, vb);
    for (int k0 = 0; k0 < p.K; k0 += KT) {
        __syncthreads();                       // the previous stage is no longer read
        stash(As, a_k, vecA, va);
        stash(Bs, b_k, vecB, vb);
        __syncthreads();
        if (k0 + KT < p.K) {                   // next stage: in flight during the MFMAs below
            fetch(A, p.lda, a_k, vecA, m0, p.M, k0 + KT, va);
            fetch(B, p.ldb, b_k, vecB, n0, p.N, k0 + KT, vb);
        }
        const float* ap = As + (wm * 32 + l31) * KS + 4 * h;
        const float* bp = Bs + (wn * 32 + l31) * KS + 4 * h;
#pragma unroll
        for (int j = 0; j < KT / 8; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ap + 8 * j);
            const f32x4 b = *reinterpret_cast<const f32x4*>(bp + 8 * j);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = mfma32(a[u], b[u], acc);
        }
    }

    const int col = n0 + wn * 32 + l31;
    if (col < p.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + acc_row(r, h);
            if (row < p.M) {
                float* c = C + (size_t)row * p.ldc + col;
                const float v = p.alpha * acc[r];
                *c = p.accumulate ? *c + v : v;
            }
        }
    }
}

}  // namespace

extern "C" int tocvp_bmm_f32(const float* A, int lda, long sA1, long sA2, int transA, const float* B,
                             int ldb, long sB1, long sB2, int transB, float* C, int ldc, long sC1,
                             long sC2, int nb1, int nb2, int M, int N, int K, float alpha,
                             int accumulate, void* stream) {
    TOCVP_CHECK_ARG(A && B && C);
    TOCVP_CHECK_ARG(nb1 >= 0 && nb2 > 0 && M >= 0 && N >= 0 && K >= 0);
    TOCVP_CHECK_ARG(lda > 0 && ldb > 0 && ldc > 0);
    TOCVP_CHECK_ARG((long)nb1 * nb2 <= 65535);
    if (nb1 == 0 || M == 0 || N == 0) return TOCVP_OK;
    const int vec = tocvp_aligned16(A) && tocvp_aligned16(B) && ((sA1 | sA2 | sB1 | sB2) & 3) == 0;
    BmmArgs a{A, B, C, sA1, sA2, sB1, sB2, sC1, sC2, lda, ldb, ldc, transA ? 1 : 0, transB ? 1 : 0, nb2,
              M, N, K, alpha, accumulate ? 1 : 0, vec};
    const dim3 grid((N + BT - 1) / BT, (M + BT - 1) / BT, nb1 * nb2);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (K >= 64 && vec && (lda & 3) == 0 && (ldb & 3) == 0) {
        switch ((transA ? 0 : 1) | (transB ? 2 : 0)) {
            case 0: hipLaunchKernelGGL((bmm_f32_kernel<64, 0>), grid, dim3(256), 0, s, a); break;
            case 1: hipLaunchKernelGGL((bmm_f32_kernel<64, 1>), grid, dim3(256), 0, s, a); break;
            case 2: hipLaunchKernelGGL((bmm_f32_kernel<64, 2>), grid, dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((bmm_f32_kernel<64, 3>), grid, dim3(256), 0, s, a); break;
        }
    } else {
        hipLaunchKernelGGL((bmm_f32_kernel<16, -1>), grid, dim3(256), 0, s, a);
    }
    return tocvp_launch_status();
}
