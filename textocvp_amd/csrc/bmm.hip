// Generic strided / batched fp32 GEMM on the exact fp32 MFMA, for the BACKWARD pass of the predictor
// training step: weight and data gradients of nn.Linear
// (dW = dY^T X, dX = dY W), and the per-head products of attention backward
// (S = Q K^T, dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q).
//
//   C[b1,b2] (M x N) = alpha * op(A[b1,b2]) (M x K) * op(B[b1,b2]) (K x N)  (+ C[b1,b2] if accumulate)
//   op(X) = X or X^T; every operand has a leading dimension and two batch strides (elements), so the
//   (batch, head) slices of a (B, T, H*dh) tensor are addressed in place.
//
// 64 x 64 tile per 4-wave workgroup (32 x 32 per wave), 16-deep k-tiles staged through LDS as
// [row][k] images for BOTH operands (20-word rows: conflict-free ds_read_b128), any M / N / K (zero
// fill on load, masked stores).  Loads are 16 bytes per lane along the operand's contiguous axis when the
// addresses allow it, element-wise otherwise; built for generality, not for the roofline.
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

constexpr int BT = 64, KT = 16, KS = KT + 4;

__global__ __launch_bounds__(256) void bmm_f32_kernel(BmmArgs p) {
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
    const bool vecA = p.vec && (p.lda & 3) == 0, vecB = p.vec && (p.ldb & 3) == 0;

    for (int k0 = 0; k0 < p.K; k0 += KT) {
        __syncthreads();
        // ---- A image: As[m][k] = op(A)[m0+m][k0+k]   (one float4 per thread = the whole 64 x 16 tile)
        if (vecA) {
            if (!p.transA) {
                const int m = t >> 2, k4 = (t & 3) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m0 + m < p.M) {
                    const float* src = A + (size_t)(m0 + m) * p.lda + k0 + k4;
                    if (k0 + k4 + 3 < p.K) v = *reinterpret_cast<const f32x4*>(src);
                    else
                        for (int u = 0; u < 4; ++u) if (k0 + k4 + u < p.K) v[u] = src[u];
                }
                *reinterpret_cast<f32x4*>(As + m * KS + k4) = v;
            } else {
                const int k = t >> 4, m4 = (t & 15) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k0 + k < p.K) {
                    const float* src = A + (size_t)(k0 + k) * p.lda + m0 + m4;
                    if (m0 + m4 + 3 < p.M) v = *reinterpret_cast<const f32x4*>(src);
                    else
                        for (int u = 0; u < 4; ++u) if (m0 + m4 + u < p.M) v[u] = src[u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) As[(m4 + u) * KS + k] = v[u];
            }
        } else {
#pragma unroll
            for (int i = 0; i < (BT * KT) / 256; ++i) {
                const int idx = t + 256 * i;
                const int m = p.transA ? idx % BT : idx / KT, k = p.transA ? idx / BT : idx % KT;
                float v = 0.f;
                if (m0 + m < p.M && k0 + k < p.K)
                    v = p.transA ? A[(size_t)(k0 + k) * p.lda + m0 + m] : A[(size_t)(m0 + m) * p.lda + k0 + k];
                As[m * KS + k] = v;
            }
        }
        // ---- B image: Bs[n][k] = op(B)[k0+k][n0+n]
        if (vecB) {
            if (p.transB) {
                const int n = t >> 2, k4 = (t & 3) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n0 + n < p.N) {
                    const float* src = B + (size_t)(n0 + n) * p.ldb + k0 + k4;
                    if (k0 + k4 + 3 < p.K) v = *reinterpret_cast<const f32x4*>(src);
                    else
                        for (int u = 0; u < 4; ++u) if (k0 + k4 + u < p.K) v[u] = src[u];
                }
                *reinterpret_cast<f32x4*>(Bs + n * KS + k4) = v;
            } else {
                const int k = t >> 4, n4 = (t & 15) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k0 + k < p.K) {
                    const float* src = B + (size_t)(k0 + k) * p.ldb + n0 + n4;
                    if (n0 + n4 + 3 < p.N) v = *reinterpret_cast<const f32x4*>(src);
                    else
                        for (int u = 0; u < 4; ++u) if (n0 + n4 + u < p.N) v[u] = src[u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) Bs[(n4 + u) * KS + k] = v[u];
            }
        } else {
#pragma unroll
            for (int i = 0; i < (BT * KT) / 256; ++i) {
                const int idx = t + 256 * i;
                const int n = p.transB ? idx / KT : idx % BT, k = p.transB ? idx % KT : idx / BT;
                float v = 0.f;
                if (n0 + n < p.N && k0 + k < p.K)
                    v = p.transB ? B[(size_t)(n0 + n) * p.ldb + k0 + k] : B[(size_t)(k0 + k) * p.ldb + n0 + n];
                Bs[n * KS + k] = v;
            }
        }
        __syncthreads();
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
    hipLaunchKernelGGL(bmm_f32_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return tocvp_launch_status();
}
