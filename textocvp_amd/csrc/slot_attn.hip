// One slot-attention iteration over the H*W feature grid (reference: attention.py:99-103).
//
//   dots[i,j] = q_i . k_j * scale ; attn = softmax_i(dots) + eps ; upd_i = sum_j attn_ij v_j / sum_j attn_ij
//
// HBM-bound: k and v (N x D fp32 each) are streamed exactly once per iteration, everything else
// lives on chip.  gfx950 mapping:
//  * the K-slot query block (padded to 32 x 128) is staged once in LDS per workgroup;
//  * each wave walks 32-location tiles: the k tile is loaded with coalesced 16-byte reads into the
//    wave's private LDS tile; dots^T (32 locations x 32 slots) = k q^T is 64 fp32 MFMAs;
//  * slots sit on LANES of the accumulator, so the softmax ACROSS SLOTS is a 32-lane __shfl_xor
//    butterfly per accumulator register (wavefront shuffle reductions, no LDS);
//  * the attention tile is consumed in place as the B operand of  upd^T += v^T attn^T  with v^T
//    operands fetched straight from global memory as 128-byte coalesced segments;
//  * per-wave partial sums go to a workspace in [d][slot] order (coalesced) and a second tiny
//    kernel reduces them in a fixed order (deterministic) and renormalises.
#include "common.h"

namespace {

constexpr int SD = 128;           // slot / feature dim handled by this kernel
constexpr int QS = SD + 4;        // padded LDS row stride
constexpr int REC = SD * 32 + 32; // floats per partial record: upd^T [d][slot] + rowsum[slot]
constexpr float NEG_BIG = -1.0e30f;

struct SaArgs {
    const float* q; const float* k; const float* v; int ldkv;
    float* attn_out; float* ws;
    int B, Ks, N; int lpw;        // locations per wave
    float scale, eps;
};

__global__ __launch_bounds__(128) void slot_attn_partial_kernel(SaArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[32 * QS + 2 * 32 * QS];
    float* qs = lds;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    float* ks = lds + 32 * QS + wave * 32 * QS;
    const int b = blockIdx.y;

    // stage q (zero rows for padded slots)
    for (int i = t; i < 32 * (SD / 4); i += 128) {
        const int r = i / (SD / 4), c = (i % (SD / 4)) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < p.Ks) v = *reinterpret_cast<const f32x4*>(p.q + ((size_t)b * p.Ks + r) * SD + c);
        *reinterpret_cast<f32x4*>(qs + r * QS + c) = v;
    }
    __syncthreads();

    const int wave_id = blockIdx.x * 2 + wave;          // partial record index within the sample
    const int loc_begin = wave_id * p.lpw;
    const float* kb = p.k + (size_t)b * p.N * p.ldkv;
    const float* vb = p.v + (size_t)b * p.N * p.ldkv;

    f32x16 uacc[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) uacc[d][r] = 0.f;
    float rowsum = 0.f;
    const bool slot_ok = l31 < p.Ks;

    for (int loc0 = loc_begin; loc0 < loc_begin + p.lpw; loc0 += 32) {
        // k tile -> LDS (wave private): 32 rows x 128 floats, coalesced float4
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int i = lane + 64 * it;
            const int r = i >> 5, c = (i & 31) * 4;
            *reinterpret_cast<f32x4*>(ks + r * QS + c) =
                *reinterpret_cast<const f32x4*>(kb + (size_t)(loc0 + r) * p.ldkv + c);
        }
        __builtin_amdgcn_wave_barrier();

        // dots^T: rows = locations, cols (lanes) = slots
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* ka = ks + l31 * QS + 4 * h;
        const float* qa = qs + l31 * QS + 4 * h;
#pragma unroll
        for (int j = 0; j < SD / 8; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ka + 8 * j);
            const f32x4 bq = *reinterpret_cast<const f32x4*>(qa + 8 * j);
#pragma unroll
            for (int u = 0; u < 4; ++u) s = mfma32(a[u], bq[u], s);
        }

        // softmax over slots (lanes of one half), + eps
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = slot_ok ? s[r] * p.scale : NEG_BIG;
            const float mx = half_max32(x);
            const float e = slot_ok ? expf(x - mx) : 0.f;
            const float sm = half_sum32(e);
            const float a = slot_ok ? e / sm + p.eps : 0.f;
            s[r] = a;
            rowsum += a;
        }
        if (p.attn_out && slot_ok) {
            float* ao = p.attn_out + ((size_t)b * p.Ks + l31) * p.N + loc0;
#pragma unroll
            for (int r = 0; r < 16; ++r) ao[acc_row(r, h)] = s[r];
        }

        // upd^T (d x slots) += v^T (d x loc) * attn^T (loc x slots)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* va = vb + (size_t)(loc0 + acc_row(r, h)) * p.ldkv + l31;
#pragma unroll
            for (int d = 0; d < 4; ++d) uacc[d] = mfma32(va[32 * d], s[r], uacc[d]);
        }
        __builtin_amdgcn_wave_barrier();
    }

    // partial record: [d][slot] (lanes = consecutive slots -> coalesced) + rowsum[slot]
    const int nrec = p.N / p.lpw;
    float* rec = p.ws + ((size_t)b * nrec + wave_id) * REC;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) rec[(d * 32 + acc_row(r, h)) * 32 + l31] = uacc[d][r];
    rowsum += __shfl_xor(rowsum, 32, 64);
    if (h == 0) rec[SD * 32 + l31] = rowsum;
}

__global__ __launch_bounds__(256) void slot_attn_finalize_kernel(const float* __restrict__ ws,
                                                                 float* __restrict__ updates,
                                                                 int Ks, int nrec) {
    __shared__ float inv[32];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* base = ws + (size_t)b * nrec * REC;
    if (t < 32) {
        float s = 0.f;
        for (int r = 0; r < nrec; ++r) s += base[(size_t)r * REC + SD * 32 + t];
        inv[t] = (t < Ks) ? 1.0f / s : 0.f;
    }
    __syncthreads();
    for (int e = t; e < SD * 32; e += 256) {
        const int d = e >> 5, slot = e & 31;
        if (slot >= Ks) continue;
        float s = 0.f;
        for (int r = 0; r < nrec; ++r) s += base[(size_t)r * REC + e];
        updates[((size_t)b * Ks + slot) * SD + d] = s * inv[slot];
    }
}

inline int pick_lpw(int B, int N) {
    int lpw = 256;
    while (lpw > N) lpw >>= 1;
    while (lpw > 32 && (long)B * N / (2 * lpw) < 512) lpw >>= 1;
    return lpw;
}

}  // namespace

extern "C" size_t tocvp_slot_attn_ws_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    const int lpw = pick_lpw(B, N);
    return (size_t)B * (N / lpw) * REC * sizeof(float);
}

extern "C" int tocvp_slot_attn_iter_f32(const float* q, const float* k, const float* v, int ldkv,
                                        float* updates, float* attn_out, int B, int Ks, int N,
                                        int D, float scale, float eps, void* ws, size_t ws_bytes,
                                        void* stream) {
    TOCVP_CHECK_ARG(q && k && v && updates && ws);
    TOCVP_CHECK_ARG(B >= 0 && B <= 65535 && Ks > 0 && Ks <= 32 && D == SD);
    TOCVP_CHECK_ARG(N >= 64 && (N % 64) == 0 && ldkv >= D);
    if ((ldkv & 3) || !tocvp_aligned16(q) || !tocvp_aligned16(k) || !tocvp_aligned16(v))
        return TOCVP_EALIGN;
    if (B == 0) return TOCVP_OK;
    const int lpw = pick_lpw(B, N);
    TOCVP_CHECK_ARG(N % (2 * lpw) == 0);
    TOCVP_CHECK_ARG(ws_bytes >= tocvp_slot_attn_ws_bytes(B, N));
    SaArgs p{q, k, v, ldkv, attn_out, static_cast<float*>(ws), B, Ks, N, lpw, scale, eps};
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(slot_attn_partial_kernel, dim3(N / (2 * lpw), B), dim3(128), 0, s, p);
    if (hipGetLastError() != hipSuccess) return TOCVP_ELAUNCH;
    hipLaunchKernelGGL(slot_attn_finalize_kernel, dim3(B), dim3(256), 0, s,
                       static_cast<const float*>(ws), updates, Ks, N / lpw);
    return tocvp_launch_status();
}
