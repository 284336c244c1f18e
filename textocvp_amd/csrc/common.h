// Shared device helpers for libtocvp (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tocvp.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TOCVP_CHECK_ARG(cond) \
    do {                      \
        if (!(cond)) return TOCVP_EINVAL; \
    } while (0)

static inline int tocvp_launch_status() {
    return hipGetLastError() == hipSuccess ? TOCVP_OK : TOCVP_ELAUNCH;
}

static inline bool tocvp_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32), exact fp32 fma chain.
//   lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
//   D: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), reg in [0,16).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of accumulator register r for lane-half h (32x32 MFMA C/D layout)
// torch.clamp(x, 0, 1): NaN stays NaN (both comparisons are false for it)
__device__ __forceinline__ float tocvp_clamp01(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// butterfly reductions over the 32 lanes of one wave half (lanes l and l^32 stay separate)
__device__ __forceinline__ float half_max32(float v) {
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float half_sum32(float v) {
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- split-plane activation stores (producer side of the split GEMMs) ---------------------------
// nsplit 2 / 3: bf16 planes hi (+ mid) + lo of v;  nsplit 22: two fp16 planes of 2^8 v ("f16x3",
// gemm_bf16.hip Elem<true>: the pre-scale keeps the lo plane out of the fp16 subnormals, values
// saturate at |v| = 255.9).  Plane s of the 4 values goes to base + s * plane_stride (elements).
constexpr float TOCVP_F16X3_ACT_SCALE = 256.f;
constexpr float TOCVP_F16X3_WEIGHT_SCALE = 1024.f;

__device__ __forceinline__ void tocvp_store_planes4(void* base, size_t elem_off, size_t plane_stride,
                                                    f32x4 v, int nsplit) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
    if (nsplit == 22) {
        _Float16* ys = static_cast<_Float16*>(base) + elem_off;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = __builtin_amdgcn_fmed3f(v[u] * TOCVP_F16X3_ACT_SCALE, -65504.f, 65504.f);
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            f16x4_t piece;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                piece[u] = (_Float16)v[u];
                v[u] -= (float)piece[u];
            }
            *reinterpret_cast<f16x4_t*>(ys + (size_t)sp * plane_stride) = piece;
        }
    } else {
        __bf16* ys = static_cast<__bf16*>(base) + elem_off;
        for (int sp = 0; sp < nsplit; ++sp) {
            bf16x4_t piece;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                piece[u] = (__bf16)v[u];
                v[u] -= (float)piece[u];
            }
            *reinterpret_cast<bf16x4_t*>(ys + (size_t)sp * plane_stride) = piece;
        }
    }
}
