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

// GELU (exact-erf form of nn.GELU / nn.TransformerEncoderLayer(activation="gelu")) for the GEMM epilogues.  erf by
// Abramowitz & Stegun 7.1.26 (|error| < 1.5e-7 in exact arithmetic), branch-free: one v_rcp_f32, one v_exp_f32, five fma.
// In fp32 the GELU's absolute error against float64 is 4.6e-7 over |v| < 11, the same as with the library's erff
// (4.5e-7: both are dominated by the rounding of 0.5 v (1 + erf)); erff evaluates BOTH of its branches in a divergent
// wave and cost 270 us of a 1090 us ViT fc1 launch (65792 x 3072 outputs).  Every kernel of the library uses this one
// function, so plane / fp32 / fused paths stay bit-identical to one another.
__device__ __forceinline__ float tocvp_gelu(float v) {
    const float x = v * 0.70710678118654752440f;
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float p = 1.061405429f;
    p = __builtin_fmaf(p, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __expf(-ax * ax);
    const float r = __builtin_fmaf(-p, e, 1.0f);                     // erf(|x|)
    float g = 0.5f * v * (1.0f + __builtin_copysignf(r, x));
    // (opaque to the optimiser: a residual added right behind the activation must not be contracted into this product in
    // one kernel and not in another -- the epilogues that stage through LDS before the add cannot fuse it)
    asm volatile("" : "+v"(g));
    return g;
}

// ---- split-plane activation stores (producer side of the split GEMMs) ---------------------------
// nsplit 2 / 3: bf16 planes hi (+ mid) + lo of v;  nsplit 22: two fp16 planes of 2^8 v ("f16x3",
// gemm_bf16.hip Elem<true>: the pre-scale keeps the lo plane out of the fp16 subnormals, values
// saturate at |v| = 255.9).  Plane s of the 4 values goes to base + s * plane_stride (elements).
constexpr float TOCVP_F16X3_ACT_SCALE = 256.f;
constexpr float TOCVP_F16X3_WEIGHT_SCALE = 1024.f;

// Two fp32 values -> their fp16 planes as packed pairs, hi = f16(x), lo = f16(x - hi), in three instructions
// (v_cvt_pk_f16_f32 + two mixed-precision FMAs that subtract in fp32 and round once): bit-identical to the C expressions
// (_Float16)x and (_Float16)(x - (float)hi), for which the compiler emits 8.5 instructions per pair (it converts hi twice
// and back once).  The caller clamps / scales.
__device__ __forceinline__ void tocvp_split2_f16(float x0, float x1, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=&v"(lo) : "v"(hi), "v"(x0));
    // (the s_nop: hipcc pads nothing behind an asm statement, and an MFMA that takes `lo` as an operand right away would read
    // the register before this vector instruction has written it -- two wait states, VALU write -> XDL read)
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1" : "+v"(lo) : "v"(hi), "v"(x1));
}

__device__ __forceinline__ void tocvp_store_planes4(void* base, size_t elem_off, size_t plane_stride,
                                                    f32x4 v, int nsplit) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
    if (nsplit == 22) {
        _Float16* ys = static_cast<_Float16*>(base) + elem_off;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = __builtin_amdgcn_fmed3f(v[u] * TOCVP_F16X3_ACT_SCALE, -65504.f, 65504.f);
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        unsigned h0, l0, h1, l1;                                     // hi = f16(v), lo = f16(v - hi): the same bits as the C form
        tocvp_split2_f16(v[0], v[1], h0, l0);
        tocvp_split2_f16(v[2], v[3], h1, l1);
        *reinterpret_cast<u32x2_t*>(ys) = u32x2_t{h0, h1};
        *reinterpret_cast<u32x2_t*>(ys + plane_stride) = u32x2_t{l0, l1};
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
