// Sustained matrix-core throughput of the shapes the hybrid conv could use, whole chip, random
// operands held in registers (power / DVFS included): f16 32x32x16 vs 16x16x32, e4m3 32x32x64 vs
// 16x16x128 (block-scaled), and the conv's 1:1 mix of f16 and fp8 matrix cycles.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/mfma_shape_rate scripts/probes/mfma_shape_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    const unsigned id = blockIdx.x * 256 + threadIdx.x;
    f16x8 ha[4], hb[2];
    i32x8 a8[4], b8[2];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 8; ++j) {
            const unsigned r = hash(id * 64 + i * 8 + j);
            ha[i][j] = (_Float16)(((int)(r & 0xffff) - 32768) * (1.f / 16384.f));
            // e4m3 bytes with exponent field 4..10: magnitudes 2^-3 .. 2^3, random sign / mantissa
            unsigned w = 0;
            for (int b = 0; b < 4; ++b) {
                const unsigned q = hash(r + b);
                w |= (((q & 0x80) | ((4 + (q >> 8) % 7) << 3) | (q & 7)) & 0xff) << (8 * b);
            }
            a8[i][j] = (int)w;
        }
    }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) {
            const unsigned r = hash(id * 64 + 32 + i * 8 + j);
            hb[i][j] = (_Float16)(((int)(r & 0xffff) - 32768) * (1.f / 16384.f));
            unsigned w = 0;
            for (int b = 0; b < 4; ++b) {
                const unsigned q = hash(r + 7 * b);
                w |= (((q & 0x80) | ((4 + (q >> 8) % 7) << 3) | (q & 7)) & 0xff) << (8 * b);
            }
            b8[i][j] = (int)w;
        }
    f32x16 c32[4][2];
    f32x4 c16[8][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) c32[i][j][r] = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) c16[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 4) {           // f16 32x32x16: 8 tiles x 4 k-steps = K 64
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        c32[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[(m + ks) & 3], hb[n], c32[m][n], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 5) {           // f16 16x16x32: 32 tiles x 2 k-steps = K 64
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        c16[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha[(m + ks) & 3], hb[n & 1], c16[m][n], 0, 0, 0);
        }
        if (MODE == 2 || MODE == 4) {           // e4m3 32x32x64, two products per tile (the conv's cross terms)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        c32[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[(m + pl) & 3], b8[n], c32[m][n], 0, 0, 0, 127, 0, 123);
        }
        if (MODE == 3 || MODE == 5) {           // e4m3 16x16x128: 32 tiles, K 128 each -> half the instructions per K
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    c16[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[m & 3], b8[n & 1], c16[m][n], 0, 0, 0, 127, 0, 123);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += c32[i][j][r];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += c16[i][j][r];
    if (s == 1.2345f) out[id] = s;
}

int main() {
    float* out; hipMalloc(&out, 1 << 24);
    const char* names[6] = {"f16 32x32x16", "f16 16x16x32", "e4m3 32x32x64 (x2 products)", "e4m3 16x16x128 (K128)",
                            "mix 32x32: f16 K64 + 2 e4m3 K64", "mix 16x16: f16 K64 + e4m3 K128"};
    // matrix "units" per iteration and wave, in 32-cycle f16 32x32x16 equivalents
    const double units[6] = {32, 32, 32, 32, 64, 64};
    for (int wpc = 1; wpc <= 2; ++wpc)
        for (int m = 0; m < 6; ++m) {
            const int iters = 20000, blocks = 256 * wpc;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (m) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, iters); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double u = units[m] * iters * 4.0 * blocks;            // f16-equivalent 32x32x16 MFMAs
            const double tf = u * 32768.0 / (ms * 1e-3) / 1e12;          // at f16 flop per unit
            printf("%d wave/SIMD  %-34s: %7.2f ms  %7.1f f16-equivalent TFLOP/s (units/s: %.3e)\n", wpc, names[m], ms, tf,
                   u / (ms * 1e-3));
        }
    return 0;
}
