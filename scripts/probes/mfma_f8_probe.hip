// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950: operand lane/byte layout (exact small
// integers), E8M0 scale semantics, and issue rate against the f16 32x32x16 form.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f8_probe scripts/probes/mfma_f8_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// e4m3 encoding of small non-negative integers 0..15 (exact)
__host__ __device__ inline uint8_t e4m3_of_int(int v) {
    // e4m3fn: bias 7; value = 1.mmm * 2^(e-7)
    if (v == 0) return 0;
    int s = v < 0; if (s) v = -v;
    int e = 0; while ((v >> (e + 1)) != 0) ++e;          // floor(log2 v)
    int m = ((v << 3) >> e) & 7;                          // exact for v < 16
    return (uint8_t)((s << 7) | ((e + 7) << 3) | m);
}

__global__ void probe(const uint8_t* A, const uint8_t* B, float* C, int scaleA_e, int scaleB_e) {
    // A: 32 rows x 64 k (row-major bytes), B: 64 k x 32 cols stored as Bt[col][k]
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    i32x8 a, b;
    const uint8_t* ap = A + r * 64 + h * 32;
    const uint8_t* bp = B + r * 64 + h * 32;
    for (int i = 0; i < 8; ++i) {
        a[i] = *reinterpret_cast<const int*>(ap + 4 * i);
        b[i] = *reinterpret_cast<const int*>(bp + 4 * i);
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scaleA_e, 0, scaleB_e);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C[row * 32 + r] = c[i];
    }
}

template <int MODE>
__global__ void rate(float* out, int iters) {
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = c1[i] = c2[i] = c3[i] = 0.f; }
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x38383838; }
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)1.0f; hb[i] = (_Float16)0.5f; }
    long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {        // f16 32x32x16
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c3, 0, 0, 0);
        } else if (MODE == 1) { // fp8 x fp8 scaled
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 127, 0, 127);
        } else if (MODE == 2) { // fp6 x fp6
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 2, 2, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 2, 2, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 2, 2, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 2, 2, 0, 127, 0, 127);
        } else if (MODE == 3) { // fp4 x fp4
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, 127, 0, 127);
        } else {                // fp8 x fp6 mixed
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 2, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 2, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 2, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 2, 0, 127, 0, 127);
        }
    }
    long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0) / (4.f * iters); }
    if (s == 12345.f) out[1] = s;
}

int main() {
    uint8_t hA[32 * 64], hB[32 * 64];
    int iA[32 * 64], iB[32 * 64];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) {
        iA[i] = rand() % 7 - 3; iB[i] = rand() % 9 - 4;
        hA[i] = e4m3_of_int(iA[i]); hB[i] = e4m3_of_int(iB[i]);
    }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    float hC[32 * 32];
    const int scales[3][2] = {{127, 127}, {130, 127}, {127, 124}};
    for (int t = 0; t < 3; ++t) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, scales[t][0], scales[t][1]);
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        const float f = ldexpf(1.f, scales[t][0] - 127 + scales[t][1] - 127);
        int bad = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                int ref = 0;
                for (int k = 0; k < 64; ++k) ref += iA[i * 64 + k] * iB[j * 64 + k];
                if (fabsf(hC[i * 32 + j] - f * ref) > 1e-3f) ++bad;
            }
        printf("layout k=32h+j, scaleA_e=%d scaleB_e=%d: %d / 1024 mismatches (C[0][0]=%g)\n",
               scales[t][0], scales[t][1], bad, hC[0]);
    }
    float* dO; hipMalloc(&dO, 8); float hO[2];
    const char* names[5] = {"f16 32x32x16", "fp8xfp8 32x32x64", "fp6xfp6 32x32x64", "fp4xfp4 32x32x64", "fp8xfp6 32x32x64"};
    for (int m = 0; m < 5; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(rate<0>, dim3(1), dim3(64), 0, 0, dO, 4096);
            if (m == 1) hipLaunchKernelGGL(rate<1>, dim3(1), dim3(64), 0, 0, dO, 4096);
            if (m == 2) hipLaunchKernelGGL(rate<2>, dim3(1), dim3(64), 0, 0, dO, 4096);
            if (m == 3) hipLaunchKernelGGL(rate<3>, dim3(1), dim3(64), 0, 0, dO, 4096);
            if (m == 4) hipLaunchKernelGGL(rate<4>, dim3(1), dim3(64), 0, 0, dO, 4096);
            hipMemcpy(hO, dO, 8, hipMemcpyDeviceToHost);
        }
        printf("%-20s : %.1f clock64 ticks per MFMA (one wave)\n", names[m], hO[0]);
    }
    return 0;
}
