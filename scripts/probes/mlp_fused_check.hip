// Fused MLP (csrc/mlp_fused.hip) against the two-GEMM path of the library on the same operand planes: bitwise comparison
// and interleaved timings.  Links libtocvp.so (C-ABI only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o scripts/probes/mlp_fused_check scripts/probes/mlp_fused_check.hip \
//         -Ltextocvp_amd/_lib -ltocvp -Wl,-rpath,'$ORIGIN/../../textocvp_amd/_lib'
//   ./mlp_fused_check [M [Hd [rounds [relu_sparse]]]]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "tocvp.h"

// the kernel under test is compiled INTO the probe (its definition of tocvp_mlp_f16x3_fused_f32 takes precedence over the
// library's), so that -DTOCVP_MLP_STAMP / -DTOCVP_MLP_ABLATE=n variants can be built without touching the library
#include "../../textocvp_amd/csrc/mlp_fused.hip"

__global__ void planes_kernel(const float* x, _Float16* out, long rows, int K) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * K) return;
    const long r = i / K; const int k = (int)(i - r * K);
    float v = __builtin_amdgcn_fmed3f(x[i] * 256.f, -65504.f, 65504.f);
    const _Float16 hi = (_Float16)v;
    out[(r * 2 + 0) * K + k] = hi;
    out[(r * 2 + 1) * K + k] = (_Float16)(v - (float)hi);
}

#define CK(x) do { int e_ = (x); if (e_) { printf("%s -> %d\n", #x, e_); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 38400, Hd = argc > 2 ? atoi(argv[2]) : 2048, rounds = argc > 3 ? atoi(argv[3]) : 5;
    const float xs = argc > 4 ? atof(argv[4]) : 1.0f;
    const int E = 512;
    float *x, *w1, *w2, *b1, *b2, *r, *y_ref, *y_fu;
    _Float16 *xp, *hp, *w1f, *w2f;
    hipMalloc(&x, (size_t)M * E * 4); hipMalloc(&r, (size_t)M * E * 4); hipMalloc(&y_ref, (size_t)M * E * 4); hipMalloc(&y_fu, (size_t)M * E * 4);
    hipMalloc(&w1, (size_t)Hd * E * 4); hipMalloc(&w2, (size_t)Hd * E * 4); hipMalloc(&b1, Hd * 4); hipMalloc(&b2, E * 4);
    hipMalloc(&xp, (size_t)M * 2 * E * 2); hipMalloc(&hp, (size_t)M * 2 * Hd * 2); hipMalloc(&w1f, (size_t)Hd * 2 * E * 2); hipMalloc(&w2f, (size_t)Hd * 2 * E * 2);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / (1 << 24) - 0.5f; };
    {   std::vector<float> hx((size_t)M * E); for (auto& v : hx) v = rnd() * 4.f * xs; hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        for (auto& v : hx) v = rnd() * 2.f; hipMemcpy(r, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> hw((size_t)Hd * E); for (auto& v : hw) v = rnd() * 0.1f; hipMemcpy(w1, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        for (auto& v : hw) v = rnd() * 0.05f; hipMemcpy(w2, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> hb(Hd); for (auto& v : hb) v = rnd() * 0.5f; hipMemcpy(b1, hb.data(), Hd * 4, hipMemcpyHostToDevice);
        hb.resize(E); for (auto& v : hb) v = rnd(); hipMemcpy(b2, hb.data(), E * 4, hipMemcpyHostToDevice); }
    hipLaunchKernelGGL(planes_kernel, dim3((unsigned)(((size_t)M * E + 255) / 256)), dim3(256), 0, 0, x, xp, (long)M, E);
    CK(tocvp_split_weights_frag_f16(w1, w1f, Hd, E, nullptr));
    CK(tocvp_split_weights_frag_f16(w2, w2f, E, Hd, nullptr));
    auto ref = [&]() {
        int e = tocvp_gemm_bf16wfrag_f32(xp, 1, E, w1f, 22, b1, nullptr, 0, nullptr, 1, 1, 0, hp, 1, Hd, M, Hd, E, TOCVP_ACT_RELU, nullptr);
        if (e) return e;
        return tocvp_gemm_bf16wfrag_f32(hp, 1, Hd, w2f, 22, b2, r, E, nullptr, 1, 1, 0, y_ref, 0, E, M, E, Hd, TOCVP_ACT_NONE, nullptr);
    };
    const int use_ws = argc > 5 ? atoi(argv[5]) : 1;
    void* ws = nullptr; const size_t wsb = tocvp_mlp_f16x3_fused_ws_bytes();
    if (use_ws) { hipMalloc(&ws, wsb); hipMemset(ws, 0, wsb); }
    auto fused = [&]() { return tocvp_mlp_f16x3_fused_f32(xp, w1f, b1, w2f, b2, r, E, y_fu, E, M, E, Hd, ws, ws ? wsb : 0, nullptr); };
    hipMemset(y_ref, 0xff, (size_t)M * E * 4); hipMemset(y_fu, 0xee, (size_t)M * E * 4);
    CK(ref()); CK(fused());
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<float> o1((size_t)M * E), o2((size_t)M * E);
    hipMemcpy(o1.data(), y_ref, o1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), y_fu, o2.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0; double maxd = 0, maxv = 0;
    for (size_t i = 0; i < o1.size(); ++i) {
        if (memcmp(&o1[i], &o2[i], 4)) { if (!bad) first = i; ++bad; }
        maxd = std::max(maxd, (double)fabsf(o1[i] - o2[i])); maxv = std::max(maxv, (double)fabsf(o1[i]));
    }
    {   // determinism of the cut tiles: a second run must reproduce the first bit for bit
        CK(fused()); hipDeviceSynchronize();
        std::vector<float> o3((size_t)M * E); hipMemcpy(o3.data(), y_fu, o3.size() * 4, hipMemcpyDeviceToHost);
        printf("second run %s the first\n", memcmp(o2.data(), o3.data(), o3.size() * 4) ? "DIFFERS from" : "equals"); }
    printf("M %d Hd %d: %zu of %zu words differ (first at row %zu col %zu), max |diff| %.3e, max |ref| %.3e\n", M, Hd, bad, o1.size(),
           first / E, first % E, maxd, maxv);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 4.0 * M * E * Hd;
    for (int rd = 0; rd < rounds; ++rd) {
        float t_ref, t_fu; const int reps = 10;
        hipEventRecord(e0); for (int i = 0; i < reps; ++i) ref(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t_ref, e0, e1);
        hipEventRecord(e0); for (int i = 0; i < reps; ++i) fused(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&t_fu, e0, e1);
        printf("round %d: two GEMMs %.1f us (%.0f TFLOP/s)   fused %.1f us (%.0f TFLOP/s)\n", rd, t_ref * 1e3 / reps,
               flop / (t_ref * 1e-3 / reps) * 1e-12, t_fu * 1e3 / reps, flop / (t_fu * 1e-3 / reps) * 1e-12);
    }
#ifdef TOCVP_MLP_STAMP
    {   fused(); hipDeviceSynchronize();
        std::vector<unsigned long long> st(1024 * 8);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(tocvp_mlp_stamps), st.size() * 8);
        const int nwg = std::min(1024, (M + 127) / 128);
        double tot = 0, g1 = 0, e1 = 0, g2 = 0, epi = 0; unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < nwg; ++i) { const auto* q = &st[i * 8]; tot += q[1] - q[0]; g1 += q[2]; e1 += q[3]; g2 += q[4]; epi += q[1] - q[5];
            t0 = std::min(t0, q[0]); t1 = std::max(t1, q[1]); }
        printf("stamps (s_memtime ticks, mean per workgroup of %d): total %.0f  product1 %.0f  epilogue1+barrier %.0f  product2 %.0f  "
               "final epilogue %.0f;  kernel span %llu ticks\n", nwg, tot / nwg, g1 / nwg, e1 / nwg, g2 / nwg, epi / nwg, t1 - t0);
    }
#endif
    return bad ? 2 : 0;
}
