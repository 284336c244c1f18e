// planes3 (persistent, interleaved) against planes2 on the same operands: bitwise comparison of the outputs
// (same products, same k order), timings, and the in-kernel stamps of planes3.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itextocvp_amd/csrc -DTOCVP_P3_STAMP \
//         -o scripts/probes/gemm16p3_check scripts/probes/gemm16p3_check.hip ;  ./gemm16p3_check [M N K [MI [skew [grid]]]]
#include "gemm_f16p.hip"
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>

static float* g_part = nullptr; static unsigned* g_flag = nullptr;
template <int MI> static void run3(const PArgs& p, int grid, int skew, int sk) {
    const int ntm = (p.M + 64 * MI - 1) / (64 * MI), ntn = p.N / 256;
    if (sk) {
        if constexpr (MI == 2)
            hipLaunchKernelGGL((gemm_f16_planes3_kernel<2, true>), dim3(grid), dim3(512), 0, 0, p, ntm, ntn, skew, g_part, g_flag);
    } else {
        hipLaunchKernelGGL((gemm_f16_planes3_kernel<MI, false>), dim3(std::min(grid, ntm * ntn)), dim3(512), 0, 0, p, ntm, ntn,
                           skew, nullptr, nullptr);
    }
}
template <int MI> static void run2(const PArgs& p) {
    const int ntm = (p.M + 64 * MI - 1) / (64 * MI), ntn = p.N / 256;
    hipLaunchKernelGGL(gemm_f16_planes2_kernel<MI>, dim3(ntm * ntn), dim3(512), 0, 0, p);
}

int main(int argc, char** argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 38400, N = argc > 3 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
    const int MI = argc > 4 ? atoi(argv[4]) : 4, skew = argc > 5 ? atoi(argv[5]) : 0, grid = argc > 6 ? atoi(argv[6]) : 256;
    const int csplit = argc > 7 ? atoi(argv[7]) : 0, useR = argc > 8 ? atoi(argv[8]) : 0, sk = argc > 9 ? atoi(argv[9]) : 0;
    {   void* ws; const size_t wsb = tocvp_gemm_f16planes_ws_bytes(); hipMalloc(&ws, wsb); hipMemset(ws, 0, wsb);
        g_flag = (unsigned*)ws; g_part = (float*)((unsigned char*)ws + 4096); }
    _Float16 *a, *w; float *c2, *c3, *b, *r;
    hipMalloc(&a, (size_t)M * 2 * K * 2); hipMalloc(&w, (size_t)N * 2 * K * 2);
    hipMalloc(&c2, (size_t)M * N * 4); hipMalloc(&c3, (size_t)M * N * 4); hipMalloc(&b, N * 4); hipMalloc(&r, (size_t)M * N * 4);
    std::vector<_Float16> h((size_t)1 << 22);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((float)(s >> 8) / (1 << 24) - 0.5f) * 64.f); }
    for (size_t o = 0; o < (size_t)M * 2 * K; o += h.size())
        hipMemcpy(a + o, h.data(), std::min(h.size(), (size_t)M * 2 * K - o) * 2, hipMemcpyHostToDevice);
    for (size_t o = 0; o < (size_t)N * 2 * K; o += h.size() - 12345)
        hipMemcpy(w + o, h.data() + 777, std::min(h.size() - 12345, (size_t)N * 2 * K - o) * 2, hipMemcpyHostToDevice);
    std::vector<float> hb(N); for (int i = 0; i < N; ++i) hb[i] = 0.01f * (i % 97) - 0.3f;
    hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
    {   std::vector<float> hr((size_t)1 << 20); for (size_t i = 0; i < hr.size(); ++i) hr[i] = 0.001f * (float)(i % 1013);
        for (size_t o = 0; o < (size_t)M * N; o += hr.size()) hipMemcpy(r + o, hr.data(), std::min(hr.size(), (size_t)M * N - o) * 4, hipMemcpyHostToDevice); }
    PArgs p{(const unsigned char*)a, (const unsigned char*)w, b, useR ? r : nullptr, N, c2, N, csplit, M, N, K, 1};
    PArgs q = p; q.C = c3;
    hipMemset(c2, 0xff, (size_t)M * N * 4); hipMemset(c3, 0xee, (size_t)M * N * 4);
    auto go2 = [&]() { if (MI == 4 || MI == 0) run2<4>(p); else run2<2>(p); };
    auto go3 = [&]() {
        if (MI == 0) {          // the library's own plan (phases of tile heights)
            tocvp_gemm_f16planes_ws_f32(q.A, q.W, q.bias, q.R, q.ldr, q.C, q.c_split, q.ldc, M, N, K, q.act, nullptr, 0, nullptr);
        } else if (MI == 4) run3<4>(q, grid, skew, sk); else if (MI == 2) run3<2>(q, grid, skew, sk); else run3<1>(q, grid, skew, 0); };
    go2(); go3();
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<float> o2((size_t)M * N), o3((size_t)M * N);
    hipMemcpy(o2.data(), c2, o2.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(o3.data(), c3, o3.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0; double maxd = 0;
    for (size_t i = 0; i < o2.size(); ++i)
        if (memcmp(&o2[i], &o3[i], 4) != 0) { if (!bad) first = i; ++bad; maxd = std::max(maxd, (double)fabsf(o2[i] - o3[i])); }
    double maxrel = 0;
    for (size_t i = 0; i < o2.size(); ++i) maxrel = std::max(maxrel, (double)fabsf(o2[i] - o3[i]) / (1.0 + fabs((double)o2[i])));
    printf("%dx%dx%d MI=%d csplit=%d R=%d sk=%d: planes3 vs planes2: %zu of %zu words differ (first at row %zu col %zu, max |d| %.3g, "
           "max |d|/(1+|x|) %.3g)\n", M, N, K, MI, csplit, useR, sk, bad, o2.size(), first / N, first % N, maxd, maxrel);
    if (sk) {   // determinism of the cut tiles: a second launch must reproduce the first bit for bit
        hipMemset(c3, 0x11, (size_t)M * N * 4); go3(); hipDeviceSynchronize();
        std::vector<float> o4((size_t)M * N); hipMemcpy(o4.data(), c3, o4.size() * 4, hipMemcpyDeviceToHost);
        printf("  stream-K rerun bit-identical: %s\n", memcmp(o3.data(), o4.data(), o4.size() * 4) == 0 ? "yes" : "NO");
        bad = maxrel > 1e-5 ? 1 : 0;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms2, ms3;
    const int reps = 10;
    for (int rep = 0; rep < 2; ++rep) {       // interleaved rounds in one process
        for (int i = 0; i < 3; ++i) go2();
        hipEventRecord(e0); for (int i = 0; i < reps; ++i) go2(); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms2, e0, e1);
        for (int i = 0; i < 3; ++i) go3();
        hipEventRecord(e0); for (int i = 0; i < reps; ++i) go3(); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms3, e0, e1);
        printf("  round %d: planes2 %.1f us (%.1f TF)   planes3 %.1f us (%.1f TF)\n", rep, ms2 / reps * 1e3,
               2.0 * M * N * K / (ms2 / reps) / 1e9, ms3 / reps * 1e3, 2.0 * M * N * K / (ms3 / reps) / 1e9);
    }
    if (MI == 0) return bad ? 2 : 0;
    std::vector<unsigned long long> st(256 * 4);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(tocvp_p3_stamps), st.size() * 8);
    std::vector<double> tot, epi;
    const int g = std::min(grid, ((M + 64 * MI - 1) / (64 * MI)) * (N / 256));
    for (int i = 0; i < g; ++i) { tot.push_back(double(st[i * 4 + 1] - st[i * 4])); epi.push_back(double(st[i * 4 + 2]) / std::max(1.0, double(st[i * 4 + 3]))); }
    std::sort(tot.begin(), tot.end()); std::sort(epi.begin(), epi.end());
    printf("  planes3 stamps: workgroup lifetime median %.0f max %.0f cycles; epilogue per tile median %.0f max %.0f\n",
           tot[tot.size() / 2], tot.back(), epi[epi.size() / 2], epi.back());
    return bad ? 2 : 0;
}
