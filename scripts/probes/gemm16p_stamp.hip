// Where a workgroup of the all-DMA planes GEMM spends its cycles: s_memtime stamps at the phase boundaries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itextocvp_amd/csrc -DTOCVP_P2_STAMP \
//         -o scripts/probes/gemm16p_stamp scripts/probes/gemm16p_stamp.hip ;  ./gemm16p_stamp [M N K]
#include "gemm_f16p.hip"
#include <stdio.h>
#include <algorithm>
#include <vector>

int main(int argc, char** argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 38400, N = argc > 3 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
    _Float16 *a, *w; float *c, *b;
    hipMalloc(&a, (size_t)M * 2 * K * 2); hipMalloc(&w, (size_t)N * 2 * K * 2); hipMalloc(&c, (size_t)M * N * 4); hipMalloc(&b, N * 4);
    std::vector<_Float16> h((size_t)1 << 22);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((float)(s >> 8) / (1 << 24) - 0.5f) * 64.f); }
    for (size_t o = 0; o < (size_t)M * 2 * K; o += h.size())
        hipMemcpy(a + o, h.data(), std::min(h.size(), (size_t)M * 2 * K - o) * 2, hipMemcpyHostToDevice);
    for (size_t o = 0; o < (size_t)N * 2 * K; o += h.size())
        hipMemcpy(w + o, h.data(), std::min(h.size(), (size_t)N * 2 * K - o) * 2, hipMemcpyHostToDevice);
    hipMemset(b, 0, N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) tocvp_gemm_f16planes_f32(a, w, b, nullptr, 0, c, 0, N, M, N, K, 1, nullptr);
    hipEventRecord(e0);
    tocvp_gemm_f16planes_f32(a, w, b, nullptr, 0, c, 0, N, M, N, K, 1, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int nwg = ((M + 255) / 256) * (N / 256);
    std::vector<unsigned long long> st(8192 * 4);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(tocvp_p2_stamps), st.size() * 8);
    const int n = std::min(nwg, 8192);
    unsigned long long tmin = ~0ull, tmax = 0;
    std::vector<double> pro, loop, epi, start, end;
    for (int i = 0; i < n; ++i) {
        const unsigned long long* t = &st[i * 4];
        tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[3]);
    }
    for (int i = 0; i < n; ++i) {
        const unsigned long long* t = &st[i * 4];
        pro.push_back(double(t[1] - t[0])); loop.push_back(double(t[2] - t[1])); epi.push_back(double(t[3] - t[2]));
        start.push_back(double(t[0] - tmin)); end.push_back(double(t[3] - tmin));
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto p90 = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() * 9 / 10]; };
    printf("%dx%dx%d: %.1f us, %d workgroups; s_memtime ticks (100 MHz? see span): span %llu\n", M, N, K, ms * 1e3, nwg, tmax - tmin);
    printf("  prologue  median %.0f  p90 %.0f\n  k-loop    median %.0f  p90 %.0f\n  epilogue  median %.0f  p90 %.0f\n",
           med(pro), p90(pro), med(loop), p90(loop), med(epi), p90(epi));
    std::sort(start.begin(), start.end());
    printf("  start times (ticks from first): ");
    for (int q = 0; q <= 10; ++q) printf("%.0f ", start[std::min<size_t>(start.size() - 1, start.size() * q / 10)]);
    printf("\n");
    return 0;
}
