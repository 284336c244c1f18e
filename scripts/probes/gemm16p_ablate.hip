// Timing ablations of the all-DMA planes GEMM (results are NOT correct for ABLATE != 0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itextocvp_amd/csrc -DTOCVP_GEMM_P2_ABLATE=n \
//         -o scripts/probes/gemm16p_ablate_n scripts/probes/gemm16p_ablate.hip ;  ./gemm16p_ablate_n [M N K]
#include "gemm_f16p.hip"
#include <stdio.h>
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
    for (int i = 0; i < 3; ++i) tocvp_gemm_f16planes_f32(a, w, b, nullptr, 0, c, 0, N, M, N, K, 1, nullptr);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) tocvp_gemm_f16planes_f32(a, w, b, nullptr, 0, c, 0, N, M, N, K, 1, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("ABLATE=%d %dx%dx%d: %.1f us per launch -> %.1f TFLOP/s algorithmic\n", TOCVP_GEMM_P2_ABLATE, M, N, K,
           ms / reps * 1e3, 2.0 * M * N * K / (ms / reps) / 1e9);
    return 0;
}
