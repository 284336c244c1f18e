// Timing ablations of the fragment-order f16x3 GEMM (results are NOT correct for ABLATE != 0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itextocvp_amd/csrc -DTOCVP_GEMM_ABLATE=n \
//         -o scripts/probes/gemm_ablate_n scripts/probes/gemm_ablate.hip
#include "../../textocvp_amd/csrc/gemm_bf16.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 38400, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 512;
    const int planes = argc > 4 ? atoi(argv[4]) : 0;      // 1: A given as fp16 operand planes (DMA kernel)
    float *A, *W, *C, *bias; void* Wf;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&bias, N * 4); hipMalloc(&Wf, (size_t)N * K * 4);
    std::vector<float> h((size_t)1 << 22);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / (1 << 24) - 0.5f) * 2.f; }
    for (size_t o = 0; o < (size_t)M * K; o += h.size())
        hipMemcpy(A + o, h.data(), std::min(h.size(), (size_t)M * K - o) * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, N * 4);
    tocvp_split_weights_frag_f16(W, Wf, N, K, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        tocvp_gemm_bf16wfrag_f32(A, planes, K, Wf, 22, bias, nullptr, 0, nullptr, 1, 1, 0, C, 0, N, M, N, K, 1, nullptr);
    hipEventRecord(e0);
    const int reps = argc > 5 ? atoi(argv[5]) : 10;
    for (int i = 0; i < reps; ++i)
        tocvp_gemm_bf16wfrag_f32(A, planes, K, Wf, 22, bias, nullptr, 0, nullptr, 1, 1, 0, C, 0, N, M, N, K, 1, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("planes=%d GEMM_ABLATE=%d %dx%dx%d: %.1f us -> %.1f TFLOP/s algorithmic\n", planes, TOCVP_GEMM_ABLATE, M, N, K,
           ms / reps * 1e3, 2.0 * M * N * K / (ms / reps) / 1e9);
    return 0;
}
