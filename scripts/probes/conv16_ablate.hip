// Timing ablations of the f16x3 decoder conv, persistent form (results are NOT correct for ABLATE != 0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itextocvp_amd/csrc -DTOCVP_ABLATE=n [-DLAYOUT=7] \
//         -o scripts/probes/conv16_ablate_n scripts/probes/conv16_ablate.hip
#include "../../textocvp_amd/csrc/conv_f16x3.hip"
#include <stdio.h>
#ifndef LAYOUT
#define LAYOUT 7            // pass-major in / out, persistent kernel (bit 2)
#endif
#include <vector>

int main() {
    const int n = 2040, H = 64, W = 64;
    const size_t act = (size_t)n * H * W * 64;
    float *x, *y, *w, *b; void* wf;
    hipMalloc(&x, act * 4); hipMalloc(&y, act * 4); hipMalloc(&w, 64 * 64 * 25 * 4); hipMalloc(&b, 256);
    hipMalloc(&wf, tocvp_conv_weights_dec_f16x3_bytes());
    std::vector<float> hx(1 << 20), hw(64 * 64 * 25);
    unsigned s = 1;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / (1 << 24); }
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / (1 << 24) - 0.5f) * 0.1f; }
    for (size_t o = 0; o < act; o += hx.size())
        hipMemcpy(x + o, hx.data(), std::min(hx.size(), act - o) * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, 256);
    tocvp_split_conv_weights_dec_f16x3(w, wf, 64, 64, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) tocvp_conv5x5_dec_f16x3_f32(x, nullptr, 0, wf, b, y, n, H, W, 64, 64, 1, LAYOUT, nullptr);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) tocvp_conv5x5_dec_f16x3_f32(x, nullptr, 0, wf, b, y, n, H, W, 64, 64, 1, LAYOUT, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("LAYOUT=%d ABLATE=%d: %.3f ms per launch (%d slot images) -> %.1f TFLOP/s algorithmic\n", LAYOUT, TOCVP_ABLATE,
           ms / reps, n, n * 0.8388608 / (ms / reps));
    return 0;
}
