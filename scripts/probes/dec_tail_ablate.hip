// Timing ablations of the decoder tail (results are NOT correct for TOCVP_DT_ABLATE != 0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itextocvp_amd/csrc -DTOCVP_DT_ABLATE=n \
//         -o scripts/probes/dec_tail_ablate_n scripts/probes/dec_tail_ablate.hip
#include "../../textocvp_amd/csrc/conv.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 68, K = 30, H = 64, W = 64;
    const size_t act = (size_t)F * K * H * W * 64;
    float *x, *w, *b, *ri, *rc, *mk, *ws;
    hipMalloc(&x, act * 4); hipMalloc(&w, 4 * 64 * 9 * 4); hipMalloc(&b, 256); hipMalloc(&ws, 9 * 64 * 4 * 4);
    hipMalloc(&ri, (size_t)F * 3 * H * W * 4); hipMalloc(&rc, (size_t)F * K * 3 * H * W * 4); hipMalloc(&mk, (size_t)F * K * H * W * 4);
    std::vector<float> hx(1 << 20), hw(4 * 64 * 9);
    unsigned s = 1;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / (1 << 24); }
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / (1 << 24) - 0.5f) * 0.1f; }
    for (size_t o = 0; o < act; o += hx.size())
        hipMemcpy(x + o, hx.data(), std::min(hx.size(), act - o) * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) tocvp_dec_tail_f32(x, w, b, ri, rc, mk, F, K, H, W, 64, ws, 9 * 64 * 4 * 4, nullptr);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) tocvp_dec_tail_f32(x, w, b, ri, rc, mk, F, K, H, W, 64, ws, 9 * 64 * 4 * 4, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("DT_ABLATE=%d: %.3f ms per launch (%d frames x %d slots) -> %.2f TB/s of input\n", TOCVP_DT_ABLATE, ms / reps, F, K,
           act * 4 / (ms / reps) / 1e9);
    return 0;
}
