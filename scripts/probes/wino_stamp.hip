// Where a workgroup of the Winograd decoder conv spends its cycles: s_memtime stamps of waves 0 and 3 at the phase boundaries
// (-DTOCVP_WINO_STAMP), or the launch time with one part of the kernel taken out (-DTOCVP_WINO_ABLATE=1..4, no stamps).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Itextocvp_amd/csrc -DTOCVP_WINO_STAMP \
//         -o scripts/probes/wino_stamp scripts/probes/wino_stamp.hip ;  ./wino_stamp [nimg]
#include "../../textocvp_amd/csrc/conv_wino.hip"
#include <stdio.h>
#include <algorithm>
#include <vector>

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2040, H = 64, W = 64;
    const size_t ne = (size_t)n * H * W * 64;
    float *x, *y, *w, *b; void* wf;
    hipMalloc(&x, ne * 4); hipMalloc(&y, ne * 4); hipMalloc(&w, 64 * 64 * 25 * 4); hipMalloc(&b, 256);
    hipMalloc(&wf, tocvp_conv_weights_wino_f16x3_bytes());
    std::vector<float> hx((size_t)1 << 22), hw(64 * 64 * 25), hb(64, 0.05f);
    unsigned s = 1;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = std::max(0.f, ((float)(s >> 8) / (1 << 24) - 0.4f) * 32.f); }
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / (1 << 24) - 0.5f) * 0.05f; }
    for (size_t off = 0; off < ne; off += hx.size()) hipMemcpy(x + off, hx.data(), std::min(hx.size(), ne - off) * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    float scales[8], coef[32];
    for (int i = 0; i < 8; ++i) scales[i] = 65536.f;
    for (int i = 0; i < 32; ++i) coef[i] = 1.f / (16.f * 65536.f);
    if (tocvp_split_conv_weights_wino_f16x3(w, wf, scales, nullptr, 64, 64, nullptr)) { printf("split failed\n"); return 1; }
    auto run = [&]() { return tocvp_conv5x5_dec_wino_f16x3_f32(x, nullptr, 0, wf, coef, b, nullptr, y, n, H, W, 1, 1, nullptr, nullptr, nullptr); };
    for (int i = 0; i < 3; ++i) if (run()) { printf("launch failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%d slot images: %.3f ms per launch (TOCVP_WINO_ABLATE %d)\n", n, ms / 5, TOCVP_WINO_ABLATE);
#ifdef TOCVP_WINO_STAMP
    const int nwg = std::min(16384, ((n + 7) / 8) * 8 * 16);
    std::vector<unsigned long long> st((size_t)16384 * 2 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(tocvp_wino_stamps), st.size() * 8);
    const char* names[8] = {"transform (4 passes)", "multiply (4 passes)", "pass barriers", "exchange write", "exchange barrier 1",
                            "read + combine", "exchange barrier 2", "epilogue + stores"};
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%d slot images: %.3f ms per launch, %d workgroups\n", n, ms / 5, nwg);
    for (int half = 0; half < 2; ++half) {
        double tot = 0;
        printf(" wave %d\n", half * 3);
        for (int ph = 0; ph < 8; ++ph) {
            std::vector<double> v;
            for (int i = 0; i < nwg; ++i) v.push_back((double)st[((size_t)i * 2 + half) * 16 + ph]);
            printf("   %-44s median %8.0f ticks\n", names[ph], med(v));
            tot += med(v);
        }
        std::vector<double> life;
        for (int i = 0; i < nwg; ++i) life.push_back((double)(st[((size_t)i * 2 + half) * 16 + 15] - st[((size_t)i * 2 + half) * 16 + 14]));
        printf("   sum %.0f; lifetime median %.0f ticks (100 MHz: x 10 ns)\n", tot, med(life));
    }
#endif
    return 0;
}
