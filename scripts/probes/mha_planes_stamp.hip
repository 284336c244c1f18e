// Where a workgroup of the plane-input attention kernel spends its cycles: s_memtime stamps of wave 0 at the phase boundaries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Iinclude -Itextocvp_amd/csrc -DTOCVP_MHAP_STAMP \
//         -o scripts/probes/mha_planes_stamp scripts/probes/mha_planes_stamp.hip ;  ./mha_planes_stamp [B H T]
#include "../../textocvp_amd/csrc/attn_planes.hip"
#include <stdio.h>
#include <algorithm>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 3 ? atoi(argv[1]) : 256, H = argc > 3 ? atoi(argv[2]) : 8, T = argc > 3 ? atoi(argv[3]) : 300;
    const int E = H * 64, ld = 3 * E;
    const size_t n = (size_t)B * T * 2 * ld;
    _Float16* qkv; float* o;
    hipMalloc(&qkv, n * 2); hipMalloc(&o, (size_t)B * T * E * 4);
    std::vector<_Float16> h((size_t)1 << 22);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((float)(s >> 8) / (1 << 24) - 0.5f) * 512.f); }
    for (size_t off = 0; off < n; off += h.size()) hipMemcpy(qkv + off, h.data(), std::min(h.size(), n - off) * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&]() { return tocvp_mha_planes_f16(qkv, ld, qkv + E, ld, qkv + 2 * E, ld, o, E, nullptr, B, H, T, T, T, 64, 0.125f, nullptr, nullptr); };
    for (int i = 0; i < 5; ++i) if (run()) { printf("launch failed\n"); return 1; }
    hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(16384 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(tocvp_mhap_stamps), st.size() * 8);
    const int nwg = std::min(16384, B * H * ((T + 127) / 128));
    const char* names[6] = {"wait vmcnt (DMA landed)", "barrier", "DMA issue", "QK^T (12 MFMA + K frags)", "softmax", "P split + PV (12 MFMA)"};
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const int nt = (T + 31) / 32;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < nwg; ++i) { tmin = std::min(tmin, st[i * 8 + 6]); tmax = std::max(tmax, st[i * 8 + 7]); }
    printf("B %d H %d T %d: %.1f us, %d workgroups, %d tiles each; ticks from first start to last end %llu\n", B, H, T, ms * 1e3, nwg, nt, tmax - tmin);
    double tot = 0;
    for (int ph = 0; ph < 6; ++ph) {
        std::vector<double> v;
        for (int i = 0; i < nwg; ++i) v.push_back((double)st[i * 8 + ph] / nt);
        printf("  %-28s median %7.0f ticks per tile\n", names[ph], med(v));
        tot += med(v);
    }
    std::vector<double> life;
    for (int i = 0; i < nwg; ++i) life.push_back((double)(st[i * 8 + 7] - st[i * 8 + 6]));
    printf("  sum of phases %.0f per tile; workgroup loop lifetime median %.0f ticks (%d tiles)\n", tot, med(life), nt);
    return 0;
}
