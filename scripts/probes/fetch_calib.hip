// Calibration of FETCH_SIZE / fetch granularity for the hybrid conv's staging pattern: every pass
// reads ONE 64-byte quarter of each pixel's 256-byte channel vector (4 lanes x 16 B), pixels 256 B apart.
//   mode 0: contiguous 16 B/lane stream over `payload` bytes          (reference)
//   mode 1: 64-byte quarter of every 256 B over 4 x payload bytes      (the conv's pattern)
//   mode 2: 128-byte half of every 256 B over 2 x payload bytes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void rd(const char* __restrict__ src, float* __restrict__ out, size_t nchunks) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
        size_t off;
        if (MODE == 0) off = i * 16;
        else if (MODE == 1) off = (i >> 2) * 256 + (i & 3) * 16;
        else off = (i >> 3) * 256 + (i & 7) * 16;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + off);
        acc += v;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) out[0] = acc[0];
}
int main() {
    const size_t payload = (size_t)1 << 30;              // 1 GiB actually used per launch
    char* buf; float* out;
    hipMalloc(&buf, payload * 4); hipMalloc(&out, 64);
    hipMemset(buf, 0, payload * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"contiguous stream", "64 B of every 256 B", "128 B of every 256 B"};
    for (int m = 0; m < 3; ++m)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            const size_t n = payload / 16;
            if (m == 0) hipLaunchKernelGGL(rd<0>, dim3(4096), dim3(256), 0, 0, buf, out, n);
            if (m == 1) hipLaunchKernelGGL(rd<1>, dim3(4096), dim3(256), 0, 0, buf, out, n);
            if (m == 2) hipLaunchKernelGGL(rd<2>, dim3(4096), dim3(256), 0, 0, buf, out, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-22s: %.3f ms for 1 GiB of payload -> %.2f TB/s payload\n", names[m], ms, payload / (ms * 1e-3) / 1e12);
        }
    return 0;
}
