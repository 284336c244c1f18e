import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K
def bench(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for (M, N) in [(9600, 512), (9600, 2048), (38400, 2048)]:
    for Kd in (64, 128, 256, 512, 1024, 2048):
        x = torch.randn(M, Kd, device="cuda"); w = torch.randn(N, Kd, device="cuda") * Kd ** -0.5
        b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda")
        t6 = bench(lambda: K.linear(x, w, b, act=K.ACT_RELU, out=out, precision="bf16x6"))
        t3 = bench(lambda: K.linear(x, w, b, act=K.ACT_RELU, out=out, precision="bf16x3"))
        t32 = bench(lambda: K.linear(x, w, b, act=K.ACT_RELU, out=out, precision="fp32"))
        print(f"M={M} N={N} K={Kd:5d}: x6 {t6:7.1f} us  x3 {t3:7.1f} us  fp32 {t32:7.1f} us", flush=True)
