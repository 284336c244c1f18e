"""Mid-size f16x3 GEMMs (B = 8 / 32 row counts): 128 x 128 against 64 x 64 tiles (TOCVP_GEMM_SMALL_BELOW), graph replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textocvp_amd import kernels as k
dev = torch.device("cuda", 0)
shapes = [(2400, 512, 512), (2400, 1536, 512), (2400, 2048, 512), (2400, 512, 2048),
          (9600, 512, 512), (9600, 1536, 512), (9600, 2048, 512), (9600, 512, 2048), (4800, 2048, 512), (4800, 512, 2048)]
def t(fn, n=100):
    for _ in range(5): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
print(f"TOCVP_GEMM_SMALL_BELOW={os.environ.get('TOCVP_GEMM_SMALL_BELOW', '192')}")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    us = t(lambda: k.linear(x, w, b, out=out, precision="f16x3"))
    print(f"  {M:5d} x {N:4d} x {K:4d}: {us:6.1f} us  ({2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s)", flush=True)
