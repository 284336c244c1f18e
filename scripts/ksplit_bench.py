"""Skinny f16x3 GEMMs: unsplit vs split-K (HIP events over 200 back-to-back launches per shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textocvp_amd import kernels as k
dev = torch.device("cuda", 0)
shapes = [(300, 512, 2048), (30, 512, 2048), (150, 512, 2048), (300, 2048, 512), (300, 512, 512), (300, 1536, 512),
          (30, 512, 512), (30, 2048, 512), (12, 512, 64), (30, 128, 128), (30, 384, 128), (2400, 512, 2048), (2400, 2048, 512),
          (2400, 512, 512), (2400, 1536, 512), (4800, 512, 2048), (4800, 512, 512)]
def t(fn, n=200):
    for _ in range(10): fn()
    if os.environ.get("KSPLIT_BENCH_GRAPH", "1") == "1":          # replay from a HIP graph: no host launch cost
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n): fn()
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        return 1e3 * a.elapsed_time(b) / n
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
print(f"TOCVP_GEMM_KSPLIT_WGS={os.environ.get('TOCVP_GEMM_KSPLIT_WGS', '512')}")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    k._GEMM_KSPLIT = False
    t0 = t(lambda: k.linear(x, w, b, out=out, precision="f16x3"))
    k._GEMM_KSPLIT = True
    t1 = t(lambda: k.linear(x, w, b, out=out, precision="f16x3"))
    print(f"  {M:5d} x {N:4d} x {K:4d}: unsplit {t0:6.1f} us   split-K {t1:6.1f} us", flush=True)
