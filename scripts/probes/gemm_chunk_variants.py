#!/usr/bin/env python
"""
Timing ablations and in-kernel stamps of the chunk-resident f16x3 GEMM (csrc/gemm_f16c.hip).  The variants are built by
scripts/probes/build_gemm_chunk_variants.sh (-DTOCVP_GC_ABLATE=n -DTOCVP_GC_STAMP): 0 full, 1 no A DMA in the loop,
2 no weight loads in the loop, 4 no MFMAs, 5 no epilogue (1 / 2 / 4 / 5 give wrong results: timing only).
    python scripts/probes/gemm_chunk_variants.py [M N K] [reps]
"""
import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from textocvp_amd import kernels as K

M, N, Kd = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (98304, 1024, 1024)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
x = torch.relu(torch.randn(M, Kd, generator=g)).to(dev)
v = torch.clamp(x * 256.0, -65504.0, 65504.0)
hi = v.to(torch.float16)
planes = torch.stack([hi, (v - hi.float()).to(torch.float16)], dim=1).contiguous()
del x, v, hi
w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev)
b = torch.randn(N, generator=g).to(dev)
wf = K._split_weight(w, 22, frag=True)
out = torch.empty(M, 2, N, device=dev, dtype=torch.float16)
stream = torch.cuda.current_stream().cuda_stream
tiles = ((M + 127) // 128) * (N // 512)

for var in (0, 1, 2, 4, 5, 6, 7):
    path = os.path.join(HERE, f"gemm_chunk_v{var}.so")
    if not os.path.exists(path):
        continue
    lib = ctypes.CDLL(path)
    fn = lib.tocvp_gemm_f16chunk_f32
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]

    def launch():
        rc = fn(planes.data_ptr(), wf.data_ptr(), b.data_ptr(), None, N, out.data_ptr(), 1, N, M, N, Kd, K.ACT_RELU, stream)
        assert rc == 0, rc
    launch()
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    n = 256
    buf = (ctypes.c_ulonglong * (4 * n))()
    lib.tocvp_gc_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.tocvp_gc_read_stamps(buf, n) == 0
    st = torch.tensor(list(buf), dtype=torch.float64).reshape(n, 4)
    st = st[st[:, 3] > 0]
    per = tiles / max(st.shape[0], 1)                # tiles per (persistent) workgroup
    pro, loop, epi, tot = st[:, 0], st[:, 1], st[:, 2], st[:, 3]
    print(f"variant {var}: {best:.0f} us ({2.0 * M * N * Kd / best / 1e6:.0f} TF/s) | {st.shape[0]} workgroups x {per:.2f} tiles; "
          f"cycles per workgroup (mean): prologue {pro.mean():.0f}, k-loops {loop.mean():.0f} ({loop.mean() / per:.0f} per tile), "
          f"epilogues {epi.mean():.0f} ({epi.mean() / per:.0f} per tile), life {tot.mean():.0f} (max {tot.max():.0f}) "
          f"= {tot.max() / best / 1e3:.2f} GHz-equivalent", flush=True)
