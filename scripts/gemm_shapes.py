#!/usr/bin/env python
""" f16x3 GEMM timings on the rollout's shapes: fp32-A kernel (in-loop split), planes kernels (TOCVP_GEMM_P2=0/1) """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K
shapes = [(38400, 2048, 512), (38400, 2048, 1024), (38400, 2048, 2048), (38400, 512, 2048), (38400, 1536, 512),
          (38400, 512, 512), (19200, 2048, 512), (7680, 2048, 512), (7680, 512, 2048)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for M, N, Kd in shapes:
    x = torch.randn(M, Kd, device="cuda"); w = torch.randn(N, Kd, device="cuda") * Kd ** -0.5
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda")
    with K.gemm_precision("f16x3"):
        xf = x
        eye = torch.eye(Kd, device="cuda")
        xp = K.linear(xf, eye, out_split=22)              # the same values as fp16 operand planes
    res = {}
    for name, a in (("fp32-A", xf), ("planes", xp)):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            K.linear(a, w, b, act=K.ACT_RELU, out=out if name == "fp32-A" else None, precision="f16x3")
        s.record()
        for _ in range(10):
            y = K.linear(a, w, b, act=K.ACT_RELU, out=out if name == "fp32-A" else None, precision="f16x3")
        e.record(); torch.cuda.synchronize()
        res[name] = (s.elapsed_time(e) / 10, y.clone())
    err = (res["planes"][1] - res["fp32-A"][1]).abs().max().item()
    print(f"{M}x{N}x{Kd}: " + "  ".join(f"{k} {v[0]*1e3:7.1f} us {2*M*N*Kd/v[0]/1e9:6.1f} TF/s" for k, v in res.items())
          + f"  |diff| {err:.1e}", flush=True)
