#!/usr/bin/env python
""" GEMM micro-benchmark: algorithmic TFLOP/s of tocvp_gemm_* per shape and arithmetic mode. """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K

shapes = [(960, 512, 512), (4800, 512, 512), (9600, 512, 512), (9600, 1536, 512), (9600, 2048, 512),
          (9600, 512, 2048), (4800, 2048, 512), (960, 2048, 512), (300, 512, 512), (38400, 2048, 512)]
modes = sys.argv[1:] or ["fp32", "bf16x3", "bf16x6"]
for (M, N, Kd) in shapes:
    x = torch.randn(M, Kd, device="cuda"); w = torch.randn(N, Kd, device="cuda") * Kd ** -0.5
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda")
    line = f"M={M:6d} N={N:5d} K={Kd:5d}: "
    for mode in modes:
        for _ in range(3):
            K.linear(x, w, b, act=K.ACT_RELU, out=out, precision=mode)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        s.record()
        for _ in range(reps):
            K.linear(x, w, b, act=K.ACT_RELU, out=out, precision=mode)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / reps
        line += f"{mode} {us:8.1f} us {2.0 * M * N * Kd / us / 1e6:7.1f} TF | "
    print(line, flush=True)
