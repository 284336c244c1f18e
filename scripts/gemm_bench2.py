#!/usr/bin/env python
""" GEMM micro-benchmark with pre-split A (SplitAct) vs on-the-fly split, x6. """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K
shapes = [(9600, 512, 512), (9600, 1536, 512), (9600, 2048, 512), (9600, 512, 2048), (4800, 2048, 512)]
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for (M, N, Kd) in shapes:
    x = torch.randn(M, Kd, device="cuda"); w = torch.randn(N, Kd, device="cuda") * Kd ** -0.5
    b = torch.randn(N, device="cuda"); g = torch.ones(Kd, device="cuda"); z = torch.zeros(Kd, device="cuda")
    out = torch.empty(M, N, device="cuda")
    xs = K.layer_norm(x, g, z, 1e-6, split=3)
    t_fly = bench(lambda: K.linear(x, w, b, act=K.ACT_RELU, out=out, precision="bf16x6"))
    t_pre = bench(lambda: K.linear(xs, w, b, act=K.ACT_RELU, out=out))
    t_pre_so = bench(lambda: K.linear(xs, w, b, act=K.ACT_RELU, out_split=3))
    t_ln = bench(lambda: K.layer_norm(x, g, z, 1e-6, split=3))
    t_ln32 = bench(lambda: K.layer_norm(x, g, z, 1e-6))
    f = 2.0 * M * N * Kd / 1e6
    print(f"M={M} N={N} K={Kd}: on-the-fly {t_fly:7.1f} us {f/t_fly:6.1f} TF | pre-split A {t_pre:7.1f} us {f/t_pre:6.1f} TF | "
          f"pre-split A + split C {t_pre_so:7.1f} us {f/t_pre_so:6.1f} TF | LN split {t_ln:6.1f} us (fp32 {t_ln32:6.1f})", flush=True)
