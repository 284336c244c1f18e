#!/usr/bin/env python
"""
GEMM timings (HIP events around back-to-back launches of tocvp_gemm_* through kernels.linear), one script for
every question the GEMM work asked:

  gemm_shapes.py                               rollout shapes, f16x3: fp32-A kernel vs operand-planes kernels
  gemm_shapes.py modes [MODE ...]              per arithmetic mode (fp32 bf16x3 bf16x6 f16x3 ...) on the step's shapes
  gemm_shapes.py ksweep                        fixed (M, N), K = 64 .. 2048: slope = k-loop rate, intercept = fixed cost
  gemm_shapes.py presplit                      bf16x6 with A split in the loop vs split by the producing LayerNorm
  gemm_shapes.py one MODE M N K [presplit]     one shape, a few launches: the target of rocprofv3 --pmc runs
  gemm_shapes.py MxNxK [MxNxK ...]             the default comparison on the given shapes
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K                                                # noqa: E402

ROLLOUT = [(38400, 2048, 512), (38400, 2048, 1024), (38400, 2048, 2048), (38400, 512, 2048), (38400, 1536, 512),
           (38400, 512, 512), (19200, 2048, 512), (7680, 2048, 512), (7680, 512, 2048)]
STEP = [(960, 512, 512), (4800, 512, 512), (9600, 512, 512), (9600, 1536, 512), (9600, 2048, 512),
        (9600, 512, 2048), (4800, 2048, 512), (960, 2048, 512), (300, 512, 512), (38400, 2048, 512)]


def bench(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps                     # microseconds


def operands(M, N, Kd):
    x = torch.randn(M, Kd, device="cuda")
    w = torch.randn(N, Kd, device="cuda") * Kd ** -0.5
    return x, w, torch.randn(N, device="cuda"), torch.empty(M, N, device="cuda")


def tf(M, N, Kd, us):
    return 2.0 * M * N * Kd / us / 1e6


def planes_vs_fp32(shapes):
    for M, N, Kd in shapes:
        x, w, b, out = operands(M, N, Kd)
        with K.gemm_precision("f16x3"):
            xp = K.linear(x, torch.eye(Kd, device="cuda"), out_split=22)       # the same values as fp16 operand planes
        res = {}
        for name, a in (("fp32-A", x), ("planes", xp)):
            o = out if name == "fp32-A" else None
            us = bench(lambda: K.linear(a, w, b, act=K.ACT_RELU, out=o, precision="f16x3"), reps=10)
            res[name] = (us, K.linear(a, w, b, act=K.ACT_RELU, precision="f16x3").clone())
        err = (res["planes"][1] - res["fp32-A"][1]).abs().max().item()
        print(f"{M}x{N}x{Kd}: " + "  ".join(f"{k} {v[0]:7.1f} us {tf(M, N, Kd, v[0]):6.1f} TF/s" for k, v in res.items())
              + f"  |diff| {err:.1e}", flush=True)


def per_mode(modes):
    for M, N, Kd in STEP:
        x, w, b, out = operands(M, N, Kd)
        line = f"M={M:6d} N={N:5d} K={Kd:5d}: "
        for mode in modes:
            us = bench(lambda: K.linear(x, w, b, act=K.ACT_RELU, out=out, precision=mode))
            line += f"{mode} {us:8.1f} us {tf(M, N, Kd, us):7.1f} TF | "
        print(line, flush=True)


def ksweep():
    for M, N in [(9600, 512), (9600, 2048), (38400, 2048)]:
        for Kd in (64, 128, 256, 512, 1024, 2048):
            x, w, b, out = operands(M, N, Kd)
            t = {m: bench(lambda: K.linear(x, w, b, act=K.ACT_RELU, out=out, precision=m), reps=50, warm=5)
                 for m in ("f16x3", "bf16x6", "bf16x3", "fp32")}
            print(f"M={M} N={N} K={Kd:5d}: " + "  ".join(f"{m} {us:7.1f} us" for m, us in t.items()), flush=True)


def presplit():
    for M, N, Kd in [(9600, 512, 512), (9600, 1536, 512), (9600, 2048, 512), (9600, 512, 2048), (4800, 2048, 512)]:
        x, w, b, out = operands(M, N, Kd)
        g, z = torch.ones(Kd, device="cuda"), torch.zeros(Kd, device="cuda")
        xs = K.layer_norm(x, g, z, 1e-6, split=3)
        t_fly = bench(lambda: K.linear(x, w, b, act=K.ACT_RELU, out=out, precision="bf16x6"))
        t_pre = bench(lambda: K.linear(xs, w, b, act=K.ACT_RELU, out=out))
        t_pre_so = bench(lambda: K.linear(xs, w, b, act=K.ACT_RELU, out_split=3))
        t_ln, t_ln32 = bench(lambda: K.layer_norm(x, g, z, 1e-6, split=3)), bench(lambda: K.layer_norm(x, g, z, 1e-6))
        print(f"M={M} N={N} K={Kd}: on-the-fly {t_fly:7.1f} us {tf(M, N, Kd, t_fly):6.1f} TF | pre-split A {t_pre:7.1f} us "
              f"{tf(M, N, Kd, t_pre):6.1f} TF | pre-split A + split C {t_pre_so:7.1f} us | LN split {t_ln:6.1f} us "
              f"(fp32 {t_ln32:6.1f})", flush=True)


def one(mode="f16x3", M=38400, N=2048, Kd=512, pre=False):
    x, w, b, out = operands(M, N, Kd)
    if pre:
        with K.gemm_precision(mode):
            x = K.layer_norm(x, torch.ones(Kd, device="cuda"), torch.zeros(Kd, device="cuda"), 1e-6,
                             split=K.active_nsplit())
    us = bench(lambda: K.linear(x, w, b, act=K.ACT_RELU, out=out, precision=mode), reps=10)
    print(f"{mode} {M}x{N}x{Kd} presplit={pre}: {us:.1f} us  {tf(M, N, Kd, us):.1f} TF/s")


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "modes":
        per_mode(a[1:] or ["fp32", "bf16x3", "bf16x6", "f16x3"])
    elif a and a[0] == "ksweep":
        ksweep()
    elif a and a[0] == "presplit":
        presplit()
    elif a and a[0] == "one":
        one(*(a[1:2] or ["f16x3"]), *(int(v) for v in a[2:5]), *([a[5] == "presplit"] if len(a) > 5 else []))
    else:
        planes_vs_fp32([tuple(int(v) for v in s.split("x")) for s in a] or ROLLOUT)
