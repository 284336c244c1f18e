#!/usr/bin/env python
""" A few launches of the training step's two dedicated kernels at the C2 shapes: target of rocprofv3 --pmc runs.
    train_kernels_one.py tn   [M N K splits]      weight gradient  (default 9600 2048 512 8)
    train_kernels_one.py attn [B H T]             fused attention backward (default 32 8 300)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K                                                # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "tn"
s = torch.cuda.current_stream().cuda_stream
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
if what == "tn":
    M, N, Kd, splits = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (9600, 2048, 512, 8)
    g, x = torch.randn(M, N, device="cuda") * 1e-3, torch.randn(M, Kd, device="cuda")
    part, bias = torch.empty(splits, N * Kd, device="cuda"), torch.empty(splits, N, device="cuda")

    def run(acc):
        K._check(K.lib().tocvp_gemm_tn_f32(g.data_ptr(), N, x.data_ptr(), Kd, part.data_ptr(), bias.data_ptr(), M, N, Kd,
                                           splits, acc, s), "tn")
    run(0)
    ev[0].record()
    for _ in range(10):
        run(1)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 100
    print(f"gemm_tn {M}x{N}x{Kd} splits {splits}: {us:.1f} us  {2.0 * M * N * Kd / us / 1e6:.1f} TFLOP/s (fp32 MFMA peak 157.3)")
else:
    B, H, T = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (32, 8, 300)
    E = H * 64
    q, k, v, o, do = (torch.randn(B, T, E, device="cuda") for _ in range(5))
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    stats = torch.empty(B, H, T, 2, device="cuda")

    def run():
        K._check(K.lib().tocvp_attn_bwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(),
                                            dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), stats.data_ptr(), None, B, H, T, T, E,
                                            0.125, s), "attn_bwd")
    run()
    ev[0].record()
    for _ in range(10):
        run()
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 100
    tiles = ((T + 31) // 32) ** 2
    print(f"attn_bwd B={B} H={H} T={T}: {us:.1f} us per attention; {B * H * tiles * 256} MFMAs of 64 cycles "
          f"= {B * H * tiles * 256 * 64 / 1024 / 2.1e3:.1f} us of matrix time at 2.1 GHz")
