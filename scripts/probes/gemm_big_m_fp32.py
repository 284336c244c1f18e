#!/usr/bin/env python
"""fp32-input split GEMM with more than 2^32 bytes of A (1.1 M rows x 1024): rows before / behind the 4 GB line against float64."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from textocvp_amd import kernels as K
M, N, Kd = 1_100_003, 512, 1024
g = torch.Generator().manual_seed(1)
x = torch.randn(M, Kd, generator=g).cuda()
w = (torch.randn(N, Kd, generator=g) / 32).cuda(); b = torch.randn(N, generator=g).cuda()
for prec in ("f16x3", "bf16x3", "fp32"):
    with K.gemm_precision(prec):
        y = K.linear(x, w, b, act=K.ACT_RELU)
    torch.cuda.synchronize()
    for name, rows in (("first", slice(0, 512)), ("around 2^32 bytes", slice(1048576 - 256, 1048576 + 256)), ("last", slice(M - 512, M))):
        ref = torch.relu(x[rows].double() @ w.double().t() + b.double())
        print(f"{prec} {name}: err {float((y[rows].double() - ref).abs().max()):.3e}")
