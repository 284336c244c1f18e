#!/usr/bin/env python
"""One shape of the mid-size chunk GEMM (tocvp_gemm_f16mid_f32) for the PMC passes of scripts/pmc_collect.sh:
    scripts/pmc_collect.sh <out_dir> gemm_f16x3_mid -- python3 scripts/gemm_mid_one.py 9600 2048 512 8"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K
M, N, Kd = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
x = torch.relu(torch.randn(M, Kd, generator=g)).to(dev)
v = torch.clamp(x * 256.0, -65504.0, 65504.0); hi = v.to(torch.float16)
xp = K.SplitAct(torch.stack([hi, (v - hi.float()).to(torch.float16)], dim=1).contiguous(), (M, Kd))
w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
K._GEMM_CHUNK, K._GEMM_MID, K._GEMM_MID_MIN_TILES = False, True, 1
with K.gemm_precision("f16x3"):
    for _ in range(reps):
        y = K.linear(xp, w, b, act=K.ACT_RELU, out_split=22)
torch.cuda.synchronize()
print("done")
