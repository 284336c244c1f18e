#!/usr/bin/env python
"""
One shape of the chunk-resident GEMM (tocvp_gemm_f16chunk_f32) for the PMC passes of scripts/pmc_collect.sh:
    scripts/pmc_collect.sh <out_dir> gemm_f16x3_chunk -- python3 scripts/gemm_chunk_one.py 98304 1024 1024 6
ReLU epilogue, plane output (the MLPPatchDecoder's hidden layers); post-ReLU random activations.
"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K

M, N, Kd = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
x = torch.relu(torch.randn(M, Kd, generator=g)).to(dev)
v = torch.clamp(x * 256.0, -65504.0, 65504.0)
hi = v.to(torch.float16)
xp = K.SplitAct(torch.stack([hi, (v - hi.float()).to(torch.float16)], dim=1).contiguous(), (M, Kd))
del x, v, hi
w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev)
b = torch.randn(N, generator=g).to(dev)
K._GEMM_CHUNK, K._GEMM_CHUNK_MIN_TILES = True, 1
with K.gemm_precision("f16x3"):
    for _ in range(reps):
        y = K.linear(xp, w, b, act=K.ACT_RELU, out_split=22)
torch.cuda.synchronize()
print("done", tuple(y.planes.shape))
