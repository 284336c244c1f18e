""" one shape of the plane-input / fp32-input attention for PMC collection: python scripts/mha_one_planes.py planes|fp32 [B H T] """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K
which = sys.argv[1]
B, H, T = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (256, 8, 300)
E = H * 64
torch.manual_seed(0)
qkv = torch.randn(B, T, 3 * E, device="cuda")
X = torch.clamp(qkv.reshape(B * T, 3 * E) * 256.0, -65504.0, 65504.0)
hi = X.to(torch.float16)
sp = K.SplitAct(torch.stack([hi, (X - hi.float()).to(torch.float16)], dim=1).contiguous(), (B, T, 3 * E))
with K.gemm_precision("f16x3"):
    for _ in range(5):
        if which == "planes":
            K.mha_planes(sp, 0, sp, E, sp, 2 * E, B, T, T, H, 0.125, out_split=22)
        else:
            K.mha(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], H, 0.125, out_split=22)
torch.cuda.synchronize()
