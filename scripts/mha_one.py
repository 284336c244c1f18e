""" A few launches of the predictor's self-attention at the headline shape (256 x 8 x 300 x 300 x 64): target of scripts/pmc_collect.sh """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K
B, H, T, dh = 256, 8, 300, 64
E = H * dh
qkv = torch.randn(B, T, 3 * E, device="cuda")
for _ in range(6):
    o = K.mha(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], H, dh ** -0.5)
torch.cuda.synchronize()
print("ok", float(o.abs().max()))
