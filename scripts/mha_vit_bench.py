import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K
B, H, T, dh = 256, 12, 257, 64
E = H * dh
qkv = torch.randn(B, T, 3 * E, device="cuda")
for tail in (False, True, False, True):
    K._MHA_TAIL_ROW = tail
    for _ in range(2): K.mha(qkv[..., :E], qkv[..., E:2*E], qkv[..., 2*E:], H, dh ** -0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.mha(qkv[..., :E], qkv[..., E:2*E], qkv[..., 2*E:], H, dh ** -0.5)
    e1.record(); torch.cuda.synchronize()
    print("tail row split", tail, e0.elapsed_time(e1) / 10 * 1e3, "us")
