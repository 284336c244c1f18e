""" Predictor / ViT self-attention in isolation (the `rooflines[2]` shape of bench.py and the ViT's), optionally through another
build of the library (MHA_LIB=path: same-box A/B of two kernel versions; prints the largest difference to a torch fp64 softmax). """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import build as _build
if os.environ.get("MHA_LIB"):
    _build.LIB_PATH = os.path.abspath(os.environ["MHA_LIB"])
from textocvp_amd import kernels as K

for B, H, T, dh in ((256, 8, 300, 64), (128, 8, 300, 64), (256, 12, 257, 64), (8, 8, 300, 64)):
    E = H * dh
    torch.manual_seed(0)
    qkv = torch.randn(B, T, 3 * E, device="cuda")
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    for _ in range(3): o = K.mha(q, k, v, H, dh ** -0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): K.mha(q, k, v, H, dh ** -0.5)
    e1.record(); torch.cuda.synchronize()
    nb = min(B, 8)
    sp = lambda x: x[:nb].double().view(nb, T, H, dh).transpose(1, 2)
    ref = (torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * dh ** -0.5, -1) @ sp(v)).transpose(1, 2).reshape(nb, T, E)
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"B {B} H {H} T {T} dh {dh}: {us:8.1f} us  {4.0 * B * H * T * T * dh / us * 1e-6:7.1f} TFLOP/s  max err {float((o[:nb].double() - ref).abs().max()):.3e}  bits {int(o.view(torch.int32).sum(dtype=torch.int64)):x}", flush=True)
