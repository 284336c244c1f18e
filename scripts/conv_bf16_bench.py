"""Data-gradient conv of the training step (bf16x3, weights direct): ms per 2040 slot images."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textocvp_amd import kernels as k
dev = torch.device("cuda", 0)
n = 2040
x = torch.rand(n, 64, 64, 64, device=dev) * (torch.rand(n, 64, 64, 64, device=dev) > 0.5)
w = (torch.rand(64, 64, 5, 5, device=dev) - 0.5) * 0.1
b = torch.zeros(64, device=dev)
ws, wf = k.split_conv_weights_bf16(w), k.split_conv_weights_frag_bf16(w)
out = torch.empty_like(x)
for _ in range(3): k.conv5x5_bf16x3(x, ws, b, relu=True, wfrag=wf, out=out)
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(10): k.conv5x5_bf16x3(x, ws, b, relu=True, wfrag=wf, out=out)
e.record(); torch.cuda.synchronize()
print(f"conv5x5_bf16x3 (weights direct): {a.elapsed_time(e) / 10:.3f} ms per {n} slot images")
