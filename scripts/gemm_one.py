import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K
M, N, Kd = 9600, 2048, 512
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
x = torch.randn(M, Kd, device="cuda"); w = torch.randn(N, Kd, device="cuda") * Kd ** -0.5
b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda")
for _ in range(5):
    K.linear(x, w, b, act=K.ACT_RELU, out=out, precision=mode)
torch.cuda.synchronize()
