""" one GEMM shape, a few launches: target of rocprofv3 --pmc runs.  gemm_one.py MODE M N K [presplit] """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K
mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
M, N, Kd = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (38400, 2048, 512)
presplit = len(sys.argv) > 5 and sys.argv[5] == "presplit"
x = torch.randn(M, Kd, device="cuda"); w = torch.randn(N, Kd, device="cuda") * Kd ** -0.5
b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda")
if presplit:
    g, be = torch.ones(Kd, device="cuda"), torch.zeros(Kd, device="cuda")
    with K.gemm_precision(mode):
        x = K.layer_norm(x, g, be, 1e-6, split=K.active_nsplit())
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    K.linear(x, w, b, act=K.ACT_RELU, out=out, precision=mode)
s.record()
for _ in range(10):
    K.linear(x, w, b, act=K.ACT_RELU, out=out, precision=mode)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f"{mode} {M}x{N}x{Kd} presplit={presplit}: {ms*1e3:.1f} us  {2*M*N*Kd/ms/1e9:.1f} TF/s")
