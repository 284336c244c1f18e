""" A few launches of the fused predictor MLP at one row count (default 76800 = the headline's full window at 256 sequences):
the target of the rocprofv3 --pmc passes behind profiles/mlp_pmc_summary.json; prints the time per launch and the largest
difference to the two-GEMM path (0 = bit-identical; TOCVP_MLP_ZIGZAG=1 changes the accumulation order of odd hidden chunks).
Usage: mlp_fused_one.py [rows] [launches] """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import build as _build
if os.environ.get("MHA_LIB"):                      # another build of the library (same-box A/B)
    _build.LIB_PATH = os.path.abspath(os.environ["MHA_LIB"])
from textocvp_amd import kernels as K
M = int(sys.argv[1]) if len(sys.argv) > 1 else 76800
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0); g = torch.Generator().manual_seed(3)
x = torch.randn(M, 512, generator=g).to(dev)
v = torch.clamp(x * 256.0, -65504.0, 65504.0); hi = v.half()
xp = K.SplitAct(torch.stack([hi, (v - hi.float()).half()], dim=1).contiguous(), (M, 512))
w1 = (torch.randn(2048, 512, generator=g) / 512 ** 0.5).to(dev); b1 = torch.randn(2048, generator=g).to(dev)
w2 = (torch.randn(512, 2048, generator=g) / 2048 ** 0.5).to(dev); b2 = torch.randn(512, generator=g).to(dev)
R = torch.randn(M, 512, generator=g).to(dev)
with K.gemm_precision("f16x3"):
    y = K.mlp_fused(xp, w1, b1, w2, b2, residual=R)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = K.mlp_fused(xp, w1, b1, w2, b2, residual=R)
    e1.record(); torch.cuda.synchronize()
    h = K.linear(xp, w1, b1, act=K.ACT_RELU, out_split=22)
    y2 = K.linear(h, w2, b2, residual=R)
ref = (torch.relu(x[:512].double() @ w1.double().T + b1.double()) @ w2.double().T + b2.double() + R[:512].double())
print(f"rows {M}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch; max |fused - two GEMMs| {float((y - y2).abs().max()):.3e}; "
      f"max err vs fp64 fused {float((y[:512].double() - ref).abs().max()):.3e} two GEMMs {float((y2[:512].double() - ref).abs().max()):.3e}")
