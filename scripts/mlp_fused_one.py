""" A few launches of the fused predictor MLP at one row count (default 76800 = the headline's full window at 256 sequences):
the target of the rocprofv3 --pmc passes behind profiles/mlp_pmc_summary.json.  Usage: mlp_fused_one.py [rows] [launches] """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
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
    for _ in range(n):
        y = K.mlp_fused(xp, w1, b1, w2, b2, residual=R)
torch.cuda.synchronize()
print("ok", float(y.abs().max()))
