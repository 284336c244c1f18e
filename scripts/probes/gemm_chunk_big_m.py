import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from textocvp_amd import kernels as K
M, N, Kd = 1_100_003, 512, 1024
g = torch.Generator().manual_seed(1)
x = torch.randn(M, Kd, generator=g).cuda()
w = (torch.randn(N, Kd, generator=g) / 32).cuda(); b = torch.randn(N, generator=g).cuda()
v = torch.clamp(x * 256.0, -65504.0, 65504.0); hi = v.half()
xp = K.SplitAct(torch.stack([hi, (v - hi.float()).half()], dim=1).contiguous(), (M, Kd)); del x, v, hi
with K.gemm_precision("f16x3"):
    K._GEMM_CHUNK = True
    a = K.linear(xp, w, b, act=K.ACT_RELU)
    K._GEMM_CHUNK = False; K._GEMM_MID = False
    c = K.linear(xp, w, b, act=K.ACT_RELU)
torch.cuda.synchronize()
print("rows", M, "equal", torch.equal(a, c), float((a - c).abs().max()))
a64 = xp.planes.double().sum(dim=1) / 256.0
for name, rows in (("first", slice(0, 512)), ("around the row-block boundary", slice(1048448 - 256, 1048448 + 256)), ("last", slice(M - 512, M))):
    ref = torch.relu(a64[rows] @ w.double().t() + b.double())
    print(f"{name}: chunk path err {float((a[rows].double() - ref).abs().max()):.3e}, two-operand path err {float((c[rows].double() - ref).abs().max()):.3e}")
bad = (a - c).abs().amax(dim=1) > 1e-3
idx = bad.nonzero().flatten()
print("rows that differ:", int(bad.sum()), "first", int(idx[0]) if len(idx) else None, "last", int(idx[-1]) if len(idx) else None)
