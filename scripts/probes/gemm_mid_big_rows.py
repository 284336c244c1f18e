import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from textocvp_amd import kernels as K
dev = torch.device("cuda", 0); g = torch.Generator().manual_seed(1)
for (M, N, Kd) in [(38400, 1536, 512), (76800, 1536, 512), (23040, 1536, 512), (38400, 512, 512), (76800, 512, 512)]:
    x = torch.relu(torch.randn(M, Kd, generator=g)).to(dev)
    v = torch.clamp(x * 256.0, -65504.0, 65504.0); hi = v.half()
    xp = K.SplitAct(torch.stack([hi, (v - hi.float()).half()], dim=1).contiguous(), (M, Kd))
    w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev) if os.environ.get("WITH_RESIDUAL") else None
    res = {}
    for rnd in range(3):
        for name, mid in (("planes kernel", False), ("mid", True)):
            K._GEMM_CHUNK, K._GEMM_MID, K._GEMM_MID_MAX_ROWS = False, mid, 10 ** 9
            with K.gemm_precision("f16x3"):
                for _ in range(2): K.linear(xp, w, b, residual=R)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): K.linear(xp, w, b, residual=R)
                e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{M}x{N}x{Kd}: " + " | ".join(f"{n} {min(t):.0f} us" for n, t in res.items()), flush=True)
