""" fp32-input f16x3 GEMMs (A split in the k-loop: gemm_bf16_wfrag_kernel) through the library named by MHA_LIB (default: the tree's):
us per launch and a checksum of the output bits, for same-box A/B of two builds. """
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
from textocvp_amd import build as _build
if os.environ.get("MHA_LIB"):
    _build.LIB_PATH = os.path.abspath(os.environ["MHA_LIB"])
from textocvp_amd import kernels as K
g = torch.Generator().manual_seed(5)
for M, N, Kd in ((76800, 512, 512), (38400, 512, 512), (9600, 512, 512), (9600, 2048, 512), (9600, 512, 2048), (2400, 2048, 512), (2400, 512, 2048), (4800, 512, 512)):
    x = torch.randn(M, Kd, generator=g).cuda(); w = (torch.randn(N, Kd, generator=g) * Kd ** -0.5).cuda()
    b = torch.randn(N, generator=g).cuda(); R = torch.randn(M, N, generator=g).cuda()
    for _ in range(3): y = K.linear(x, w, b, residual=R, precision="f16x3")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = K.linear(x, w, b, residual=R, precision="f16x3")
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('MHA_LIB', 'tree')[-12:]:12s} {M}x{N}x{Kd}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us  bits {int(y.view(torch.int32).sum(dtype=torch.int64)):x}", flush=True)
