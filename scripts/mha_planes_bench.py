""" Plane-input self-attention (csrc/attn_planes.hip) against the fp32-input kernel at the predictor / ViT shapes, HIP events,
same box.  MHA_LIB=path runs another build of the library. """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import build as _build
if os.environ.get("MHA_LIB"):
    _build.LIB_PATH = os.path.abspath(os.environ["MHA_LIB"])
from textocvp_amd import kernels as K


def planes_of(x2):
    X = torch.clamp(x2 * 256.0, -65504.0, 65504.0)
    hi = X.to(torch.float16)
    return torch.stack([hi, (X - hi.float()).to(torch.float16)], dim=1).contiguous()


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, H, T in ((256, 8, 300), (128, 8, 300), (256, 12, 257), (256, 12, 256), (32, 8, 300), (8, 8, 300), (1, 8, 300)):
    E = H * 64
    torch.manual_seed(0)
    qkv = torch.randn(B, T, 3 * E, device="cuda")
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    sp = K.SplitAct(planes_of(qkv.reshape(B * T, 3 * E)), (B, T, 3 * E))
    with K.gemm_precision("f16x3"):
        old = timeit(lambda: K.mha(q, k, v, H, 0.125, out_split=22))
        new = timeit(lambda: K.mha_planes(sp, 0, sp, E, sp, 2 * E, B, T, T, H, 0.125, out_split=22))
        a, b = K.mha(q, k, v, H, 0.125), K.mha_planes(sp, 0, sp, E, sp, 2 * E, B, T, T, H, 0.125)
    fl = 4.0 * B * H * T * T * 64
    print(f"B {B:3d} H {H:2d} T {T}: fp32-input {old:7.1f} us ({fl / old * 1e-6:6.1f} TF/s)   planes {new:7.1f} us ({fl / new * 1e-6:6.1f} TF/s)"
          f"   equal rows {bool(torch.equal(a[:, :T - 1], b[:, :T - 1]))}  max diff {float((a - b).abs().max()):.2e}", flush=True)
