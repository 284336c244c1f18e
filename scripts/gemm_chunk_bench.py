#!/usr/bin/env python
"""
Chunk-resident f16x3 GEMM (csrc/gemm_f16c.hip, tocvp_gemm_f16chunk_f32) against the in-loop-split / planes kernels of
gemm_bf16.hip on the same fp16 operand planes: bitwise comparison (fp32 and plane outputs, bias / ReLU / GELU / residual,
ragged row counts) and interleaved timings (HIP events, `reps` launches per round).
    python scripts/gemm_chunk_bench.py [check|time|all] [reps]
"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(7)


def operands(M, N, Kd, dense=True):
    x = torch.randn(M, Kd, generator=g).to(dev)
    if not dense:
        x = torch.relu(x)
    w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev)
    return x, w, b, r, None


def as_planes(x):
    """ fp16 operand planes (rows, 2, D) of 2^8 x, the expression of tocvp_store_planes4 """
    v = torch.clamp(x * 256.0, -65504.0, 65504.0)
    hi = v.to(torch.float16)
    lo = (v - hi.float()).to(torch.float16)
    return K.SplitAct(torch.stack([hi, lo], dim=1).contiguous(), x.shape)


def run(x_planes, w, b, act, r, out_split, chunk):
    K._GEMM_CHUNK, K._GEMM_CHUNK_MIN_TILES = chunk, 1
    with K.gemm_precision("f16x3"):
        y = K.linear(x_planes, w, b, act=act, residual=r, out_split=out_split)
    return y.planes if out_split else y


if mode in ("check", "all"):
    bad = 0
    for (M, N, Kd) in [(4096, 1024, 1024), (1000, 512, 128), (300, 512, 512), (128 * 9 + 5, 1536, 512), (2048, 2048, 2048),
                       (77, 1024, 3072), (1500, 768, 768), (700, 2304, 768), (260, 384, 128), (3000, 768, 3072)]:
        x, w, b, r, _ = operands(M, N, Kd)
        xp = as_planes(x)
        for act in (K.ACT_NONE, K.ACT_RELU, K.ACT_GELU):
            for res in (None, r):
                for osplit in (0, 22):
                    if osplit and res is not None:
                        continue
                    a = run(xp, w, b, act, res, osplit, False)
                    c = run(xp, w, b, act, res, osplit, True)
                    torch.cuda.synchronize()
                    same = torch.equal(a.view(torch.int16) if osplit else a, c.view(torch.int16) if osplit else c)
                    if not same:
                        bad += 1
                        d = (a.float() - c.float()).abs().max().item()
                        print(f"MISMATCH {M}x{N}x{Kd} act {act} res {res is not None} split {osplit}: max diff {d:.3e}", flush=True)
        ref = (x.double() @ w.double().t() + b.double()).float()
        got = run(xp, w, b, K.ACT_NONE, None, 0, True)
        print(f"{M}x{N}x{Kd}: chunk kernel vs fp64 product: max abs err {(got - ref).abs().max().item():.3e} "
              f"(max |y| {ref.abs().max().item():.2f})", flush=True)
    print("bitwise check:", "ALL EQUAL" if bad == 0 else f"{bad} mismatches", flush=True)

if mode in ("time", "all"):
    shapes = [(98304, 1024, 1024), (98304, 1024, 128), (38400, 1536, 512), (38400, 2048, 512), (38400, 512, 2048),
              (38400, 512, 512), (65792, 3072, 768), (65792, 2304, 768), (65792, 768, 3072), (9600, 1536, 512), (9600, 2048, 512), (9600, 512, 2048)]
    for (M, N, Kd) in shapes:
        x, w, b, r, _ = operands(M, N, Kd, dense=False)
        xp = as_planes(x)
        del x
        res = {}
        for rnd in range(3):
            for name, chunk in (("planes", False), ("chunk", True)):
                run(xp, w, b, K.ACT_RELU, None, 22, chunk)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run(xp, w, b, K.ACT_RELU, None, 22, chunk)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) / reps * 1e3)
        fl = 2.0 * M * N * Kd
        print(f"{M}x{N}x{Kd} (ReLU, plane output): " + " | ".join(
            f"{n} {min(v):.0f} us ({fl / min(v) / 1e6:.0f} TF/s; rounds {', '.join(f'{t:.0f}' for t in v)})" for n, v in res.items()),
            flush=True)
