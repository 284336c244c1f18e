#!/usr/bin/env python
"""
Mid-size form of the chunk-resident GEMM (tocvp_gemm_f16mid_f32) against the two-operand kernels on the shapes of the
predictor at small evaluation batches: bitwise comparison without split-K, deterministic and fp32-class with it, interleaved
timings of (fp32 input, in-loop split) / (plane input, two-operand kernel) / (plane input, mid kernel).
    python scripts/gemm_mid_bench.py [check|time|all] [reps]
"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(11)


def as_planes(x):
    v = torch.clamp(x * 256.0, -65504.0, 65504.0)
    hi = v.to(torch.float16)
    return K.SplitAct(torch.stack([hi, (v - hi.float()).to(torch.float16)], dim=1).contiguous(), x.shape)


def run(x, w, b, act, r, out_split, mid, splitk=True):
    K._GEMM_MID, K._GEMM_MID_SPLITK, K._GEMM_MID_MIN_TILES, K._GEMM_CHUNK = mid, splitk, 1, False
    with K.gemm_precision("f16x3"):
        y = K.linear(x, w, b, act=act, residual=r, out_split=out_split)
    return y.planes if out_split else y


if mode in ("check", "all"):
    bad = 0
    for (M, N, Kd) in [(2400, 512, 2048), (2400, 2048, 512), (960, 1536, 512), (300, 512, 512), (3841, 256, 128), (9600, 512, 2048),
                       (65, 768, 1024)]:
        x = torch.randn(M, Kd, generator=g).to(dev)
        w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev)
        b, r = torch.randn(N, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
        xp = as_planes(x)
        ref = x.double() @ w.double().t() + b.double()
        for act in (K.ACT_NONE, K.ACT_RELU, K.ACT_GELU):
            for res, osplit in ((None, 0), (r, 0), (None, 22)):
                a = run(xp, w, b, act, res, osplit, False)
                c = run(xp, w, b, act, res, osplit, True, splitk=False)
                d1 = run(xp, w, b, act, res, osplit, True, splitk=True)
                d2 = run(xp, w, b, act, res, osplit, True, splitk=True)
                torch.cuda.synchronize()
                v = (lambda t_: t_.view(torch.int16)) if osplit else (lambda t_: t_)
                if not torch.equal(v(a), v(c)):
                    bad += 1
                    print(f"MISMATCH (no split-K) {M}x{N}x{Kd} act {act} res {res is not None} split {osplit}: "
                          f"{(a.float() - c.float()).abs().max().item():.3e}", flush=True)
                if not torch.equal(v(d1), v(d2)):
                    bad += 1
                    print(f"NOT REPEATABLE (split-K) {M}x{N}x{Kd} act {act}", flush=True)
                if not osplit and (d1 - a).abs().max().item() > 2e-5 * max(1.0, a.abs().max().item()):
                    bad += 1
                    print(f"split-K far from unsplit {M}x{N}x{Kd}: {(d1 - a).abs().max().item():.3e}", flush=True)
        got = run(xp, w, b, K.ACT_NONE, None, 0, True)
        print(f"{M}x{N}x{Kd}: mid kernel (split-K) vs fp64: {(got.double() - ref).abs().max().item():.3e} (max |y| {ref.abs().max().item():.2f})",
              flush=True)
    print("check:", "ALL OK" if bad == 0 else f"{bad} problems", flush=True)

if mode in ("time", "all"):
    shapes = [(2400, 512, 2048), (2400, 2048, 512), (2400, 1536, 512), (2400, 512, 512), (960, 512, 2048), (960, 2048, 512),
              (3840, 512, 2048), (3840, 2048, 512), (3840, 1536, 512), (9600, 512, 2048), (9600, 2048, 512), (9600, 1536, 512),
              (9600, 512, 512), (240, 512, 2048), (240, 2048, 512)]
    for (M, N, Kd) in shapes:
        x = torch.relu(torch.randn(M, Kd, generator=g)).to(dev)
        w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        xp = as_planes(x)
        res = {}
        variants = (("fp32 in", x, False, True), ("planes", xp, False, True), ("mid", xp, True, True), ("mid no split-K", xp, True, False))
        for rnd in range(3):
            for name, inp, mid, sk in variants:
                run(inp, w, b, K.ACT_RELU, None, 0, mid, sk)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for _ in range(reps):
                        run(inp, w, b, K.ACT_RELU, None, 0, mid, sk)
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) / reps * 1e3)
        fl = 2.0 * M * N * Kd
        print(f"{M}x{N}x{Kd}: " + " | ".join(f"{n} {min(v):.1f} us ({fl / min(v) / 1e6:.0f} TF/s)" for n, v in res.items()), flush=True)
