#!/usr/bin/env python
"""
Decoder conv 64 -> 64, 5 x 5, at the headline shape (2040 slot images of 64 x 64): the direct split-fp16 kernel
(operand planes in / out, as the decoder chains it) against the Winograd form (x 16 pass-major in / out).

    python scripts/wino_bench.py [nimg] [reps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textocvp_amd import kernels as K           # noqa: E402
from textocvp_amd import synth                  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2040
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    x = torch.relu(synth.synth_tensor("wbx", (n, 64, 64, 64))).to(dev)
    w = synth.synth_tensor("wbw", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5).to(dev)
    b = synth.synth_tensor("wbb", (64,), "uniform", 0.1).to(dev)
    wf = K.split_conv_weights_dec_f16x3(w)
    wp = K.split_conv_weights_wino_f16x3(w)
    flops = 2.0 * n * 64 * 64 * 64 * 64 * 25
    planes = K.conv5x5_dec_f16x3(x, wf, b, relu=True, pm_out=True, planes=True)
    out_d = torch.empty_like(planes)
    t_d = timeit(lambda: K.conv5x5_dec_f16x3(planes, wf, b, relu=True, out=out_d, pm_in=True, pm_out=True, planes=True), reps)
    x16 = K.conv5x5_dec_wino(x, wp, b, relu=True, out_mode=1)
    out_w = torch.empty_like(x16)
    t_w = timeit(lambda: K.conv5x5_dec_wino(x16, wp, b, relu=True, out=out_w, in_mode=0, out_mode=1), reps)
    out_p = torch.empty_like(x16)
    t_wp = timeit(lambda: K.conv5x5_dec_wino(x16, wp, b, relu=True, out=out_p, in_mode=0, out_mode=2), reps)
    print(f"{n} slot images: direct planes->planes {t_d:.3f} ms ({flops / t_d * 1e-9:.0f} TFLOP/s algorithmic) | "
          f"Winograd x16->x16 {t_w:.3f} ms ({flops / t_w * 1e-9:.0f}) | Winograd x16->planes {t_wp:.3f} ms")
    ref = K.conv5x5_dec_f16x3(x, wf, b, relu=True)
    got = K.conv5x5_dec_wino(x, wp, b, relu=True)
    print("max |Winograd - direct| =", float((got - ref).abs().max()), "at scale", float(ref.abs().max()))


if __name__ == "__main__":
    main()
