#!/usr/bin/env python
""" GB/s of one slot-attention iteration (k and v, fp32, read once: B * 2 * N * D * 4 bytes) by HIP events """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K, synth
N, D, Ks = 4096, 128, 30
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 32, 128, 256]:
    q = synth.synth_tensor("b.q", (B, Ks, D), "normal").cuda()
    kv = synth.synth_tensor("b.kv", (B, N, 2 * D), "normal").cuda()
    k, v = kv[..., :D], kv[..., D:]
    ws = K.slot_attn_workspace(B, N, "cuda")
    X = kv * 256.0
    hi = X.half()
    planes = torch.stack([hi, (X - hi.float()).half()], dim=2).contiguous()
    for name, fn in (("fp32 rows", lambda: K.slot_attn_iter(q, k, v, D ** -0.5, 1e-8, ws=ws)),
                     ("f16 planes", lambda: K.slot_attn_iter_planes(q, planes, D ** -0.5, 1e-8, ws=ws))):
        for _ in range(3):
            fn()
        reps = 20
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
        med = ts[reps // 2]
        gb = B * 2 * N * D * 4 / 1e9
        print(f"B={B:4d} {name:10s}: median {med * 1e3:8.1f} us  min {ts[0] * 1e3:8.1f} us  -> {gb / med * 1e3:7.1f} GB/s "
              f"(median)  {gb / ts[0] * 1e3:7.1f} GB/s (best)", flush=True)
