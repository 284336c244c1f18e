#!/usr/bin/env python
"""
Per-call timing of SAVi._encode_kv (conv encoder -> position embedding + LayerNorm -> per-pixel MLP -> slot-attention
input LayerNorm -> fused k/v projection) for one chunk of images: HIP events around every kernels.* call.
    python scripts/encode_kv_breakdown.py [images]
"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K, synth                                  # noqa: E402
from textocvp_amd.setup_model import default_exp_params, setup_model          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
savi = setup_model(default_exp_params(num_slots=30)["model"]).eval()
synth.fill_module_(savi, prefix="savi.")
savi = savi.cuda()
imgs = synth.synth_videos(n // 4 + 1, 4, seed=3).reshape(-1, 3, 64, 64)[:n].contiguous().cuda()
rec = collections.OrderedDict()


def wrap(name):
    fn = getattr(K, name)

    def w(*a, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(*a, **kw)
        e.record()
        shape = tuple(a[0].shape) if hasattr(a[0], "shape") else ()
        rec.setdefault((name, shape), []).append((s, e))
        return out
    setattr(K, name, w)


for name in ("linear", "layer_norm", "conv5x5_in3", "conv5x5", "conv5x5_bf16x3", "pos_embed"):
    if hasattr(K, name):
        wrap(name)
with torch.no_grad():
    for it in range(3):
        rec.clear()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        kv = savi._encode_kv(imgs)
        e.record()
        torch.cuda.synchronize()
print(f"{n} images: _encode_kv {s.elapsed_time(e):.2f} ms")
for (name, shape), evs in rec.items():
    print(f"  {name:12s} {str(shape):28s} x{len(evs)}  {sum(a.elapsed_time(b) for a, b in evs):7.3f} ms")
