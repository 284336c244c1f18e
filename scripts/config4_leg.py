#!/usr/bin/env python
"""
bench.py's configs[3] leg on its own (ExtendedDINOSAUR from pixels + TextOCVP_T5, 16 sequences, 1 seed + 29 preds) with
the per-shape table of its split GEMMs: A/B of the GEMM knobs (TOCVP_GEMM_CHUNK, TOCVP_VIT_PLANES, TOCVP_DINO_MLP_PLANES).
    python scripts/config4_leg.py [batch] [reps]
"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from textocvp_amd import kernels

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
shapes = {}
_orig = kernels.LaunchTimer.summary


def summary(self):
    out = _orig(self)
    shapes.update(out)
    return out


kernels.LaunchTimer.summary = summary
res = bench.leg_config4(torch.device("cuda", 0), kernels, batch=batch, reps=reps)
print(json.dumps(res))
for name, v in sorted(shapes.items(), key=lambda kv: -kv[1]["total_ms"])[:14]:
    if v["launches"]:
        print(f"  {name:40s} x {v['launches']:4d}  {v['total_ms']:8.2f} ms  {v['total_ms'] / v['launches'] * 1e3:8.1f} us  "
              f"{v['units'] / v['total_ms'] / 1e9:6.1f} TF/s")
