#!/usr/bin/env python
"""Which strided copies (tocvp_copy4d_f32) of a configs[3] pass are slow: HIP events around every kernels.copy_strided call."""
import sys, os, collections, traceback
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import torch
from textocvp_amd import kernels as K
recs = []
orig = K.copy_strided
def timed(src, dst):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(src, dst)
    e1.record()
    st = traceback.extract_stack(limit=5)[:-1]
    recs.append((e0, e1, tuple(src.shape), src.stride(), dst.stride(),
                 " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in st)))
    return r
K.copy_strided = timed
import bench
res = bench.leg_config4(torch.device("cuda", 0), K, batch=16, reps=1)
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
for e0, e1, shape, ss, ds, where in recs:
    a = acc[(shape, ss, ds, where)]
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
for k, (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"{n:4d} calls {us / n:8.1f} us avg  shape {k[0]} src strides {k[1]} dst strides {k[2]}  {k[3]}")
