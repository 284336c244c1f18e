#!/usr/bin/env python
"""
Per-shape timing of every kernels.linear / mha / layer_norm call of ONE rollout (predictor only) at
batch B: HIP events around each call, summed per (op, shape).  Usage: rollout_breakdown.py [B]
"""
import collections
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels, synth                                        # noqa: E402
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
exp = default_exp_params(num_slots=30, num_context=1, num_preds=19)
savi = setup_model(exp["model"]).eval()
pred = setup_predictor(exp).eval()
synth.fill_module_(savi, prefix="savi.")
synth.fill_module_(pred, prefix="pred.")
savi, pred = savi.cuda(), pred.cuda()
videos = synth.synth_videos(B, 20, seed=100).cuda()
tokens, lengths = synth.synth_captions(B, max_len=12, seed=100)
tokens, lengths = tokens.cuda(), lengths.cuda()
noise = synth.synth_noise(B, 30, 128, seed=200).cuda()

records = collections.defaultdict(list)


def timed(name, fn, keyfn):
    def wrapper(*a, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(*a, **kw)
        e.record()
        records[(name,) + keyfn(*a, **kw)].append((s, e))
        return out
    return wrapper


def lin_key(x, weight, *a, **kw):
    shape = x.shape if not isinstance(x, kernels.SplitAct) else x.shape
    M = 1
    for d in shape[:-1]:
        M *= d
    return (M, weight.shape[0], weight.shape[1], kernels._GEMM_PRECISION if kw.get("precision") is None else kw["precision"])


with torch.no_grad():
    out = savi(mode="decomp", x=videos, num_imgs=20, decode=False, init_noise=noise)
    for it in range(2):
        if it == 1:
            kernels.linear = timed("linear", kernels.linear, lin_key)
            kernels.mha = timed("mha", kernels.mha, lambda q, k, v, *a, **kw: (tuple(q.shape), tuple(k.shape)))
            kernels.layer_norm = timed("layer_norm", kernels.layer_norm, lambda x, *a, **kw: (tuple(x.shape),))
            kernels.mlp_fused = timed("mlp_fused", kernels.mlp_fused, lambda x, w1, *a, **kw: (x.planes.shape[0], w1.shape[0]))
            kernels.xattn_collapsed = timed("xattn", kernels.xattn_collapsed, lambda x, *a, **kw: (tuple(x.shape),))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ps = pred(out["slot_history"], caption_tokens=tokens, caption_lengths=lengths)
        e.record()
        torch.cuda.synchronize()
        print(f"rollout pass {it}: {s.elapsed_time(e):.1f} ms")

rows = []
for key, evs in records.items():
    ms = sum(a.elapsed_time(b) for a, b in evs)
    tf = ""
    if key[0] == "linear":
        M, N, Kd = key[1:4]
        tf = f"{2.0 * M * N * Kd * len(evs) / ms / 1e9:7.1f} TF/s"
    rows.append((ms, key, len(evs), tf))
total = sum(r[0] for r in rows)
for ms, key, n, tf in sorted(rows, reverse=True)[:40]:
    print(f"{ms:8.2f} ms {100 * ms / total:5.1f}%  x{n:4d}  {ms / n * 1e3:8.1f} us  {tf}  {key}")
print(f"sum of timed calls {total:.1f} ms")
