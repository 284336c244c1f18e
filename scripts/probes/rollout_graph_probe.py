#!/usr/bin/env python
"""How host-bound is the rollout at small batches?  The predictor's 19-step rollout launched from Python against the same
rollout replayed from ONE captured HIP graph (device time only), B = 8 / 32."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from textocvp_amd import synth
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
dev = torch.device("cuda", 0)
exp = default_exp_params(num_slots=30, num_context=1, num_preds=19)
savi = setup_model(exp["model"]).eval().to(dev); pred = setup_predictor(exp).eval().to(dev)
synth.fill_module_(savi, prefix="savi."); synth.fill_module_(pred, prefix="pred.")
for B in (8, 32):
    t, l = synth.synth_captions(B, max_len=12, seed=1)
    t, l = t.to(dev), l.to(dev)
    hist = torch.randn(B, 20, 30, 128, device=dev)
    with torch.no_grad():
        for _ in range(2): ref = pred(hist, caption_tokens=t, caption_lengths=l)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): pred(hist, caption_tokens=t, caption_lengths=l)
        torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 5 * 1e3
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = pred(hist, caption_tokens=t, caption_lengths=l)
        g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): g.replay()
        torch.cuda.synchronize(); rep = (time.perf_counter() - t0) / 5 * 1e3
    print(f"B = {B}: rollout eager {eager:.1f} ms, replayed from a graph {rep:.1f} ms, identical {torch.equal(out, ref)}", flush=True)
