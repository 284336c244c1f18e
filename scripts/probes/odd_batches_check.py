#!/usr/bin/env python
"""configs[1] at batch sizes that cut the encoder / decoder chunks and the GEMM tile plans unevenly (77, 129, 200, 33):
samples of the batch against their own batch-of-1 runs; overlapped decode against the serial order."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from textocvp_amd import synth
from textocvp_amd.evaluator import forward_eval
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
dev = torch.device("cuda", 0)
exp = default_exp_params(num_slots=30, num_context=1, num_preds=19)
savi = setup_model(exp["model"]).eval().to(dev); pred = setup_predictor(exp).eval().to(dev)
synth.fill_module_(savi, prefix="savi."); synth.fill_module_(pred, prefix="pred.")
P = 19
with torch.no_grad():
    for B in (77, 129, 200, 33):
        v = synth.synth_videos(B, 1 + P, seed=100 + B).to(dev)
        t, l = synth.synth_captions(B, max_len=12, seed=B)
        t, l, n = t.to(dev), l.to(dev), synth.synth_noise(B, 30, 128, seed=B).to(dev)
        out = forward_eval(savi, pred, v, 1, P, caption_tokens=t, caption_lengths=l, init_noise=n, overlap_decode=False)
        ov = forward_eval(savi, pred, v, 1, P, caption_tokens=t, caption_lengths=l, init_noise=n, overlap_decode=True)
        same = torch.equal(ov["pred_imgs"], out["pred_imgs"]) and torch.equal(ov["pred_slots"], out["pred_slots"])
        worst = 0.0
        for b in (0, B // 3, B - 1):
            one = forward_eval(savi, pred, v[b:b + 1], 1, P, caption_tokens=t[b:b + 1], caption_lengths=l[b:b + 1], init_noise=n[b:b + 1])
            worst = max(worst, float((one["pred_imgs"] - out["pred_imgs"][b:b + 1]).abs().max()),
                        float((one["pred_slots"] - out["pred_slots"][b:b + 1]).abs().max()),
                        float((one["masks"] - out["masks"][b * P:(b + 1) * P]).abs().max()))
        print(f"B = {B}: overlapped == serial {same}; worst |batch - batch-of-1| {worst:.2e}; finite {bool(torch.isfinite(out['pred_imgs']).all())}", flush=True)
