#!/usr/bin/env python
""" forward_eval with / without decoding predicted frames on a second stream during the rollout """
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import synth
from textocvp_amd.evaluator import forward_eval
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
exp = default_exp_params(num_slots=30, num_context=1, num_preds=19)
savi = setup_model(exp["model"]).eval(); pred = setup_predictor(exp).eval()
synth.fill_module_(savi, prefix="savi."); synth.fill_module_(pred, prefix="pred.")
savi, pred = savi.cuda(), pred.cuda()
videos = synth.synth_videos(B, 20, seed=100).cuda()
tokens, lengths = synth.synth_captions(B, max_len=12, seed=100); tokens, lengths = tokens.cuda(), lengths.cuda()
noise = synth.synth_noise(B, 30, 128, seed=200).cuda()
def run(ov):
    return forward_eval(savi, pred, videos, 1, 19, overlap_decode=ov, caption_tokens=tokens, caption_lengths=lengths, init_noise=noise)
res = {}
for ov in (False, True, False, True):
    for _ in range(2): out = run(ov)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): out = run(ov)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    res[ov] = out
    print(f"B={B} overlap={ov}: {dt*1e3:.1f} ms/step -> {B*19/dt:.0f} frames/s", flush=True)
d = (res[True]["pred_imgs"] - res[False]["pred_imgs"]).abs().max().item()
dm = (res[True]["masks"] - res[False]["masks"]).abs().max().item()
print("max|pred_imgs diff|", d, "max|masks diff|", dm)
