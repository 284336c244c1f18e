#!/usr/bin/env python
""" eager vs HIP-graph replay of one forward_eval step at small batch (launch-bound regime) """
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import synth
from textocvp_amd.evaluator import forward_eval
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
exp = default_exp_params(num_slots=30, num_context=1, num_preds=19)
savi = setup_model(exp["model"]).eval(); pred = setup_predictor(exp).eval()
synth.fill_module_(savi, prefix="savi."); synth.fill_module_(pred, prefix="pred.")
savi, pred = savi.cuda(), pred.cuda()
videos = synth.synth_videos(B, 20, seed=100).cuda()
tokens, lengths = synth.synth_captions(B, max_len=12, seed=100); tokens, lengths = tokens.cuda(), lengths.cuda()
noise = synth.synth_noise(B, 30, 128, seed=200).cuda()
def step():
    return forward_eval(savi, pred, videos, 1, 19, caption_tokens=tokens, caption_lengths=lengths, init_noise=noise)
def timeit(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.no_grad():
    for _ in range(2): out_e = step()
    te = timeit(step)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out_g = step()
    tg = timeit(g.replay)
    g.replay(); torch.cuda.synchronize()
    err = (out_g["pred_imgs"] - out_e["pred_imgs"]).abs().max().item()
print(f"B={B}: eager {te*1e3:.1f} ms ({B*19/te:.0f} f/s)  graph replay {tg*1e3:.1f} ms ({B*19/tg:.0f} f/s)  max|diff| {err:.1e}")
