#!/usr/bin/env python
""" cProfile of the HOST side of one evaluation step at batch B (default 8): where the Python time of the ~2400 launches goes.
    python scripts/host_profile.py [B] """
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import synth                                                        # noqa: E402
from textocvp_amd.evaluator import forward_eval                                       # noqa: E402
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
exp = default_exp_params(num_slots=30, num_context=1, num_preds=19)
savi, pred = setup_model(exp["model"]).eval(), setup_predictor(exp).eval()
synth.fill_module_(savi, prefix="savi.")
synth.fill_module_(pred, prefix="pred.")
savi, pred = savi.cuda(), pred.cuda()
videos = synth.synth_videos(B, 20, seed=100).cuda()
tokens, lengths = synth.synth_captions(B, max_len=12, seed=100)
tokens, lengths = tokens.cuda(), lengths.cuda()
noise = synth.synth_noise(B, 30, 128, seed=200).cuda()


def step():
    return forward_eval(savi, pred, videos, 1, 19, caption_tokens=tokens, caption_lengths=lengths, init_noise=noise)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3 * (t1 - t0) / 5:.1f} ms per step, until the device is done {1e3 * (t2 - t0) / 5:.1f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
