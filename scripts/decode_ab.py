#!/usr/bin/env python
"""SAVi.decode at the bench's chunk shape with the library given as argv[1] (A/B of two builds on one box)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import textocvp_amd.build as B
B.LIB_PATH = os.path.abspath(sys.argv[1])
B.build = lambda *a, **k: B.LIB_PATH
from textocvp_amd import synth
from textocvp_amd.setup_model import default_exp_params, setup_model
exp = default_exp_params(num_slots=30, num_preds=19)
savi = setup_model(exp["model"]).eval(); synth.fill_module_(savi, prefix="savi."); savi = savi.cuda()
slots = synth.synth_tensor("dec.slots", (68, 30, 128), "normal", 2.0).cuda()
with torch.no_grad():
    for _ in range(2): savi(mode="decode", slots=slots)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): savi(mode="decode", slots=slots)
    e1.record(); torch.cuda.synchronize()
print(os.path.basename(sys.argv[1]), f"{e0.elapsed_time(e1) / 10:.3f} ms per decode of 2040 slot images")
