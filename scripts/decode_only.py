#!/usr/bin/env python
""" runs SAVi.decode a few times at the bench shape (for PMC / trace collection of the conv kernel) """
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import synth
from textocvp_amd.setup_model import default_exp_params, setup_model
F = int(sys.argv[1]) if len(sys.argv) > 1 else 68
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
exp = default_exp_params(num_slots=30, num_preds=19)
savi = setup_model(exp["model"]).eval(); synth.fill_module_(savi, prefix="savi."); savi = savi.cuda()
slots = synth.synth_tensor("dec.slots", (F, 30, 128), "normal", 2.0).cuda()
with torch.no_grad():
    for _ in range(reps):
        savi(mode="decode", slots=slots)
torch.cuda.synchronize()
print("done")
