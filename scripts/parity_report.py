#!/usr/bin/env python
""" Prints the measured max-abs deviations of the HIP path from the reference goldens (C1, C2). """
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from textocvp_amd import synth
from textocvp_amd.evaluator import forward_eval
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor

def run(K, B, P, golden, lengths=None, sub=1):
    exp = default_exp_params(num_slots=K, num_context=1, num_preds=P)
    savi, pred = setup_model(exp["model"]).eval(), setup_predictor(exp).eval()
    synth.fill_module_(savi, prefix="savi."); synth.fill_module_(pred, prefix="pred.")
    savi, pred = savi.cuda(), pred.cuda()
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", golden)))
    videos = synth.synth_videos(B, 1 + P, seed=0)
    tokens, lens = synth.synth_captions(B, max_len=12, lengths=lengths, seed=0)
    noise = synth.synth_noise(B, K, 128, seed=1)
    with torch.no_grad():
        out = forward_eval(savi, pred, videos.cuda(), 1, P, caption_tokens=tokens.cuda(),
                           caption_lengths=lens.cuda(), init_noise=noise)
    d = lambda a, b: float((a.cpu().double() - torch.from_numpy(b).double()).abs().max())
    imgs = out["pred_imgs"][..., ::sub, ::sub]
    key = "pred_imgs" if sub == 1 else f"pred_imgs_sub{sub}"
    ps = out["pred_slots"].cpu()
    per_step = [float((ps[:, t].double() - torch.from_numpy(g["pred_slots"][:, t]).double()).abs().max()) for t in range(P)]
    print(f"{golden}: slot_history {d(out['slot_history'], g['slot_history']):.2e}  pred_slots {d(out['pred_slots'], g['pred_slots']):.2e}  pred_imgs {d(imgs, g[key]):.2e}")
    print("   per-step pred_slots err:", " ".join(f"{e:.1e}" for e in per_step))

run(7, 2, 4, "e2e_c1.npz", lengths=[9, 12])
run(30, 1, 19, "e2e_c2.npz", sub=2)
