"""Probe: eager vs HIP-graph replay of one evaluation step (forward_eval + PSNR/SSIM) at small batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textocvp_amd import kernels, synth
from textocvp_amd.evaluator import forward_eval
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor

dev = torch.device("cuda", 0)
S, C, P = 30, 1, 19
exp = default_exp_params(num_slots=S, num_context=C, num_preds=P)
savi = setup_model(exp["model"]).eval(); pred = setup_predictor(exp).eval()
synth.fill_module_(savi, prefix="savi."); synth.fill_module_(pred, prefix="pred.")
savi, pred = savi.to(dev), pred.to(dev)

def inputs(B):
    v = synth.synth_videos(B, C + P, seed=100).to(dev)
    t, l = synth.synth_captions(B, max_len=12, seed=100)
    return v, t.to(dev), l.to(dev), synth.synth_noise(B, S, 128, seed=200).to(dev)

def step(inp, **kw):
    v, t, l, n = inp
    out = forward_eval(savi, pred, v, C, P, caption_tokens=t, caption_lengths=l, init_noise=n, **kw)
    B, Pn, Ch, H, W = out["pred_imgs"].shape
    ps, ss = kernels.psnr_ssim(out["pred_imgs"].reshape(B * Pn, Ch, H, W), out["targets"].reshape(B * Pn, Ch, H, W), clamp01=True)
    return torch.stack([ps.view(B, Pn), ss.view(B, Pn)], -1), out["pred_imgs"]

def wall(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n

for B in [int(a) for a in (sys.argv[1:] or ["1", "8", "32"])]:
    inp = inputs(B)
    for ov in (True, False):
        for _ in range(2): m0, img0 = step(inp, overlap_decode=ov)
        m0, img0 = m0.clone(), img0.clone()
        e = wall(lambda: step(inp, overlap_decode=ov), 5)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                m1, img1 = step(inp, overlap_decode=ov)
            g.replay(); torch.cuda.synchronize()
            same = bool(torch.equal(m1, m0) and torch.equal(img1, img0))
            r = wall(g.replay, 5)
            print(f"B={B} overlap={ov}: eager {e:.2f} ms  graph {r:.2f} ms  identical={same}", flush=True)
            del g
        except Exception as err:
            print(f"B={B} overlap={ov}: eager {e:.2f} ms  capture failed: {type(err).__name__}: {str(err)[:300]}", flush=True)
            torch.cuda.synchronize()
