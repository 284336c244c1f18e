#!/usr/bin/env python
"""configs[3] pass (16 sequences) with and without the decode / rollout overlap, and its phases alone (wall clock)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from textocvp_amd import synth, kernels
from textocvp_amd.evaluator import forward_eval
from textocvp_amd.setup_model import default_dinosaur_params, default_exp_params, setup_model, setup_predictor
dev = torch.device("cuda", 0)
B, K4, P4 = 16, 24, 29
model = setup_model(default_dinosaur_params(num_slots=K4, img_size=224)).eval()
exp = default_exp_params(num_slots=K4, num_context=1, num_preds=P4, predictor_name="TextOCVP_T5")
pred = setup_predictor(exp).eval()
synth.fill_module_(model, prefix="dino."); synth.fill_module_(pred, prefix="pred.")
model, pred = model.to(dev), pred.to(dev)
videos = synth.synth_videos(B, 1 + P4, height=224, width=224, seed=4).to(dev)
g = torch.Generator().manual_seed(5)
ids = torch.randint(1, 32000, (B, 16), generator=g).to(dev)
mask = torch.ones(B, 16, dtype=torch.int64, device=dev)
noise = synth.synth_noise(B, K4, 128, seed=3).to(dev)
def run(**kw):
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = forward_eval(model, pred, videos, 1, P4, caption_tokens=ids, attn_masks=mask, init_noise=noise, **kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts[1:]) * 1e3, out
import ctypes
from textocvp_amd import evaluator as EV
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(pattern, total=256, priority=None):
    nwords = (total + 31) // 32
    mask = (ctypes.c_uint32 * nwords)()
    for i in range(total):
        if pattern(i):
            mask[i // 32] |= 1 << (i % 32)
    s_ = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s_), nwords, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s_.value)


ndec = int(os.environ.get("DEC_CUS_PER_XCD", "0"))
if ndec:
    dec_s = masked_stream(lambda i: i % 32 < ndec)
    rol_s = masked_stream(lambda i: i % 32 >= ndec)
    EV._side_stream = lambda device: dec_s
    EV._rollout_stream = lambda device: rol_s
with torch.no_grad():
    a, oa = run(overlap_decode=True)
    b, ob = run(overlap_decode=False)
    print(f"overlap {a:.1f} ms | serial {b:.1f} ms | identical {torch.equal(oa['pred_imgs'], ob['pred_imgs'])}")
    feats = model.encoder(videos); torch.cuda.synchronize()
    for name, fn in (("ViT", lambda: model.encoder(videos)),
                     ("rollout", lambda: pred(ob["slot_history"][:, :1].contiguous(), caption_tokens=ids, attn_masks=mask, num_preds=P4) if False else pred(ob["slot_history"], caption_tokens=ids, attn_masks=mask)),
                     ("decode", lambda: model(mode="decode", slots=ob["pred_slots"].reshape(B * P4, K4, 128)[:16 * 8]))):
        ts = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{name}: {statistics.median(ts[1:]) * 1e3:.1f} ms" + (" (8 of 29 steps)" if name == "decode" else ""))
if os.environ.get("TRY_GRAPH", "0") != "0":
    from textocvp_amd.evaluator import GraphedEval
    ge = GraphedEval(model, pred, 1, P4, overlap_decode=os.environ.get("TRY_GRAPH") == "2")
    with torch.no_grad():
        ts = []
        for it in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            og = ge(videos, caption_tokens=ids, attn_masks=mask, init_noise=noise)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("graph replay:", [f"{t * 1e3:.1f}" for t in ts], "ms; identical", torch.equal(og["pred_imgs"], ob["pred_imgs"]))
