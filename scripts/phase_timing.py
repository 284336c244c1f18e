import sys, time, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import synth, kernels
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
B = int(sys.argv[1])
exp = default_exp_params(num_slots=30, num_context=1, num_preds=19)
savi = setup_model(exp["model"]).eval(); pred = setup_predictor(exp).eval()
synth.fill_module_(savi, prefix="savi."); synth.fill_module_(pred, prefix="pred.")
savi, pred = savi.cuda(), pred.cuda()
videos = synth.synth_videos(B, 20, seed=100).cuda()
tokens, lengths = synth.synth_captions(B, max_len=12, seed=100); tokens, lengths = tokens.cuda(), lengths.cuda()
noise = synth.synth_noise(B, 30, 128, seed=200).cuda()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    for it in range(2):
        t0 = sync()
        out = savi(mode="decomp", x=videos, num_imgs=20, decode=False, init_noise=noise)
        t1 = sync()
        ps = pred(out["slot_history"], caption_tokens=tokens, caption_lengths=lengths)
        t2 = sync()
        dec = savi(mode="decode", slots=ps.reshape(B * 19, 30, 128))
        t3 = sync()
        print(f"B={B} iter{it}: encode {t1-t0:.3f}s rollout {t2-t1:.3f}s decode {t3-t2:.3f}s total {t3-t0:.3f}s -> {B*19/(t3-t0):.1f} frames/s", flush=True)
