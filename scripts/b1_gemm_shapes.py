"""Per-shape GEMM time of one evaluation step at a small batch (HIP events around every launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textocvp_amd import kernels, synth
from textocvp_amd.evaluator import forward_eval
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S, C, P = 30, 1, 19
exp = default_exp_params(num_slots=S, num_context=C, num_preds=P)
savi = setup_model(exp["model"]).eval().to(dev); pred = setup_predictor(exp).eval().to(dev)
synth.fill_module_(savi, prefix="savi."); synth.fill_module_(pred, prefix="pred.")
v = synth.synth_videos(B, C + P, seed=100).to(dev)
t, l = synth.synth_captions(B, max_len=12, seed=100)
t, l, n = t.to(dev), l.to(dev), synth.synth_noise(B, S, 128, seed=200).to(dev)
def run():
    forward_eval(savi, pred, v, C, P, caption_tokens=t, caption_lengths=l, init_noise=n, overlap_decode=False)
    torch.cuda.synchronize()
run(); run()
kernels.TIMER = kernels.LaunchTimer(only=("gemm_", "mha_", "xattn_", "slot_attn_"))
run()
summ = kernels.TIMER.summary(); kernels.TIMER = None
tot = sum(x["total_ms"] for x in summ.values())
print(f"B={B}: timed launches {sum(x['launches'] for x in summ.values())}, total {tot:.2f} ms")
for k, x in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:40]:
    print(f"  {k:48s} x{x['launches']:5d}  {x['total_ms']:8.3f} ms  avg {1e3 * x['total_ms'] / x['launches']:7.1f} us")
