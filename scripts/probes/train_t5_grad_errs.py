import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import torch, torch.nn.functional as F
from conftest import load_golden
from oracle import slot_rollout_oracle as O
from textocvp_amd import synth
from textocvp_amd.setup_model import default_exp_params, setup_predictor
from textocvp_amd.train import autograd as ag
from textocvp_amd.train.predictor import TrainablePredictor
DEV = torch.device("cuda", 0)
def rel_err(got, ref):
    ref = ref.to(torch.float64); got = got.detach().cpu().to(torch.float64)
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
g = load_golden("t5_encoder.npz")
exp = default_exp_params(num_slots=7, num_context=1, num_preds=3, predictor_name="TextOCVP_T5")
pred = setup_predictor(exp)
synth.fill_module_(pred.predictor.text_encoder, prefix="t5.")
for part in ("predictor", "mlp_in", "mlp_out", "pe"):
    synth.fill_module_(getattr(pred.predictor, part), prefix=f"pred.predictor.{part}.")
ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
B, P = ids.shape[0], 3
hist = synth.synth_tensor("train.hist_t5", (B, 1 + P, 7, 128), "normal")
pred = pred.to(DEV); tp = TrainablePredictor(pred); tape = ag.Tape()
preds = tp.rollout(tape, hist.to(DEV), ids.to(DEV), None, P, attn_masks=mask.to(DEV))
stacked = ag.stack_frames(tape, preds); target = hist[:, 1:1 + P]
total, sc = ag.mse(tape, stacked, target.to(DEV)); tape.backward()
text = pred.encode_text_caption(caption_tokens=ids.to(DEV), attn_masks=mask.to(DEV)).detach().cpu().double()
sd = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in pred.state_dict().items()
      if not k.startswith("predictor.text_encoder.") and v.dtype.is_floating_point}
p = O.sub(sd, "predictor."); window, ref = hist[:, :1].double().clone(), []
for t in range(P):
    cur = O.text_ocvp_step(p, window, text); window = torch.cat([window, cur.unsqueeze(1)], dim=1); ref.append(cur)
ref_preds = torch.stack(ref, dim=1); F.mse_loss(ref_preds, target.double()).backward()
print("forward rel err", rel_err(stacked.data, ref_preds))
errs = sorted(((rel_err(var.grad, sd[name].grad), name) for name, var in tp.names.items() if sd[name].grad is not None and var.grad is not None), reverse=True)
for e, n in errs[:8]: print(f"{e:.3e} {n}")
