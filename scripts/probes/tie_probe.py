import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import slot_rollout_oracle as O
from textocvp_amd import synth
from textocvp_amd.evaluator import forward_eval
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
exp = default_exp_params(num_slots=7, num_context=1, num_preds=4)
savi, pred = setup_model(exp["model"]).eval(), setup_predictor(exp).eval()
synth.fill_module_(savi, prefix="savi."); synth.fill_module_(pred, prefix="pred.")
ssd = {k: v.clone() for k, v in savi.state_dict().items()}; psd = {k: v.clone() for k, v in pred.state_dict().items()}
savi, pred = savi.cuda(), pred.cuda()
videos = synth.synth_videos(3, 5, seed=11)
tokens, lengths = synth.synth_captions(3, max_len=15, lengths=[4, 15, 9], seed=5)
noise = synth.synth_noise(3, 7, 128, seed=12)
with torch.no_grad():
    hist, preds, imgs, masks = O.forward_eval(ssd, psd, videos, tokens, lengths, noise, 1, 4)
    out = forward_eval(savi, pred, videos.cuda(), 1, 4, caption_tokens=tokens.cuda(), caption_lengths=lengths.cuda(), init_noise=noise)
m = out["masks"].cpu()
d = (m.argmax(1) != masks.argmax(1))
print("diff pixels", d.nonzero().tolist())
for idx in d.nonzero().tolist():
    f, _, y, x = idx
    ours = m[f, :, 0, y, x]; ref = masks[f, :, 0, y, x]
    print("ours top2", ours.topk(2), "\noracle top2", ref.topk(2), "\nmax |mask diff| at pixel", float((ours - ref).abs().max()))
print("max mask err overall", float((m - masks).abs().max()))
