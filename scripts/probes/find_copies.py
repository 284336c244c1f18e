"""Probe: which python lines issue device-to-device copies during one eager training step."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from textocvp_amd import synth
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor
from textocvp_amd.train.step import PredictorTrainStep

seen = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(s in name for s in ("clone", "copy_", "_to_copy", "contiguous", "zeros", "fill", "zero_", "cat", "stack")):
            st = [f for f in traceback.extract_stack(limit=12) if "textocvp_amd" in f.filename]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:])
            numel = next((a.numel() for a in args if isinstance(a, torch.Tensor)), -1)
            seen[(name, numel, where)] += 1
        return func(*args, **(kwargs or {}))

dev = torch.device("cuda")
exp = default_exp_params(num_slots=30, num_context=1, num_preds=19)
savi, pred = setup_model(exp["model"]).eval(), setup_predictor(exp)
synth.fill_module_(savi, prefix="savi."); synth.fill_module_(pred, prefix="pred.")
ts = PredictorTrainStep(savi.to(dev), pred.to(dev))
B = 8
videos = synth.synth_videos(B, 20, seed=100).to(dev)
tokens, lengths = synth.synth_captions(B, max_len=12, seed=100); tokens, lengths = tokens.to(dev), lengths.to(dev)
noise = synth.synth_noise(B, 30, 128, seed=200).to(dev)
ts.step(videos, tokens, lengths, init_noise=noise)
with Spy():
    ts.step(videos, tokens, lengths, init_noise=noise)
for k, v in seen.most_common(25):
    print(v, k)
