""" decoder of 68 frames x 30 slots (one chunk of 2040 slot images) through the library named by MHA_LIB (default: the tree's): ms per decode.
Same-box check that an unrelated change of the library (code layout) did not move the conv kernel. """
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
from textocvp_amd import build as _build
if os.environ.get("MHA_LIB"):
    _build.LIB_PATH = os.path.abspath(os.environ["MHA_LIB"])
from textocvp_amd import synth
from textocvp_amd.setup_model import default_exp_params, setup_model
exp = default_exp_params(num_slots=30, num_preds=19)
savi = setup_model(exp["model"]).eval(); synth.fill_module_(savi, prefix="savi."); savi = savi.cuda()
slots = synth.synth_tensor("dec.slots", (68, 30, 128), "normal", 2.0).cuda()
with torch.no_grad():
    for _ in range(3): savi(mode="decode", slots=slots)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): savi(mode="decode", slots=slots)
    e1.record(); torch.cuda.synchronize()
print(os.environ.get("MHA_LIB", "tree"), f"{e0.elapsed_time(e1) / 20:.3f} ms per decode of 2040 slot images")
