""" bench.py's configs[3] leg through the library named by MHA_LIB (default: the tree's): same-box A/B of two builds. """
import json, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, root)
from textocvp_amd import build as _build
if os.environ.get("MHA_LIB"):
    _build.LIB_PATH = os.path.abspath(os.environ["MHA_LIB"])
import torch
import bench
from textocvp_amd import kernels
res = bench.leg_config4(torch.device("cuda", 0), kernels, batch=int(sys.argv[1]) if len(sys.argv) > 1 else 16, reps=2)
print(os.environ.get("MHA_LIB", "tree")[-12:], res["value"], res["ms_per_step"])
