#!/usr/bin/env python
""" A few launches of the Winograd decoder conv at the headline shape, for rocprofv3 counter passes (scripts/pmc_collect.sh).
    python scripts/wino_one.py [nimg] [launches] """
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textocvp_amd import kernels as K           # noqa: E402
from textocvp_amd import synth                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2040
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
x = torch.relu(synth.synth_tensor("wbx", (n, 64, 64, 64))).to(dev)
w = synth.synth_tensor("wbw", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5).to(dev)
b = synth.synth_tensor("wbb", (64,), "uniform", 0.1).to(dev)
wp = K.split_conv_weights_wino_f16x3(w)
x16 = K.conv5x5_dec_wino(x, wp, b, relu=True, out_mode=1)
out = torch.empty_like(x16)
for _ in range(reps):
    K.conv5x5_dec_wino(x16, wp, b, relu=True, out=out, in_mode=0, out_mode=1)
torch.cuda.synchronize()
