"""
One 64 -> 64 5 x 5 layer on the GPU against fp64: the direct split-fp16 kernel, the Winograd kernel and torch's own fp32
convolution (CPU), signed and post-ReLU inputs.  python scripts/probes/wino_vs_direct_error.py
"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from textocvp_amd import kernels as K, synth
dev = torch.device("cuda:0")
for kind, post in (("normal", False), ("relu(normal)", True)):
    x = synth.synth_tensor("cmpx", (4, 64, 64, 64))
    if post: x = torch.relu(x)
    w = synth.synth_tensor("cmpw", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5)
    b = synth.synth_tensor("cmpb", (64,), "uniform", 0.1)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=2).permute(0, 2, 3, 1)
    d = K.conv5x5_dec_f16x3(x.to(dev), K.split_conv_weights_dec_f16x3(w.to(dev)), b.to(dev), relu=False).cpu().double()
    g = K.conv5x5_dec_wino(x.to(dev), K.split_conv_weights_wino_f16x3(w.to(dev)), b.to(dev), relu=False).cpu().double()
    f32 = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=2).permute(0, 2, 3, 1).double()
    sc = ref.abs().max().item()
    print(f"{kind}: scale {sc:.3g}; max abs err / scale: direct f16x3 {(d-ref).abs().max().item()/sc:.2e}, Winograd {(g-ref).abs().max().item()/sc:.2e}, torch fp32 (CPU) {(f32-ref).abs().max().item()/sc:.2e}; rms: direct {(d-ref).pow(2).mean().sqrt().item()/sc:.2e} Winograd {(g-ref).pow(2).mean().sqrt().item()/sc:.2e} fp32 {(f32-ref).pow(2).mean().sqrt().item()/sc:.2e}")
