#!/usr/bin/env python
"""
MLPPatchDecoder (BASELINE configs[3], reference decoders.py:264-307) -- its four Linear layers at the bench's chunk size
(16 frames x 24 slots x 256 patches = 98304 rows; 128 -> 1024 -> 1024 -> 1024 -> 769) under the hand-over variants:
  fp32      fp32 activations, split in the consumer's k-loop (gemm_bf16_wfrag_kernel)
  planes    fp16 operand planes written by the producing epilogue, in-loop kernel's plane-input form
  planes+chunk the same planes into the chunk-resident persistent GEMM (gemm_f16c.hip)
Prints the time per forward of the MLP chain (HIP events, interleaved rounds) and checks the variants bit for bit.
"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import synth, kernels as K
from textocvp_amd.setup_model import default_dinosaur_params, setup_model
from textocvp_amd.models.EncodersDecoders import decoders as D

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model = setup_model(default_dinosaur_params(num_slots=24, img_size=224)).eval()
synth.fill_module_(model, prefix="dino.")
dec = model.decoder.cuda()
dec.reconstruct_images = False
slots = synth.synth_noise(frames, 24, 128, seed=5).cuda()

VARIANTS = {"fp32": (False, False), "planes": (True, False), "planes+chunk": (True, True)}


def run(name):
    D._MLP_PLANES, K._GEMM_CHUNK = VARIANTS[name]
    with torch.no_grad():
        return dec(slots)


class Timer:
    def __init__(self):
        self.rows = {}

    def wrap(self, name, units, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.rows.setdefault(name, []).append((a, b, units))
        return r


ref = None
for name in VARIANTS:
    out = run(name)
    torch.cuda.synchronize()
    if ref is None:
        ref = out
    else:
        for k_ in ("recons_feats", "masks"):
            same = torch.equal(ref[k_], out[k_])
            print(f"{name}: {k_} bit-identical to fp32 hand-over: {same}"
                  + ("" if same else f" (max diff {(ref[k_] - out[k_]).abs().max().item():.3e})"), flush=True)

for r in range(rounds):
    for name in VARIANTS:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        K.TIMER = Timer()
        ev[0].record()
        run(name)
        ev[1].record()
        torch.cuda.synchronize()
        t = K.TIMER
        K.TIMER = None
        parts = []
        for nm, rows in t.rows.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in rows)
            fl = sum(u for _, _, u in rows)
            parts.append(f"{nm} {ms * 1e3:.0f} us ({fl / ms / 1e9:.0f} TF/s)")
        print(f"round {r} {name:10s} decoder {ev[0].elapsed_time(ev[1]):.3f} ms | " + " | ".join(parts), flush=True)
