#!/usr/bin/env python
""" does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) restrict a kernel to a subset of the CUs on this pool?
    Times SAVi.decode on masked streams of different widths. """
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from textocvp_amd import synth
from textocvp_amd.setup_model import default_exp_params, setup_model

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(pattern, total=256):
    nwords = (total + 31) // 32
    mask = (ctypes.c_uint32 * nwords)()
    for i in range(total):
        if pattern(i):
            mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), nwords, mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value)


exp = default_exp_params(num_slots=30, num_preds=19)
savi = setup_model(exp["model"]).eval()
synth.fill_module_(savi, prefix="savi.")
savi = savi.cuda()
slots = synth.synth_tensor("dec.slots", (34, 30, 128), "normal", 2.0).cuda()
with torch.no_grad():
    savi(mode="decode", slots=slots)
    torch.cuda.synchronize()
    pats = [("all 256", lambda i: True)]
    for lo, hi in ((0, 32), (0, 64), (0, 128), (0, 192), (128, 256), (64, 128), (32, 64), (0, 16), (0, 8)):
        pats.append((f"bits [{lo}, {hi})", (lambda lo, hi: lambda i: lo <= i < hi)(lo, hi)))
    pats.append(("low 16 of every word", lambda i: i % 32 < 16))
    pats.append(("low 8 of every word", lambda i: i % 32 < 8))
    pats.append(("words 0,2,4,6", lambda i: (i // 32) % 2 == 0))
    for name, pat in pats:
        st = masked_stream(pat)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            savi(mode="decode", slots=slots)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                savi(mode="decode", slots=slots)
            e1.record()
        torch.cuda.synchronize()
        print(f"{name:24s}: {e0.elapsed_time(e1) / 3:.2f} ms per decode of 1020 slot images")
