#!/usr/bin/env python
"""
Collapsed text cross-attention (csrc/xattn.hip) against the four-kernel path (LayerNorm, q GEMM, attention over the
caption, output GEMM + residual) on one TransformerDecoderBlock: HIP events around back-to-back launches.
    python scripts/xattn_bench.py [B Tq Lt]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import kernels as K                                              # noqa: E402
from textocvp_amd.models.Blocks.attention import TextKV, TransformerDecoderBlock   # noqa: E402

B, Tq, Lt = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 300, 12)
E, H, dh = 512, 8, 64
blk = TransformerDecoderBlock(embed_dim=E, head_dim=dh, kv_dim=E, num_heads=H, mlp_size=2048).eval().cuda()
x = torch.randn(B, Tq, E, device="cuda")
text = torch.randn(B, Lt, E, device="cuda")


def bench(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


with torch.no_grad(), K.gemm_precision("f16x3"):
    tkv = blk.project_text(text)
    Gf, Hf, _ = tkv.collapsed
    lnq, ca = blk.ln_cross_att_q, blk.cross_attn
    fused = lambda: K.xattn_collapsed(x, lnq.weight, lnq.bias, lnq.eps, Gf, Hf, ca.out_projection.bias, H, Lt, dh ** -0.5)
    four = lambda: ca(None, query_embs=K.layer_norm(x, lnq.weight, lnq.bias, lnq.eps), residual=x, kv=tkv.kv)
    t_f, t_4 = bench(fused), bench(four)
    t_prep = bench(lambda: blk.project_text(text), reps=5, warm=1)
print(f"B={B} Tq={Tq} Lt={Lt}: fused {t_f:.1f} us, four kernels {t_4:.1f} us; operands per caption batch {t_prep:.1f} us; "
      f"algorithmic bytes x + y = {2 * B * Tq * E * 4 / 1e6:.0f} MB -> {2 * B * Tq * E * 4 / t_f / 1e3:.0f} GB/s")
