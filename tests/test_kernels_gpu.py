"""
Kernel-level parity: every C-ABI entry point of libtocvp.so against the CPU oracle / plain fp32
torch-CPU math on the same seeded inputs.  Needs a real MI355X (pytest -m gpu).

Tolerance: all kernels compute in fp32 (exact-fp32 MFMA fma chains), so the only differences to
the CPU are summation order and libm (expf/erff/tanhf) -> 2e-5 relative to the output scale.
"""

import math

import os

import pytest
import torch
import torch.nn.functional as F

from oracle import slot_rollout_oracle as O
from textocvp_amd import synth

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _k():
    from textocvp_amd import kernels
    return kernels


def rnd(name, shape, kind="normal", scale=1.0):
    return synth.synth_tensor("ktest." + name, shape, kind, scale)


def close(got, ref, tol=2e-5):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= tol * scale, f"max abs err {err:.3e} (scale {scale:.3g})"


def test_library_loads_and_version():
    k = _k()
    assert k.lib().tocvp_version() == 100


@pytest.mark.parametrize("M,N,K", [(30, 512, 512), (300, 1536, 512), (5000, 2048, 512),
                                   (1000, 128, 32), (7, 36, 4), (129, 100, 68),
                                   (20000, 256, 128), (210, 128, 2048)])
def test_gemm_plain_bias_act(M, N, K):
    k = _k()
    x, w, b = rnd("gx", (M, K)), rnd("gw", (N, K), "uniform", K ** -0.5), rnd("gb", (N,))
    ref = x @ w.t() + b
    close(k.linear(x.to(DEV), w.to(DEV), b.to(DEV)), ref)
    close(k.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=k.ACT_RELU), torch.relu(ref))
    close(k.linear(x.to(DEV), w.to(DEV), None, act=k.ACT_GELU), O.gelu(x @ w.t()))


def test_gemm_residual_and_flipped_rowvec():
    k = _k()
    B, w_, Ks, D, E = 3, 4, 7, 128, 512
    x = rnd("rx", (B, w_, Ks, D))
    w, b = rnd("rw", (E, D), "uniform", D ** -0.5), rnd("rb", (E,))
    pe = rnd("rpe", (w_, E))
    res = rnd("rres", (B, w_, Ks, E))
    ref = x @ w.t() + b + torch.flip(pe, dims=(0,))[None, :, None, :] + res
    got = k.linear(x.to(DEV), w.to(DEV), b.to(DEV), residual=res.to(DEV), rowvec=pe.to(DEV),
                   rv_div=Ks, rv_flip=True)
    close(got, ref)
    got2 = k.linear(x.to(DEV), w.to(DEV), b.to(DEV), rowvec=pe.to(DEV), rv_div=Ks, rv_flip=False)
    close(got2, x @ w.t() + b + pe[None, :, None, :])


def test_gemm_rejects_bad_arguments():
    k = _k()
    x = torch.zeros(4, 6, device=DEV)          # K % 4 != 0
    w = torch.zeros(8, 6, device=DEV)
    with pytest.raises(k.TocvpError):
        k.linear(x, w)
    with pytest.raises(k.TocvpError):          # CPU tensors are refused: no CPU path
        k.linear(torch.zeros(4, 8), torch.zeros(8, 8))


@pytest.mark.parametrize("rows,D,eps,with_add", [(4096 * 2, 32, 1e-5, True), (61, 128, 1e-3, False),
                                                 (300, 512, 1e-6, False), (5, 128, 1e-8, False),
                                                 (4096 + 13, 128, 1e-3, True), (77, 64, 1e-5, False),
                                                 (9, 32, 1e-5, False), (33, 256, 1e-5, False)])
def test_layernorm(rows, D, eps, with_add):
    k = _k()
    x, g, b = rnd("lx", (rows, D), "normal", 3.0), 1 + rnd("lg", (D,), "uniform", 0.3), rnd("lb", (D,))
    add = rnd("la", (4096, D)) if with_add else None
    xin = x + add.repeat(rows // 4096 + 1, 1)[:rows] if with_add else x
    ref = O.layer_norm(xin, g, b, eps)
    got = k.layer_norm(x.to(DEV), g.to(DEV), b.to(DEV), eps, add=add.to(DEV) if with_add else None)
    close(got, ref)


@pytest.mark.parametrize("B,H,Tq,Tk,dh", [(2, 8, 300, 300, 64), (3, 8, 70, 20, 64), (2, 4, 7, 7, 32),
                                          (2, 4, 30, 30, 32), (1, 8, 129, 33, 64)])
def test_mha(B, H, Tq, Tk, dh):
    k = _k()
    E = H * dh
    q, kk, v = rnd("aq", (B, Tq, E)), rnd("ak", (B, Tk, E)), rnd("av", (B, Tk, E))
    ref = O.attention(q, kk, v, H, dh ** -0.5)
    close(k.mha(q.to(DEV), kk.to(DEV), v.to(DEV), H, dh ** -0.5), ref)


def test_mha_fused_qkv_views_and_key_lengths():
    k = _k()
    B, H, T, dh = 3, 4, 20, 32
    E = H * dh
    qkv = rnd("fqkv", (B, T, 3 * E))
    lengths = torch.tensor([5, 12, 20], dtype=torch.int64)
    key_pad = torch.arange(1, T + 1)[None, :] > lengths[:, None]
    ref = O.attention(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], H, dh ** -0.5,
                      key_mask=key_pad)
    d = qkv.to(DEV)
    got = k.mha(d[..., :E], d[..., E:2 * E], d[..., 2 * E:], H, dh ** -0.5,
                key_len=lengths.to(torch.int32).to(DEV))
    close(got, ref)


@pytest.mark.parametrize("B,Ks,N", [(2, 7, 4096), (1, 30, 4096), (3, 24, 256), (40, 30, 4096)])
def test_slot_attn_iter(B, Ks, N):
    k = _k()
    D = 128
    q = rnd("sq", (B, Ks, D))
    kv = rnd("skv", (B, N, 2 * D))
    kk, v = kv[..., :D], kv[..., D:]
    scale, eps = D ** -0.5, 1e-8
    dots = (q @ kk.transpose(1, 2)) * scale
    attn = torch.softmax(dots, dim=1) + eps
    ref = (attn / attn.sum(-1, keepdim=True)) @ v
    d = kv.to(DEV)
    attn_out = torch.empty((B, Ks, N), device=DEV)
    got = k.slot_attn_iter(q.to(DEV), d[..., :D], d[..., D:], scale, eps, attn_out=attn_out)
    close(got, ref)
    close(attn_out, attn, tol=1e-5)
    # same iteration fed with the fp16 operand planes the fused kv projection writes (2^8 x = hi + lo)
    X = kv * 256.0
    hi = X.half()
    lo = (X - hi.float()).half()
    planes = torch.stack([hi, lo], dim=2).contiguous().to(DEV)            # (B, N, 2, 2 D)
    attn_p = torch.empty((B, Ks, N), device=DEV)
    got_p = k.slot_attn_iter_planes(q.to(DEV), planes, scale, eps, attn_out=attn_p)
    close(got_p, ref)
    close(attn_p, attn, tol=1e-5)
    # deterministic: the cross-workgroup reduction runs in record order whoever arrives last
    again = k.slot_attn_iter_planes(q.to(DEV), planes, scale, eps)
    assert torch.equal(again, got_p)


def test_gru_gates():
    k = _k()
    rows, D = 60, 128
    x, h = rnd("gx2", (rows, D)), rnd("gh2", (rows, D))
    wi, wh = rnd("gwi", (3 * D, D), "uniform", 0.1), rnd("gwh", (3 * D, D), "uniform", 0.1)
    bi, bh = rnd("gbi", (3 * D,), "uniform", 0.1), rnd("gbh", (3 * D,), "uniform", 0.1)
    ref = O.gru_cell(x, h, wi, wh, bi, bh)
    cell = torch.nn.GRUCell(D, D)
    with torch.no_grad():
        cell.weight_ih.copy_(wi), cell.weight_hh.copy_(wh), cell.bias_ih.copy_(bi), cell.bias_hh.copy_(bh)
        close(ref, cell(x, h), tol=1e-6)     # the oracle itself is torch's GRUCell
    gi = k.linear(x.to(DEV), wi.to(DEV), bi.to(DEV))
    gh = k.linear(h.to(DEV), wh.to(DEV), bh.to(DEV))
    close(k.gru_gates(gi, gh, h.to(DEV)), ref)


@pytest.mark.parametrize("C", [32, 128])
def test_pos_embed(C):
    k = _k()
    w, b = rnd("pw", (C, 4, 1, 1)), rnd("pb", (C,))
    ref = O.soft_pos_embed(w, b, (64, 64))
    close(k.pos_embed(w.to(DEV), b.to(DEV), 64, 64), ref, tol=2e-6)


def test_conv_in3():
    k = _k()
    vid = rnd("cvid", (2, 3, 3, 64, 64), "unit")          # (B, L, C, H, W): frames 1.. of a video
    w, b = rnd("cw0", (32, 3, 5, 5), "uniform", 0.2), rnd("cb0", (32,), "uniform", 0.1)
    x = vid[:, 1]                                           # strided view (B, 3, H, W)
    ref = torch.relu(F.conv2d(x, w, b, padding=2)).permute(0, 2, 3, 1)
    close(k.conv5x5_in3(vid.to(DEV)[:, 1], w.to(DEV), b.to(DEV)), ref)


@pytest.mark.parametrize("Cin,Cout,n,relu", [(32, 32, 3, True), (64, 64, 5, True), (128, 64, 1, False),
                                             (64, 64, 37, True)])
def test_conv5x5(Cin, Cout, n, relu):
    k = _k()
    x = rnd("cx", (n, 64, 64, Cin))
    w = rnd("cw", (Cout, Cin, 5, 5), "uniform", (25 * Cin) ** -0.5)
    b = rnd("cb", (Cout,), "uniform", 0.1)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=2)
    ref = (torch.relu(ref) if relu else ref).permute(0, 2, 3, 1)
    wp = k.pack_conv_weights(w.to(DEV))
    close(wp, w.permute(2, 3, 0, 1).reshape(25, Cout, Cin), tol=0)
    close(k.conv5x5(x.to(DEV), wp, b.to(DEV), relu=relu), ref)


def test_decoder_layer0_collapse_matches_direct_conv():
    """ relu(conv0(broadcast(s)+pos)) computed analytically == the reference's explicit path. """
    k = _k()
    n, D, C0 = 6, 128, 64
    slots = rnd("dslots", (n, D))
    pw, pb = rnd("dpw", (D, 4, 1, 1)), rnd("dpb", (D,))
    w0 = rnd("dw0", (C0, D, 5, 5), "uniform", (25 * D) ** -0.5)
    b0 = rnd("db0", (C0,), "uniform", 0.1)
    w1 = rnd("dw1", (64, C0, 5, 5), "uniform", (25 * C0) ** -0.5)
    b1 = rnd("db1", (64,), "uniform", 0.1)
    pos = O.soft_pos_embed(pw, pb, (64, 64))
    x0 = (slots[:, None, None, :] + pos[None]).permute(0, 3, 1, 2)
    a0 = torch.relu(F.conv2d(x0, w0, b0, padding=2))
    ref = torch.relu(F.conv2d(a0, w1, b1, padding=2)).permute(0, 2, 3, 1)

    pos_d = k.pos_embed(pw.to(DEV), pb.to(DEV), 64, 64)
    cpos = k.conv5x5(pos_d[None].contiguous(), k.pack_conv_weights(w0.to(DEV)), b0.to(DEV),
                     relu=False)[0]
    tapsum = k.dec_tapsum(w0.to(DEV))                                  # (25, C0, D)
    S = k.linear(slots.to(DEV), tapsum.reshape(25 * C0, D)).reshape(n, 25, C0)
    got = k.conv5x5_collapsed(cpos.contiguous(), S.contiguous(), k.pack_conv_weights(w1.to(DEV)),
                              b1.to(DEV), relu=True)
    close(got, ref, tol=3e-5)


@pytest.mark.parametrize("Fr,K", [(2, 7), (1, 30)])
def test_dec_tail(Fr, K):
    k = _k()
    x = rnd("tx", (Fr * K, 64, 64, 64))
    w = rnd("tw", (4, 64, 3, 3), "uniform", 0.1)
    b = rnd("tb", (4,), "uniform", 0.3)
    y = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1).reshape(Fr, K, 4, 64, 64)
    recons, alpha = y[:, :, :3], y[:, :, 3:]
    masks = torch.softmax(alpha, dim=1)
    imgs = (recons * masks).sum(1)
    g_imgs, g_recons, g_masks = k.dec_tail(x.to(DEV), w.to(DEV), b.to(DEV), Fr, K)
    close(g_recons, recons)
    close(g_masks, masks, tol=1e-5)
    close(g_imgs, imgs)


@pytest.mark.parametrize("Fr,K", [(2, 7), (1, 30)])
def test_decoder_tail_folded_into_the_last_layer(Fr, K):
    """ last 5 x 5 layer + tail: the tap products of the folded epilogue (36 per pixel, split-fp16 arithmetic) summed by
    tocvp_dec_tail_sum_f32 against (i) fp64 conv -> conv3x3 -> softmax -> compositing and (ii) the unfused kernels
    (f16x3 layer, exact-fp32 tail), for NHWC and for operand-plane input; image borders carry the zero padding of both
    convolutions """
    k = _k()
    n = Fr * K
    x = rnd("fx", (n, 64, 64, 64))
    x[0, :3, :5] = 0.0
    w3 = rnd("fw3", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5)
    b3 = rnd("fb3", (64,), "uniform", 0.1)
    wt = rnd("fwt", (4, 64, 3, 3), "uniform", 0.1)
    bt = rnd("fbt", (4,), "uniform", 0.3)
    y3 = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w3.double(), b3.double(), padding=2))
    y = F.conv2d(y3, wt.double(), bt.double(), padding=1).reshape(Fr, K, 4, 64, 64)
    ref_rec, ref_masks = y[:, :, :3], torch.softmax(y[:, :, 3:], dim=1)
    ref_imgs = (ref_rec * ref_masks).sum(1)
    xd, wf, bd = x.to(DEV), k.split_conv_weights_dec_f16x3(w3.to(DEV)), b3.to(DEV)
    taps = k.pack_tail_taps_f16x3(wt.to(DEV))
    # unfused
    y3_d = k.conv5x5_dec_f16x3(xd, wf, bd, relu=True)
    u_imgs, u_rec, u_masks = k.dec_tail(y3_d, wt.to(DEV), bt.to(DEV), Fr, K)
    # folded, NHWC input
    P = k.conv5x5_dec_f16x3_tail(xd, wf, bd, taps, relu=True)
    imgs, rec, masks = k.dec_tail_sum(P, bt.to(DEV), Fr, K)
    for got, un, ref, name in ((rec, u_rec, ref_rec, "recons"), (masks, u_masks, ref_masks, "masks"),
                               (imgs, u_imgs, ref_imgs, "imgs")):
        e_f = (got.cpu().double() - ref.reshape(got.shape)).abs().max().item()
        e_u = (un.cpu().double() - ref.reshape(un.shape)).abs().max().item()
        print(f"folded tail {name}: err {e_f:.2e} (unfused {e_u:.2e})")
        assert e_f < max(3.0 * e_u, 3e-6 * max(1.0, ref.abs().max().item()))
    # folded, operand-plane input written by a previous layer == the same through fp32 pass-major
    w2 = rnd("fw2", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5)
    wf2 = k.split_conv_weights_dec_f16x3(w2.to(DEV))
    mid_p = k.conv5x5_dec_f16x3(xd, wf2, bd, relu=True, pm_out=True, planes=True)
    mid_f = k.conv5x5_dec_f16x3(xd, wf2, bd, relu=True, pm_out=True, planes=False)
    P_p = k.conv5x5_dec_f16x3_tail(mid_p, wf, bd, taps, relu=True, pm_in=True, planes=True)
    P_f = k.conv5x5_dec_f16x3_tail(mid_f, wf, bd, taps, relu=True, pm_in=True, planes=False)
    assert torch.equal(P_p, P_f)


def test_text_embed():
    k = _k()
    tokens, _ = synth.synth_captions(3, max_len=20, lengths=[5, 12, 20], seed=3)
    te, pe = rnd("te", (50, 128)), rnd("tpe", (50, 128))
    g, b = 1 + rnd("tg", (128,), "uniform", 0.2), rnd("tb2", (128,), "uniform", 0.1)
    ref = O.layer_norm(te[tokens] + pe[:20][None], g, b, 1e-8) * (tokens != 0).unsqueeze(-1)
    got = k.text_embed(tokens.to(DEV), te.to(DEV), pe.to(DEV), g.to(DEV), b.to(DEV), 1e-8)
    close(got, ref)


@pytest.mark.parametrize("n", [3, 37])
def test_conv5x5_bf16x3_accuracy(n):
    """ split-bf16 conv: error relative to the fp64 result must be ~1e-5 of the output scale """
    k = _k()
    x = rnd("bx", (n, 64, 64, 64))
    w = rnd("bw", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5)
    b = rnd("bb", (64,), "uniform", 0.1)
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=2))
    ref = ref.permute(0, 2, 3, 1)
    ws = k.split_conv_weights_bf16(w.to(DEV))
    got = k.conv5x5_bf16x3(x.to(DEV), ws, b.to(DEV), relu=True)
    # weights-direct variant (fragment-order weights, no LDS weight image) must agree bit for bit:
    # same products in the same order
    got_wd = k.conv5x5_bf16x3(x.to(DEV), ws, b.to(DEV), relu=True,
                              wfrag=k.split_conv_weights_frag_bf16(w.to(DEV)))
    assert torch.equal(got, got_wd)
    err = (got.cpu().double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f"bf16x3 conv: max abs err {err:.3e} at scale {scale:.3g}")
    assert err < 4e-5 * scale
    # fp32 MFMA path on the same data for comparison
    got32 = k.conv5x5(x.to(DEV), k.pack_conv_weights(w.to(DEV)), b.to(DEV), relu=True)
    err32 = (got32.cpu().double() - ref).abs().max().item()
    assert err32 < 5e-6 * scale


@pytest.mark.parametrize("mode,tol", [("bf16x3", 3e-5), ("bf16x6", 2e-6)])
@pytest.mark.parametrize("M,N,K", [(30, 512, 512), (9600, 2048, 512), (301, 512, 2048)])
def test_gemm_split_bf16(mode, tol, M, N, K):
    """ split-bf16 GEMMs against an fp64 reference; bf16x6 must be fp32-class """
    k = _k()
    x, w, b = rnd("sx", (M, K)), rnd("sw", (N, K), "uniform", K ** -0.5), rnd("sb", (N,))
    res = rnd("sres", (M, N))
    ref = torch.relu(x.double() @ w.double().t() + b.double()) + res.double()
    got = k.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=k.ACT_RELU, residual=res.to(DEV),
                   precision=mode)
    err = (got.cpu().double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    got32 = k.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=k.ACT_RELU, residual=res.to(DEV))
    err32 = (got32.cpu().double() - ref).abs().max().item()
    print(f"{mode} gemm {M}x{N}x{K}: err {err:.2e} (fp32 mfma {err32:.2e}) at scale {scale:.3g}")
    assert err < tol * scale


@pytest.mark.parametrize("N,H,W", [(5, 64, 64), (2, 32, 48)])
def test_psnr_ssim_kernel(N, H, W):
    """ metric kernel (fused clamp) vs the float64 restatement of piqa's PSNR / SSIM definitions """
    from oracle import metrics_oracle as MO
    k = _k()
    x = rnd("mx", (N, 3, H, W), "unit", 1.3) - 0.1            # exercises the clamp on both sides
    y = (x + rnd("my", (N, 3, H, W), "normal", 0.05)).clamp(-0.2, 1.2)
    y[0] = x[0]                                               # identical pair: ssim = 1, psnr = 80 dB
    p, s = k.psnr_ssim(x.to(DEV), y.to(DEV), clamp01=True)
    xc, yc = x.clamp(0, 1), y.clamp(0, 1)
    close(p, MO.psnr(xc, yc), tol=2e-6)
    close(s, MO.ssim(xc, yc), tol=2e-5)
    assert abs(s[0].item() - 1.0) < 1e-5 and abs(p[0].item() - 80.0) < 1e-3


def test_metric_tracker_surface(tmp_path):
    from textocvp_amd.metrics import MetricTracker
    from oracle import metrics_oracle as MO
    mt = MetricTracker(metrics=["psnr", "ssim"])
    a, b = rnd("ta", (2, 4, 3, 64, 64), "unit"), rnd("tb", (2, 4, 3, 64, 64), "unit")
    mt.accumulate(a.to(DEV), b.to(DEV))
    mt.accumulate(b.to(DEV), a.to(DEV))
    mt.aggregate()
    res = mt.get_results()
    assert res["psnr"]["framewise"].shape == (4,)
    ref = MO.ssim(a.reshape(8, 3, 64, 64), b.reshape(8, 3, 64, 64)).mean().item()
    assert abs(res["ssim"]["mean"] - ref) < 1e-5
    mt.save_results(str(tmp_path), "r")
    assert (tmp_path / "results" / "r" / "results.json").exists()
    with pytest.raises(NotImplementedError):
        MetricTracker(metrics=["lpips"])


@pytest.mark.parametrize("Cin,Cout,S,up", [(128, 64, 24, False), (64, 128, 16, True), (192, 32, 40, False),
                                           (128, 64, 16, False), (64, 32, 16, False), (128, 64, 8, True)])   # 16-wide: narrow tiles
def test_conv3x3_scale_shift_upsample(Cin, Cout, S, up):
    k = _k()
    x = rnd("c3x", (2, S, S, Cin))
    w = rnd("c3w", (Cout, Cin, 3, 3), "uniform", (9 * Cin) ** -0.5)
    sc, sf = 1 + rnd("c3s", (Cout,), "uniform", 0.3), rnd("c3f", (Cout,), "uniform", 0.2)
    xin = x.permute(0, 3, 1, 2)
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, w, None, padding=1) * sc[None, :, None, None] + sf[None, :, None, None]
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    wp = k.pack_conv_weights(w.to(DEV))
    got = k.conv3x3(x.to(DEV), wp, sc.to(DEV), sf.to(DEV), relu=True, upsample2=up)
    close(got, ref)
    # split-fp16 form: fp32-class, compared with the fp64 result
    ref64 = F.conv2d(xin.double(), w.double(), None, padding=1) * sc.double()[None, :, None, None] \
        + sf.double()[None, :, None, None]
    ref64 = torch.relu(ref64).permute(0, 2, 3, 1)
    got16 = k.conv3x3(x.to(DEV), wp, sc.to(DEV), sf.to(DEV), relu=True, upsample2=up, precision="f16x3")
    err16 = (got16.cpu().double() - ref64).abs().max().item()
    err32 = (got.cpu().double() - ref64).abs().max().item()
    assert err16 < max(3 * err32, 5e-6 * ref64.abs().max().item()), (err16, err32)


def test_slot_composite_and_bilinear():
    k = _k()
    dec = rnd("scd", (2, 24, 50, 97))
    feats, alpha = dec[..., :-1], dec[..., -1:]
    a = torch.softmax(alpha, dim=1)
    rec, masks = k.slot_composite(dec.to(DEV))
    close(rec, (feats * a).sum(1))
    close(masks, a[..., 0], tol=1e-6)
    # padded rows (the producing GEMM rounds its width up to a multiple of 32): padding is ignored
    padded = torch.cat([dec, torch.full((2, 24, 50, 31), 7.0)], dim=-1)
    rec_p, masks_p = k.slot_composite(padded.to(DEV), feat_dim=96)
    assert torch.equal(rec_p, rec) and torch.equal(masks_p, masks)
    x = rnd("bil", (2, 48, 48, 32))
    ref = F.interpolate(x[..., :3].permute(0, 3, 1, 2), size=(42, 42), mode="bilinear", align_corners=False)
    close(k.bilinear_resize_nhwc_to_nchw(x.to(DEV), 3, 42, 42), ref, tol=2e-6)
    ref_up = F.interpolate(x[..., :3].permute(0, 3, 1, 2), size=(60, 60), mode="bilinear", align_corners=False)
    close(k.bilinear_resize_nhwc_to_nchw(x.to(DEV), 3, 60, 60), ref_up, tol=2e-6)


@pytest.mark.parametrize("M,N,K", [(30, 512, 512), (9600, 2048, 512), (301, 512, 2048)])
def test_gemm_f16x3_is_fp32_class(M, N, K):
    """ two fp16 planes, 3 products: must be as accurate as the exact-fp32 MFMA kernel """
    k = _k()
    x, w, b = rnd("sx", (M, K)), rnd("sw", (N, K), "uniform", K ** -0.5), rnd("sb", (N,))
    x[0, :8] = torch.tensor([1e-6, -3e-5, 2e-4, 6e-5, -1e-7, 5e-3, 90.0, -200.0])   # tiny + large values
    x[1] = x[1] * 1e-3                       # a whole row of small activations (fp16-subnormal lo planes)
    res = rnd("sres", (M, N))
    ref = torch.relu(x.double() @ w.double().t() + b.double()) + res.double()
    got = k.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=k.ACT_RELU, residual=res.to(DEV),
                   precision="f16x3")
    got32 = k.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=k.ACT_RELU, residual=res.to(DEV))
    err = (got.cpu().double() - ref).abs().max().item()
    err32 = (got32.cpu().double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print(f"f16x3 gemm {M}x{N}x{K}: err {err:.2e} (fp32 mfma {err32:.2e}) at scale {scale:.3g}")
    assert err < max(2.5 * err32, 2e-6 * scale)
    # the small row on its own, without the residual: relative accuracy must hold at its own scale
    small_ref = torch.relu(x[1:2].double() @ w.double().t() + b.double())
    small = k.linear(x[1:2].to(DEV), w.to(DEV), b.to(DEV), act=k.ACT_RELU, precision="f16x3")
    assert (small.cpu().double() - small_ref).abs().max().item() < 3e-6


@pytest.mark.parametrize("M,N,K", [(300, 512, 2048), (30, 512, 2048), (300, 2048, 512), (270, 512, 512),
                                   (301, 1536, 512), (12, 512, 64), (600, 512, 2048), (5, 128, 256)])
def test_gemm_f16x3_split_k_of_the_small_batches(M, N, K, monkeypatch):
    """ skinny f16x3 GEMMs split K over idle CUs (gemm_bf16.hip, SK): fp32-class against fp64, the same value as the
    unsplit kernel to rounding, identical on every repetition (slices are added in slice order by whichever
    workgroup arrives last), epilogue forms intact, arrival counters back at zero """
    k = _k()
    x, w, b = rnd("kx", (M, K)), rnd("kw", (N, K), "uniform", K ** -0.5), rnd("kb", (N,))
    res, rv = rnd("kres", (M, N)), rnd("krv", (7, N))
    xd, wd, bd, rd, rvd = (t.to(DEV) for t in (x, w, b, res, rv))
    ref = torch.relu(x.double() @ w.double().t() + b.double()) + res.double()
    monkeypatch.setattr(k, "_GEMM_KSPLIT", False)
    plain = k.linear(xd, wd, bd, act=k.ACT_RELU, residual=rd, precision="f16x3")
    plain_rv = k.linear(xd, wd, bd, rowvec=rvd, rv_div=3, precision="f16x3")
    monkeypatch.setattr(k, "_GEMM_KSPLIT", True)
    got = k.linear(xd, wd, bd, act=k.ACT_RELU, residual=rd, precision="f16x3")
    scale = ref.abs().max().item()
    err, err_plain = (got.cpu().double() - ref).abs().max().item(), (plain.cpu().double() - ref).abs().max().item()
    print(f"split-K gemm {M}x{N}x{K}: err {err:.2e} (unsplit {err_plain:.2e}) at scale {scale:.3g}")
    assert err < max(2.5 * err_plain, 2e-6 * scale)
    for _ in range(20):
        assert torch.equal(k.linear(xd, wd, bd, act=k.ACT_RELU, residual=rd, precision="f16x3"), got)
    close(k.linear(xd, wd, bd, rowvec=rvd, rv_div=3, precision="f16x3"), plain_rv, tol=2e-6)
    planes = k.linear(xd, wd, bd, precision="f16x3", out_split=22)
    planes_ref = k.linear(xd, wd, bd, precision="f16x3")
    hi_lo = planes.planes.float().view(M, 2, N).sum(1) / 256.0
    close(hi_lo, planes_ref, tol=2e-6)
    wk = k._ksplit_workspace(xd.device)[0]
    assert int(wk[:1024].view(torch.int32).abs().sum()) == 0


def test_absmax_reduction_of_the_range_check():
    """ tocvp_absmax_f32 (the reduction behind the checked pass): exact max |x| on odd sizes, strided views,
    negative extremes, zeros; NaN reads as inf so that a check can never pass on it """
    k = _k()
    for n in (1, 63, 257, 100003, 4 * 1024 * 1024 + 5):
        x = rnd(f"am{n}", (n,), "normal", 3.0)
        x[n // 2] = -1234.5
        assert k.absmax(x.to(DEV)) == 1234.5
    y = rnd("am2d", (300, 96)).to(DEV)
    assert k.absmax(y[:, 7:40]) == float(y[:, 7:40].abs().max())
    assert k.absmax(torch.zeros(1000, device=DEV)) == 0.0
    z = torch.ones(5000, device=DEV)
    z[4321] = float("nan")
    assert k.absmax(z) == float("inf")


def test_gemm_f16x3_range_behaviour(monkeypatch):
    """ outside |x| < 255 the fp16 planes saturate: 11-bit accuracy up to 511, finite (never inf/nan)
    beyond; TOCVP_CHECK_RANGE turns the silent saturation into an error """
    k = _k()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, 128, generator=g)
    w = torch.randn(64, 128, generator=g) * 0.1
    x[3, 7] = 400.0
    x[9, 1] = -5000.0
    ref = x.double() @ w.double().t()
    got = k.linear(x.to(DEV), w.to(DEV), precision="f16x3").cpu().double()
    assert torch.isfinite(got).all()
    rows = [r for r in range(64) if r not in (3, 9)]
    assert (got[rows] - ref[rows]).abs().max().item() < 1e-5
    assert (got[3] - ref[3]).abs().max().item() < 400 * 0.1 * 4 * 2 ** -11
    monkeypatch.setattr(k, "_CHECK_RANGE", True)
    with pytest.raises(k.TocvpRangeError, match="out of the fp16-plane range"):
        k.linear(x.to(DEV), w.to(DEV), precision="f16x3")
    w_big = w.clone()
    w_big[5, 5] = 64.0
    with pytest.raises(k.TocvpRangeError, match="weight out of the fp16-plane range"):
        k.linear(x[:3].contiguous().to(DEV), w_big.to(DEV), precision="f16x3")
    k.linear(x[:3].contiguous().to(DEV), w.to(DEV), precision="f16x3")


@pytest.mark.parametrize("n", [3, 37])
def test_conv5x5_f16f8_accuracy(n):
    """ hybrid f16 + fp8 conv: f16 main term + e4m3 cross terms -> ~1e-5 of the output scale """
    k = _k()
    x = rnd("bx", (n, 64, 64, 64))
    x[0, :4, :4] = 0.0                                   # exact zeros, tiny and large activations
    x[0, 5, 5, :8] = torch.tensor([1e-6, -3e-5, 2e-4, 1e-3, 40.0, -90.0, 200.0, 0.25])
    w = rnd("bw", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5)
    b = rnd("bb", (64,), "uniform", 0.1)
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=2))
    ref = ref.permute(0, 2, 3, 1)
    wi = k.split_conv_weights_f16f8(w.to(DEV))
    got = k.conv5x5_f16f8(x.to(DEV), wi, b.to(DEV), relu=True)
    err = (got.cpu().double() - ref).abs()
    scale = ref[1:].abs().max().item()
    print(f"f16f8 conv: max abs err {err[1:].max().item():.3e} at scale {scale:.3g} "
          f"(image with planted values: {err[0].max().item():.3e})")
    assert err[1:].max().item() < 8e-5 * scale
    assert err[0].max().item() < 8e-5 * max(scale, ref[0].abs().max().item())
    # no ReLU: the sign structure must survive too
    got_lin = k.conv5x5_f16f8(x.to(DEV), wi, b.to(DEV), relu=False)
    ref_lin = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=2).permute(0, 2, 3, 1)
    assert (got_lin.cpu().double() - ref_lin).abs().max().item() < 8e-5 * ref_lin.abs().max().item()


def test_conv5x5_f16f8_collapsed_input():
    """ layer-1 mode of the hybrid conv: input = relu(cpos + S[border class]) built on the fly """
    k = _k()
    n, D, C0 = 6, 128, 64
    slots = rnd("dslots", (n, D))
    pw, pb = rnd("dpw", (D, 4, 1, 1)), rnd("dpb", (D,))
    w0 = rnd("dw0", (C0, D, 5, 5), "uniform", (25 * D) ** -0.5)
    b0 = rnd("db0", (C0,), "uniform", 0.1)
    w1 = rnd("dw1", (64, C0, 5, 5), "uniform", (25 * C0) ** -0.5)
    b1 = rnd("db1", (64,), "uniform", 0.1)
    pos = O.soft_pos_embed(pw, pb, (64, 64))
    x0 = (slots[:, None, None, :] + pos[None]).permute(0, 3, 1, 2)
    a0 = torch.relu(F.conv2d(x0, w0, b0, padding=2))
    ref = torch.relu(F.conv2d(a0.double(), w1.double(), b1.double(), padding=2)).permute(0, 2, 3, 1)
    pos_d = k.pos_embed(pw.to(DEV), pb.to(DEV), 64, 64)
    cpos = k.conv5x5(pos_d[None].contiguous(), k.pack_conv_weights(w0.to(DEV)), b0.to(DEV),
                     relu=False)[0]
    tapsum = k.dec_tapsum(w0.to(DEV))
    S = k.linear(slots.to(DEV), tapsum.reshape(25 * C0, D)).reshape(n, 25, C0)
    got = k.conv5x5_f16f8(None, k.split_conv_weights_f16f8(w1.to(DEV)), b1.to(DEV), relu=True,
                          collapsed=(cpos.contiguous(), S.contiguous()))
    err = (got.cpu().double() - ref).abs().max().item()
    print(f"f16f8 collapsed conv: max abs err {err:.3e} at scale {ref.abs().max().item():.3g}")
    assert err < 8e-5 * ref.abs().max().item()


def _to_pass_major(x):
    """ NHWC (n, H, W, 64) -> the pass-major layout (n, 4, H, W, 16) stored in a (n, H, W, 64) buffer """
    n, H, W, C = x.shape
    return x.reshape(n, H, W, 4, 16).permute(0, 3, 1, 2, 4).contiguous().reshape(n, H, W, C)


@pytest.mark.parametrize("n", [3, 37])
def test_conv5x5_dec_f16x3_is_fp32_class(n):
    """ default decoder conv (3 fp16 products): as accurate as the exact-fp32 MFMA kernel, every layout """
    k = _k()
    x = rnd("bx", (n, 64, 64, 64))
    x[0, :4, :4] = 0.0                                   # exact zeros, tiny and large activations
    x[0, 5, 5, :8] = torch.tensor([1e-6, -3e-5, 2e-4, 1e-3, 40.0, -90.0, 200.0, 0.25])
    x[1] = x[1] * 1e-3                                   # a whole image of small activations
    w = rnd("bw", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5)
    b = rnd("bb", (64,), "uniform", 0.1)
    lin = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=2).permute(0, 2, 3, 1)
    ref = torch.relu(lin)
    wf = k.split_conv_weights_dec_f16x3(w.to(DEV))
    got = k.conv5x5_dec_f16x3(x.to(DEV), wf, b.to(DEV), relu=True)
    got32 = k.conv5x5(x.to(DEV), k.pack_conv_weights(w.to(DEV)), b.to(DEV), relu=True)
    err = (got.cpu().double() - ref).abs()
    err32 = (got32.cpu().double() - ref).abs().max().item()
    scale = ref[2:].abs().max().item()
    print(f"f16x3 decoder conv: max abs err {err.max().item():.3e} (fp32 MFMA {err32:.3e}) at scale {scale:.3g}; "
          f"small image {err[1].max().item():.3e}")
    assert err.max().item() < max(2.5 * err32, 2e-6 * scale)
    assert err[1].max().item() < 3e-6                    # relative accuracy at the small image's own scale
    got_lin = k.conv5x5_dec_f16x3(x.to(DEV), wf, b.to(DEV), relu=False)
    assert (got_lin.cpu().double() - lin).abs().max().item() < max(2.5 * err32, 2e-6 * lin.abs().max().item())
    # pass-major input / output: same values, other layout
    xpm = _to_pass_major(x).to(DEV)
    for pm_in, pm_out in ((True, False), (False, True), (True, True)):
        y = k.conv5x5_dec_f16x3(xpm if pm_in else x.to(DEV), wf, b.to(DEV), relu=True, pm_in=pm_in, pm_out=pm_out)
        want = _to_pass_major(got.cpu()) if pm_out else got.cpu()
        assert torch.equal(y.cpu(), want), (pm_in, pm_out)


@pytest.mark.parametrize("n,HW", [(3, 64), (21, 64), (2, 128)])
def test_conv5x5_dec_f16x3_operand_planes_between_layers(n, HW):
    """ two layers chained through fp16 operand planes (producer epilogue writes [Xh | Xl] per pixel and pass, the
    consumer stages them by LDS-DMA) == the same layers chained through fp32 pass-major buffers, bit for bit, and a
    three-layer chain planes -> planes -> NHWC as the decoder runs it; image borders, saturating values and exact
    zeros included; 128 x 128 is the largest image whose DMA source offsets fit their 16-bit form """
    k = _k()
    x = rnd(f"px{HW}", (n, HW, HW, 64))
    x[0, :4, :4] = 0.0
    x[0, 5, 5, :8] = torch.tensor([1e-6, -3e-5, 2e-4, 1e-3, 40.0, -90.0, 200.0, 0.25])
    x[1] = x[1] * 1e-3
    ws = [rnd(f"pw{i}", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5) for i in range(3)]
    ws[0] *= 40.0                                         # drives some layer-1 outputs past the fp16-plane range (255)
    bs = [rnd(f"pb{i}", (64,), "uniform", 0.1) for i in range(3)]
    wf = [k.split_conv_weights_dec_f16x3(w.to(DEV)) for w in ws]
    bd = [b.to(DEV) for b in bs]
    xd = x.to(DEV)

    def chain(planes):
        y1 = k.conv5x5_dec_f16x3(xd, wf[0], bd[0], relu=True, pm_out=True, planes=planes)
        y2 = k.conv5x5_dec_f16x3(y1, wf[1], bd[1], relu=True, pm_in=True, pm_out=True, planes=planes)
        return y1, k.conv5x5_dec_f16x3(y2, wf[2], bd[2], relu=True, pm_in=True, planes=planes)

    y1_f, y3_f = chain(False)
    y1_p, y3_p = chain(True)
    assert torch.equal(y3_p, y3_f)
    assert float(y1_f.max()) > 255.0                     # the saturating case is really in the data
    # the planes buffer itself: hi + lo of 2^8 y, saturated at the fp16 range, per pixel and pass
    pl = y1_p.view(torch.float16).view(n, 4, HW, HW, 2, 16).float()
    want = (y1_f.view(n, 4, HW, HW, 16) * 256.0).clamp(-65504.0, 65504.0)
    assert (pl.sum(4) - want).abs().max().item() <= 2.0 ** -10 * 256.0 * 255.0 * 2.0 ** -11
    # against fp64 through all three layers (inputs of layer 2 / 3 saturate where layer 1 exceeded the range, so the
    # reference clamps the same way)
    ref = x.double()
    for li, (w, b) in enumerate(zip(ws, bs)):
        ref = torch.relu(F.conv2d(ref.permute(0, 3, 1, 2), w.double(), b.double(), padding=2).permute(0, 2, 3, 1))
        if li < 2:
            ref = ref.clamp(max=65504.0 / 256.0)
    got = y3_p.cpu().double()
    assert (got - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


def test_conv5x5_dec_f16x3_collapsed_input_and_range():
    """ layer-1 mode of the default decoder conv + the weight range check at split time """
    k = _k()
    n, D, C0 = 6, 128, 64
    slots = rnd("dslots", (n, D))
    pw, pb = rnd("dpw", (D, 4, 1, 1)), rnd("dpb", (D,))
    w0 = rnd("dw0", (C0, D, 5, 5), "uniform", (25 * D) ** -0.5)
    b0 = rnd("db0", (C0,), "uniform", 0.1)
    w1 = rnd("dw1", (64, C0, 5, 5), "uniform", (25 * C0) ** -0.5)
    b1 = rnd("db1", (64,), "uniform", 0.1)
    pos = O.soft_pos_embed(pw, pb, (64, 64))
    x0 = (slots[:, None, None, :] + pos[None]).permute(0, 3, 1, 2)
    a0 = torch.relu(F.conv2d(x0.double(), w0.double(), b0.double(), padding=2))
    ref = torch.relu(F.conv2d(a0, w1.double(), b1.double(), padding=2)).permute(0, 2, 3, 1)
    pos_d = k.pos_embed(pw.to(DEV), pb.to(DEV), 64, 64)
    cpos = k.conv5x5(pos_d[None].contiguous(), k.pack_conv_weights(w0.to(DEV)), b0.to(DEV), relu=False)[0]
    tapsum = k.dec_tapsum(w0.to(DEV))
    S = k.linear(slots.to(DEV), tapsum.reshape(25 * C0, D)).reshape(n, 25, C0)
    got = k.conv5x5_dec_f16x3(None, k.split_conv_weights_dec_f16x3(w1.to(DEV)), b1.to(DEV), relu=True,
                              collapsed=(cpos.contiguous(), S.contiguous()))
    err = (got.cpu().double() - ref).abs().max().item()
    print(f"f16x3 collapsed conv: max abs err {err:.3e} at scale {ref.abs().max().item():.3g}")
    assert err < 3e-6 * ref.abs().max().item()
    w_big = w1.clone()
    w_big[3, 5, 2, 2] = 70.0
    with pytest.raises(k.TocvpError, match="weight out of the fp16-plane range"):
        k.split_conv_weights_dec_f16x3(w_big.to(DEV))


@pytest.mark.parametrize("n,HW", [(3, 64), (37, 64), (3, 128)])
def test_conv5x5_dec_wino_is_fp32_class(n, HW):
    """ the Winograd form of the decoder conv (vertical F(4, 5), split-fp16 products): same error class against fp64 as
    the direct split-fp16 kernel; NHWC and x 16 pass-major layouts; borders, exact zeros, tiny and large activations """
    k = _k()
    x = rnd(f"wx{HW}", (n, HW, HW, 64))
    x[0, :4, :4] = 0.0
    x[0, 5, 5, :8] = torch.tensor([1e-6, -3e-5, 2e-4, 1e-3, 40.0, -90.0, 200.0, 0.25])
    x[1] = x[1] * 1e-3
    w = rnd("ww", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5)
    b = rnd("wb", (64,), "uniform", 0.1)
    lin = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=2).permute(0, 2, 3, 1)
    ref = torch.relu(lin)
    wp = k.split_conv_weights_wino_f16x3(w.to(DEV))
    got = k.conv5x5_dec_wino(x.to(DEV), wp, b.to(DEV), relu=True)
    direct = k.conv5x5_dec_f16x3(x.to(DEV), k.split_conv_weights_dec_f16x3(w.to(DEV)), b.to(DEV), relu=True)
    err = (got.cpu().double() - ref).abs()
    err_d = (direct.cpu().double() - ref).abs().max().item()
    scale = ref[2:].abs().max().item()
    print(f"Winograd decoder conv {HW}: max abs err {err.max().item():.3e} (direct f16x3 {err_d:.3e}) at scale {scale:.3g}; "
          f"small image {err[1].max().item():.3e}")
    assert err.max().item() < max(4.0 * err_d, 3e-6 * scale)      # outliers (200 next to O(1)) cost the transform ~3x
    assert err[1].max().item() < 3e-6
    got_lin = k.conv5x5_dec_wino(x.to(DEV), wp, b.to(DEV), relu=False)
    assert (got_lin.cpu().double() - lin).abs().max().item() < max(4.0 * err_d, 3e-6 * lin.abs().max().item())
    # x 16 pass-major out, then in: same values in the other layout / a second layer on top
    y16 = k.conv5x5_dec_wino(x.to(DEV), wp, b.to(DEV), relu=True, out_mode=1)
    assert torch.equal(y16.cpu().reshape(n, HW, HW, 64), _to_pass_major(got.cpu() * 16.0))
    w2 = rnd("ww2", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5)
    wp2 = k.split_conv_weights_wino_f16x3(w2.to(DEV))
    z = k.conv5x5_dec_wino(y16, wp2, b.to(DEV), relu=True, in_mode=0)
    z_ref = torch.relu(F.conv2d(got.cpu().double().permute(0, 3, 1, 2), w2.double(), b.double(), padding=2)).permute(0, 2, 3, 1)
    assert (z.cpu().double() - z_ref).abs().max().item() < 3e-6 * z_ref.abs().max().item()
    assert torch.equal(z, k.conv5x5_dec_wino(got, wp2, b.to(DEV), relu=True, in_mode=2))
    # operand planes out == the planes the direct kernel writes for the same values
    pl = k.conv5x5_dec_wino(x.to(DEV), wp, b.to(DEV), relu=True, out_mode=2)
    plf = pl.view(torch.float16).view(n, 4, HW, HW, 2, 16).float().sum(4)
    want = (got.cpu().view(n, HW, HW, 4, 16).permute(0, 3, 1, 2, 4) * 256.0).clamp(-65504.0, 65504.0)
    assert (plf.cpu() - want).abs().max().item() <= 2.0 ** -10 * 256.0 * 255.0 * 2.0 ** -11


def test_conv5x5_dec_wino_rows_in_fours_and_one_image():
    """ the Winograd entry tiles four rows at a time: heights that are a multiple of 4 but not of 8 (the direct kernel's
    tile), one and nine images (grids that do not fill the eight-XCD rounding), a 4-row image whose every tile touches both
    the top and the bottom border """
    k = _k()
    w = rnd("w4w", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5)
    b = rnd("w4b", (64,), "uniform", 0.1)
    wp = k.split_conv_weights_wino_f16x3(w.to(DEV))
    for n, H, W in ((1, 20, 64), (9, 4, 64), (2, 12, 128)):
        x = rnd(f"w4x{H}", (n, H, W, 64))
        ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=2)).permute(0, 2, 3, 1)
        got = k.conv5x5_dec_wino(x.to(DEV), wp, b.to(DEV), relu=True)
        err = (got.cpu().double() - ref).abs().max().item()
        print(f"Winograd conv {n} x {H} x {W}: max abs err {err:.2e} at scale {ref.abs().max().item():.3g}")
        assert err < 3e-6 * ref.abs().max().item(), (n, H, W)


def test_conv5x5_dec_wino_collapsed_tail_and_wide_weights():
    """ first-layer (collapsed) mode, the folded decoder tail in the epilogue, and weights far outside the direct
    kernel's |w| < 63 range (the per-row scales are picked from the weights) """
    k = _k()
    n, D, C0 = 6, 128, 64
    slots = rnd("dslots", (n, D))
    pw, pb = rnd("dpw", (D, 4, 1, 1)), rnd("dpb", (D,))
    w0 = rnd("dw0", (C0, D, 5, 5), "uniform", (25 * D) ** -0.5)
    b0 = rnd("db0", (C0,), "uniform", 0.1)
    w1 = rnd("dw1", (64, C0, 5, 5), "uniform", (25 * C0) ** -0.5)
    b1 = rnd("db1", (64,), "uniform", 0.1)
    pos = O.soft_pos_embed(pw, pb, (64, 64))
    x0 = (slots[:, None, None, :] + pos[None]).permute(0, 3, 1, 2)
    a0 = torch.relu(F.conv2d(x0.double(), w0.double(), b0.double(), padding=2))
    ref = torch.relu(F.conv2d(a0, w1.double(), b1.double(), padding=2)).permute(0, 2, 3, 1)
    pos_d = k.pos_embed(pw.to(DEV), pb.to(DEV), 64, 64)
    cpos = k.conv5x5(pos_d[None].contiguous(), k.pack_conv_weights(w0.to(DEV)), b0.to(DEV), relu=False)[0]
    tapsum = k.dec_tapsum(w0.to(DEV))
    S = k.linear(slots.to(DEV), tapsum.reshape(25 * C0, D)).reshape(n, 25, C0)
    wp = k.split_conv_weights_wino_f16x3(w1.to(DEV))
    got = k.conv5x5_dec_wino(None, wp, b1.to(DEV), relu=True, collapsed=(cpos.contiguous(), S.contiguous()))
    err = (got.cpu().double() - ref).abs().max().item()
    print(f"Winograd collapsed conv: max abs err {err:.3e} at scale {ref.abs().max().item():.3g}")
    assert err < 3e-6 * ref.abs().max().item()
    # folded tail: products of this layer's output with the tap matrix == the direct kernel's, to rounding
    Fr, K = 2, 3
    wt = rnd("fwt", (4, 64, 3, 3), "uniform", 0.1)
    bt = rnd("fbt", (4,), "uniform", 0.3)
    taps = k.pack_tail_taps_f16x3(wt.to(DEV))
    P = k.conv5x5_dec_wino(got, wp, b1.to(DEV), relu=True, in_mode=2, out_mode=3, tail_taps=taps)
    imgs, rec, masks = k.dec_tail_sum(P, bt.to(DEV), Fr, K)
    y3 = torch.relu(F.conv2d(got.cpu().double().permute(0, 3, 1, 2), w1.double(), b1.double(), padding=2))
    y = F.conv2d(y3, wt.double(), bt.double(), padding=1).reshape(Fr, K, 4, 64, 64)
    ref_rec, ref_masks = y[:, :, :3], torch.softmax(y[:, :, 3:], dim=1)
    ref_imgs = (ref_rec * ref_masks).sum(1)
    for g_, r_, name in ((rec, ref_rec, "recons"), (masks, ref_masks, "masks"), (imgs, ref_imgs, "imgs")):
        e = (g_.cpu().double() - r_.reshape(g_.shape)).abs().max().item()
        print(f"Winograd folded tail {name}: err {e:.2e}")
        assert e < 3e-6 * max(1.0, r_.abs().max().item())
    # inputs of any magnitude (data gradients): the operand scale follows max |x| measured on the device; ReLU gate in the store
    gsmall = rnd("wgs", (3, 64, 64, 64)) * 3e-6
    gsmall[0, 7, 9, :4] = torch.tensor([4e-3, -2e-9, 1e-12, 0.0])
    gate = rnd("wgate", (3, 64, 64, 64))
    zb = torch.zeros(64)
    got_g = k.conv5x5_dec_wino(gsmall.to(DEV), wp, zb.to(DEV), relu=False, auto_scale=True, gate=gate.to(DEV))
    ref_g = F.conv2d(gsmall.permute(0, 3, 1, 2).double(), w1.double(), None, padding=2).permute(0, 2, 3, 1) * (gate > 0)
    err_g = (got_g.cpu().double() - ref_g).abs().max().item()
    print(f"Winograd conv on 1e-6-sized input with device-side scale: err {err_g:.2e} at scale {ref_g.abs().max().item():.2e}")
    assert err_g < 1e-5 * ref_g.abs().max().item()           # a 1300 x outlier sets the scale; bf16x3 (the form it replaces): ~1e-4
    assert torch.equal(got_g.cpu() == 0, (gate <= 0) | (got_g.cpu() == 0))
    # weights of any range
    w_big = w1 * 3000.0
    got_big = k.conv5x5_dec_wino(x0.permute(0, 2, 3, 1).contiguous().to(DEV)[:, :, :, :64].contiguous(),
                                 k.split_conv_weights_wino_f16x3(w_big.to(DEV)), b1.to(DEV), relu=False)
    ref_big = F.conv2d(x0[:, :64].double(), w_big.double(), b1.double(), padding=2).permute(0, 2, 3, 1)
    assert (got_big.cpu().double() - ref_big).abs().max().item() < 3e-6 * ref_big.abs().max().item()


@pytest.mark.parametrize("M", [333, 5003])
def test_f16_operand_planes_from_producers(M):
    """ LayerNorm / attention / GEMM epilogues emitting fp16 operand planes (2^8 x, hi + lo) feed the
    f16x3 GEMM directly: same result as splitting inside the GEMM.  M = 5003 is large enough for the
    256x128-tile DMA kernel (gemm_f16_planes_kernel), with a ragged last row block """
    k = _k()
    D, Hd = 512, 2048
    x = rnd("px", (M, D))
    g, b = 1 + rnd("pg", (D,), "uniform", 0.2), rnd("pb", (D,), "uniform", 0.1)
    w1, b1 = rnd("pw1", (Hd, D), "uniform", D ** -0.5), rnd("pb1", (Hd,), "uniform", 0.1)
    w2, b2 = rnd("pw2", (D, Hd), "uniform", Hd ** -0.5), rnd("pb2", (D,), "uniform", 0.1)
    xd = x.to(DEV)
    # reference chain in fp64
    ln = torch.nn.functional.layer_norm(x.double(), (D,), g.double(), b.double(), 1e-6)
    hid = torch.relu(ln @ w1.double().t() + b1.double())
    ref = hid @ w2.double().t() + b2.double() + x.double()
    with k.gemm_precision("f16x3"):
        planes = k.layer_norm(xd, g.to(DEV), b.to(DEV), 1e-6, split=22)
        assert isinstance(planes, k.SplitAct) and planes.planes.dtype == torch.float16
        assert planes.nsplit == 22 and planes.planes.shape == (M, 2, D)
        rebuilt = planes.planes.float().sum(dim=1) / 256.0
        assert (rebuilt.cpu().double() - ln).abs().max().item() < 2e-6
        h = k.linear(planes, w1.to(DEV), b1.to(DEV), act=k.ACT_RELU, out_split=22)
        assert isinstance(h, k.SplitAct) and h.planes.shape == (M, 2, Hd)
        got = k.linear(h, w2.to(DEV), b2.to(DEV), residual=xd)
        # same chain with the split inside the GEMMs
        y32 = k.layer_norm(xd, g.to(DEV), b.to(DEV), 1e-6)
        got_in = k.linear(k.linear(y32, w1.to(DEV), b1.to(DEV), act=k.ACT_RELU), w2.to(DEV), b2.to(DEV),
                          residual=xd)
    err = (got.cpu().double() - ref).abs().max().item()
    err_in = (got_in.cpu().double() - ref).abs().max().item()
    print(f"f16 planes from producers: err {err:.2e} (split in the GEMM: {err_in:.2e})")
    assert err < 2e-5 and err < 3 * err_in + 1e-6
    # attention epilogue -> out-projection
    B, T, E, H = 3, 70, 512, 8
    qkv = rnd("pqkv", (B, T, 3 * E)).to(DEV)
    wo = rnd("pwo", (E, E), "uniform", E ** -0.5).to(DEV)
    with k.gemm_precision("f16x3"):
        o32 = k.mha(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], H, (E // H) ** -0.5)
        osp = k.mha(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], H, (E // H) ** -0.5, out_split=22)
        assert (osp.planes.float().sum(dim=1).reshape(B, T, E) / 256.0 - o32).abs().max().item() < 2e-6
        a, bb = k.linear(osp, wo), k.linear(o32, wo)
    assert (a - bb).abs().max().item() < 1e-5


@pytest.mark.parametrize("B,Tq,Lt", [(3, 300, 12), (2, 37, 5), (1, 30, 16), (61, 270, 12), (130, 140, 9),
                                     (2, 70, 17), (3, 300, 24), (40, 290, 32), (130, 100, 29)])
def test_cross_attention_collapsed_over_the_caption(B, Tq, Lt, monkeypatch):
    """
    csrc/xattn.hip: LayerNorm + q projection + attention over the caption + output projection + residual in ONE
    kernel, with the projections folded into per-sample caption operands.  Against (a) an fp64 evaluation of the
    reference's TransformerDecoderBlock cross-attention half (attention.py:445-463, 303-319) and (b) this repo's
    four-kernel path on the same module (TextKV without collapsed operands); ragged Tq, Lt < 16 (masked slots).
    Captions of 17-32 tokens (round 4) take 32 caption slots per head (xattn_collapsed_kernel<32>, any batch size);
    33-50 tokens (the text encoder admits 50, text_encoders.py:36) keep the four-kernel path (checked at the end).
    The last two shapes fill the chip and take the 64-token-workgroup variant (B not a multiple of 8: idle ids of the
    XCD-aware numbering; Tq not a multiple of 64), the first three the 32-token variant.
    """
    from textocvp_amd.models.Blocks import attention as A
    from textocvp_amd.models.Blocks.attention import TextKV, TransformerDecoderBlock
    k = _k()
    E, H, dh = 512, 8, 64
    blk = TransformerDecoderBlock(embed_dim=E, head_dim=dh, kv_dim=E, num_heads=H, mlp_size=2048).eval()
    with torch.no_grad():
        for n_, p_ in blk.named_parameters():
            if p_.dim() > 1:
                p_.copy_(rnd("xa." + n_, tuple(p_.shape), "uniform", p_.shape[1] ** -0.5))
            else:
                p_.copy_((1.0 if "weight" in n_ else 0.0) + rnd("xa." + n_, tuple(p_.shape), "uniform", 0.2))
    x = rnd("xa.x", (B, Tq, E), "normal") * 1.5
    text = rnd("xa.text", (B, Lt, E), "normal")
    # (a) fp64 reference of the cross-attention half: z = x + out_proj(softmax(q k^T / sqrt(dh)) v)
    P64 = {n_: p_.detach().double() for n_, p_ in blk.named_parameters()}
    ln = torch.nn.functional.layer_norm
    qn = ln(x.double(), (E,), P64["ln_cross_att_q.weight"], P64["ln_cross_att_q.bias"], 1e-6)
    tn = ln(text.double(), (E,), P64["ln_cross_att_kv.weight"], P64["ln_cross_att_kv.bias"], 1e-6)
    q = (qn @ P64["cross_attn.q.weight"].t()).view(B, Tq, H, dh).transpose(1, 2)
    kk = (tn @ P64["cross_attn.k.weight"].t()).view(B, Lt, H, dh).transpose(1, 2)
    vv = (tn @ P64["cross_attn.v.weight"].t()).view(B, Lt, H, dh).transpose(1, 2)
    att = torch.softmax(q @ kk.transpose(-1, -2) * dh ** -0.5, dim=-1) @ vv
    z_ref = x.double() + att.transpose(1, 2).reshape(B, Tq, E) @ P64["cross_attn.out_projection.weight"].t() \
        + P64["cross_attn.out_projection.bias"]
    blk = blk.to(DEV)
    xd, td = x.to(DEV), text.to(DEV)
    with torch.no_grad(), k.gemm_precision("f16x3"):
        tkv = blk.project_text(td)
        assert isinstance(tkv, TextKV) and tkv.collapsed is not None and tkv.collapsed[2] == Lt
        Gf, Hf, _ = tkv.collapsed
        lnq = blk.ln_cross_att_q
        z = k.xattn_collapsed(xd, lnq.weight, lnq.bias, lnq.eps, Gf, Hf, blk.cross_attn.out_projection.bias, H, Lt,
                              dh ** -0.5)
        z4 = blk.cross_attn(None, query_embs=k.layer_norm(xd, lnq.weight, lnq.bias, lnq.eps), residual=xd, kv=tkv.kv)
        full = blk(xd, td, text_kv=tkv)                                    # fused path inside the block
        full4 = blk(xd, td, text_kv=TextKV(tkv.kv, None))                  # four-kernel path
    err = (z.cpu().double() - z_ref).abs().max().item()
    err4 = (z4.cpu().double() - z_ref).abs().max().item()
    print(f"collapsed cross-attention B={B} Tq={Tq} Lt={Lt}: |fused - fp64| {err:.2e}, |four kernels - fp64| {err4:.2e}, "
          f"|block fused - block four| {(full - full4).abs().max().item():.2e}")
    assert err < 5e-6 * max(1.0, float(z_ref.abs().max())) and err < 3 * err4 + 2e-6
    assert (full - full4).abs().max().item() < 2e-5
    assert torch.isfinite(z).all()
    # captions behind the limit (32 tokens) fall back to the four-kernel path
    with torch.no_grad(), k.gemm_precision("f16x3"):
        assert blk.project_text(rnd("xa.long", (B, 50, E), "normal").to(DEV)).collapsed is None
        assert blk.project_text(rnd("xa.long", (B, 33, E), "normal").to(DEV)).collapsed is None


def test_conv5x5_f16f8_pass_major_layout():
    """ the private (n, 4, H, W, 16) layout between decoder layers holds exactly the NHWC values """
    k = _k()
    n = 5
    x = rnd("bx", (n, 64, 64, 64))
    w = rnd("bw", (64, 64, 5, 5), "uniform", (25 * 64) ** -0.5)
    b = rnd("bb", (64,), "uniform", 0.1)
    wi = k.split_conv_weights_f16f8(w.to(DEV))

    def to_pm(t):
        return t.view(n, 64, 64, 4, 16).permute(0, 3, 1, 2, 4).contiguous().view(n, 64, 64, 64)

    ref = k.conv5x5_f16f8(x.to(DEV), wi, b.to(DEV), relu=True)
    out_pm = k.conv5x5_f16f8(x.to(DEV), wi, b.to(DEV), relu=True, pm_out=True)
    assert torch.equal(out_pm, to_pm(ref))
    in_pm = k.conv5x5_f16f8(to_pm(x.to(DEV)), wi, b.to(DEV), relu=True, pm_in=True)
    assert torch.equal(in_pm, ref)
    both = k.conv5x5_f16f8(to_pm(x.to(DEV)), wi, b.to(DEV), relu=True, pm_in=True, pm_out=True)
    assert torch.equal(both, to_pm(ref))


@pytest.mark.parametrize("Cin,Cout,n", [(32, 32, 5), (64, 64, 2)])
def test_conv5x5_f16x3_generic(Cin, Cout, n):
    """ generic split-fp16 5x5 conv (SAVi encoder shapes): fp32-class against the fp64 result """
    k = _k()
    x = rnd("gx", (n, 64, 64, Cin))
    w = rnd("gw", (Cout, Cin, 5, 5), "uniform", (25 * Cin) ** -0.5)
    b = rnd("gb", (Cout,), "uniform", 0.1)
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=2)).permute(0, 2, 3, 1)
    wp = k.pack_conv_weights(w.to(DEV))
    got = k.conv5x5(x.to(DEV), wp, b.to(DEV), relu=True, precision="f16x3")
    got32 = k.conv5x5(x.to(DEV), wp, b.to(DEV), relu=True)
    err, err32 = (got.cpu().double() - ref).abs().max().item(), (got32.cpu().double() - ref).abs().max().item()
    print(f"conv5x5 f16x3 {Cin}->{Cout}: err {err:.2e} (fp32 mfma {err32:.2e})")
    assert err < max(3 * err32, 5e-6 * ref.abs().max().item())


@pytest.mark.parametrize("M", [32768, 38400, 12000, 11123])
def test_mlp_fused_against_the_two_gemms(M, monkeypatch):
    """
    csrc/mlp_fused.hip: relu(x W1^T + b1) W2^T + b2 + residual in one launch (the predictor's nn.Linear -> ReLU ->
    nn.Linear pairs, attention.py:355-359 / :428-432), hidden activation kept on the CU.  Rows of whole-tile
    workgroups are BIT-IDENTICAL to the two f16x3 GEMMs with plane hand-over (same planes, same k order); the tiles of a
    partly filled last round are cut along the hidden dimension and meet in the workspace: those rows agree to fp32
    re-association (< 1e-5 here), every repetition is bit-identical, the arrival counters end at zero, and the whole
    result is fp32-class against float64.  Rows that are not a multiple of 128 (11123) exercise the clamped tail tile.
    """
    k = _k()
    E, Hd = 512, 2048
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, E, generator=g)
    w1, b1 = (torch.rand(Hd, E, generator=g) - 0.5) * 2 * E ** -0.5, torch.randn(Hd, generator=g) * 0.1
    w2, b2 = (torch.rand(E, Hd, generator=g) - 0.5) * 2 * Hd ** -0.5, torch.randn(E, generator=g) * 0.1
    res = torch.randn(M, E, generator=g)
    xd, w1d, b1d, w2d, b2d, rd = (t.to(DEV) for t in (x, w1, b1, w2, b2, res))
    gamma, beta = torch.ones(E, device=DEV), torch.zeros(E, device=DEV)
    with k.gemm_precision("f16x3"):
        xp = k.layer_norm(xd, gamma, beta, 1e-6, split=22)                 # the producer the predictor uses
        assert isinstance(xp, k.SplitAct) and k.mlp_fused_ok(xp, w1d, w2d)
        monkeypatch.setattr(k, "_MLP_FUSED", False)
        assert not k.mlp_fused_ok(xp, w1d, w2d)
        hid = k.linear(xp, w1d, b1d, act=k.ACT_RELU, out_split=22)
        two = k.linear(hid, w2d, b2d, residual=rd)
        del hid
        monkeypatch.setattr(k, "_MLP_FUSED", True)
        one = k.mlp_fused(xp, w1d, b1d, w2d, b2d, residual=rd)
        for _ in range(5):
            assert torch.equal(k.mlp_fused(xp, w1d, b1d, w2d, b2d, residual=rd), one)
        no_res = k.mlp_fused(xp, w1d, b1d, w2d, b2d)
    tiles, cus = (M + 127) // 128, torch.cuda.get_device_properties(0).multi_processor_count
    whole = (tiles - tiles % cus) * 128 if tiles % cus and cus // (tiles % cus) >= 2 else M
    assert torch.equal(one[:whole], two[:whole]), "whole-tile rows must equal the two-GEMM path bit for bit"
    assert (one - two).abs().max().item() < 1e-5
    close(no_res + rd, one, tol=1e-6)
    xn = torch.nn.functional.layer_norm(x.double(), (E,), eps=1e-6)
    ref = torch.relu(xn @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double() + res.double()
    err, err2 = (one.cpu().double() - ref).abs().max().item(), (two.cpu().double() - ref).abs().max().item()
    print(f"fused MLP {M} rows: {err:.2e} vs float64 (two GEMMs {err2:.2e}); whole-tile rows {whole}")
    assert err < max(2.0 * err2, 5e-6)
    wk = k._mlp_workspace(xd.device)[0]
    assert int(wk[:1024].view(torch.int32).abs().sum()) == 0


def test_mlp_fused_zigzag_order_is_the_same_sum():
    """
    TOCVP_MLP_ZIGZAG=1 (opt-in, read once per process: a child process): odd hidden chunks walk the X k-tiles back to front --
    the same products in another fp32 order.  33000 rows = 256 whole tiles + 2 sliced ones (slices beginning at odd and
    even chunks): within 2e-5 of the two-GEMM path and as close to float64 as it (scripts/mlp_fused_one.py prints both).
    """
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TOCVP_MLP_ZIGZAG="1")
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "mlp_fused_one.py"), "33000", "2"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"two GEMMs\| ([0-9.e+-]+); max err vs fp64 fused ([0-9.e+-]+) two GEMMs ([0-9.e+-]+)", out.stdout)
    assert m, out.stdout
    diff, err, err2 = (float(v) for v in m.groups())
    print(out.stdout.strip())
    assert 0.0 < diff < 2e-5 and err < max(2.0 * err2, 1e-5)


def test_copy_strided_matches_torch_index_copies():
    """ tocvp_copy4d_f32 behind kernels.copy_strided / contiguous / stack1: the window slices, last-frame slices, stacks and
    the time-major copy of the frames that the hot path used to leave to torch.cat / torch.stack / .contiguous() """
    k = _k()
    g = torch.Generator().manual_seed(3)
    hist = torch.randn(5, 20, 30, 128, generator=g).to(DEV)
    assert torch.equal(k.contiguous(hist[:, 3:13]), hist[:, 3:13].contiguous())
    assert torch.equal(k.contiguous(hist[:, -1]), hist[:, -1].contiguous())
    buf = torch.zeros(5, 7, 30, 128, device=DEV)
    k.copy_strided(hist[:, 4], buf[:, 2])
    assert torch.equal(buf[:, 2], hist[:, 4]) and float(buf[:, :2].abs().sum()) == 0 and float(buf[:, 3:].abs().sum()) == 0
    vids = torch.rand(3, 9, 3, 64, 64, generator=g).to(DEV)
    assert torch.equal(k.contiguous(vids[:, 2:8].transpose(0, 1)), vids[:, 2:8].transpose(0, 1).contiguous())
    toks = torch.randn(4, 300, 512, generator=g).to(DEV)
    assert torch.equal(k.contiguous(toks.reshape(4, 10, 30, 512)[:, -1]), toks.reshape(4, 10, 30, 512)[:, -1].contiguous())
    parts = [torch.randn(6, 7, 128, generator=g).to(DEV) for _ in range(4)]
    assert torch.equal(k.stack1(parts), torch.stack(parts, dim=1))
    # layouts the kernel does not take fall through to torch's device copy instead of failing (round 5): runs of 6 floats,
    # a base pointer that is not 16-byte aligned (an odd image width), four strided dimensions
    odd = k.copy_strided(hist[..., :6], torch.empty(5, 20, 30, 6, device=DEV))
    assert torch.equal(odd, hist[..., :6])
    flat = torch.randn(4 * 33 * 33 + 1, generator=g).to(DEV)
    assert torch.equal(k.contiguous(flat[1:].reshape(4, 33, 33).transpose(1, 2)), flat[1:].reshape(4, 33, 33).transpose(1, 2).contiguous())
    assert torch.equal(k.contiguous(hist[::2, ::3, ::5, ::4]), hist[::2, ::3, ::5, ::4].contiguous())
    misaligned = flat[1:1 + 4 * 32 * 8].reshape(4, 32, 8)
    assert misaligned.data_ptr() % 16 != 0 and torch.equal(k.contiguous(misaligned.transpose(0, 1)), misaligned.transpose(0, 1).contiguous())


@pytest.mark.parametrize("M,N,Kd", [(4096, 1024, 1024), (1157, 1536, 512), (1000, 512, 128), (77, 1024, 3072),
                                    (1500, 768, 768), (700, 2304, 768), (260, 384, 128)])
def test_gemm_chunk_resident_equals_the_in_loop_split_kernel(M, N, Kd, monkeypatch):
    """
    csrc/gemm_f16c.hip (tocvp_gemm_f16chunk_f32): the persistent f16x3 GEMM whose A operand is walked in 128-deep chunks
    resident in LDS while the weights stream from L2 in fragment order -- nn.Linear 1024 -> 1024 of the MLPPatchDecoder
    (reference decoders.py:264-307) and the wide projections of the ViT / predictor blocks (attention.py:167-175).  Same
    operand planes, same k order, same epilogue expressions as tocvp_gemm_bf16wfrag_f32: BIT-IDENTICAL for fp32 and plane
    outputs, every activation, with and without the residual, on ragged row counts, for 512- and 384-wide column tiles
    (N % 512 == 0 / N % 384 == 0); fp32-class against float64; repeatable.
    """
    k = _k()
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, Kd, generator=g)
    w, b = torch.randn(N, Kd, generator=g) / Kd ** 0.5, torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    xd, wd, bd, rd = (t.to(DEV) for t in (x, w, b, r))
    monkeypatch.setattr(k, "_GEMM_CHUNK_MIN_TILES", 1)
    # the activation as fp16 operand planes of 2^8 x (the expressions of tocvp_store_planes4)
    v = torch.clamp(xd * 256.0, -65504.0, 65504.0)
    hi = v.to(torch.float16)
    xp = k.SplitAct(torch.stack([hi, (v - hi.float()).to(torch.float16)], dim=1).contiguous(), (M, Kd))
    with k.gemm_precision("f16x3"):
        for act in (k.ACT_NONE, k.ACT_RELU, k.ACT_GELU):
            for res, osplit in ((None, 0), (rd, 0), (None, 22)):
                monkeypatch.setattr(k, "_GEMM_CHUNK", False)
                monkeypatch.setattr(k, "_GEMM_MID", False)
                a = k.linear(xp, wd, bd, act=act, residual=res, out_split=osplit)
                monkeypatch.setattr(k, "_GEMM_CHUNK", True)
                names = []
                k.TIMER = type("T", (), {"wrap": staticmethod(lambda name, units, fn: (names.append(name), fn())[1])})()
                try:
                    c = k.linear(xp, wd, bd, act=act, residual=res, out_split=osplit)
                    c2 = k.linear(xp, wd, bd, act=act, residual=res, out_split=osplit, chunk_ok=False)
                finally:
                    k.TIMER = None
                assert len(names) == 2
                if osplit:
                    a, c, c2 = a.planes.view(torch.int16), c.planes.view(torch.int16), c2.planes.view(torch.int16)
                assert torch.equal(a, c), (act, res is not None, osplit)
                assert torch.equal(a, c2)
        got = k.linear(xp, wd, bd)
        assert torch.equal(got, k.linear(xp, wd, bd))
    ref = x.double() @ w.double().t() + b.double()
    err = (got.cpu().double() - ref).abs().max().item()
    print(f"chunk-resident GEMM {M}x{N}x{Kd}: {err:.2e} vs float64 (max |y| {ref.abs().max().item():.1f})")
    assert err < 5e-6 * max(1.0, ref.abs().max().item())


def test_gemm_chunk_resident_rejects_bad_arguments():
    k = _k()
    lib = k.lib()
    a = torch.zeros(256, 2, 128, device=DEV, dtype=torch.float16)
    w = torch.zeros(512 * 2 * 128, device=DEV, dtype=torch.float16)
    c = torch.zeros(256, 512, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    ok = lib.tocvp_gemm_f16chunk_f32(a.data_ptr(), w.data_ptr(), None, None, 0, c.data_ptr(), 0, 512, 256, 512, 128, 0, st)
    assert ok == 0
    for (N, Kd, act, ldc) in ((500, 128, 0, 512), (512, 100, 0, 512), (512, 128, 7, 512), (512, 128, 0, 256)):
        assert lib.tocvp_gemm_f16chunk_f32(a.data_ptr(), w.data_ptr(), None, None, 0, c.data_ptr(), 0, ldc, 256, N, Kd, act,
                                           st) != 0
    assert lib.tocvp_gemm_f16chunk_f32(a.data_ptr() + 2, w.data_ptr(), None, None, 0, c.data_ptr(), 0, 512, 256, 512, 128, 0,
                                       st) != 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("Cin,Cout,S,n,relu", [(128, 64, 16, 3, True), (64, 128, 8, 2, True), (128, 32, 40, 2, False),
                                               (64, 32, 16, 2, False), (96, 64, 32, 1, True)])
def test_conv3x3_up2_phases_equal_the_conv_over_the_upsampled_image(Cin, Cout, S, n, relu):
    """
    tocvp_conv3x3_up2_f16x3_f32: "Upsample(scale_factor=2) -> Conv2d(k3, p1)" of the DINOSAUR image head (reference
    decoders.py:325-365 with the Upsample of model_blocks.py:23-45) as four 2x2 phase convolutions over the SOURCE image
    (the 3x3 taps that read the same source pixel summed beforehand: 2.25x fewer FLOPs).  Against float64
    F.interpolate(nearest) + conv2d + scale / shift (+ ReLU), at the accuracy of the 3x3 kernel with the upsampling fused
    into its loader; every output pixel of every phase is written (borders included: the zero padding carries over).
    """
    k = _k()
    x = rnd("u2x", (n, S, S, Cin), "normal")
    w = rnd("u2w", (Cout, Cin, 3, 3), "uniform", (9 * Cin) ** -0.5)
    sc, sf = 1.0 + 0.2 * rnd("u2sc", (Cout,), "uniform", 1.0), rnd("u2sf", (Cout,), "uniform", 0.3)
    xd, wd, scd, sfd = x.to(DEV), w.to(DEV), sc.to(DEV), sf.to(DEV)
    y = torch.full((1,), float("nan"))
    got = k.conv3x3_up2(xd, k.pack_conv3x3_up2_weights(wd), scd, sfd, relu=relu)
    old = k.conv3x3(xd, k.pack_conv_weights(wd), scd, sfd, relu=relu, upsample2=True, precision="f16x3")
    assert got.shape == (n, 2 * S, 2 * S, Cout) and bool(torch.isfinite(got).all())
    up = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2).double(), scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(up, w.double(), padding=1) * sc.double()[None, :, None, None] + sf.double()[None, :, None, None]
    if relu:
        ref = torch.relu(ref)
    ref = ref.permute(0, 2, 3, 1)
    err, err_old = (got.cpu().double() - ref).abs().max().item(), (old.cpu().double() - ref).abs().max().item()
    print(f"conv3x3_up2 {Cin}->{Cout} @ {S}: {err:.2e} vs float64 (3x3 over the upsampled image {err_old:.2e}), max |y| {ref.abs().max().item():.2f}")
    assert err < max(2.0 * err_old, 2e-6 * max(1.0, ref.abs().max().item()))
    # no scale (the final RGB conv): scale = NULL means 1
    got1 = k.conv3x3_up2(xd, k.pack_conv3x3_up2_weights(wd), None, sfd, relu=False)
    ref1 = (torch.nn.functional.conv2d(up, w.double(), padding=1) + sf.double()[None, :, None, None]).permute(0, 2, 3, 1)
    assert (got1.cpu().double() - ref1).abs().max().item() < 2e-6 * max(1.0, ref1.abs().max().item())


@pytest.mark.parametrize("M,N,Kd", [(2400, 2048, 512), (9600, 512, 2048), (960, 1536, 512), (3841, 256, 128), (65, 768, 1024)])
def test_gemm_chunk_mid_size_form(M, N, Kd, monkeypatch):
    """
    tocvp_gemm_f16mid_f32 (csrc/gemm_f16c.hip, mid-size form: 64 x 256 tiles, two workgroups per CU, A chunks by LDS-DMA,
    weights streamed) -- the predictor's products at small evaluation batches (reference attention.py:167-175, 355-359,
    428-432).  BIT-IDENTICAL to tocvp_gemm_bf16wfrag_f32 on the same planes (every activation, residual, plane output,
    ragged rows), repeatable, fp32-class against float64.
    """
    k = _k()
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, Kd, generator=g)
    w, b = torch.randn(N, Kd, generator=g) / Kd ** 0.5, torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    xd, wd, bd, rd = (t.to(DEV) for t in (x, w, b, r))
    v = torch.clamp(xd * 256.0, -65504.0, 65504.0)
    hi = v.to(torch.float16)
    xp = k.SplitAct(torch.stack([hi, (v - hi.float()).to(torch.float16)], dim=1).contiguous(), (M, Kd))
    monkeypatch.setattr(k, "_GEMM_CHUNK", False)
    monkeypatch.setattr(k, "_GEMM_MID_MIN_TILES", 1)
    ref = x.double() @ w.double().t() + b.double()
    with k.gemm_precision("f16x3"):
        for act in (k.ACT_NONE, k.ACT_RELU, k.ACT_GELU):
            for res, osplit in ((None, 0), (rd, 0), (None, 22)):
                monkeypatch.setattr(k, "_GEMM_MID", False)
                a = k.linear(xp, wd, bd, act=act, residual=res, out_split=osplit)
                monkeypatch.setattr(k, "_GEMM_MID", True)
                c = k.linear(xp, wd, bd, act=act, residual=res, out_split=osplit)
                c2 = k.linear(xp, wd, bd, act=act, residual=res, out_split=osplit)
                if osplit:
                    a, c, c2 = (t.planes.view(torch.int16) for t in (a, c, c2))
                assert torch.equal(a, c), (act, res is not None, osplit)
                assert torch.equal(c, c2)
        got = k.linear(xp, wd, bd)
    err = (got.cpu().double() - ref).abs().max().item()
    assert err < 5e-6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,H,T,dh", [(24, 12, 257, 64), (70, 4, 129, 32)])
def test_mha_tail_row_of_the_vit_sequence_length(B, H, T, dh, monkeypatch):
    """
    Self-attention at 128 n + 1 tokens (the ViT's 256 patches + class token, timm_encoders.py:59-70): the tile kernel runs the
    first T - 1 query rows of tensors that hold T rows per sample (tocvp_mha_qk16_rows_f32) and the last row goes through
    tocvp_mha_one_query_f32 (exact fp32, a wave per (sample, head)) instead of a third 128-query tile that stages every key
    and value again for that one row.  Fused-qkv views (row stride 3 E), per-sample key lengths; against the oracle's
    attention and against the one-launch path.
    """
    k = _k()
    E = H * dh
    qkv = rnd("tqkv", (B, T, 3 * E))
    lengths = torch.full((B,), T, dtype=torch.int64)
    lengths[1], lengths[B - 1] = T - 7, 131
    key_pad = torch.arange(1, T + 1)[None, :] > lengths[:, None]
    ref = O.attention(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], H, dh ** -0.5, key_mask=key_pad)
    d = qkv.to(DEV)
    kl = lengths.to(torch.int32).to(DEV)
    names = []
    real = k.lib()
    monkeypatch.setattr(k, "_MHA_TAIL_ROW", True)
    k.TIMER = type("T", (), {"wrap": staticmethod(lambda name, units, fn: (names.append(name), fn())[1])})()
    try:
        got = k.mha(d[..., :E], d[..., E:2 * E], d[..., 2 * E:], H, dh ** -0.5, key_len=kl)
    finally:
        k.TIMER = None
    monkeypatch.setattr(k, "_MHA_TAIL_ROW", False)
    one = k.mha(d[..., :E], d[..., E:2 * E], d[..., 2 * E:], H, dh ** -0.5, key_len=kl)
    close(got, ref)
    close(one, ref)
    assert torch.equal(got[:, :T - 1], one[:, :T - 1]), "the first T - 1 rows come from the same tile kernel"
    assert (got[:, T - 1] - one[:, T - 1]).abs().max().item() < 3e-6
    # O as fp16 operand planes (tocvp_mha_qk16_rows_split_f16 + tocvp_mha_one_query_split_f16): the split of the fp32 output
    monkeypatch.setattr(k, "_MHA_TAIL_ROW", True)
    pl = k.mha(d[..., :E], d[..., E:2 * E], d[..., 2 * E:], H, dh ** -0.5, key_len=kl, out_split=22)
    v_ = torch.clamp(got * 256.0, -65504.0, 65504.0).reshape(B * T, E)
    hi_ = v_.to(torch.float16)
    assert torch.equal(pl.planes[:, 0], hi_) and torch.equal(pl.planes[:, 1], (v_ - hi_.float()).to(torch.float16))
    # direct C-ABI call of the one-row kernel on another row, plain (B, T, E) tensors
    q2, k2, v2 = (t.contiguous() for t in (d[..., :E], d[..., E:2 * E], d[..., 2 * E:]))
    o2 = torch.zeros(B, T, E, device=DEV)
    rc = real.tocvp_mha_one_query_f32(q2.data_ptr(), E, k2.data_ptr(), E, v2.data_ptr(), E, o2.data_ptr(), E, B, H, T, 5, T, dh,
                                      dh ** -0.5, kl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert (o2[:, 5].cpu() - ref[:, 5]).abs().max().item() < 3e-6 and float(o2[:, 6].abs().max()) == 0.0


def test_plane_input_gemms_beyond_four_gigabytes_of_planes(monkeypatch):
    """
    The plane-input kernels address their A operand with 32-bit byte offsets (LDS-DMA sources).  More than 2^32 bytes of
    planes (here 263 000 rows at K = 4096; configs[3] decoded in one call hands 2.85 M rows of 1024 to the MLPPatchDecoder's
    layers) must go through in row blocks: every path (chunk-resident, mid-size, two-operand) against float64 on rows before,
    around and behind the limit, and bit-identical to one another.  (Before round 4's guard the two-operand planes kernel
    returned wrong rows behind the limit, silently.)
    """
    k = _k()
    M, N, Kd = 263_000, 512, 4096
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    x = torch.randn(M, Kd, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    v = torch.clamp(x * 256.0, -65504.0, 65504.0)
    hi = v.to(torch.float16)
    planes = torch.empty(M, 2, Kd, device=DEV, dtype=torch.float16)
    planes[:, 0], planes[:, 1] = hi, (v - hi.float()).to(torch.float16)
    del v, hi
    xp = k.SplitAct(planes, (M, Kd))
    assert planes.numel() * 2 > 2 ** 32
    outs = {}
    with k.gemm_precision("f16x3"):
        for name, chunk, mid in (("chunk", True, False), ("two-operand", False, False)):
            monkeypatch.setattr(k, "_GEMM_CHUNK", chunk)
            monkeypatch.setattr(k, "_GEMM_MID", mid)
            outs[name] = k.linear(xp, w, b, act=k.ACT_RELU)
        monkeypatch.setattr(k, "_GEMM_CHUNK", False)
        pl = k.linear(xp, w, b, act=k.ACT_RELU, out_split=22)
    limit = 2 ** 32 // (4 * Kd)
    for rows in (slice(0, 256), slice(limit - 128, limit + 128), slice(M - 256, M)):
        ref = torch.relu(x[rows].double() @ w.double().t() + b.double())
        for name, o in outs.items():
            err = (o[rows].double() - ref).abs().max().item()
            assert err < 1e-5 * max(1.0, ref.abs().max().item()), (name, rows, err)
        rebuilt = pl.planes[rows].double().sum(dim=1) / 256.0
        assert (rebuilt - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(outs["chunk"], outs["two-operand"])


@pytest.mark.parametrize("prec", ["f16x3", "bf16x3"])
def test_fp32_input_split_gemms_beyond_four_gigabytes_of_activations(prec):
    """
    1 100 003 rows x 1024 fp32 activations (4.5 GB; configs[3] decoded in ONE call hands the MLPPatchDecoder 2.85 M such
    rows, reference 05_evaluate_predictor.py:88-96): the split kernels' 32-bit byte offsets wrapped behind 2^32 bytes and
    rows >= 1 048 576 came back wrong in rounds 1-3 -- kernels.linear cuts the rows into blocks now.  Against float64 on
    rows before, around and behind the line.
    """
    k = _k()
    M, N, Kd = 1_100_003, 256, 1024
    x = torch.randn(M, Kd, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8))
    g = torch.Generator().manual_seed(9)
    w, b = (torch.randn(N, Kd, generator=g) / 32).to(DEV), torch.randn(N, generator=g).to(DEV)
    with k.gemm_precision(prec):
        y = k.linear(x, w, b, act=k.ACT_RELU)
    tol = 2e-5 if prec == "f16x3" else 2e-4
    for rows in (slice(0, 256), slice(1048576 - 128, 1048576 + 128), slice(M - 256, M)):
        ref = torch.relu(x[rows].double() @ w.double().t() + b.double())
        assert (y[rows].double() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item()), rows


def _planes_of(x2):
    """ fp16 operand planes (rows, 2, D) of 2^8 x, the split of tocvp_store_planes4 (common.h) spelled in torch """
    X = torch.clamp(x2 * 256.0, -65504.0, 65504.0)
    hi = X.to(torch.float16)
    return torch.stack([hi, (X - hi.float()).to(torch.float16)], dim=1).contiguous()


@pytest.mark.skipif(os.environ.get("TOCVP_PRECISION") == "fp32" or os.environ.get("TOCVP_ATTN_QK") == "fp32",
                    reason="operand planes exist in the f16x3 arithmetic only")
@pytest.mark.parametrize("B,H,Tq,Tk,lens", [(5, 8, 300, 300, False), (3, 8, 30, 300, False), (2, 8, 77, 77, True),
                                            (24, 12, 257, 257, True), (1, 8, 30, 30, False), (9, 6, 257, 257, False)])
def test_mha_planes_equals_the_fp32_input_kernel(B, H, Tq, Tk, lens, monkeypatch):
    """
    csrc/attn_planes.hip (round 5): self-attention of the predictor / ViT blocks (reference attention.py:183-215, 245-265;
    timm_encoders.py:59-70) with q / k / v handed over as the fp16 operand planes a projection epilogue writes.  The kernel
    copies planes where tocvp_mha_qk16_f32 splits and transposes fp32 rows, and keeps its arithmetic instruction for
    instruction: the fp32 output and the plane output must equal the fp32-input kernel's BIT FOR BIT on the same values --
    whole sequences, the last-frame queries of the final predictor layer (30 query rows against 300 keys, q from its own
    projection), ragged key lengths, the ViT's 128 n + 1 rows, fused-projection column offsets; and fp32-class against the
    float64 attention of the oracle.
    """
    k = _k()
    dh, E = 64, H * 64
    monkeypatch.setattr(k, "_MHA_TAIL_ROW", False)               # reference: the tile kernel on every row (no one-row kernel)
    kv = rnd("pkv", (B, Tk, 3 * E)).to(DEV)                       # fused projection: columns [q | k | v]
    qsep = rnd("pq", (B, Tq, E)).to(DEV) if Tq != Tk else None     # a separate query projection (forward_last)
    kl = None
    if lens:
        lengths = torch.full((B,), Tk, dtype=torch.int64)
        lengths[0], lengths[B - 1] = max(1, Tk - 7), max(1, Tk // 2 + 3)
        kl = lengths.to(torch.int32).to(DEV)
    q32 = qsep if qsep is not None else kv[..., :E]
    ref = k.mha(q32, kv[..., E:2 * E], kv[..., 2 * E:], H, dh ** -0.5, key_len=kl)
    ref_pl = k.mha(q32, kv[..., E:2 * E], kv[..., 2 * E:], H, dh ** -0.5, key_len=kl, out_split=22)
    kvp = k.SplitAct(_planes_of(kv.reshape(B * Tk, 3 * E)), (B, Tk, 3 * E))
    qp, qcol = (kvp, 0) if qsep is None else (k.SplitAct(_planes_of(qsep.reshape(B * Tq, E)), (B, Tq, E)), 0)
    with k.gemm_precision("f16x3"):
        assert k.mha_planes_ok(H, E)
        got = k.mha_planes(qp, qcol, kvp, E, kvp, 2 * E, B, Tq, Tk, H, dh ** -0.5, key_len=kl)
        got_pl = k.mha_planes(qp, qcol, kvp, E, kvp, 2 * E, B, Tq, Tk, H, dh ** -0.5, key_len=kl, out_split=22)
    assert torch.equal(got, ref), float((got - ref).abs().max())
    assert torch.equal(got_pl.planes, ref_pl.planes)
    pad = None if kl is None else (torch.arange(1, Tk + 1)[None, :] > kl.cpu().long()[:, None])
    o64 = O.attention(q32.cpu().double(), kv[..., E:2 * E].cpu().double(), kv[..., 2 * E:].cpu().double(), H, dh ** -0.5, key_mask=pad)
    assert (got.cpu().double() - o64).abs().max().item() < 2e-6
    again = k.mha_planes(qp, qcol, kvp, E, kvp, 2 * E, B, Tq, Tk, H, dh ** -0.5, key_len=kl)
    assert torch.equal(again, got)
