"""
Module-level and end-to-end parity of the MI355X path (textocvp_amd.models through the C-ABI)
against (a) the golden vectors produced by the reference itself and (b) the CPU oracle on the same
seeded inputs / weights.  Needs a real MI355X (pytest -m gpu).

Tolerance = the north-star bar: 1e-4 absolute on slots, masks and rendered pixels, identical
argmax_K(masks) maps (slot-index permutation).  Unit fixtures use 5e-5 where the arithmetic is
fp32-class.  The suite runs in the default arithmetic (decoder convs and predictor GEMMs f16x3, the
rest per DESIGN.md section 3) and, with TOCVP_PRECISION=fp32, in the all-fp32 mode.
Three weight families (textocvp_amd/synth.py): "damped" (round-1 fixtures), "undamped" (O(1) RGB
head) and "xavier" (the distribution of the reference's own init) -- the last two at FULL resolution.
"""

import os

import numpy as np
import pytest
import torch

from conftest import load_golden, max_abs
from oracle import slot_rollout_oracle as O
from textocvp_amd import synth
from textocvp_amd.evaluator import forward_eval
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(num_slots, num_preds, savi_family="damped"):
    exp = default_exp_params(num_slots=num_slots, num_context=1, num_preds=num_preds)
    savi = setup_model(exp["model"]).eval()
    pred = setup_predictor(exp).eval()
    synth.fill_module_(savi, prefix="savi.", family=savi_family)
    synth.fill_module_(pred, prefix="pred.")
    return savi.to(DEV), pred.to(DEV)


@pytest.fixture(scope="module")
def k7():
    return build(7, 4)


@pytest.fixture(scope="module")
def k30():
    return build(30, 19)


def gpu(t):
    return t.to(DEV)


ARGMAX_REPORT = {}                    # name -> {"pixels", "differ", "max_margin"}; dumped for profiles/r03_parity_by_mode.md


def slot_assignment_diff(masks, ref_argmax, name):
    """
    Slot-index permutation check (north star: "slot-index permutation bit-exact"): number of pixels whose
    argmax_K(masks) differs from the reference map, and the top-1 minus top-2 mask margin at those pixels.
    masks (F, K, 1, H, W); every call is recorded in ARGMAX_REPORT / gpurun_out/r03_argmax_report.json.
    """
    am = masks.argmax(dim=1).cpu()
    ref = torch.as_tensor(ref_argmax).to(am.dtype).reshape(am.shape)
    diff = am != ref
    n = int(diff.sum())
    margins = torch.zeros(0)
    if n:
        top2 = masks.topk(2, dim=1).values.cpu()
        margins = (top2[:, 0] - top2[:, 1])[diff]
    mode = os.environ.get("TOCVP_PRECISION", "default")
    ARGMAX_REPORT[f"{name} [{mode}]"] = {"pixels": diff.numel(), "differ": n,
                                         "max_margin": float(margins.max()) if n else 0.0}
    print(f"argmax_K(masks) {name} [{mode}]: {n} of {diff.numel()} pixels differ"
          + (f", margins {[f'{m:.1e}' for m in margins.tolist()[:8]]}" if n else ""))
    try:
        import json
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "r03_argmax_report.json"), "w") as f:
            json.dump(ARGMAX_REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass
    return n, margins


def assert_same_slot_assignment(masks, ref_argmax, name, ties=0, tie_margin=1e-6):
    """
    ZERO differing pixels, except ``ties`` pixels NAMED by the calling test: pixels where the reference's own
    top-2 masks are equal to within a few fp32 ulps (``tie_margin``), so that two fp32 evaluation orders on
    the CPU already disagree (tests/test_oracle_golden.py::test_decomp_only_eval_c1 pins one such pixel between
    the reference and the oracle).  No tolerance band beyond that.
    """
    n, margins = slot_assignment_diff(masks, ref_argmax, name)
    assert n <= ties, f"{name}: {n} argmax pixels differ from the reference (allowed: {ties} named ties)"
    if n:
        assert float(margins.max()) < tie_margin, f"{name}: differing pixel with margin {float(margins.max()):.2e}"


@torch.no_grad()
def test_units_k7_against_reference_goldens(k7):
    savi, pred = k7
    core = pred.predictor
    g = load_golden("units_k7.npz")
    B, K, D = 2, 7, 128

    imgs = synth.synth_tensor("unit.imgs", (B, 3, 64, 64), "unit")
    assert max_abs(savi.encode(gpu(imgs))[:, ::16].cpu(), g["encoder_feats_sub16"]) < 5e-5

    sa_in = synth.synth_tensor("unit.sa_feats", (B, 4096, D), "normal")
    slots0 = synth.synth_tensor("unit.sa_slots", (B, K, D), "normal")
    savi.slot_attention.store_attention_masks = True
    s0 = savi.slot_attention(gpu(sa_in), gpu(slots0), step=0)
    assert max_abs(s0.cpu(), g["sa_step0"]) < 1e-4
    attn = savi.slot_attention.get_attention_masks()
    assert max_abs(attn[:, :, ::16].cpu(), g["sa_step0_attn_sub16"]) < 1e-5
    savi.slot_attention.store_attention_masks = False
    s1 = savi.slot_attention(gpu(sa_in), gpu(slots0), step=1)
    assert max_abs(s1.cpu(), g["sa_step1"]) < 5e-5

    assert max_abs(savi.transition_module(gpu(slots0)).cpu(), g["transition"]) < 2e-5

    tokens, lengths = torch.from_numpy(g["text_tokens"]), torch.from_numpy(g["text_lengths"])
    text = core.text_encoder(text=gpu(tokens), text_length=gpu(lengths))
    assert max_abs(text.cpu(), g["text_emb"]) < 5e-5

    text_ref = gpu(torch.from_numpy(g["text_emb"]))
    x = synth.synth_tensor("unit.block_x", (3, 2 * K, 512), "normal")
    assert max_abs(core.predictor[0](gpu(x), text_ref).cpu(), g["block0"]) < 1e-4

    win1 = synth.synth_tensor("unit.win1", (3, 1, K, D), "normal")
    win10 = synth.synth_tensor("unit.win10", (3, 10, K, D), "normal")
    assert max_abs(core(slots=gpu(win1), text_embeddings=text_ref).cpu(), g["pred_step_w1"]) < 1e-4
    assert max_abs(core(slots=gpu(win10), text_embeddings=text_ref).cpu(), g["pred_step_w10"]) < 1e-4

    dslots = synth.synth_tensor("unit.dec_slots", (2, K, D), "normal")
    out = savi(mode="decode", slots=gpu(dslots))
    assert out["recons"].shape == (2, K, 3, 64, 64) and out["masks"].shape == (2, K, 1, 64, 64)
    assert max_abs(out["recons_imgs"].cpu(), g["dec7_recons_imgs"]) < 5e-5
    assert max_abs(out["recons"][..., ::4, ::4].cpu(), g["dec7_recons_sub4"]) < 5e-5
    # masks are softmax outputs of O(5) alpha logits: hold them to the 1e-4 north-star bar
    assert max_abs(out["masks"][..., ::4, ::4].cpu(), g["dec7_masks_sub4"]) < 1e-4


@torch.no_grad()
def test_decoder_k30_against_reference_golden(k30):
    savi, _ = k30
    g = load_golden("units_k30.npz")
    dslots = synth.synth_tensor("unit.dec_slots30", (2, 30, 128), "normal")
    out = savi(mode="decode", slots=gpu(dslots))
    assert max_abs(out["recons_imgs"].cpu(), g["dec30_recons_imgs"]) < 5e-5
    assert max_abs(out["recons"][..., ::8, ::8].cpu(), g["dec30_recons_sub8"]) < 5e-5
    assert max_abs(out["masks"][..., ::8, ::8].cpu(), g["dec30_masks_sub8"]) < 1e-4


# margin demanded of the decoder arithmetic on the un-flattering families: HALF the north-star bar
PARITY_TOL = 5e-5


@torch.no_grad()
@pytest.mark.parametrize("family", ["undamped", "xavier"])
def test_parity_families_k7_full_resolution(family):
    """ decode + e2e config 1 on weights that do not attenuate pixel errors; every pixel of recons /
    masks / recons_imgs, values not argmax, in the default arithmetic """
    g = load_golden("parity_k7.npz")
    savi, pred = build(7, 4, savi_family=family)
    dslots = synth.synth_tensor("unit.dec_slots", (2, 7, 128), "normal")
    out = savi(mode="decode", slots=gpu(dslots))
    errs = {k_: max_abs(out[k_].cpu(), g[f"{family}_dec7_{k_}"]) for k_ in ("recons_imgs", "recons", "masks")}
    videos = synth.synth_videos(2, 5, seed=0)
    tokens, lengths = synth.synth_captions(2, max_len=12, lengths=[9, 12], seed=0)
    noise = synth.synth_noise(2, 7, 128, seed=1)
    e2e = forward_eval(savi, pred, gpu(videos), 1, 4, caption_tokens=gpu(tokens),
                       caption_lengths=gpu(lengths), init_noise=noise)
    errs["c1_slot_history"] = max_abs(e2e["slot_history"].cpu(), g[f"{family}_c1_slot_history"])
    errs["c1_pred_slots"] = max_abs(e2e["pred_slots"].cpu(), g[f"{family}_c1_pred_slots"])
    errs["c1_recons_imgs"] = max_abs(e2e["recons_imgs"].cpu(), g[f"{family}_c1_recons_imgs"])
    errs["c1_masks"] = max_abs(e2e["masks"][:4].cpu(), g[f"{family}_c1_masks_s0"])
    errs["c1_recons"] = max_abs(e2e["recons"][3].cpu(), g[f"{family}_c1_recons_s0f3"])
    print(family, {k_: f"{v:.2e}" for k_, v in errs.items()})
    for name, err in errs.items():
        # decode-only rows isolate the decoder arithmetic: half the bar.  The e2e rows add the image of the
        # ~1e-5 slot deviation through an O(1) head (the exact-fp32 decoder measures the same 4-5e-5 there,
        # profiles/r02_parity_by_mode.md): the north-star bar itself
        assert err < (1e-4 if name.startswith("c1_") else PARITY_TOL), (family, name, err)


@torch.no_grad()
def test_parity_family_undamped_k30_full_resolution():
    """ K = 30: decode of one frame and the e2e config 2 (north star) with the O(1) RGB head """
    g = load_golden("parity_k30.npz")
    savi, pred = build(30, 19, savi_family="undamped")
    dslots = synth.synth_tensor("unit.dec_slots30", (1, 30, 128), "normal")
    out = savi(mode="decode", slots=gpu(dslots))
    errs = {k_: max_abs(out[k_].cpu(), g[f"undamped_dec30_{k_}"]) for k_ in ("recons_imgs", "recons", "masks")}
    videos = synth.synth_videos(1, 20, seed=0)
    tokens, lengths = synth.synth_captions(1, max_len=12, seed=0)
    noise = synth.synth_noise(1, 30, 128, seed=1)
    e2e = forward_eval(savi, pred, gpu(videos), 1, 19, caption_tokens=gpu(tokens),
                       caption_lengths=gpu(lengths), init_noise=noise)
    errs["c2_recons_imgs"] = max_abs(e2e["recons_imgs"].cpu(), g["undamped_c2_recons_imgs"])
    errs["c2_masks_f18"] = max_abs(e2e["masks"][18].cpu(), g["undamped_c2_masks_f18"])
    errs["c2_recons_f18"] = max_abs(e2e["recons"][18].cpu(), g["undamped_c2_recons_f18"])
    print("undamped K=30", {k_: f"{v:.2e}" for k_, v in errs.items()})
    for name, err in errs.items():
        # the 19-step rollout itself deviates by ~1e-5 on pred_slots (same in every decoder mode); its
        # image of that through the O(1) head is part of the e2e figures, hence the full bar there
        assert err < (1e-4 if name.startswith("c2_") else PARITY_TOL), (name, err)
    assert_same_slot_assignment(e2e["masks"], g["undamped_c2_masks_argmax"], "e2e config 2, undamped family (parity_k30)")


@torch.no_grad()
def test_e2e_config1_against_reference_golden(k7):
    """ config 1: K=7, B=2, 1 seed + 4 preds, ragged captions padded per batch """
    savi, pred = k7
    g = load_golden("e2e_c1.npz")
    videos = synth.synth_videos(2, 5, seed=0)
    tokens, lengths = synth.synth_captions(2, max_len=12, lengths=[9, 12], seed=0)
    noise = synth.synth_noise(2, 7, 128, seed=1)
    out = forward_eval(savi, pred, gpu(videos), 1, 4, caption_tokens=gpu(tokens),
                       caption_lengths=gpu(lengths), init_noise=noise)
    assert out["slot_history"].shape == (2, 5, 7, 128)
    assert max_abs(out["slot_history"].cpu(), g["slot_history"]) < 1e-4
    assert max_abs(out["pred_slots"].cpu(), g["pred_slots"]) < 1e-4
    assert max_abs(out["pred_imgs"].cpu(), g["pred_imgs"]) < 1e-4
    # zero differing pixels (rounds 3-4 allowed one pixel whose two largest masks were bitwise equal in this
    # implementation's output; it has resolved the reference's way since the end of round 3: the allowance is gone)
    assert_same_slot_assignment(out["masks"], g["masks_argmax"], "e2e config 1 (e2e_c1)")


@torch.no_grad()
def test_e2e_config2_against_reference_golden(k30):
    """ config 2 (north star): K=30, B=1, 1 seed + 19 preds """
    savi, pred = k30
    g = load_golden("e2e_c2.npz")
    videos = synth.synth_videos(1, 20, seed=0)
    tokens, lengths = synth.synth_captions(1, max_len=12, seed=0)
    noise = synth.synth_noise(1, 30, 128, seed=1)
    out = forward_eval(savi, pred, gpu(videos), 1, 19, caption_tokens=gpu(tokens),
                       caption_lengths=gpu(lengths), init_noise=noise)
    assert max_abs(out["slot_history"].cpu(), g["slot_history"]) < 1e-4
    assert max_abs(out["pred_slots"].cpu(), g["pred_slots"]) < 1e-4
    assert max_abs(out["pred_imgs"][..., ::2, ::2].cpu(), g["pred_imgs_sub2"]) < 1e-4
    assert_same_slot_assignment(out["masks"][..., ::2, ::2], g["masks_argmax_sub2"], "e2e config 2 (e2e_c2, every 2nd pixel)")


@torch.no_grad()
@torch.no_grad()
@pytest.mark.parametrize("tag", ["l40", "l50"])
def test_e2e_long_captions_against_reference_golden(tag):
    """
    Captions beyond 16 tokens (up to the text encoder's 50, text_encoders.py:36; padded to the longest of the batch,
    padded positions attend, attention.py:303-319): a whole rollout + decode against the reference's own output on the
    "undamped" SAVi family (O(1) RGB head), lengths (24, 40) and (50, 17).  1e-4 on slots and every rendered pixel,
    identical slot-index maps.
    """
    g = load_golden("long_captions_k7.npz")
    savi, pred = build(7, 4, savi_family="undamped")
    videos = synth.synth_videos(2, 5, seed=0)
    noise = synth.synth_noise(2, 7, 128, seed=1)
    tokens, lengths = torch.from_numpy(g[f"{tag}_tokens"]), torch.from_numpy(g[f"{tag}_lengths"])
    out = forward_eval(savi, pred, gpu(videos), 1, 4, caption_tokens=gpu(tokens), caption_lengths=gpu(lengths),
                       init_noise=noise)
    errs = {"slot_history": max_abs(out["slot_history"].cpu(), g["slot_history"]),
            "pred_slots": max_abs(out["pred_slots"].cpu(), g[f"{tag}_pred_slots"]),
            "recons_imgs": max_abs(out["recons_imgs"].cpu(), g[f"{tag}_recons_imgs"]),
            "masks_s0": max_abs(out["masks"][:4].cpu(), g[f"{tag}_masks_s0"])}
    print(f"long captions {tag} (lengths {lengths.tolist()}):", {k_: f"{v:.2e}" for k_, v in errs.items()})
    assert all(v < 1e-4 for v in errs.values()), errs
    assert_same_slot_assignment(out["masks"], g[f"{tag}_masks_argmax"], f"e2e long captions {tag} (K=7, undamped)")


@torch.no_grad()
def test_fp16_plane_range_check(k7, monkeypatch):
    """ TOCVP_CHECK_RANGE: every fp16-plane kernel on the path sees operands inside |x| < 255 on the
    synthetic model, and an out-of-range activation is reported instead of saturating silently """
    from textocvp_amd import kernels as K
    monkeypatch.setattr(K, "_CHECK_RANGE", True)
    savi, pred = k7
    videos = gpu(synth.synth_videos(2, 5, seed=0))
    tokens, lengths = synth.synth_captions(2, max_len=12, lengths=[9, 12], seed=0)
    forward_eval(savi, pred, videos, 1, 4, caption_tokens=gpu(tokens), caption_lengths=gpu(lengths),
                 init_noise=synth.synth_noise(2, 7, 128, seed=1))
    w = synth.synth_tensor("rc.w", (64, 64, 5, 5), "uniform", 0.02).to(DEV)
    x = torch.full((1, 64, 64, 64), 300.0, device=DEV)
    with pytest.raises(K.TocvpError, match="out of the fp16-plane range"):
        K.conv5x5_f16f8(x, K.split_conv_weights_f16f8(w), torch.zeros(64, device=DEV))


@pytest.mark.skipif(os.environ.get("TOCVP_PRECISION") == "fp32", reason="all-fp32 mode has no fp16-plane modules")
@torch.no_grad()
def test_calibrate_precision_moves_out_of_range_modules():
    """ a checkpoint whose decoder activations leave the fp16-plane range is moved to the range-free
    arithmetic by the calibration pass, and the result is then finite and close to the fp32 path """
    from textocvp_amd.setup_model import calibrate_precision
    savi, pred = build(7, 4)
    with torch.no_grad():
        savi.decoder.decoder[1].conv.weight.mul_(40.0)          # layer-1 activations far above 255
    videos = gpu(synth.synth_videos(2, 5, seed=0))
    tokens, lengths = synth.synth_captions(2, max_len=12, lengths=[9, 12], seed=0)
    kw = dict(caption_tokens=gpu(tokens), caption_lengths=gpu(lengths),
              init_noise=synth.synth_noise(2, 7, 128, seed=1))
    saturated = forward_eval(savi, pred, videos, 1, 4, **kw)       # f16f8 planes pinned at 255.9
    changed = calibrate_precision(savi, pred, videos, 1, 4, **kw)
    assert changed == {("ConvDecoder", "conv_precision"): "bf16x3"}
    out = forward_eval(savi, pred, videos, 1, 4, **kw)
    savi.decoder.conv_precision = "fp32"
    ref = forward_eval(savi, pred, videos, 1, 4, **kw)
    assert torch.isfinite(out["pred_imgs"]).all()
    err_cal = max_abs(out["pred_imgs"].cpu(), ref["pred_imgs"].cpu())
    err_sat = max_abs(saturated["pred_imgs"].cpu(), ref["pred_imgs"].cpu())
    print(f"pixels vs fp32 path: saturating fp16 planes {err_sat:.2e}, after calibration {err_cal:.2e}")
    assert err_cal < 2e-2 and err_cal < err_sat / 5
    # an in-range model is left alone
    savi2, pred2 = build(7, 4)
    assert calibrate_precision(savi2, pred2, videos, 1, 4, **kw) == {}


@pytest.mark.skipif(os.environ.get("TOCVP_PRECISION") == "fp32", reason="all-fp32 mode has no fp16-plane modules")
@torch.no_grad()
def test_checked_pass_sees_the_activations_that_only_exist_as_operand_planes(monkeypatch):
    """
    Two activations never reach HBM as fp32 on the default path: the decoder's last hidden layer (its epilogue feeds
    the folded tail with fp16 planes) and the predictor MLP's hidden layer (the up-projection's epilogue writes the
    planes the down-projection reads).  A checkpoint that is in range everywhere EXCEPT there must still be caught
    by the checked pass: it runs those two hand-overs in fp32, verifies them, and moves the owner to its fallback.
    """
    from textocvp_amd import kernels as K
    from textocvp_amd.models.Blocks import attention as A
    from textocvp_amd.setup_model import calibrate_precision
    videos = gpu(synth.synth_videos(2, 5, seed=0))
    tokens, lengths = synth.synth_captions(2, max_len=12, lengths=[9, 12], seed=0)
    kw = dict(caption_tokens=gpu(tokens), caption_lengths=gpu(lengths),
              init_noise=synth.synth_noise(2, 7, 128, seed=1))
    # (a) decoder: only the LAST hidden layer's output is large (its input, layer 2's output, stays small)
    savi, pred = build(7, 4)
    assert savi.decoder.tail_fold and savi.decoder.conv_precision == "f16x3"
    last = savi.decoder.decoder[len(savi.decoder.hidden_dims) - 1].conv
    last.bias.add_(400.0)                                            # relu(y3) ~ 400 > 255.9
    dslots = gpu(synth.synth_tensor("unit.dec_slots", (2, 7, 128), "normal"))
    with K.check_range(True):
        with pytest.raises(K.TocvpRangeError, match="last hidden activation") as ei:
            savi(mode="decode", slots=dslots)
    assert ei.value.owner == (savi.decoder, "conv_precision")
    assert calibrate_precision(savi, pred, videos, 1, 4, **kw) == {("ConvDecoder", "conv_precision"): "bf16x3"}
    # (b) predictor: only the MLP's hidden activation is large; planes forced on at this small row count
    monkeypatch.setattr(A, "_PRESPLIT_MIN_ROWS", 0)
    savi, pred = build(7, 4)
    blk = pred.predictor.predictor[0]
    blk.mlp[0].bias.add_(400.0)                                      # relu(hidden) ~ 400
    blk.mlp[2].weight.mul_(1e-3)                                     # keeps everything downstream in range
    assert calibrate_precision(savi, pred, videos, 1, 4, **kw) == {
        (type(pred.predictor).__name__, "gemm_precision"): "bf16x6"}


@pytest.mark.skipif(os.environ.get("TOCVP_PRECISION") == "fp32", reason="all-fp32 mode has no fp16-plane modules")
@torch.no_grad()
def test_loaded_weights_are_range_checked_once():
    """ load_state_dict marks the module unchecked: a direct call then fails LOUDLY on out-of-range
    operands (never silent saturation), forward_eval re-calibrates by itself and says so """
    from textocvp_amd import kernels as K
    savi, pred = build(7, 4)
    sd = {k_: v.clone() for k_, v in savi.state_dict().items()}
    sd["decoder.decoder.1.block.0.weight"] *= 40.0                # layer-1 activations far above 255
    assert not savi._range_unchecked
    savi.load_state_dict(sd)
    assert savi._range_unchecked and not pred._range_unchecked
    dslots = gpu(synth.synth_tensor("unit.dec_slots", (2, 7, 128), "normal"))
    with pytest.raises(K.TocvpRangeError, match="out of the fp16-plane range") as ei:
        savi(mode="decode", slots=dslots)
    assert ei.value.owner == (savi.decoder, "conv_precision") and savi._range_unchecked
    videos = gpu(synth.synth_videos(2, 5, seed=0))
    tokens, lengths = synth.synth_captions(2, max_len=12, lengths=[9, 12], seed=0)
    with pytest.warns(UserWarning, match="arithmetic changed"):
        out = forward_eval(savi, pred, videos, 1, 4, caption_tokens=gpu(tokens), caption_lengths=gpu(lengths),
                           init_noise=synth.synth_noise(2, 7, 128, seed=1))
    assert savi.decoder.conv_precision == "bf16x3" and not savi._range_unchecked
    assert torch.isfinite(out["pred_imgs"]).all()
    # in-range weights: the checked first call passes and clears the mark, nothing is changed
    savi2, _ = build(7, 4)
    savi2.load_state_dict({k_: v.clone() for k_, v in savi2.state_dict().items()})
    assert savi2._range_unchecked
    savi2(mode="decode", slots=dslots)
    assert not savi2._range_unchecked and savi2.decoder.conv_precision == "f16x3"
    # a weight beyond the direct kernel's |w| < 63 is caught when its fp16 weight image is built; the Winograd form scales
    # its weight rows from the weights at hand (no limit), and the range check sees what such a weight does to the activations
    sd2 = {k_: v.clone() for k_, v in savi2.state_dict().items()}
    sd2["decoder.decoder.2.block.0.weight"][0, 0, 0, 0] = 100.0
    savi2.load_state_dict(sd2)
    with pytest.raises(K.TocvpRangeError, match="out of the fp16-plane range" if savi2.decoder.conv_wino
                       else "weight out of the fp16-plane range"):
        savi2(mode="decode", slots=dslots)


@torch.no_grad()
def test_decode_overlap_is_bit_identical(k7):
    """ decoding on the second stream (the default below 96 sequences) runs the same kernels on the
    same data as the serial order: every output must match bit for bit """
    savi, pred = k7
    videos = gpu(synth.synth_videos(2, 6, seed=3))
    tokens, lengths = synth.synth_captions(2, max_len=10, seed=3)
    noise = synth.synth_noise(2, 7, 128, seed=5)
    outs = [forward_eval(savi, pred, videos, 2, 4, overlap_decode=ov, caption_tokens=gpu(tokens),
                         caption_lengths=gpu(lengths), init_noise=noise) for ov in (False, True)]
    for key in ("slot_history", "pred_slots", "pred_imgs", "masks", "recons", "recons_imgs", "targets"):
        assert torch.equal(outs[0][key], outs[1][key]), key
    # round 4: the tail kernel places every decode straight into the (B * P, ...) results and writes the clamped
    # frames itself -- it must give what ONE plain decode call + reshape + torch clamp give (05_evaluate_predictor.py:88-96)
    o = outs[1]
    plain = savi(mode="decode", slots=o["pred_slots"].reshape(2 * 4, 7, 128))
    for key in ("recons_imgs", "recons", "masks"):
        assert torch.equal(plain[key], o[key]), key
    assert torch.equal(plain["recons_imgs"].view(2, 4, 3, 64, 64).clamp(0, 1), o["pred_imgs"])
    assert torch.equal(videos[:, 2:6].clamp(0, 1), o["targets"])
    bad = videos.clone()
    bad[0, 3, 1, 5, 7], bad[1, 2, 0, 0, 0], bad[1, 5, 2, 63, 63] = float("nan"), -0.25, 1.5
    from textocvp_amd import kernels as K
    got, want = K.clamp01_rows(bad[:, 2:6]), bad[:, 2:6].clamp(0, 1)
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(got.nan_to_num(7.0), want.nan_to_num(7.0))


def test_e2e_against_oracle_fresh_inputs(k7):
    """ seeds not covered by the fixtures, B=3 with three different caption lengths """
    savi, pred = k7
    ssd = {k: v.cpu() for k, v in savi.state_dict().items()}
    psd = {k: v.cpu() for k, v in pred.state_dict().items()}
    videos = synth.synth_videos(3, 5, seed=11)
    tokens, lengths = synth.synth_captions(3, max_len=15, lengths=[4, 15, 9], seed=5)
    noise = synth.synth_noise(3, 7, 128, seed=12)
    hist, preds, imgs, masks = O.forward_eval(ssd, psd, videos, tokens, lengths, noise, 1, 4)
    out = forward_eval(savi, pred, gpu(videos), 1, 4, caption_tokens=gpu(tokens),
                       caption_lengths=gpu(lengths), init_noise=noise)
    assert max_abs(out["slot_history"].cpu(), hist) < 1e-4
    assert max_abs(out["pred_slots"].cpu(), preds) < 1e-4
    assert max_abs(out["pred_imgs"].cpu(), imgs) < 1e-4
    # ZERO differing slot-index pixels, except ONE NAMED pixel: frame 11 (sample 2, prediction 3), row 23, column 2, where
    # slots 1 and 6 tie -- the ORACLE's own two largest masks there are 0.3329 and 0.3329, 5.4e-6 apart, so two fp32
    # evaluations that each hold the masks to a few 1e-6 may rank them either way (history: 2.1e-7 apart in this implementation's output with split-K, 6.3e-7
    # with the folded tail, 1.3e-6 since the attention's exponentials are one v_exp_f32 each (round 5); the masks themselves agree
    # to 4e-6 there, the bar is 1e-4).  Every reference-generated fixture stays at zero differing pixels.
    got_am, ref_am = out["masks"].argmax(dim=1).cpu(), masks.argmax(dim=1)
    named = (11, 0, 23, 2)
    differ = [tuple(i) for i in (got_am != ref_am).nonzero().tolist()]
    assert differ in ([], [named]), f"slot-index maps differ at {differ}"
    top2 = masks[named[0], :, 0, named[2], named[3]].topk(2)
    # the oracle's own tie: its two largest masks there are 5.4e-6 apart (a twentieth of the 1e-4 bar; masks of ~0.33)
    assert float(top2.values[0] - top2.values[1]) < 1e-5 and set(top2.indices.tolist()) == {1, 6}
    if differ:
        assert {int(got_am[named]), int(ref_am[named])} == {1, 6}
    slot_assignment_diff(out["masks"], ref_am, "e2e fresh inputs vs oracle (K=7, B=3)")      # recorded in the argmax report


@torch.no_grad()
def test_full_size_properties(k30):
    """
    Size-independent properties at the bench shape (K=30, 1+19, B=4):
      * samples do not interact: a batch of 4 equals four batch-of-1 runs (same caption length);
      * masks are a partition of unity over slots and rendered frames are their convex blend;
      * decode=True in forward_decomp equals decode() of the returned slot_history.
    """
    savi, pred = k30
    B = 4
    videos = gpu(synth.synth_videos(B, 20, seed=21))
    tokens, lengths = synth.synth_captions(B, max_len=12, seed=21)
    noise = synth.synth_noise(B, 30, 128, seed=22)
    out = forward_eval(savi, pred, videos, 1, 19, caption_tokens=gpu(tokens),
                       caption_lengths=gpu(lengths), init_noise=noise)
    for b in (0, 3):
        one = forward_eval(savi, pred, videos[b:b + 1], 1, 19, caption_tokens=gpu(tokens[b:b + 1]),
                           caption_lengths=gpu(lengths[b:b + 1]), init_noise=noise[b:b + 1])
        assert max_abs(one["pred_slots"].cpu(), out["pred_slots"][b:b + 1].cpu()) < 2e-5
        assert max_abs(one["pred_imgs"].cpu(), out["pred_imgs"][b:b + 1].cpu()) < 2e-5
    assert max_abs(out["masks"].sum(dim=1).cpu(), torch.ones(B * 19, 1, 64, 64)) < 1e-5
    assert torch.isfinite(out["pred_slots"]).all() and torch.isfinite(out["pred_imgs"]).all()

    dec = savi(mode="decomp", x=videos[:1], num_imgs=3, decode=True, init_noise=noise[:1])
    assert dec["recons_imgs"].shape == (1, 3, 3, 64, 64)
    assert dec["recons_objs"].shape == (1, 3, 30, 3, 64, 64) and dec["masks"].shape == (1, 3, 30, 1, 64, 64)
    again = savi(mode="decode", slots=dec["slot_history"].reshape(3, 30, 128))
    assert max_abs(again["recons_imgs"].cpu(), dec["recons_imgs"][0].cpu()) == 0.0
    nodec = savi(mode="decomp", x=videos[:1], num_imgs=3, decode=False, init_noise=noise[:1])
    assert nodec["recons_imgs"].shape == (0, 3)
    assert max_abs(nodec["slot_history"].cpu(), dec["slot_history"].cpu()) == 0.0


@pytest.mark.parametrize("B", [128, 256])
@torch.no_grad()
def test_bench_shape_b128_against_b1_runs_and_oracle(k30, B):
    """
    The MEASURED shapes (bench.py: B = 256 sequences per GPU since the end of round 4, B = 128 before and as its
    `extra.batch_128` leg; K = 30, 1 seed + 19 preds): the encoder runs in 3 / 5 chunks of <= 1024 images, the decoder in
    36 / 72 chunks of 2040 slot images, slot attention splits every sample over 2 workgroups / takes one each, the GEMMs
    see their largest M (38400 / 76800 rows: fused MLP with a cut last round, mid-size and chunk-stream kernels on the
    short windows).  None of the B <= 4 tests crosses those boundaries.
      * samples 0, B / 2 - 1 and B - 1 of the batch equal their own batch-of-1 runs (<= 2e-5: GEMM tile shapes differ
        with M, every per-sample kernel is batch-invariant);
      * sample 0 equals the CPU oracle (<= 1e-4, the north-star bar) incl. the slot-index map of every frame.
    """
    savi, pred = k30
    P = 19
    videos = gpu(synth.synth_videos(B, 1 + P, seed=61))
    tokens, lengths = synth.synth_captions(B, max_len=12, seed=61)
    noise = synth.synth_noise(B, 30, 128, seed=62)
    out = forward_eval(savi, pred, videos, 1, P, caption_tokens=gpu(tokens), caption_lengths=gpu(lengths),
                       init_noise=noise, overlap_decode=False)
    assert out["pred_imgs"].shape == (B, P, 3, 64, 64) and out["masks"].shape == (B * P, 30, 1, 64, 64)
    assert torch.isfinite(out["pred_slots"]).all() and torch.isfinite(out["pred_imgs"]).all()
    worst = {}
    for b in (0, B // 2 - 1, B - 1):
        one = forward_eval(savi, pred, videos[b:b + 1], 1, P, caption_tokens=gpu(tokens[b:b + 1]),
                           caption_lengths=gpu(lengths[b:b + 1]), init_noise=noise[b:b + 1])
        errs = {"slot_history": max_abs(one["slot_history"].cpu(), out["slot_history"][b:b + 1].cpu()),
                "pred_slots": max_abs(one["pred_slots"].cpu(), out["pred_slots"][b:b + 1].cpu()),
                "pred_imgs": max_abs(one["pred_imgs"].cpu(), out["pred_imgs"][b:b + 1].cpu()),
                "masks": max_abs(one["masks"].cpu(), out["masks"][b * P:(b + 1) * P].cpu())}
        print(f"B={B} sample {b} vs its B=1 run:", {k_: f"{v:.2e}" for k_, v in errs.items()})
        for k_, v in errs.items():
            worst[k_] = max(worst.get(k_, 0.0), v)
    assert all(v <= 2e-5 for v in worst.values()), worst
    ssd = {k_: v.cpu() for k_, v in savi.state_dict().items()}
    psd = {k_: v.cpu() for k_, v in pred.state_dict().items()}
    hist, preds, imgs, masks = O.forward_eval(ssd, psd, videos[:1].cpu(), tokens[:1], lengths[:1], noise[:1], 1, P)
    errs = {"slot_history": max_abs(out["slot_history"][:1].cpu(), hist),
            "pred_slots": max_abs(out["pred_slots"][:1].cpu(), preds),
            "pred_imgs": max_abs(out["pred_imgs"][:1].cpu(), imgs),
            "masks": max_abs(out["masks"][:P].cpu(), masks)}
    print(f"B={B} sample 0 vs CPU oracle:", {k_: f"{v:.2e}" for k_, v in errs.items()})
    assert all(v < 1e-4 for v in errs.values()), errs
    assert_same_slot_assignment(out["masks"][:P], masks.argmax(dim=1), f"B={B} sample 0 vs oracle (K=30, 19 frames)")
    # overlapped decode (second stream) at this size: bit-identical to the serial order
    ov = forward_eval(savi, pred, videos, 1, P, caption_tokens=gpu(tokens), caption_lengths=gpu(lengths),
                      init_noise=noise, overlap_decode=True)
    assert torch.equal(ov["pred_imgs"], out["pred_imgs"]) and torch.equal(ov["pred_slots"], out["pred_slots"])


@pytest.mark.skipif(os.environ.get("TOCVP_PRECISION") == "fp32", reason="the whole suite already runs in that mode")
@torch.no_grad()
def test_fp32_mode_in_process_slot_assignment_is_exact(monkeypatch):
    """
    The all-fp32 arithmetic (TOCVP_PRECISION=fp32: every GEMM / conv / attention product on the exact fp32 MFMA;
    the slot-attention iteration keeps its split-fp16 contraction) selected IN THIS PROCESS before the models are
    built: the reference goldens hold at the bar and the slot-index maps are identical on config 1.  On the
    config-2 fixture of the "undamped" family ONE of 77824 pixels is a tie: measured top-2 margin 2.4e-6 (the
    masks themselves agree with the reference to 1.6e-5 there), resolved the reference's way by the default
    arithmetic and the other way by this one -- named here, with its margin bounded, instead of a tolerance band.
    """
    from textocvp_amd import kernels as K
    monkeypatch.setenv("TOCVP_PRECISION", "fp32")
    monkeypatch.setattr(K, "_ATTN_QK16", False)
    savi, pred = build(7, 4)
    assert savi.decoder.conv_precision == "fp32" and pred.predictor.gemm_precision == "fp32"
    g = load_golden("e2e_c1.npz")
    videos = synth.synth_videos(2, 5, seed=0)
    tokens, lengths = synth.synth_captions(2, max_len=12, lengths=[9, 12], seed=0)
    noise = synth.synth_noise(2, 7, 128, seed=1)
    out = forward_eval(savi, pred, gpu(videos), 1, 4, caption_tokens=gpu(tokens), caption_lengths=gpu(lengths),
                       init_noise=noise)
    assert max_abs(out["slot_history"].cpu(), g["slot_history"]) < 1e-4
    assert max_abs(out["pred_slots"].cpu(), g["pred_slots"]) < 1e-4
    assert max_abs(out["pred_imgs"].cpu(), g["pred_imgs"]) < 1e-4
    assert_same_slot_assignment(out["masks"], g["masks_argmax"], "e2e config 1 (e2e_c1), fp32 mode in-process")
    savi30, pred30 = build(30, 19, savi_family="undamped")
    g30 = load_golden("parity_k30.npz")
    videos = synth.synth_videos(1, 20, seed=0)
    tokens, lengths = synth.synth_captions(1, max_len=12, seed=0)
    noise = synth.synth_noise(1, 30, 128, seed=1)
    e2e = forward_eval(savi30, pred30, gpu(videos), 1, 19, caption_tokens=gpu(tokens), caption_lengths=gpu(lengths),
                       init_noise=noise)
    assert max_abs(e2e["recons_imgs"].cpu(), g30["undamped_c2_recons_imgs"]) < 1e-4
    assert_same_slot_assignment(e2e["masks"], g30["undamped_c2_masks_argmax"],
                                "e2e config 2, undamped family (parity_k30), fp32 mode in-process")


@torch.no_grad()
def test_slot_permutation_equivariance(k7):
    """ permuting the initial slots permutes slot_history identically (slot-index bookkeeping) """
    savi, _ = k7
    videos = gpu(synth.synth_videos(1, 3, seed=31))
    noise = synth.synth_noise(1, 7, 128, seed=32)
    perm = torch.tensor([3, 0, 6, 1, 5, 2, 4])
    a = savi(mode="decomp", x=videos, num_imgs=3, decode=False, init_noise=noise)["slot_history"]
    b = savi(mode="decomp", x=videos, num_imgs=3, decode=False,
             init_noise=noise[:, perm])["slot_history"]
    assert max_abs(a[:, :, perm].cpu(), b.cpu()) < 5e-5


@torch.no_grad()
def test_rollout_options_against_oracle():
    """
    Wrapper options the fixtures do not cover: 3 seed frames, a 4-frame sliding buffer (slides
    after the second step), teacher forcing (the reference applies the config value in eval mode
    too, predictor_wrapper.py:136-139) and the num_preds override.
    """
    exp = default_exp_params(num_slots=7, num_context=3, num_preds=5, input_buffer_size=4)
    savi = setup_model(exp["model"]).eval()
    synth.fill_module_(savi, prefix="savi.")
    videos = synth.synth_videos(2, 8, seed=41)
    tokens, lengths = synth.synth_captions(2, max_len=10, lengths=[10, 6], seed=42)
    noise = synth.synth_noise(2, 7, 128, seed=43)
    ssd = {k: v.clone() for k, v in savi.state_dict().items()}
    hist_ref = O.savi_decomp(ssd, videos, noise, 8)
    savi = savi.to(DEV)
    hist = savi(mode="decomp", x=gpu(videos), num_imgs=8, decode=False, init_noise=noise)["slot_history"]
    assert max_abs(hist.cpu(), hist_ref) < 1e-4
    for tf in (False, True):
        exp["prediction_params"]["teacher_force"] = tf
        pred = setup_predictor(exp).eval()
        synth.fill_module_(pred, prefix="pred.")
        psd = {k: v.clone() for k, v in pred.state_dict().items()}
        ref = O.rollout(psd, hist_ref, tokens, lengths, 3, 5, buffer_size=4, teacher_force=tf)
        pred = pred.to(DEV)
        got = pred(hist, caption_tokens=gpu(tokens), caption_lengths=gpu(lengths))
        assert got.shape == (2, 5, 7, 128)
        assert max_abs(got.cpu(), ref) < 1e-4, f"teacher_force={tf}"
        got2 = pred(hist, num_preds=2, caption_tokens=gpu(tokens), caption_lengths=gpu(lengths))
        assert max_abs(got2.cpu(), ref[:, :2]) < 1e-4


@torch.no_grad()
def test_forward_eval_with_teacher_forcing_reads_the_whole_history():
    """
    A wrapper with teacher_force=True reads slot_history[:, num_context + t] inside the rollout
    (predictor_wrapper.py:74-82; the reference applies the config value in eval mode too), and a wrapper whose
    own num_context is larger than the evaluator's reads more context frames.  forward_eval must then hand the
    predictor the FULL decomposition instead of the context-only cut of its overlapped-encode branch
    (evaluator._reads_context_only), and equal the serial order bit for bit and the oracle at the bar.
    """
    from textocvp_amd.evaluator import _reads_context_only
    exp = default_exp_params(num_slots=7, num_context=2, num_preds=3, input_buffer_size=4)
    exp["prediction_params"]["teacher_force"] = True
    savi = setup_model(exp["model"]).eval()
    pred = setup_predictor(exp).eval()
    synth.fill_module_(savi, prefix="savi.")
    synth.fill_module_(pred, prefix="pred.")
    videos = synth.synth_videos(2, 5, seed=51)
    tokens, lengths = synth.synth_captions(2, max_len=9, lengths=[9, 5], seed=52)
    noise = synth.synth_noise(2, 7, 128, seed=53)
    ssd = {k: v.clone() for k, v in savi.state_dict().items()}
    psd = {k: v.clone() for k, v in pred.state_dict().items()}
    hist_ref = O.savi_decomp(ssd, videos, noise, 5)
    ref = O.rollout(psd, hist_ref, tokens, lengths, 2, 3, buffer_size=4, teacher_force=True)
    savi, pred = savi.to(DEV), pred.to(DEV)
    assert not _reads_context_only(pred, 2)
    kw = dict(caption_tokens=gpu(tokens), caption_lengths=gpu(lengths), init_noise=noise)
    outs = [forward_eval(savi, pred, gpu(videos), 2, 3, overlap_decode=ov, **kw) for ov in (False, True)]
    for key in ("slot_history", "pred_slots", "pred_imgs", "masks"):
        assert torch.equal(outs[0][key], outs[1][key]), key
    assert max_abs(outs[1]["pred_slots"].cpu(), ref) < 1e-4
    assert max_abs(outs[1]["slot_history"].cpu(), hist_ref) < 1e-4
    # without teacher forcing the cut is taken -- unless the wrapper itself wants more context than the evaluator
    exp["prediction_params"]["teacher_force"] = False
    assert _reads_context_only(pred, 2) and not _reads_context_only(pred, 1)
    out = forward_eval(savi, pred, gpu(videos), 1, 3, overlap_decode=True, **kw)     # wrapper context 2 > 1
    assert out["pred_slots"].shape == (2, 3, 7, 128)


@torch.no_grad()
def test_learned_initializer_and_identity_transition():
    """ the other initialiser / transition choices of the SAVi factory (initializers.py:39-61) """
    exp = default_exp_params(num_slots=7)
    exp["model"]["model_params"]["initializer"] = "Learned"
    exp["model"]["model_params"]["transition_module"] = {"model_name": ""}
    savi = setup_model(exp["model"]).eval()
    synth.fill_module_(savi, prefix="savi2.")
    assert "initializer.slots" in savi.state_dict() and not any(
        k.startswith("transition_module") for k in savi.state_dict())
    videos = synth.synth_videos(2, 3, seed=51)
    ssd = {k: v.clone() for k, v in savi.state_dict().items()}
    ref = O.savi_decomp(ssd, videos, None, 3)
    got = savi.to(DEV)(mode="decomp", x=gpu(videos), num_imgs=3, decode=False)["slot_history"]
    assert max_abs(got.cpu(), ref) < 1e-4


@torch.no_grad()
@pytest.mark.parametrize("name", ["VanillaTransformer", "OCVPSeq"])
def test_unconditioned_predictors_against_reference_golden(name):
    """ the unconditioned OCVP predictors behind the same wrapper """
    g = load_golden("uncond_k7.npz")
    K, D = 7, 128
    exp = default_exp_params(num_slots=K, num_context=2, num_preds=4, predictor_name=name)
    pred = setup_predictor(exp).eval()
    synth.fill_module_(pred, prefix=f"{name}.")
    pred = pred.to(DEV)
    win = synth.synth_tensor("unit.win3", (2, 3, K, D), "normal")
    hist = synth.synth_tensor("unit.hist6", (2, 6, K, D), "normal")
    tokens, lengths = synth.synth_captions(2, max_len=8, seed=9)
    step = pred.predictor(slots=gpu(win))
    assert max_abs(step.cpu(), g[f"{name}_step_w3"]) < 5e-5
    roll = pred(gpu(hist), caption_tokens=gpu(tokens), caption_lengths=gpu(lengths))
    assert roll.shape == (2, 4, K, D)
    assert max_abs(roll.cpu(), g[f"{name}_rollout"]) < 1e-4


@torch.no_grad()
def test_dinosaur_decode_side_config4():
    """
    BASELINE config 4 downstream of the ViT: 24 slots, 224x224 (256 patches), decoder incl. CNN
    image head against the reference golden; recurrence from patch features against the oracle.
    """
    from textocvp_amd.setup_model import default_dinosaur_params
    g = load_golden("dinosaur_dec.npz")
    model = setup_model(default_dinosaur_params(num_slots=24, img_size=224)).eval()
    synth.fill_module_(model, prefix="dino.")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    slots = synth.synth_tensor("unit.dino_slots", (1, 24, 128), "normal")
    out = model(mode="decode", slots=gpu(slots))
    assert out["recons_imgs"].shape == (1, 3, 224, 224) and out["masks"].shape == (1, 24, 1, 16, 16)
    assert max_abs(out["recons_feats"][:, ::4, ::4].cpu(), g["recons_feats_sub4"]) < 1e-4
    assert max_abs(out["masks"].cpu(), g["masks"]) < 1e-5
    assert max_abs(out["recons_imgs"][..., ::2, ::2].cpu(), g["recons_imgs_sub2"]) < 2e-6

    feats = synth.synth_tensor("unit.dino_feats", (2, 3, 256, 768), "normal")
    noise = synth.synth_noise(2, 24, 128, seed=61)
    ref = O.dinosaur_decomp(sd, feats, noise)
    got = model(mode="decomp", num_imgs=3, decode=True, encoded_img_feats=gpu(feats), init_noise=noise)
    assert max_abs(got["slot_history"].cpu(), ref) < 1e-4
    assert got["recons_imgs"].shape == (2, 3, 3, 224, 224)
    assert got["recons_feats"].shape == (2, 3, 256, 768) and got["masks"].shape == (2, 3, 24, 1, 16, 16)
    ref_imgs, ref_feats, _ = O.mlp_patch_decoder(O.sub(sd, "decoder."), ref[0], img_size=224)
    assert max_abs(got["recons_feats"][0].cpu(), ref_feats) < 2e-4
    assert max_abs(got["recons_imgs"][0].cpu(), ref_imgs) < 5e-6


@torch.no_grad()
def test_vit_backbone_and_decomp_from_pixels_against_oracle():
    """ DINOv2 ViT-B/14 (timm_encoders.py:59-70, std := mean) on the kernels vs the CPU restatement of
    timm's published algorithm (parity UNPINNED: timm itself is absent), then ExtendedDINOSAUR.forward_decomp
    from pixels (the whole row a11) """
    from textocvp_amd.setup_model import default_dinosaur_params
    model = setup_model(default_dinosaur_params(num_slots=24, img_size=224)).eval()
    synth.fill_module_(model, prefix="dino.")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    videos = synth.synth_videos(2, 2, height=224, width=224, seed=71)
    vit_sd = O.sub(sd, "encoder.vit_backbone.")
    ref_feats = torch.stack([O.vit_encoder(vit_sd, videos[b]) for b in range(2)])        # (2, 2, 256, 768)
    feats = model.encoder(gpu(videos))
    scale = float(ref_feats.abs().max())
    err = max_abs(feats.cpu(), ref_feats)
    print(f"ViT-B/14 features: max abs err {err:.2e} at scale {scale:.3g}")
    assert feats.shape == (2, 2, 256, 768) and err < 2e-5 * max(scale, 1.0)
    # one frame alone (4-dim input) equals its row of the batch
    assert max_abs(model.encoder(gpu(videos[1, 0:1])).cpu(), feats[1, 0:1].cpu()) < 1e-5 * max(scale, 1.0)
    noise = synth.synth_noise(2, 24, 128, seed=61)
    ref_hist = O.dinosaur_decomp(sd, ref_feats, noise)
    got = model(mode="decomp", x=gpu(videos), num_imgs=2, decode=True, init_noise=noise)
    assert max_abs(got["slot_history"].cpu(), ref_hist) < 1e-4
    assert max_abs(got["encoded_img_feats"].cpu(), ref_feats) < 2e-5 * max(scale, 1.0)
    assert got["recons_imgs"].shape == (2, 2, 3, 224, 224) and got["masks"].shape == (2, 2, 24, 1, 16, 16)


@torch.no_grad()
def test_vit_backbone_against_dinov2_golden():
    """
    Row a11, the backbone, PINNED (round 5): the DINOv2 ViT-B/14 forward of timm_encoders.py:59-70 on the kernels
    against transformers.Dinov2Model run through the reference's ViTEncoder wrapper (dinov2_vit.npz,
    make_golden.py::dinov2_fixtures; the oracle holds the same vectors to 6.5e-7 relative,
    test_oracle_golden.py::test_vit_oracle_vs_dinov2).  224 x 224 and the reference's default 336 x 336 (577 tokens).
    Bar: 2e-5 of the feature scale (~40: LayerScale gains are O(1) in the synthetic weights, 1e-5 in a checkpoint).
    """
    from textocvp_amd.setup_model import default_dinosaur_params
    g = load_golden("dinov2_vit.npz")
    for size, frames, step in ((224, synth.synth_videos(2, 2, height=224, width=224, seed=71), 4),
                               (336, synth.synth_videos(1, 1, height=336, width=336, seed=72), 8)):
        model = setup_model(default_dinosaur_params(num_slots=24, img_size=size)).eval()
        synth.fill_module_(model, prefix="dino.")
        enc = model.encoder.to(DEV)
        ref = torch.from_numpy(g[f"feats{size}_sub{step}"])
        scale = float(ref.abs().max())
        feats = enc(gpu(frames)).flatten(0, 1)
        err = max_abs(feats[:, ::step].cpu(), ref)
        print(f"ViT-B/14 {size}: kernels vs transformers.Dinov2Model {err:.2e} at scale {scale:.3g}")
        assert feats.shape == (frames.shape[0] * frames.shape[1], (size // 14) ** 2, 768) and err < 2e-5 * scale
        del model, enc


def _config4_pair(P):
    from textocvp_amd.setup_model import default_dinosaur_params
    model = setup_model(default_dinosaur_params(num_slots=24, img_size=224)).eval()
    exp = default_exp_params(num_slots=24, num_context=1, num_preds=P, predictor_name="TextOCVP_T5")
    pred = setup_predictor(exp).eval()
    synth.fill_module_(model, prefix="dino.", family="undamped")         # sharp alpha masks (synth.py)
    synth.fill_module_(pred, prefix="pred.")
    return model, pred


@pytest.mark.parametrize("tag,B,P,seed,sub", [("c4s", 2, 3, 83, 2), ("c4", 1, 29, 81, 4)])
@torch.no_grad()
def test_e2e_config4_against_reference_golden(tag, B, P, seed, sub):
    """
    BASELINE configs[3] END TO END FROM PIXELS against the reference's own glue (e2e_c4.npz: ExtendedDINOSAUR.forward_decomp
    models/ExtendedDINOSAUR.py:139-208 -> PredictorWrapper(TextOCVP_T5) predictor_wrapper.py:50-87 -> decode :211-214 ->
    clamp, the three calls of 05_evaluate_predictor.py:82-96; transformers' Dinov2Model / T5EncoderModel as the two
    third-party encoders):  "c4" = the workload itself, B = 1, 24 slots, 224 x 224, 1 seed + 29 preds;  "c4s" = B = 2,
    1 + 3, ragged T5 masks.  Bar: 1e-4 on slots, predicted frames and alpha masks, IDENTICAL argmax_K(masks) maps.
    The same inputs through the CPU oracle (O.forward_eval_dinosaur) hold the golden to 1.3e-5
    (test_oracle_golden.py::test_e2e_c4_*); at "c4s" this test runs the oracle as well (every pixel, recons_feats).
    """
    g = load_golden("e2e_c4.npz")
    model, pred = _config4_pair(P)
    dsd = {k: v.clone() for k, v in model.state_dict().items()}
    psd = {k: v.clone() for k, v in pred.state_dict().items()}
    model, pred = model.to(DEV), pred.to(DEV)
    videos = synth.synth_videos(B, 1 + P, height=224, width=224, seed=seed)
    noise = synth.synth_noise(B, 24, 128, seed=seed + 1)
    ids, mask = torch.from_numpy(g[f"{tag}_ids"]), torch.from_numpy(g[f"{tag}_mask"])
    out = forward_eval(model, pred, gpu(videos), 1, P, caption_tokens=gpu(ids), attn_masks=gpu(mask), init_noise=noise)
    fs = float(np.abs(g[f"{tag}_feats_f0_sub8"]).max())
    errs = {"slot_history": max_abs(out["slot_history"].cpu(), g[f"{tag}_slot_history"]),
            "pred_slots": max_abs(out["pred_slots"].cpu(), g[f"{tag}_pred_slots"]),
            "pred_imgs": max_abs(out["pred_imgs"][..., ::sub, ::sub].cpu(), g[f"{tag}_pred_imgs_sub{sub}"])}
    if tag == "c4s":
        errs["masks"] = max_abs(out["masks"].cpu(), g["c4s_masks"])
    else:
        errs["masks"] = max_abs(out["masks"][[0, 14, 28]].cpu(), g["c4_masks_f0_14_28"])
    print(f"e2e_{tag} vs reference golden:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert out["pred_imgs"].shape == (B, P, 3, 224, 224) and out["masks"].shape == (B * P, 24, 1, 16, 16)
    assert max(errs.values()) < 1e-4, errs
    assert_same_slot_assignment(out["masks"], g[f"{tag}_masks_argmax"], f"e2e_{tag} (reference golden)")
    if tag == "c4s":
        ref = O.forward_eval_dinosaur(dsd, psd, videos, ids, mask, noise, 1, P)
        assert max_abs(out["pred_imgs"].cpu(), ref["pred_imgs"]) < 1e-4          # every pixel
        assert max_abs(out["pred_slots"].cpu(), ref["pred_slots"]) < 1e-4
        assert_same_slot_assignment(out["masks"], ref["masks"].argmax(dim=1), f"e2e_{tag} (oracle)")


@torch.no_grad()
def test_e2e_config4_dinosaur_from_pixels_properties():
    """
    BASELINE configs[3] at its workload: ExtendedDINOSAUR (ViT-B/14 backbone) 24 slots, 224x224, 1 seed + 29
    preds, TextOCVP_T5 predictor, from PIXELS through the reference's forward_eval glue.  No reference vector
    exists at this size (timm / hub weights absent), so the assertions are the size-independent ones:
    samples do not interact, masks are a partition of unity, decode is idempotent, everything finite; the
    pieces are pinned separately (dinosaur_dec.npz, t5_encoder.npz, the ViT test above).
    """
    from textocvp_amd.setup_model import default_dinosaur_params
    K_, P = 24, 29
    model = setup_model(default_dinosaur_params(num_slots=K_, img_size=224)).eval()
    exp = default_exp_params(num_slots=K_, num_context=1, num_preds=P, predictor_name="TextOCVP_T5")
    pred = setup_predictor(exp).eval()
    synth.fill_module_(model, prefix="dino.")
    synth.fill_module_(pred, prefix="pred.")
    model, pred = model.to(DEV), pred.to(DEV)
    B = 2
    videos = gpu(synth.synth_videos(B, 1 + P, height=224, width=224, seed=81))
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(1, 32000, (B, 16), generator=g)
    mask = torch.ones(B, 16, dtype=torch.int64)
    mask[1, 11:] = 0
    ids = ids * mask
    noise = synth.synth_noise(B, K_, 128, seed=82)
    out = forward_eval(model, pred, videos, 1, P, caption_tokens=gpu(ids), attn_masks=gpu(mask), init_noise=noise)
    assert out["slot_history"].shape == (B, 1 + P, K_, 128) and out["pred_slots"].shape == (B, P, K_, 128)
    assert out["pred_imgs"].shape == (B, P, 3, 224, 224) and out["masks"].shape == (B * P, K_, 1, 16, 16)
    for key in ("slot_history", "pred_slots", "pred_imgs", "masks"):
        assert torch.isfinite(out[key]).all(), key
    assert float(out["pred_imgs"].min()) >= 0.0 and float(out["pred_imgs"].max()) <= 1.0
    assert max_abs(out["masks"].sum(dim=1).cpu(), torch.ones(B * P, 1, 16, 16)) < 1e-5
    # sample independence (captions are padded to the same length, so the batch rows must not interact)
    one = forward_eval(model, pred, videos[1:2], 1, P, caption_tokens=gpu(ids[1:2]), attn_masks=gpu(mask[1:2]),
                       init_noise=noise[1:2])
    assert max_abs(one["pred_slots"].cpu(), out["pred_slots"][1:2].cpu()) < 5e-5
    assert max_abs(one["pred_imgs"].cpu(), out["pred_imgs"][1:2].cpu()) < 5e-5
    # decode idempotence: decoding the returned slots again (other batch composition) reproduces the frames
    again = model(mode="decode", slots=out["pred_slots"][0].contiguous())
    assert max_abs(again["recons_imgs"].clamp(0, 1).cpu(), out["pred_imgs"][0].cpu()) < 2e-5
    assert max_abs(again["masks"].cpu(), out["masks"][:P].cpu()) < 2e-5


@torch.no_grad()
def test_textocvp_t5_against_transformers_golden_and_oracle():
    """ TextOCVP_T5: T5-small text encoder on the kernels + rollout conditioned on it """
    g = load_golden("t5_encoder.npz")
    exp = default_exp_params(num_slots=7, num_context=2, num_preds=3, predictor_name="TextOCVP_T5")
    pred = setup_predictor(exp).eval()
    synth.fill_module_(pred.predictor.text_encoder, prefix="t5.")
    synth.fill_module_(pred.predictor.predictor, prefix="pred.predictor.predictor.")
    synth.fill_module_(pred.predictor.mlp_in, prefix="pred.predictor.mlp_in.")
    synth.fill_module_(pred.predictor.mlp_out, prefix="pred.predictor.mlp_out.")
    psd = {k: v.clone() for k, v in pred.state_dict().items()}
    pred = pred.to(DEV)
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    text = pred.predictor.text_encoder(input_ids=gpu(ids), attention_mask=gpu(mask))
    assert max_abs(text.cpu(), g["last_hidden_state"]) < 5e-5
    # rollout conditioned on the T5 embeddings vs the oracle predictor step
    hist = synth.synth_tensor("unit.hist6", (3, 6, 7, 128), "normal")
    got = pred(gpu(hist), caption_tokens=gpu(ids), attn_masks=gpu(mask))
    p = O.sub(psd, "predictor.")
    text_ref = torch.from_numpy(g["last_hidden_state"])
    window, ref = hist[:, :2].clone(), []
    for t in range(3):
        cur = O.text_ocvp_step(p, window, text_ref)
        window = torch.cat([window, cur.unsqueeze(1)], dim=1)
        ref.append(cur)
    assert max_abs(got.cpu(), torch.stack(ref, dim=1)) < 1e-4


@pytest.mark.skipif(os.environ.get("TOCVP_PRECISION") == "fp32", reason="operand planes exist in the f16x3 arithmetic only")
@torch.no_grad()
def test_config4_plane_handovers_and_chunk_gemm_are_bit_identical(monkeypatch):
    """
    BASELINE configs[3] at row counts that take the round-4 GEMM path: the MLPPatchDecoder (reference decoders.py:264-307)
    and the ViT blocks (timm_encoders.py:59-70) hand LayerNorm outputs / hidden activations to their wide projections as
    fp16 operand planes, which run the chunk-resident persistent GEMM (csrc/gemm_f16c.hip; the decoder's 769-wide head
    zero-padded to 1024 columns).  Producer-written planes are the split the consumer would compute and the kernel keeps
    the k order: features, masks and ViT tokens must equal the fp32 hand-over path BIT FOR BIT.
    """
    from textocvp_amd import kernels as K
    from textocvp_amd.setup_model import default_dinosaur_params
    from textocvp_amd.models.EncodersDecoders import decoders as D, timm_encoders as T
    model = setup_model(default_dinosaur_params(num_slots=24, img_size=224)).eval()
    synth.fill_module_(model, prefix="dino.")
    model = model.to(DEV)
    slots = gpu(synth.synth_noise(4, 24, 128, seed=9))                     # 4 x 24 x 256 = 24576 rows
    videos = gpu(synth.synth_videos(1, 64, height=224, width=224, seed=72))  # 64 x 257 = 16448 rows
    names = []

    class Names:
        def wrap(self, name, units, fn):
            names.append(name)
            return fn()

    def run(planes):
        monkeypatch.setattr(D, "_MLP_PLANES", planes)
        monkeypatch.setattr(T, "_VIT_PLANES", planes)
        monkeypatch.setattr(K, "_GEMM_CHUNK", planes)
        dec = model(mode="decode", slots=slots)
        feats = model.encoder(videos)
        return dec["recons_feats"], dec["masks"], dec["recons_imgs"], feats

    ref = run(False)
    K.TIMER = Names()
    try:
        got = run(True)
    finally:
        K.TIMER = None
    assert any(n == "gemm_split22_24576x1024x1024" for n in names), names[:8]     # incl. the padded head
    assert sum(n == "gemm_split22_24576x1024x1024" for n in names) == 3
    assert any(n == "gemm_split22_16448x2304x768" for n in names) and any(n == "gemm_split22_16448x768x3072" for n in names)
    for a, b, what in zip(ref, got, ("recons_feats", "masks", "recons_imgs", "vit tokens")):
        assert torch.equal(a, b), what


@pytest.mark.parametrize("planes", [True, False])
@torch.no_grad()
def test_dinosaur_decode_of_many_frames_in_one_call(planes, monkeypatch):
    """
    The reference's evaluator decodes ALL predicted frames in one call (05_evaluate_predictor.py:88-96).  For configs[3]
    that hands the MLPPatchDecoder more than 2^32 bytes of activations per layer (here 200 frames x 24 slots x 256 patches =
    1 228 800 rows of 1024): the split GEMMs' 32-bit operand offsets wrapped behind the 4 GB line until the end of round 4
    (rows >= 1 048 576 wrong, silently).  One call must equal the same frames decoded eight at a time, bit for bit, with
    plane hand-overs (chunk-resident GEMM) and with fp32 hand-overs (two-operand kernels).
    """
    from textocvp_amd import kernels as K
    from textocvp_amd.setup_model import default_dinosaur_params
    from textocvp_amd.models.EncodersDecoders import decoders as D
    monkeypatch.setattr(D, "_MLP_PLANES", planes)
    model = setup_model(default_dinosaur_params(num_slots=24, img_size=224)).eval()
    synth.fill_module_(model, prefix="dino.")
    model = model.to(DEV)
    model.decoder.reconstruct_images = False                        # the MLP + compositing (the image head is per frame)
    slots = gpu(synth.synth_noise(200, 24, 128, seed=13))
    one = model(mode="decode", slots=slots)
    for lo in (0, 96, 168, 192):                                    # frames before, across and behind the 4 GB line
        part = model(mode="decode", slots=slots[lo:lo + 8].contiguous())
        assert torch.equal(one["recons_feats"][lo:lo + 8], part["recons_feats"]), lo
        assert torch.equal(one["masks"][lo:lo + 8], part["masks"]), lo


@torch.no_grad()
def test_dinosaur_reference_default_resolution_336():
    """
    The reference's shipped ExtendedDINOSAUR config is 336 x 336 (configs/models/ExtendedDINOSAUR.json:2,26: 576 patches, a
    24 x 24 grid, 577 ViT tokens); BASELINE configs[3] and every other test run 224 x 224.  Here: the MLPPatchDecoder + image
    head (24-wide layers: regular conv tiles, phase convolutions, bilinear resize to 336) and the ViT on one frame against
    the oracle, and a small end-to-end evaluation (10 slots, 1 seed + 3 preds) for shapes and finiteness.
    """
    from textocvp_amd.setup_model import default_dinosaur_params
    S, Kk, P, B = 336, 10, 3, 2
    model = setup_model(default_dinosaur_params(num_slots=Kk, img_size=S)).eval()
    exp = default_exp_params(num_slots=Kk, num_context=1, num_preds=P, predictor_name="TextOCVP_T5")
    pred = setup_predictor(exp).eval()
    synth.fill_module_(model, prefix="dino.")
    synth.fill_module_(pred, prefix="pred.")
    sd = {k_: v.clone() for k_, v in model.state_dict().items()}
    model, pred = model.to(DEV), pred.to(DEV)
    videos = gpu(synth.synth_videos(B, 1 + P, height=S, width=S, seed=4))
    ids = torch.randint(1, 32000, (B, 9), generator=torch.Generator().manual_seed(2)).to(DEV)
    mask = torch.ones(B, 9, dtype=torch.int64, device=DEV)
    noise = synth.synth_noise(B, Kk, 128, seed=3)
    out = forward_eval(model, pred, videos, 1, P, caption_tokens=ids, attn_masks=mask, init_noise=gpu(noise))
    assert out["pred_imgs"].shape == (B, P, 3, S, S) and out["masks"].shape == (B * P, Kk, 1, 24, 24)
    assert all(bool(torch.isfinite(v).all()) for v in out.values() if torch.is_tensor(v) and v.numel())
    slots = out["pred_slots"][0, :1].contiguous()
    dec = model(mode="decode", slots=slots)
    ref_imgs, ref_feats, _ = O.mlp_patch_decoder(O.sub(sd, "decoder."), slots.cpu(), img_size=S)
    assert max_abs(dec["recons_feats"].cpu(), ref_feats) < 2e-4 and max_abs(dec["recons_imgs"].cpu(), ref_imgs) < 5e-6
    ref_tok = O.vit_encoder(O.sub(sd, "encoder.vit_backbone."), videos[0, :1].cpu())
    got_tok = model.encoder(videos[0, :1])
    assert got_tok.shape == (1, 576, 768)
    assert max_abs(got_tok.cpu(), ref_tok) < 2e-5 * max(1.0, float(ref_tok.abs().max()))
