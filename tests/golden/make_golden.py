"""
Golden-vector generator: runs the REFERENCE modules (imported from /root/reference, CPU, fp32)
on deterministic synthetic weights/inputs and stores small input/output fixtures as .npz.

Only runs in the build container (the reference never travels to the GPU box).  Recipe for the
import follows SURVEY.md section 8(c): two inert stubs for packages the image lacks (timm, nltk),
modules constructed directly from the reference's JSON configs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The fixtures hold DATA only (outputs, and the few inputs that are not regenerated from
textocvp_amd.synth): no reference source text.
"""

import copy
import json
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("TOCVP_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from textocvp_amd import synth  # noqa: E402


def import_reference():
    """ Make `models.*` of the reference importable (stubs for timm / nltk only). """
    sys.path.insert(0, os.path.join(REF, "src"))
    from transformers import T5EncoderModel  # noqa: F401  (must precede the timm stub)

    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:  # placeholder for timm's VisionTransformer symbols
        pass

    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", PatchEmbed=_Dummy, trunc_normal_=lambda *a, **k: None)
    _stub("timm.models.resnet", ResNet=_Dummy, Bottleneck=_Dummy, BasicBlock=_Dummy)
    _stub("timm.models.vision_transformer", VisionTransformer=_Dummy,
          _create_vision_transformer=lambda *a, **k: None)
    _stub("nltk", download=lambda *a, **k: True, word_tokenize=lambda s: s.split())

    from models.SAVi import SAVi
    from models.Predictors.text_cond_OCVP import TextOCVP_CustomTF
    from models.Predictors.predictor_wrapper import PredictorWrapper
    return SAVi, TextOCVP_CustomTF, PredictorWrapper


def load_cfg(rel):
    with open(os.path.join(REF, "src", "configs", rel)) as f:
        return json.load(f)


def build_reference(num_slots, num_context, num_preds, buffer_size=10, seed=0, savi_family="damped"):
    """ SAVi + PredictorWrapper(TextOCVP_CustomTF) with synthetic weights, eval mode. """
    SAVi, TextOCVP_CustomTF, PredictorWrapper = import_reference()
    savi_cfg = load_cfg("models/SAVi.json")
    savi_cfg["num_slots"] = num_slots
    pred_cfg = load_cfg("predictors/TextOCVP_CustomTF.json")
    savi = SAVi(**copy.deepcopy(savi_cfg)).eval()
    pp = copy.deepcopy(pred_cfg["predictor_params"])
    pp["predictor_params"]["input_buffer_size"] = buffer_size
    core = TextOCVP_CustomTF(
        slot_dim=savi_cfg["slot_dim"],
        predictor_params=pp["predictor_params"],
        fusion_params=pp["fusion_params"],
        text_encoder_params=pp["text_encoder_params"],
    )
    exp_params = {
        "model": {"model_name": "SAVi", "model_params": copy.deepcopy(savi_cfg)},
        "predictor": copy.deepcopy(pred_cfg),
        "prediction_params": {"num_context": num_context, "num_preds": num_preds,
                              "teacher_force": False, "input_buffer_size": buffer_size},
    }
    wrapper = PredictorWrapper(exp_params=exp_params, predictor=core).eval()
    synth.fill_module_(savi, seed=seed, prefix="savi.", family=savi_family)
    synth.fill_module_(wrapper, seed=seed, prefix="pred.")
    # the learned temporal PE is a plain attribute holding a Parameter (not in the state_dict
    # of every torch version): fill it explicitly so both sides agree.
    pe = wrapper.predictor.pe.pe
    with torch.no_grad():
        pe.copy_(synth.synth_tensor("pred.predictor.pe.pe", pe.shape, "normal",
                                    pe.shape[-1] ** -0.5, seed))
    return savi, wrapper


class FixedNoise:
    """ Patches torch.randn so the reference's LearnedRandom draws OUR noise tensor. """

    def __init__(self, noise):
        self.noise = noise
        self._p = None

    def __enter__(self):
        noise = self.noise

        def _randn(*size, **kw):
            shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else size
            assert tuple(shape) == tuple(noise.shape), (shape, noise.shape)
            return noise.clone()
        self._p = mock.patch("torch.randn", _randn)
        self._p.__enter__()
        return self

    def __exit__(self, *a):
        self._p.__exit__(*a)


def sub(x, step=16):
    """ sub-sample the last axis-but-one heavy tensors to keep fixtures small """
    return x[..., ::step, :]


@torch.no_grad()
def unit_fixtures(out_dir):
    """ Per-unit goldens at K=7 (SURVEY.md section 4 list). """
    savi, wrapper = build_reference(num_slots=7, num_context=1, num_preds=4)
    core = wrapper.predictor
    B, K, D = 2, 7, 128
    fx = {}

    # --- a3 encoder ------------------------------------------------------------------------
    imgs = synth.synth_tensor("unit.imgs", (B, 3, 64, 64), "unit")
    feats = savi.encode(imgs)                                         # (B, 4096, 128)
    fx["encoder_feats_sub16"] = feats[:, ::16].numpy()

    # --- a4 slot attention: frame step with 3 iterations (t=0) and 1 iteration (t>0) --------
    sa_in = synth.synth_tensor("unit.sa_feats", (B, 4096, D), "normal")
    slots0 = synth.synth_tensor("unit.sa_slots", (B, K, D), "normal")
    fx["sa_step0"] = savi.slot_attention(sa_in, slots0, step=0).numpy()
    fx["sa_step0_attn_sub16"] = savi.slot_attention.attention_masks[:, :, ::16].numpy()
    fx["sa_step1"] = savi.slot_attention(sa_in, slots0, step=1).numpy()

    # --- a5 transition ----------------------------------------------------------------------
    fx["transition"] = savi.transition_module(slots0).numpy()

    # --- a7 text encoder with ragged lengths {5, 12, 12 padded to 20} ------------------------
    tokens, lengths = synth.synth_captions(3, max_len=20, lengths=[5, 12, 12], seed=3)
    fx["text_tokens"] = tokens.numpy()
    fx["text_lengths"] = lengths.numpy()
    text_emb = core.text_encoder(text=tokens, text_length=lengths)    # (3, 20, 512)
    fx["text_emb"] = text_emb.numpy()

    # --- a9 one AdaptedEncoderBlock ----------------------------------------------------------
    x = synth.synth_tensor("unit.block_x", (3, 2 * K, 512), "normal")
    fx["block0"] = core.predictor[0](x, text_emb).numpy()

    # --- a8 one predictor step at window 1 and window 10 -------------------------------------
    win1 = synth.synth_tensor("unit.win1", (3, 1, K, D), "normal")
    win10 = synth.synth_tensor("unit.win10", (3, 10, K, D), "normal")
    fx["pred_step_w1"] = core(slots=win1, text_embeddings=text_emb).numpy()
    fx["pred_step_w10"] = core(slots=win10, text_embeddings=text_emb).numpy()

    # --- a10 decoder at K=7 -------------------------------------------------------------------
    dslots = synth.synth_tensor("unit.dec_slots", (2, K, D), "normal")
    out = savi(mode="decode", slots=dslots)
    fx["dec7_recons_imgs"] = out["recons_imgs"].numpy()
    fx["dec7_recons_sub4"] = out["recons"][..., ::4, ::4].numpy()
    fx["dec7_masks_sub4"] = out["masks"][..., ::4, ::4].numpy()
    np.savez(os.path.join(out_dir, "units_k7.npz"), **fx)
    print("units_k7:", {k: v.shape for k, v in fx.items()})

    # --- a10 decoder at K=30 ------------------------------------------------------------------
    savi30, _ = build_reference(num_slots=30, num_context=1, num_preds=19)
    dslots = synth.synth_tensor("unit.dec_slots30", (2, 30, D), "normal")
    out = savi30(mode="decode", slots=dslots)
    fx30 = {
        "dec30_recons_imgs": out["recons_imgs"].numpy(),
        "dec30_recons_sub8": out["recons"][..., ::8, ::8].numpy(),
        "dec30_masks_sub8": out["masks"][..., ::8, ::8].numpy(),
    }
    np.savez(os.path.join(out_dir, "units_k30.npz"), **fx30)
    print("units_k30:", {k: v.shape for k, v in fx30.items()})


@torch.no_grad()
def forward_eval(savi, wrapper, videos, tokens, lengths, noise, num_context, num_preds):
    """ The three calls of the reference evaluator's forward_eval, on CPU. """
    B, L, C, H, W = videos.shape
    with FixedNoise(noise):
        out_model = savi(mode="decomp", x=videos, num_imgs=num_context + num_preds, decode=False,
                         caption_tokens=tokens, caption_lengths=lengths)
    slot_history = out_model["slot_history"]
    pred_slots = wrapper(slot_history, caption_tokens=tokens, caption_lengths=lengths)
    K, D = savi.num_slots, savi.slot_dim
    out_dec = savi(mode="decode", slots=pred_slots.reshape(B * num_preds, K, D))
    pred_imgs = out_dec["recons_imgs"].view(B, num_preds, C, H, W).clamp(0, 1)
    return slot_history, pred_slots, pred_imgs, out_dec


@torch.no_grad()
def e2e_fixtures(out_dir):
    # C1: K=7, B=2, 1 seed + 4 preds, ragged captions (lengths 9 and 12, padded to 12)
    savi, wrapper = build_reference(num_slots=7, num_context=1, num_preds=4)
    videos = synth.synth_videos(2, 5, seed=0)
    tokens, lengths = synth.synth_captions(2, max_len=12, lengths=[9, 12], seed=0)
    noise = synth.synth_noise(2, 7, 128, seed=1)
    sh, ps, pi, od = forward_eval(savi, wrapper, videos, tokens, lengths, noise, 1, 4)
    np.savez(os.path.join(out_dir, "e2e_c1.npz"), slot_history=sh.numpy(), pred_slots=ps.numpy(),
             pred_imgs=pi.numpy(), tokens=tokens.numpy(), lengths=lengths.numpy(),
             masks_argmax=od["masks"].argmax(dim=1).to(torch.uint8).numpy())
    print("e2e_c1:", sh.shape, ps.shape, pi.shape)

    # C2: K=30, B=1, 1 seed + 19 preds
    savi, wrapper = build_reference(num_slots=30, num_context=1, num_preds=19)
    videos = synth.synth_videos(1, 20, seed=0)
    tokens, lengths = synth.synth_captions(1, max_len=12, seed=0)
    noise = synth.synth_noise(1, 30, 128, seed=1)
    sh, ps, pi, od = forward_eval(savi, wrapper, videos, tokens, lengths, noise, 1, 19)
    np.savez(os.path.join(out_dir, "e2e_c2.npz"), slot_history=sh.numpy(), pred_slots=ps.numpy(),
             pred_imgs_sub2=pi[..., ::2, ::2].numpy(), tokens=tokens.numpy(),
             lengths=lengths.numpy(),
             masks_argmax_sub2=od["masks"].argmax(dim=1)[..., ::2, ::2].to(torch.uint8).numpy())
    print("e2e_c2:", sh.shape, ps.shape, pi.shape)


@torch.no_grad()
def parity_fixtures(out_dir):
    """
    Second and third weight family: the decoder arithmetic is qualified on
    weights that do NOT attenuate pixel errors -- "undamped" (O(1) RGB head) and "xavier" (the
    distribution of the reference's own _init_model) -- with FULL-RESOLUTION recons / masks / recons_imgs.
    """
    D = 128
    fx = {}
    for fam in ("undamped", "xavier"):
        savi7, wrapper7 = build_reference(num_slots=7, num_context=1, num_preds=4, savi_family=fam)
        dslots = synth.synth_tensor("unit.dec_slots", (2, 7, D), "normal")
        out = savi7(mode="decode", slots=dslots)
        fx[f"{fam}_dec7_recons_imgs"] = out["recons_imgs"].numpy()
        fx[f"{fam}_dec7_recons"] = out["recons"].numpy()
        fx[f"{fam}_dec7_masks"] = out["masks"].numpy()
        # e2e config 1 (K=7, B=2, 1 seed + 4 preds, ragged captions) with this SAVi family
        videos = synth.synth_videos(2, 5, seed=0)
        tokens, lengths = synth.synth_captions(2, max_len=12, lengths=[9, 12], seed=0)
        noise = synth.synth_noise(2, 7, 128, seed=1)
        sh, ps, pi, od = forward_eval(savi7, wrapper7, videos, tokens, lengths, noise, 1, 4)
        fx[f"{fam}_c1_slot_history"] = sh.numpy()
        fx[f"{fam}_c1_pred_slots"] = ps.numpy()
        fx[f"{fam}_c1_recons_imgs"] = od["recons_imgs"].numpy()            # unclamped
        fx[f"{fam}_c1_masks_s0"] = od["masks"][:4].numpy()                 # sample 0, 4 frames, full res
        fx[f"{fam}_c1_recons_s0f3"] = od["recons"][3].numpy()
        print(fam, "K=7 ranges: recons", float(out["recons"].min()), float(out["recons"].max()),
              "masks max", float(out["masks"].max()), "slots |max|", float(ps.abs().max()))
    np.savez(os.path.join(out_dir, "parity_k7.npz"), **fx)
    print("parity_k7:", {k: v.shape for k, v in fx.items()})

    fx = {}
    savi30, wrapper30 = build_reference(num_slots=30, num_context=1, num_preds=19, savi_family="undamped")
    dslots = synth.synth_tensor("unit.dec_slots30", (1, 30, D), "normal")
    out = savi30(mode="decode", slots=dslots)
    fx["undamped_dec30_recons_imgs"] = out["recons_imgs"].numpy()
    fx["undamped_dec30_recons"] = out["recons"].numpy()
    fx["undamped_dec30_masks"] = out["masks"].numpy()
    # e2e config 2 (north star: K=30, B=1, 1 seed + 19 preds): rendered frames at full resolution,
    # per-slot outputs of the LAST predicted frame at full resolution
    videos = synth.synth_videos(1, 20, seed=0)
    tokens, lengths = synth.synth_captions(1, max_len=12, seed=0)
    noise = synth.synth_noise(1, 30, 128, seed=1)
    sh, ps, pi, od = forward_eval(savi30, wrapper30, videos, tokens, lengths, noise, 1, 19)
    fx["undamped_c2_recons_imgs"] = od["recons_imgs"].numpy()
    fx["undamped_c2_masks_f18"] = od["masks"][18].numpy()
    fx["undamped_c2_recons_f18"] = od["recons"][18].numpy()
    fx["undamped_c2_masks_argmax"] = od["masks"].argmax(dim=1).to(torch.uint8).numpy()
    np.savez(os.path.join(out_dir, "parity_k30.npz"), **fx)
    print("parity_k30:", {k: v.shape for k, v in fx.items()})


@torch.no_grad()
def long_caption_fixtures(out_dir):
    """
    Captions longer than 16 tokens (the text encoder admits 50, text_encoders.py:36, and pads to the longest caption of
    the batch, :171-175; padded positions take part in the cross-attention, attention.py:303-319): e2e config-1 shapes
    (K=7, B=2, 1 seed + 4 preds) on the "undamped" SAVi family, two caption batches -- lengths (24, 40) padded to 40,
    and (50, 17) padded to the 50-token maximum.  Stored per batch: tokens, lengths, pred_slots, the rendered frames
    (unclamped, full resolution), the masks of sample 0 and the argmax_K(masks) map of every frame.
    """
    savi, wrapper = build_reference(num_slots=7, num_context=1, num_preds=4, savi_family="undamped")
    videos = synth.synth_videos(2, 5, seed=0)
    noise = synth.synth_noise(2, 7, 128, seed=1)
    fx = {}
    for tag, lens in (("l40", [24, 40]), ("l50", [50, 17])):
        tokens, lengths = synth.synth_captions(2, max_len=max(lens), lengths=lens, seed=7)
        sh, ps, pi, od = forward_eval(savi, wrapper, videos, tokens, lengths, noise, 1, 4)
        fx[f"{tag}_tokens"], fx[f"{tag}_lengths"] = tokens.numpy(), lengths.numpy()
        fx[f"{tag}_pred_slots"] = ps.numpy()
        fx[f"{tag}_recons_imgs"] = od["recons_imgs"].numpy()
        fx[f"{tag}_masks_s0"] = od["masks"][:4].numpy()
        fx[f"{tag}_masks_argmax"] = od["masks"].argmax(dim=1).to(torch.uint8).numpy()
        if tag == "l40":
            fx["slot_history"] = sh.numpy()
    np.savez(os.path.join(out_dir, "long_captions_k7.npz"), **fx)
    print("long_captions_k7:", {k: v.shape for k, v in fx.items()})


@torch.no_grad()
def decomp_fixtures(out_dir):
    """
    Decomposition-only evaluation (03_evaluate_decomp_model.py:22-46): ``model(x=videos, num_imgs=L)`` with
    the default ``mode`` / ``decode=True`` on config-1 shapes (K=7, B=2, L=5), "undamped" SAVi family (O(1)
    RGB head: nothing attenuates pixel errors).  Stored: the rendered frames (B, L, 3, H, W) unclamped, the
    masks of sample 0 (all frames, full resolution), the per-slot reconstructions of sample 1's last frame,
    the slot history and the argmax_K(masks) map of every frame.
    """
    savi, _ = build_reference(num_slots=7, num_context=1, num_preds=4, savi_family="undamped")
    videos = synth.synth_videos(2, 5, seed=0)
    noise = synth.synth_noise(2, 7, 128, seed=1)
    with FixedNoise(noise):
        out = savi(x=videos, num_imgs=videos.shape[1], caption=["a", "b"])      # kwargs as unwrap_batch_data yields
    fx = {"recons_imgs": out["recons_imgs"].numpy(), "masks_s0": out["masks"][0].numpy(),
          "recons_objs_s1f4": out["recons_objs"][1, 4].numpy(), "slot_history": out["slot_history"].numpy(),
          "masks_argmax": out["masks"].argmax(dim=2).to(torch.uint8).numpy()}
    np.savez(os.path.join(out_dir, "decomp_c1.npz"), **fx)
    print("decomp_c1:", {k: v.shape for k, v in fx.items()},
          "recons range", float(out["recons_imgs"].min()), float(out["recons_imgs"].max()))


@torch.no_grad()
def uncond_fixtures(out_dir):
    """ VanillaTransformer / OCVPSeq (models/Predictors/OCVP.py): one step + a 4-step rollout, K=7 """
    import_reference()
    from models.Predictors.OCVP import OCVPSeq, VanillaTransformerPredictor
    from models.Predictors.predictor_wrapper import PredictorWrapper
    K, D, buf = 7, 128, 10
    fx, man = {}, {}
    for name, cls in (("VanillaTransformer", VanillaTransformerPredictor), ("OCVPSeq", OCVPSeq)):
        cfg = load_cfg(f"predictors/{name}.json")
        core = cls(num_slots=K, slot_dim=D, input_buffer_size=buf, **cfg["predictor_params"])
        exp_params = {"predictor": cfg,
                      "prediction_params": {"num_context": 2, "num_preds": 4, "teacher_force": False,
                                            "input_buffer_size": buf}}
        wrapper = PredictorWrapper(exp_params=exp_params, predictor=core).eval()
        synth.fill_module_(wrapper, seed=0, prefix=f"{name}.")
        man[name] = {k: list(v.shape) for k, v in wrapper.state_dict().items()}
        win = synth.synth_tensor("unit.win3", (2, 3, K, D), "normal")
        fx[f"{name}_step_w3"] = core(slots=win).numpy()
        hist = synth.synth_tensor("unit.hist6", (2, 6, K, D), "normal")
        tokens, lengths = synth.synth_captions(2, max_len=8, seed=9)
        fx[f"{name}_rollout"] = wrapper(hist, caption_tokens=tokens, caption_lengths=lengths).numpy()
    np.savez(os.path.join(out_dir, "uncond_k7.npz"), **fx)
    with open(os.path.join(out_dir, "state_dict_manifest_uncond.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("uncond_k7:", {k: v.shape for k, v in fx.items()})


@torch.no_grad()
def dinosaur_fixtures(out_dir):
    """
    ExtendedDINOSAUR decode side (BASELINE config 4: 24 slots, 224x224 -> 256 patches): the
    reference's MLPPatchDecoder incl. the CNN image head (decoders.py:203-365), B = 1.
    The ViT backbone (timm) is not importable here; everything upstream of the decoder reuses
    units already pinned (slot attention, transition).
    """
    import_reference()
    from models.EncodersDecoders.decoders import get_decoder
    cfg = load_cfg("models/ExtendedDINOSAUR.json")
    dp = copy.deepcopy(cfg["decoder"])
    dp["decoder_params"]["num_patches"] = 256
    dp["decoder_params"]["img_size"] = 224
    dec = get_decoder(in_channels=3, decoder=dp).eval()
    synth.fill_module_(dec, seed=0, prefix="dino.decoder.")
    slots = synth.synth_tensor("unit.dino_slots", (1, 24, 128), "normal")
    out = dec(slots)
    fx = {"recons_imgs_sub2": out["recons_imgs"][..., ::2, ::2].numpy(),
          "recons_feats_sub4": out["recons_feats"][:, ::4, ::4].numpy(),
          "masks": out["masks"].numpy()}
    np.savez(os.path.join(out_dir, "dinosaur_dec.npz"), **fx)
    man = {k: list(v.shape) for k, v in dec.state_dict().items()}
    with open(os.path.join(out_dir, "state_dict_manifest_dinosaur_decoder.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("dinosaur_dec:", {k: v.shape for k, v in fx.items()},
          {k: (float(v.min()), float(v.max())) for k, v in fx.items()})


@torch.no_grad()
def t5_fixtures(out_dir):
    """
    T5-small encoder as called by the reference for TextOCVP_T5 (predictor_wrapper.py:101-111):
    transformers.T5EncoderModel(input_ids, attention_mask).last_hidden_state.  The reference loads
    hub weights (network); the architecture is instantiated from its published t5-small config and
    filled with synthetic weights, which pins the ARITHMETIC of the third-party encoder.
    """
    from transformers import T5Config, T5EncoderModel
    cfg = T5Config(vocab_size=32128, d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_heads=8,
                   relative_attention_num_buckets=32, relative_attention_max_distance=128,
                   dropout_rate=0.1, layer_norm_epsilon=1e-6, feed_forward_proj="relu")
    m = T5EncoderModel(cfg).eval()
    synth.fill_module_(m, seed=0, prefix="t5.")
    ids = torch.from_numpy(synth._rng("inputs.t5ids", 0).integers(1, 32128, size=(3, 24)))
    lengths = [24, 9, 17]
    mask = torch.zeros(3, 24, dtype=torch.int64)
    for b, n in enumerate(lengths):
        mask[b, :n] = 1
    ids = ids * mask                                             # pad id 0
    out = m(input_ids=ids, attention_mask=mask, return_dict=True).last_hidden_state
    np.savez(os.path.join(out_dir, "t5_encoder.npz"), ids=ids.numpy(), mask=mask.numpy(),
             last_hidden_state=out.numpy())
    man = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(out_dir, "state_dict_manifest_t5.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("t5_encoder:", out.shape, float(out.abs().max()))


def train_fixtures(out_dir):
    """
    Predictor training step of the reference (04_train_predictor.py:57-108 without the optimiser):
    frozen SAVi decomp (no grad) -> PredictorWrapper rollout -> SAVi.decode -> nn.MSELoss on images +
    nn.MSELoss on slots (lib/loss.py:150-191, weights 1 / 1, CONFIG.py:42-51), loss.backward() through
    torch.autograd, with dropout inactive (eval-mode modules, gradients still flow).  K=7, B=2,
    1 seed + 2 preds.  Stored: the two losses, the L2 norm of every
    parameter gradient, and four gradients in full.
    """
    P = 2
    savi, wrapper = build_reference(num_slots=7, num_context=1, num_preds=P)
    wrapper.eval()      # deterministic: the text encoder's dropout (p = 0.1, text_encoders.py:36,64,107) is off
    for p_ in savi.parameters():
        p_.requires_grad_(False)
    videos = synth.synth_videos(2, 1 + P, seed=0)
    tokens, lengths = synth.synth_captions(2, max_len=12, lengths=[9, 12], seed=0)
    noise = synth.synth_noise(2, 7, 128, seed=1)
    B, L, C, H, W = videos.shape
    with torch.no_grad(), FixedNoise(noise):
        hist = savi(mode="decomp", x=videos, num_imgs=1 + P, decode=False,
                    caption_tokens=tokens, caption_lengths=lengths)["slot_history"]
    pred_slots = wrapper(hist, caption_tokens=tokens, caption_lengths=lengths)
    dec = savi(mode="decode", slots=pred_slots.clone().reshape(B * P, 7, 128))
    pred_imgs = dec["recons_imgs"].view(B, P, C, H, W)
    mse = torch.nn.MSELoss()
    l_img = mse(pred_imgs, videos[:, 1:1 + P])
    l_slot = mse(pred_slots, hist[:, 1:1 + P])
    (l_img + l_slot).backward()
    names, norms, full = [], [], {}
    keep = ("predictor.mlp_out.weight", "predictor.pe.pe", "predictor.predictor.0.attn.q.weight",
            "predictor.text_encoder.position_embedding.weight")
    for name, p_ in wrapper.named_parameters():
        g = torch.zeros_like(p_) if p_.grad is None else p_.grad
        names.append(name)
        norms.append(float(g.norm()))
        if name in keep:                             # large matrices: every 4th row / column
            full["grad::" + name] = (g[::4, ::4] if g.dim() == 2 and g.numel() > 40000 else g).detach().numpy()
    np.savez(os.path.join(out_dir, "train_c5.npz"), loss_img=l_img.item(), loss_slot=l_slot.item(),
             names=np.array(names), grad_norms=np.array(norms, dtype=np.float64),
             pred_slots=pred_slots.detach().numpy(), **full)
    print("train_c5: losses", l_img.item(), l_slot.item(), "params", len(names), "kept", sorted(full))


@torch.no_grad()
def manifest(out_dir):
    """ state_dict key/shape manifest = the checkpoint-layout contract. """
    savi, wrapper = build_reference(num_slots=30, num_context=1, num_preds=19)
    man = {
        "SAVi": {k: list(v.shape) for k, v in savi.state_dict().items()},
        "PredictorWrapper": {k: list(v.shape) for k, v in wrapper.state_dict().items()},
    }
    with open(os.path.join(out_dir, "state_dict_manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("manifest:", len(man["SAVi"]), len(man["PredictorWrapper"]))


if __name__ == "__main__":
    torch.set_num_threads(8)
    what = sys.argv[1:] or ["manifest", "units", "e2e", "parity", "longcap", "decomp", "uncond", "dinosaur", "t5",
                            "train"]
    if "manifest" in what:
        manifest(HERE)
    if "units" in what:
        unit_fixtures(HERE)
    if "e2e" in what:
        e2e_fixtures(HERE)
    if "parity" in what:
        parity_fixtures(HERE)
    if "longcap" in what:
        long_caption_fixtures(HERE)
    if "decomp" in what:
        decomp_fixtures(HERE)
    if "uncond" in what:
        uncond_fixtures(HERE)
    if "dinosaur" in what:
        dinosaur_fixtures(HERE)
    if "t5" in what:
        t5_fixtures(HERE)
    if "train" in what:
        train_fixtures(HERE)
