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


IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)          # timm.data.constants: default_cfg["mean"] of the DINOv2 ViTs
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


class HFDinov2AsTimm(torch.nn.Module):
    """
    transformers.Dinov2Model (an independent third-party implementation of the published DINOv2 ViT) behind the
    attribute names the reference's ViTEncoder calls on a timm VisionTransformer (timm_encoders.py:59-70: patch_embed,
    _pos_embed, patch_drop, norm_pre, blocks, default_cfg).  timm itself is absent from the image; with
    this object the REFERENCE's own wrapper code (std := mean normalisation, class token dropped, final norm skipped)
    runs here, on HF's arithmetic for the backbone.  HF's ``embeddings`` module does the patch projection, the class
    token and the position table in one call, so ``patch_embed`` is the identity and ``_pos_embed`` is that module.
    """

    def __init__(self, img_size=224, patch_size=14, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                 **unused):
        super().__init__()
        from transformers import Dinov2Config, Dinov2Model
        cfg = Dinov2Config(hidden_size=embed_dim, num_hidden_layers=depth, num_attention_heads=num_heads,
                           mlp_ratio=mlp_ratio, image_size=img_size, patch_size=patch_size, qkv_bias=qkv_bias,
                           layer_norm_eps=1e-6, hidden_act="gelu", layerscale_value=1e-5,
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, drop_path_rate=0.0)
        cfg._attn_implementation = "eager"
        self.hf = Dinov2Model(cfg)
        self.default_cfg = {"mean": IMAGENET_DEFAULT_MEAN, "std": IMAGENET_DEFAULT_STD}
        self.patch_embed = torch.nn.Identity()
        self.patch_drop = torch.nn.Identity()
        self.norm_pre = torch.nn.Identity()
        self.blocks = torch.nn.Sequential(*self.hf.encoder.layer)

    def _pos_embed(self, x):
        return self.hf.embeddings(x)


def hf_to_timm_key(k):
    """ transformers.Dinov2Model state_dict name -> (timm VisionTransformer name, row slice index or None) """
    fixed = {"embeddings.cls_token": "cls_token", "embeddings.position_embeddings": "pos_embed",
             "embeddings.patch_embeddings.projection.weight": "patch_embed.proj.weight",
             "embeddings.patch_embeddings.projection.bias": "patch_embed.proj.bias",
             "layernorm.weight": "norm.weight", "layernorm.bias": "norm.bias"}
    if k in fixed:
        return fixed[k], None
    if not k.startswith("encoder.layer."):
        return None, None                                   # embeddings.mask_token: pre-training only
    i, rest = k[len("encoder.layer."):].split(".", 1)
    for j, name in enumerate(("query", "key", "value")):
        if rest.startswith(f"attention.attention.{name}."):
            return f"blocks.{i}.attn.qkv.{rest.rsplit('.', 1)[1]}", j        # fused qkv rows [q | k | v]
    rest = rest.replace("attention.output.dense.", "attn.proj.")
    rest = rest.replace("layer_scale1.lambda1", "ls1.gamma").replace("layer_scale2.lambda1", "ls2.gamma")
    return f"blocks.{i}.{rest}", None


@torch.no_grad()
def fill_hf_dinov2_(hf, prefix, seed=0):
    """
    Fill a transformers.Dinov2Model with the synthetic values of the timm-named parameters
    (``prefix`` + timm name: what synth.fill_module_ gives this repo's ExtendedDINOSAUR / a reference checkpoint layout).
    Returns the HF -> timm key map that was applied.
    """
    applied = {}
    E = hf.config.hidden_size
    for k, t in hf.state_dict().items():
        tk, part = hf_to_timm_key(k)
        if tk is None:
            continue
        shape = tuple(t.shape)
        if part is not None:
            shape = (3 * E,) + shape[1:]
        if tk == "pos_embed":
            shape = tuple(t.shape)
        vals = torch.from_numpy(synth._param_values(prefix + tk, shape, seed, "damped"))
        t.copy_(vals if part is None else vals[part * E:(part + 1) * E])
        applied[k] = tk if part is None else f"{tk}[{part}E:{part + 1}E]"
    return applied


_REFERENCE = {}


def import_reference():
    """ Make `models.*` of the reference importable (stubs for timm / nltk only). """
    if _REFERENCE:
        return _REFERENCE["classes"]
    sys.path.insert(0, os.path.join(REF, "src"))
    from transformers import T5EncoderModel  # noqa: F401  (must precede the timm stub)

    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:  # placeholder for timm symbols nothing here calls
        pass

    # timm.create_model is what the reference's ViT factories call (timm_encoders.py:215-267); the DINOv2 ones get
    # the HF model behind timm's attribute names (dinov2 / e2e_c4 fixtures); everything else stays inert
    def _create_model(name, pretrained=False, **kw):
        assert "dinov2" in name, name
        kw.pop("norm_layer", None), kw.pop("num_classes", None)
        return HFDinov2AsTimm(**kw)

    _stub("timm", create_model=_create_model)
    _stub("timm.models")
    _stub("timm.models.layers", PatchEmbed=_Dummy, trunc_normal_=lambda *a, **k: None)
    _stub("timm.models.resnet", ResNet=_Dummy, Bottleneck=_Dummy, BasicBlock=_Dummy)
    _stub("timm.models.vision_transformer", VisionTransformer=HFDinov2AsTimm,
          _create_vision_transformer=lambda *a, **k: None)
    _stub("nltk", download=lambda *a, **k: True, word_tokenize=lambda s: s.split())

    from models.SAVi import SAVi
    from models.Predictors.text_cond_OCVP import TextOCVP_CustomTF
    from models.Predictors.predictor_wrapper import PredictorWrapper
    _REFERENCE["classes"] = (SAVi, TextOCVP_CustomTF, PredictorWrapper)
    return _REFERENCE["classes"]


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
    cfg = T5Config(**T5_SMALL)
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


T5_SMALL = dict(vocab_size=32128, d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_heads=8,
                relative_attention_num_buckets=32, relative_attention_max_distance=128,
                dropout_rate=0.1, layer_norm_epsilon=1e-6, feed_forward_proj="relu")      # published t5-small config

VIT_PREFIX = "dino.encoder.vit_backbone."


@torch.no_grad()
def dinov2_fixtures(out_dir):
    """
    DINOv2 ViT-B/14 backbone as the reference calls it (timm_encoders.py:59-70, call site ExtendedDINOSAUR.py:187-188):
    ``(img - mean) / mean`` (std := mean, :54-56) -> patch_embed -> class token + positions -> 12 blocks -> drop the class
    token, NO final norm.  timm is absent, so the arithmetic of the backbone comes from transformers.Dinov2Model, an
    independent implementation of the same published model: ``hidden_states[-1][:, 1:]`` of the normalised frames.  The
    same frames also go through the REFERENCE's ViTEncoder wrapper with that model behind timm's attribute names
    (HFDinov2AsTimm); both must agree bit for bit.  Weights: the timm-named synthetic values (prefix
    'dino.encoder.vit_backbone.'), mapped to the HF names by hf_to_timm_key (map stored next to the fixture).
    Stored: features at 224 (4 frames, every 4th token), the token stream after block 0 (every 16th token), features at
    the reference's default resolution 336 (1 frame, every 8th token).
    """
    import_reference()
    from models.EncodersDecoders.timm_encoders import ViTEncoder
    mean = torch.tensor(IMAGENET_DEFAULT_MEAN).view(1, 3, 1, 1)
    fx, key_map = {}, None
    for size, frames, step in ((224, synth.synth_videos(2, 2, height=224, width=224, seed=71).flatten(0, 1), 4),
                               (336, synth.synth_videos(1, 1, height=336, width=336, seed=72).flatten(0, 1), 8)):
        backbone = HFDinov2AsTimm(img_size=size).eval()
        key_map = fill_hf_dinov2_(backbone.hf, VIT_PREFIX)
        out = backbone.hf(pixel_values=(frames - mean) / mean, output_hidden_states=True)
        feats = out.hidden_states[-1][:, 1:]                              # before the final norm, class token dropped
        wrapped = ViTEncoder(vit_backbone=backbone).eval()(frames)        # the reference's own forward
        assert torch.equal(wrapped, feats), float((wrapped - feats).abs().max())
        fx[f"feats{size}_sub{step}"] = feats[:, ::step].numpy()
        if size == 224:
            fx["after_block0_224_sub16"] = out.hidden_states[1][:, ::16].numpy()
        print(f"dinov2 {size}: feats {tuple(feats.shape)} scale {float(feats.abs().max()):.3g}")
    np.savez(os.path.join(out_dir, "dinov2_vit.npz"), **fx)
    with open(os.path.join(out_dir, "dinov2_key_map.json"), "w") as f:
        json.dump(key_map, f, indent=0, sort_keys=True)
    print("dinov2_vit:", {k: v.shape for k, v in fx.items()}, "keys mapped", len(key_map))


def build_reference_c4(num_slots, num_context, num_preds, img_size=224, buffer_size=10, seed=0, family="undamped"):
    """
    The reference's ExtendedDINOSAUR (configs/models/ExtendedDINOSAUR.json at ``img_size``) and
    PredictorWrapper(TextOCVP_T5) (configs/predictors/TextOCVP_T5.json), constructed the way lib/setup_model.py:21-53,
    57-132 does, with synthetic weights.  The two third-party encoders the reference downloads are instantiated from
    their published configs instead: timm's DINOv2 ViT-B/14 as transformers.Dinov2Model behind timm's attribute
    names (HFDinov2AsTimm), ``T5EncoderModel.from_pretrained("t5-small")`` (text_cond_OCVP.py:148) as
    ``T5EncoderModel(T5Config(t5-small))``.
    """
    import_reference()
    from transformers import T5Config, T5EncoderModel
    from models.ExtendedDINOSAUR import ExtendedDINOSAUR
    from models.Predictors.text_cond_OCVP import TextOCVP_T5
    from models.Predictors.predictor_wrapper import PredictorWrapper
    cfg = load_cfg("models/ExtendedDINOSAUR.json")
    cfg["img_size"], cfg["num_slots"] = img_size, num_slots
    cfg["decoder"]["decoder_params"]["num_patches"] = (img_size // 14) ** 2
    model = ExtendedDINOSAUR(**copy.deepcopy(cfg)).eval()
    synth.fill_module_(model, seed=seed, prefix="dino.", family=family)     # "undamped": sharp alpha masks, RGB bias 0.5
    fill_hf_dinov2_(model.encoder.vit_backbone.hf, VIT_PREFIX, seed=seed)
    pred_cfg = load_cfg("predictors/TextOCVP_T5.json")
    pp = copy.deepcopy(pred_cfg["predictor_params"])
    pp["predictor_params"]["input_buffer_size"] = buffer_size
    with mock.patch.object(T5EncoderModel, "from_pretrained",
                           classmethod(lambda cls, *a, **k: T5EncoderModel(T5Config(**T5_SMALL)))):
        core = TextOCVP_T5(slot_dim=cfg["slot_dim"], predictor_params=pp["predictor_params"],
                           fusion_params=pp["fusion_params"], text_encoder_params=pp["text_encoder_params"])
    exp_params = {
        "model": {"model_name": "ExtendedDINOSAUR", "model_params": copy.deepcopy(cfg)},
        "predictor": copy.deepcopy(pred_cfg),
        "prediction_params": {"num_context": num_context, "num_preds": num_preds,
                              "teacher_force": False, "input_buffer_size": buffer_size},
    }
    wrapper = PredictorWrapper(exp_params=exp_params, predictor=core).eval()
    synth.fill_module_(wrapper, seed=seed, prefix="pred.")
    return model, wrapper


def c4_inputs(B, L, img_size, seed):
    """ frames, T5 token ids / attention masks (16 tokens, sample 1 padded from 11) and init noise of the c4 fixtures """
    videos = synth.synth_videos(B, L, height=img_size, width=img_size, seed=seed)
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, 32000, (B, 16), generator=g)
    mask = torch.ones(B, 16, dtype=torch.int64)
    if B > 1:
        mask[1, 11:] = 0
    return videos, ids * mask, mask


@torch.no_grad()
def e2e_c4_fixtures(out_dir):
    """
    BASELINE configs[3] END TO END through the reference's own glue: the three calls of Evaluator.forward_eval
    (05_evaluate_predictor.py:82-96) on ExtendedDINOSAUR (24 slots, 224 x 224, forward_decomp ExtendedDINOSAUR.py:139-208,
    decode :211-214 -> MLPPatchDecoder decoders.py:264-365) and PredictorWrapper(TextOCVP_T5), from PIXELS.
      ``c4``      the workload itself: B = 1, 1 seed + 29 preds;
      ``c4s``     B = 2, 1 seed + 3 preds, ragged T5 masks (16 / 11 tokens): small enough for the CPU oracle test.
    Stored per tag: ids, mask, slot_history, pred_slots, the predicted frames (clamped; every 4th pixel at the full
    workload), the alpha masks (all for c4s; frames 0 / 14 / 28 for c4) and the argmax_K(masks) map of every frame.
    """
    fx = {}
    for tag, B, P, seed in (("c4s", 2, 3, 83), ("c4", 1, 29, 81)):
        model, wrapper = build_reference_c4(num_slots=24, num_context=1, num_preds=P)
        videos, ids, mask = c4_inputs(B, 1 + P, 224, seed)
        noise = synth.synth_noise(B, 24, 128, seed=seed + 1)
        with FixedNoise(noise):
            out_model = model(mode="decomp", x=videos, num_imgs=1 + P, decode=False, caption_tokens=ids, attn_masks=mask)
        hist = out_model["slot_history"]
        preds = wrapper(hist, caption_tokens=ids, attn_masks=mask)
        out_dec = model(mode="decode", slots=preds.reshape(B * P, 24, 128))
        raw = out_dec["recons_imgs"].view(B, P, 3, 224, 224)
        imgs = raw.clamp(0, 1)
        masks = out_dec["masks"]                                            # (B * P, 24, 1, 16, 16)
        fx[f"{tag}_ids"], fx[f"{tag}_mask"] = ids.numpy(), mask.numpy()
        fx[f"{tag}_slot_history"], fx[f"{tag}_pred_slots"] = hist.numpy(), preds.numpy()
        fx[f"{tag}_feats_f0_sub8"] = out_model["encoded_img_feats"][:, 0, ::8].numpy()
        if tag == "c4s":
            fx[f"{tag}_pred_imgs_sub2"] = imgs[..., ::2, ::2].numpy()
            fx[f"{tag}_masks"] = masks.numpy()
            fx[f"{tag}_recons_feats_sub8"] = out_dec["recons_feats"][:, ::8, ::8].numpy()
        else:
            fx[f"{tag}_pred_imgs_sub4"] = imgs[..., ::4, ::4].numpy()
            fx[f"{tag}_masks_f0_14_28"] = masks[[0, 14, 28]].numpy()
        fx[f"{tag}_masks_argmax"] = masks.argmax(dim=1).to(torch.uint8).numpy()
        top2 = masks.topk(2, dim=1).values
        print(f"e2e_{tag}: hist {tuple(hist.shape)} preds {tuple(preds.shape)} imgs [{float(raw.min()):.3f}, "
              f"{float(raw.max()):.3f}] clamped share {float(((raw < 0) | (raw > 1)).float().mean()):.3f} masks max {float(masks.max()):.3f} slots |max| {float(preds.abs().max()):.3g} "
              f"min top-2 mask margin {float((top2[:, 0] - top2[:, 1]).min()):.3g}", flush=True)
    np.savez(os.path.join(out_dir, "e2e_c4.npz"), **fx)
    # checkpoint-layout contract of the configs[3] pair: the reference modules' own state_dict names and shapes; the
    # backbone's entries are listed under timm's names (what a reference checkpoint holds), derived from the HF
    # model through hf_to_timm_key
    dsd, E = model.state_dict(), model.encoder.vit_backbone.hf.config.hidden_size
    man_d = {k: list(v.shape) for k, v in dsd.items() if not k.startswith("encoder.vit_backbone.")}
    for k, v in model.encoder.vit_backbone.hf.state_dict().items():
        tk, part = hf_to_timm_key(k)
        if tk is not None:
            man_d["encoder.vit_backbone." + tk] = ([3 * E] if part is not None else [v.shape[0]]) + list(v.shape[1:])
    man = {"ExtendedDINOSAUR": man_d, "PredictorWrapper_T5": {k: list(v.shape) for k, v in wrapper.state_dict().items()}}
    with open(os.path.join(out_dir, "state_dict_manifest_c4.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("manifest c4:", len(man["ExtendedDINOSAUR"]), len(man["PredictorWrapper_T5"]))
    print("e2e_c4:", {k: v.shape for k, v in fx.items()})


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
                            "train", "dinov2", "e2e_c4"]
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
    if "dinov2" in what:
        dinov2_fixtures(HERE)
    if "e2e_c4" in what:
        e2e_c4_fixtures(HERE)
