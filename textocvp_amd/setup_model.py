"""
Factories and checkpoint loading with the reference's contract (lib/setup_model.py:21-53
setup_model, :57-132 setup_predictor, :189-240 load_checkpoint), minus the optimiser/scheduler
parts (training is a later SURVEY 8f row).
"""

import copy
import os

import torch

from .models.SAVi import SAVi
from .models.ExtendedDINOSAUR import ExtendedDINOSAUR
from .models.Predictors.predictor_wrapper import PredictorWrapper
from .models.Predictors.text_cond_OCVP import TextOCVP_CustomTF, TextOCVP_T5
from .models.Predictors.OCVP import OCVPSeq, VanillaTransformerPredictor

__all__ = ["setup_model", "setup_predictor", "load_checkpoint", "default_exp_params",
           "default_dinosaur_params", "calibrate_precision"]

# configs/models/SAVi.json + configs/predictors/TextOCVP_CustomTF.json + CONFIG.py:66-71 defaults
_SAVI_DEFAULT = {
    "num_slots": 8, "slot_dim": 128, "num_iterations_first": 3, "num_iterations": 1,
    "in_channels": 3, "mlp_hidden": 256, "mlp_encoder_dim": 128, "initializer": "LearnedRandom",
    "transition_module": {"model_name": "TransformerBlock", "num_heads": 4, "mlp_size": 512},
    "encoder": {"encoder_name": "ConvEncoder", "encoder_params": {
        "num_channels": [32, 32, 32, 32], "kernel_size": 5, "resolution": [64, 64],
        "downsample_encoder": False, "downsample": 2}},
    "decoder": {"decoder_name": "ConvDecoder", "decoder_params": {
        "num_channels": [64, 64, 64, 64], "kernel_size": 5, "resolution": [64, 64],
        "downsample_decoder": False, "upsample": 1}},
}
# configs/models/ExtendedDINOSAUR.json
_DINOSAUR_DEFAULT = {
    "img_size": 336, "in_channels": 3, "num_slots": 10, "slot_dim": 128, "num_iterations_first": 3,
    "num_iterations": 1, "mlp_hidden": 512, "mlp_encoder_dim": 768, "initializer": "LearnedRandom",
    "transition_module": {"model_name": "TransformerBlock", "num_heads": 4, "mlp_size": 512},
    "encoder": {"encoder_name": "vit_base_patch14_dinov2", "encoder_params": {"encoder_num_blocks": 12}},
    "decoder": {"decoder_name": "MLPPatchDecoder", "decoder_params": {
        "patch_size": 14, "num_patches": 576, "in_dim": 128, "hidden_dim": 1024, "out_dim": 769,
        "num_layers": 4, "initial_layer_norm": True, "reconstruct_images": True, "num_layers_cnn": 4}},
}


def default_dinosaur_params(num_slots=10, img_size=336):
    """ ``exp_params["model"]`` for ExtendedDINOSAUR (num_patches follows img_size / 14) """
    p = copy.deepcopy(_DINOSAUR_DEFAULT)
    p["num_slots"], p["img_size"] = num_slots, img_size
    p["decoder"]["decoder_params"]["num_patches"] = (img_size // 14) ** 2
    return {"model_name": "ExtendedDINOSAUR", "model_params": p}


_PREDICTOR_DEFAULT = {
    "predictor_name": "TextOCVP_CustomTF",
    "predictor_params": {
        "predictor_params": {"token_dim": 512, "n_heads": 8, "hidden_dim": 2048, "num_layers": 8,
                             "residual": True},
        "fusion_params": {"num_heads": 8, "head_dim": 64, "mlp_size": 2048},
        "text_encoder_params": {"input_dim": 128, "num_layers": 2, "num_heads": 4, "vocab_size": 50},
    },
}


# configs/predictors/VanillaTransformer.json, OCVPSeq.json
_UNCOND_DEFAULT = {"token_dim": 128, "hidden_dim": 256, "num_layers": 2, "n_heads": 4, "residual": True}


def default_exp_params(num_slots=8, num_context=1, num_preds=9, input_buffer_size=10,
                       teacher_force=False, predictor_name="TextOCVP_CustomTF"):
    """ experiment_params.json-shaped dict with the shipped SAVi / predictor configs """
    model = copy.deepcopy(_SAVI_DEFAULT)
    model["num_slots"] = num_slots
    if predictor_name == "TextOCVP_CustomTF":
        predictor = copy.deepcopy(_PREDICTOR_DEFAULT)
    elif predictor_name == "TextOCVP_T5":
        predictor = copy.deepcopy(_PREDICTOR_DEFAULT)
        predictor["predictor_name"] = "TextOCVP_T5"
        predictor["predictor_params"]["text_encoder_params"] = {}
    else:
        predictor = {"predictor_name": predictor_name, "predictor_params": dict(_UNCOND_DEFAULT)}
    return {
        "model": {"model_name": "SAVi", "model_params": model},
        "predictor": predictor,
        "prediction_params": {"num_context": num_context, "num_preds": num_preds,
                              "teacher_force": teacher_force,
                              "input_buffer_size": input_buffer_size},
    }


def setup_model(model_params):
    """ ``exp_params["model"]`` -> decomposition model (lib/setup_model.py:21-53) """
    name = model_params["model_name"]
    # the reference's encoder/decoder factories pop keys from the dict; keep the caller's intact
    params = copy.deepcopy(model_params["model_params"])
    if name == "SAVi":
        return SAVi(**params)
    if name == "ExtendedDINOSAUR":
        return ExtendedDINOSAUR(**params)
    raise NotImplementedError(f"'{name = }' not in supported models: ['SAVi', 'ExtendedDINOSAUR']")


def setup_predictor(exp_params):
    """ experiment params -> PredictorWrapper around the predictor (lib/setup_model.py:57-132) """
    model_params = exp_params["model"]["model_params"]
    name = exp_params["predictor"]["predictor_name"]
    pp = exp_params["predictor"]["predictor_params"]
    if name in ("TextOCVP_CustomTF", "TextOCVP_T5"):
        inner = copy.deepcopy(pp["predictor_params"])
        inner["input_buffer_size"] = exp_params["prediction_params"]["input_buffer_size"]
        cls = TextOCVP_CustomTF if name == "TextOCVP_CustomTF" else TextOCVP_T5
        core = cls(slot_dim=model_params["slot_dim"], predictor_params=inner,
                   fusion_params=pp.get("fusion_params"),
                   text_encoder_params=pp.get("text_encoder_params"))
    elif name in ("VanillaTransformer", "OCVPSeq"):
        cls = VanillaTransformerPredictor if name == "VanillaTransformer" else OCVPSeq
        core = cls(num_slots=model_params["num_slots"], slot_dim=model_params["slot_dim"],
                   input_buffer_size=exp_params["prediction_params"]["input_buffer_size"], **pp)
    else:
        raise NameError(f"Predictor '{name}' not in recognized predictors")
    return PredictorWrapper(exp_params=exp_params, predictor=core)


def load_checkpoint(checkpoint_path, model, only_model=True, map_cpu=False, **kwargs):
    """
    Strict load of ``torch.load(path)['model_state_dict']`` (lib/setup_model.py:189-227),
    including the 'predictor.' prefix shim for bare-predictor checkpoints (:214-221).
    """
    if checkpoint_path is None:
        return model
    if not os.path.exists(checkpoint_path):
        raise FileNotFoundError(f"Checkpoint {checkpoint_path} does not exist ...")
    ckpt = torch.load(checkpoint_path, map_location="cpu" if map_cpu else None)
    sd = ckpt["model_state_dict"]
    first_model = next(iter(model.state_dict().keys()))
    first_ckpt = next(iter(sd.keys()))
    if first_model.startswith("predictor") and not first_ckpt.startswith("predictor"):
        sd = {f"predictor.{k}": v for k, v in sd.items()}
    # load_state_dict marks the module "range-unchecked" (RangeGuard): its first forward verifies every
    # fp16-plane operand and fails loudly (or forward_eval re-calibrates) instead of saturating silently
    model.load_state_dict(sd)
    if only_model:
        return model
    # training state, same tuple as the reference (lib/setup_model.py:228-240)
    optimizer, scheduler = kwargs["optimizer"], kwargs.get("scheduler")
    lr_warmup = kwargs.get("lr_warmup")
    if hasattr(optimizer, "load_training_state"):
        # textocvp_amd.train.PredictorTrainStep: Adam moments + step counter (schedule is a function of it)
        optimizer.load_training_state(ckpt)
        return model, optimizer, scheduler, lr_warmup, ckpt["epoch"] + 1
    optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    if scheduler is not None and "scheduler_state_dict" in ckpt:
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    if lr_warmup is not None and "lr_warmup" in ckpt:
        lr_warmup.load_state_dict(ckpt["lr_warmup"])
    epoch = ckpt["epoch"] + 1
    return model, optimizer, scheduler, lr_warmup, epoch


def calibrate_precision(decomp_model, predictor, videos, num_context, num_preds, **others):
    """
    One checked pass over a representative batch that makes the fast arithmetic safe for a given
    checkpoint.  The fp16-plane modes (f16x3 GEMMs / convs / attention products) are fp32-class only
    while |activation| < 255 and |weight| < 63 (operands saturate beyond).  This runs the path once with
    every such kernel verifying its operands; a ``kernels.TocvpRangeError`` names the arithmetic knob
    (module, attribute) that governs the failing kernel, and exactly THAT module is moved to its
    range-free arithmetic (``type(module).range_fallbacks``: decoder convs -> bf16x3, predictor GEMMs ->
    bf16x6, encoder / DINOSAUR decoder -> fp32 MFMA, attention products -> fp32 MFMA), then the pass
    is repeated.  Returns {(module class, attribute): new mode} of what was changed and clears the
    modules' "unchecked" marks (models/Blocks/model_utils.py::RangeGuard).
    """
    from . import kernels as K
    from .evaluator import forward_eval
    changed = {}
    for _ in range(16):
        try:
            with K.check_range(True):
                forward_eval(decomp_model, predictor, videos, num_context, num_preds,
                             overlap_decode=False, _calibrating=True, **others)
            for m in (decomp_model, predictor):
                if hasattr(m, "_range_unchecked"):
                    m._range_unchecked = False
            return changed
        except K.TocvpRangeError as err:
            owner = err.owner
            if owner == ("kernels", "_ATTN_QK16") and K._ATTN_QK16:
                K._ATTN_QK16 = False                     # process-wide: exact fp32 attention products
                changed[("kernels", "_ATTN_QK16")] = False
                continue
            if owner is None:
                raise
            mod, attr = owner
            table = getattr(type(mod), "range_fallbacks", {}).get(attr, {})
            cur = getattr(mod, attr, None)
            if cur not in table:
                raise
            setattr(mod, attr, table[cur])
            changed[(type(mod).__name__, attr)] = table[cur]
    raise K.TocvpError("calibrate_precision: operands still out of range after every fallback")
