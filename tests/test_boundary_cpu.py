"""
CPU checks of the drop-in boundary: checkpoint layout (state_dict keys + shapes vs the manifest
captured from the reference), factories, error behaviour, C-ABI symbol table.  No GPU compute.
"""

import ctypes
import os
import sys
import re

import pytest
import torch

from textocvp_amd import build as tbuild
from textocvp_amd import synth
from textocvp_amd.setup_model import (default_exp_params, load_checkpoint, setup_model,
                                      setup_predictor)

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_state_dict_layout_matches_reference(manifest):
    exp = default_exp_params(num_slots=30, num_preds=19)
    savi = setup_model(exp["model"])
    pred = setup_predictor(exp)
    got_s = {k: list(v.shape) for k, v in savi.state_dict().items()}
    got_p = {k: list(v.shape) for k, v in pred.state_dict().items()}
    assert got_s == manifest["SAVi"]
    assert got_p == manifest["PredictorWrapper"]
    assert len(got_s) == 62 and len(got_p) == 253
    # attributes the reference evaluator reads (05_evaluate_predictor.py:71-72, baseEvaluator.py:154)
    assert savi.num_slots == 30 and savi.slot_dim == 128
    assert hasattr(pred, "predictor") and next(iter(got_p)).startswith("predictor")


def test_factory_does_not_consume_callers_config():
    exp = default_exp_params()
    setup_model(exp["model"])
    setup_model(exp["model"])        # the reference pops keys; a second build must still work here
    assert "num_channels" in exp["model"]["model_params"]["encoder"]["encoder_params"]


def test_checkpoint_roundtrip_and_prefix_shim(tmp_path):
    exp = default_exp_params(num_slots=7, num_preds=4)
    savi = setup_model(exp["model"])
    synth.fill_module_(savi, prefix="savi.")
    path = tmp_path / "checkpoint_epoch_final.pth"
    torch.save({"epoch": 3, "model_state_dict": savi.state_dict()}, path)
    other = load_checkpoint(str(path), setup_model(exp["model"]), only_model=True)
    # freshly loaded weights are marked for one range-checked forward (no silent fp16-plane saturation)
    assert other._range_unchecked and not savi._range_unchecked
    for (k, a), (_, b) in zip(savi.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k
    # bare predictor checkpoint (keys without 'predictor.') loads into the wrapper
    pred = setup_predictor(exp)
    bare = {k[len("predictor."):]: v for k, v in pred.state_dict().items()}
    p2 = tmp_path / "pred.pth"
    torch.save({"epoch": 0, "model_state_dict": bare}, p2)
    load_checkpoint(str(p2), setup_predictor(exp), only_model=True)
    with pytest.raises(FileNotFoundError):
        load_checkpoint(str(tmp_path / "nope.pth"), savi)


def test_error_behaviour_matches_reference():
    exp = default_exp_params(num_slots=7, num_preds=4)
    savi = setup_model(exp["model"])
    with pytest.raises(NameError):
        savi(mode="nonsense")
    pred = setup_predictor(exp)
    with torch.no_grad(), pytest.raises(KeyError):
        pred(torch.zeros(1, 5, 7, 128))                       # no caption_tokens
    with pytest.raises(ValueError):
        from textocvp_amd.models.Blocks.initializers import get_initializer
        get_initializer("bogus", 128, 7)


def test_product_refuses_cpu_tensors():
    """ no CPU fallback: a CPU forward must fail loudly, never silently compute """
    from textocvp_amd import kernels
    exp = default_exp_params(num_slots=7, num_preds=4)
    savi = setup_model(exp["model"]).eval()
    with torch.no_grad(), pytest.raises((kernels.TocvpError, RuntimeError)):
        savi(mode="decode", slots=torch.zeros(1, 7, 128))


def test_c_abi_exports_every_declared_symbol():
    """ the library loads and exports exactly the entry points include/tocvp.h declares """
    from textocvp_amd import kernels
    path = tbuild.build()
    handle = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "tocvp.h")).read()
    declared = set(re.findall(r"\b(tocvp_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in tocvp.h but not exported"
    assert declared == set(kernels.EXPORTED_SYMBOLS)
    handle.tocvp_version.restype = ctypes.c_int
    assert handle.tocvp_version() == 100


def test_product_never_imports_the_oracle():
    """ oracle/ is test infrastructure: nothing under textocvp_amd/ may reference it """
    pkg = os.path.join(ROOT, "textocvp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower() or f == "synth.py", os.path.join(dirpath, f)


def test_unconditioned_predictor_layouts():
    """ VanillaTransformer / OCVPSeq wrappers expose the reference's state_dict keys and shapes """
    import json
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest_uncond.json")))
    for name in ("VanillaTransformer", "OCVPSeq"):
        exp = default_exp_params(num_slots=7, num_context=2, num_preds=4, predictor_name=name)
        pred = setup_predictor(exp)
        got = {k: list(v.shape) for k, v in pred.state_dict().items()}
        assert got == man[name], name
    bad = default_exp_params(predictor_name="OCVPSeq")
    bad["predictor"]["predictor_name"] = "NoSuchPredictor"
    with pytest.raises(NameError):
        setup_predictor(bad)
    # the reference wrapper demands caption tokens even for unconditioned predictors
    pred = setup_predictor(default_exp_params(num_slots=7, predictor_name="OCVPSeq"))
    with torch.no_grad(), pytest.raises(KeyError):
        pred(torch.zeros(1, 3, 7, 128))


def test_dinosaur_layout_and_vit_backbone_keys():
    """ ExtendedDINOSAUR: decoder keys from the reference's manifest; encoder.vit_backbone.* follows timm's
    published VisionTransformer names / shapes for vit_base_patch14_dinov2 at 224 px (timm is not importable
    here, so this half of the layout is pinned by the literal list below, not by a captured manifest) """
    import json
    from textocvp_amd.setup_model import default_dinosaur_params
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest_dinosaur_decoder.json")))
    model = setup_model(default_dinosaur_params(num_slots=24, img_size=224))
    sd = model.state_dict()
    got = {k[len("decoder."):]: list(v.shape) for k, v in sd.items() if k.startswith("decoder.")}
    assert got == man
    for k in ("linear_feat_proj.0.weight", "linear_feat_proj.3.bias", "slot_attention.to_q.weight",
              "initializer.slots_mu", "transition_module.attn.q.weight"):
        assert k in sd, k
    vit = {k[len("encoder.vit_backbone."):]: tuple(v.shape) for k, v in sd.items() if k.startswith("encoder.")}
    expect = {"cls_token": (1, 1, 768), "pos_embed": (1, 257, 768), "patch_embed.proj.weight": (768, 3, 14, 14),
              "patch_embed.proj.bias": (768,), "norm.weight": (768,), "norm.bias": (768,)}
    for i in range(12):
        b = f"blocks.{i}."
        expect.update({b + "norm1.weight": (768,), b + "norm1.bias": (768,), b + "attn.qkv.weight": (2304, 768),
                       b + "attn.qkv.bias": (2304,), b + "attn.proj.weight": (768, 768), b + "attn.proj.bias": (768,),
                       b + "ls1.gamma": (768,), b + "norm2.weight": (768,), b + "norm2.bias": (768,),
                       b + "mlp.fc1.weight": (3072, 768), b + "mlp.fc1.bias": (3072,),
                       b + "mlp.fc2.weight": (768, 3072), b + "mlp.fc2.bias": (768,), b + "ls2.gamma": (768,)})
    assert vit == expect
    assert not any(p.requires_grad for p in model.encoder.parameters())          # frozen (ExtendedDINOSAUR.py:92)
    from textocvp_amd import kernels
    with torch.no_grad(), pytest.raises((kernels.TocvpError, RuntimeError)):
        model(mode="decomp", x=torch.zeros(1, 2, 3, 224, 224), num_imgs=2)       # no CPU path
    with torch.no_grad(), pytest.raises(ValueError):
        model.encoder(torch.zeros(1, 3, 64, 64))                                # timm: input size must match


def test_config4_pair_layout_matches_the_reference_modules():
    """ configs[3]: the WHOLE state_dict of ExtendedDINOSAUR (24 slots, 224 px) and of PredictorWrapper(TextOCVP_T5)
    equals, name by name and shape by shape, the manifest captured from the reference's own modules
    (state_dict_manifest_c4.json, make_golden.py::e2e_c4_fixtures; the backbone's entries under timm's names,
    mapped from transformers.Dinov2Model through dinov2_key_map.json) """
    import json
    from textocvp_amd.setup_model import default_dinosaur_params
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest_c4.json")))
    model = setup_model(default_dinosaur_params(num_slots=24, img_size=224))
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == man["ExtendedDINOSAUR"]
    pred = setup_predictor(default_exp_params(num_slots=24, num_preds=29, predictor_name="TextOCVP_T5"))
    assert {k: list(v.shape) for k, v in pred.state_dict().items()} == man["PredictorWrapper_T5"]
    assert len(man["ExtendedDINOSAUR"]) == 255 and len(man["PredictorWrapper_T5"]) == 273


def test_t5_predictor_layout():
    """ TextOCVP_T5: text_encoder.* keys are those of transformers.T5EncoderModel (t5-small) """
    import json
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest_t5.json")))
    pred = setup_predictor(default_exp_params(num_slots=7, predictor_name="TextOCVP_T5"))
    sd = pred.state_dict()
    got = {k[len("predictor.text_encoder."):]: list(v.shape) for k, v in sd.items()
           if k.startswith("predictor.text_encoder.")}
    assert got == man
    assert "predictor.predictor.7.cross_attention.cross_attn.q.weight" in sd
    assert not any(p.requires_grad for p in pred.predictor.text_encoder.parameters())
    with torch.no_grad(), pytest.raises(KeyError):
        pred(torch.zeros(1, 3, 7, 128), caption_tokens=torch.zeros(1, 4, dtype=torch.int64))


def test_precision_knobs(monkeypatch):
    """ per-module arithmetic knobs and the all-fp32 master switch (host logic, no kernels) """
    from textocvp_amd.precision import knob
    monkeypatch.delenv("TOCVP_PRECISION", raising=False)
    monkeypatch.delenv("TOCVP_DECODER_PRECISION", raising=False)
    assert knob("TOCVP_DECODER_PRECISION", "f16f8") == "f16f8"
    monkeypatch.setenv("TOCVP_PRECISION", "fp32")
    assert knob("TOCVP_DECODER_PRECISION", "f16f8") == "fp32"
    monkeypatch.setenv("TOCVP_DECODER_PRECISION", "bf16x3")          # a specific knob wins over the master switch
    assert knob("TOCVP_DECODER_PRECISION", "f16f8") == "bf16x3"


def test_training_lr_schedule_matches_reference_driver():
    """
    The LR of every optimiser step equals what the reference trainer sets: LRWarmUp + CosineAnnealingLR driven
    by WarmupVSScehdule with the 0-based ``iter_`` BEFORE the step (base/basePredictorTrainer.py:280-286,
    lib/schedulers.py:88-157; restated below from those lines, torch's own CosineAnnealingLR included).
    """
    import types
    from textocvp_amd.train.step import PredictorTrainStep
    for W, T in ((5, 40), (-1, 30)):
        cfg = types.SimpleNamespace(lr=1e-4, warmup_steps=W if W > 0 else None, scheduler_steps=T, eta_min=1e-7)
        opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=T, eta_min=1e-7)
        active, final_step = True, -1
        for iter_ in range(0, 30):
            # WarmupVSScehdule.__call__ / LRWarmUp.__call__ / update_scheduler
            if active:
                if iter_ > W:
                    final_step, active = iter_, False
                elif iter_ >= 0:
                    for g in opt.param_groups:
                        g["lr"] = 1e-4 * (iter_ / W)
            else:
                sched.step()
            used = opt.param_groups[0]["lr"]
            assert PredictorTrainStep.lr_at(cfg, iter_) == pytest.approx(used, rel=1e-6, abs=1e-15), (W, iter_)
            opt.step()                                    # keeps torch's scheduler-order warning quiet
    cfg = types.SimpleNamespace(lr=1e-4, warmup_steps=2000, scheduler_steps=1e6, eta_min=1e-7)
    assert PredictorTrainStep.lr_at(cfg, 0) == 0.0        # the reference's first step runs at lr = 0
    assert PredictorTrainStep.lr_at(cfg, 2000) == pytest.approx(1e-4)
    assert PredictorTrainStep.lr_at(cfg, 2001) == pytest.approx(1e-4)


def test_graphed_step_result_is_a_lazy_mapping():
    """ PredictorTrainStep.step_graphed returns a StepResult: the device snapshot {sum sq slot, sum sq img, grad
    norm} is read on first access only, keys and arithmetic as the eager step's dict (04_train_predictor.py:96-104
    logs loss / pred_slot_mse / pred_img_mse) """
    from textocvp_amd.train.step import StepResult

    class Snap:
        reads = 0

        def tolist(self):
            Snap.reads += 1
            return [8.0, 6.0, 3.5]
    r = StepResult(Snap(), 0.25, 0.5, 1e-4)
    assert Snap.reads == 0 and list(r) == ["loss", "pred_slot_mse", "pred_img_mse", "grad_norm", "lr"] and len(r) == 5
    assert r["pred_slot_mse"] == 2.0 and r["pred_img_mse"] == 3.0 and r["loss"] == 5.0
    assert r["grad_norm"] == 3.5 and r["lr"] == 1e-4 and dict(r)["loss"] == 5.0
    assert Snap.reads == 1                                    # one read-back, however often the numbers are used


def test_dataparallel_replication_is_refused():
    """ nn.DataParallel over several devices shallow-copies modules per forward; the mirrors hold per-device
    caches (derived weights, caption K/V, kernel workspaces), so replication raises instead of sharing them """
    exp = default_exp_params(num_slots=7)
    for m in (setup_model(exp["model"]), setup_predictor(exp)):
        with pytest.raises(RuntimeError, match="one process per visible GPU"):
            m._replicate_for_data_parallel()
        # a single-device wrapper never replicates: construction and attribute access as in baseEvaluator.py
        dp = torch.nn.DataParallel(m.eval(), device_ids=None)
        assert dp.module is m


def test_dropin_alias_package_resolves_the_reference_imports():
    """ PYTHONPATH=<repo>/dropin:<repo>: every `from models.X import Y` of the reference's lib/setup_model.py:43-127,
    base/basePredictorTrainer.py:20 and data/*.py resolves to the textocvp_amd.models module OBJECTS (zero-edit) """
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "from models.SAVi import SAVi\n"
        "from models.ExtendedDINOSAUR import ExtendedDINOSAUR\n"
        "from models.Predictors.OCVP import VanillaTransformerPredictor, OCVPSeq\n"
        "from models.Predictors.text_cond_OCVP import TextOCVP_CustomTF, TextOCVP_T5\n"
        "from models.Predictors.predictor_wrapper import PredictorWrapper\n"
        "from models.Blocks.model_utils import freeze_params\n"
        "from models.EncodersDecoders.text_encoders import CustomTokenizer\n"
        "import models, textocvp_amd.models.SAVi as S, textocvp_amd.models.Predictors.predictor_wrapper as W\n"
        "assert SAVi is S.SAVi and PredictorWrapper is W.PredictorWrapper and models.__name__ == 'textocvp_amd.models'\n"
        "print('alias ok')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "dropin"), root]))
    res = subprocess.run([sys.executable, "-c", code], env=env, cwd="/tmp", capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "alias ok" in res.stdout, res.stderr[-2000:]


def test_cached_module_parameters_follow_every_way_of_replacing_them():
    """
    ``model_utils.cached_params`` (the per-module parameter tuples of the host-bound small-batch path) must never hand
    out a REPLACED parameter: attribute assignment, sub-module swap, ``load_state_dict(assign=True)``, parametrizations
    and ``.to()`` under ``set_overwrite_module_params_on_conversion`` each give a fresh tuple; plain ``load_state_dict`` /
    in-place updates keep the objects (their data changes in place, which the derived-weight caches follow).
    """
    import torch.nn as nn
    import torch.nn.utils.parametrize as parametrize
    from textocvp_amd.models.Blocks.attention import TransformerBlock
    from textocvp_amd.models.Blocks.model_utils import cached_params

    def get_ln(m):
        return cached_params(m.layernorm_query, lambda ln: (ln.weight, ln.bias, ln.eps))

    def get_mlp(m):
        return cached_params(m.mlp, lambda s: (s[0].weight, s[0].bias, s[2].weight, s[2].bias))

    blk = TransformerBlock(embed_dim=128, num_heads=4, mlp_size=256, pre_norm=False)
    w0 = get_ln(blk)[0]
    assert get_ln(blk)[0] is w0 is blk.layernorm_query.weight
    blk.load_state_dict({k: v + 1 for k, v in blk.state_dict().items()})             # in place: same objects
    assert get_ln(blk)[0] is w0
    blk.layernorm_query.weight = nn.Parameter(torch.full((128,), 3.0))              # attribute assignment
    assert get_ln(blk)[0] is blk.layernorm_query.weight and float(get_ln(blk)[0][0]) == 3.0
    old = get_mlp(blk)[0]
    blk.mlp[0] = nn.Linear(128, 256)                                                 # sub-module swap
    assert get_mlp(blk)[0] is blk.mlp[0].weight and get_mlp(blk)[0] is not old
    old = get_mlp(blk)[2]
    blk.load_state_dict({k: v.clone() for k, v in blk.state_dict().items()}, assign=True)
    assert get_mlp(blk)[2] is blk.mlp[2].weight and get_mlp(blk)[2] is not old

    class Double(nn.Module):
        def forward(self, w):
            return 2.0 * w
    parametrize.register_parametrization(blk.mlp[2], "weight", Double())            # recomputed per access: never cached
    with torch.no_grad():
        a = get_mlp(blk)[2]
        blk.mlp[2].parametrizations.weight.original.add_(1.0)
        b = get_mlp(blk)[2]
    assert torch.equal(b, a + 2.0)

    blk2 = TransformerBlock(embed_dim=128, num_heads=4, mlp_size=256, pre_norm=False)
    w0 = get_ln(blk2)[0]
    prev = torch.__future__.get_overwrite_module_params_on_conversion()
    torch.__future__.set_overwrite_module_params_on_conversion(True)
    try:
        blk2.double()
    finally:
        torch.__future__.set_overwrite_module_params_on_conversion(prev)
    assert blk2.layernorm_query.weight is not w0 and get_ln(blk2)[0] is blk2.layernorm_query.weight


def test_every_knob_is_documented():
    """ KNOBS.md lists every TOCVP_* environment variable the sources read, and nothing else """
    import subprocess
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "list_knobs.py"), "--check"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr


def test_winograd_matrices_of_the_decoder_conv():
    """ host side of csrc/conv_wino.hip: the output transform the Python side folds into the kernel's coefficients
    (kernels._WINO_AT), the input transform the kernel applies (BT, restated here from its source comments / code) and the
    weight transform of its packing kernel (WINO_G, parsed from the source) satisfy the F(4, 5) identity
    y[a] = sum_k g[k] d[a + k] for every a, and the vertical-only convolution built from them equals torch's conv2d """
    import re
    import numpy as np
    import torch.nn.functional as F
    from textocvp_amd import kernels as K
    AT = np.array(K._WINO_AT, dtype=np.float64)
    BT = np.array([[1, 0, -5.25, 0, 5.25, 0, -1, 0],
                   [0, 1, 1, -4.25, -4.25, 1, 1, 0],
                   [0, -1, 1, 4.25, -4.25, -1, 1, 0],
                   [0, 0.5, 0.25, -2.5, -1.25, 2, 1, 0],
                   [0, -0.5, 0.25, 2.5, -1.25, -2, 1, 0],
                   [0, 2, 4, -2.5, -5, 0.5, 1, 0],
                   [0, -2, 4, 2.5, -5, -0.5, 1, 0],
                   [0, -1, 0, 5.25, 0, -5.25, 0, 1]], dtype=np.float64)
    src = open(os.path.join(ROOT, "textocvp_amd", "csrc", "conv_wino.hip")).read()
    body = src[src.index("WINO_G[NXI][5] = {"):]
    body = body[:body.index("};")]
    rows = re.findall(r"\{([^{}]+)\}", body)
    G = np.array([[eval(tok) for tok in r.split(",")] for r in rows], dtype=np.float64)
    assert AT.shape == (4, 8) and G.shape == (8, 5)
    rng = np.random.default_rng(5)
    g, d = rng.standard_normal(5), rng.standard_normal(8)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(g[k] * d[a + k] for k in range(5)) for a in range(4)])
    assert np.abs(y - ref).max() < 1e-12
    # |BT| row sums bound the growth of the transformed operand: 15 x 16 x 255.9 < 65504 (the kernel's x 16 scale)
    assert np.abs(BT).sum(1).max() == 15.0 and 15.0 * 16.0 * 255.9 < 65504.0
    # the nested form the kernel evaluates (vertical transform, five direct horizontal taps) on a small image
    w = torch.from_numpy(rng.standard_normal((3, 2, 5, 5)))
    x = torch.from_numpy(rng.standard_normal((1, 2, 8, 9)))
    ref2 = F.conv2d(x, w, padding=2)
    xp = F.pad(x, (2, 2, 2, 2))
    U = torch.einsum("ak,ockl->aloc", torch.from_numpy(G), w)                     # (8, 5, O, C)
    out = torch.zeros_like(ref2)
    for t in range(2):
        V = torch.einsum("ai,ncix->ancx", torch.from_numpy(BT), xp[:, :, 4 * t:4 * t + 8])      # (8, n, C, W + 4)
        M = torch.zeros((8, 1, 3, 9), dtype=torch.float64)
        for dx in range(5):
            M += torch.einsum("ancx,aoc->anox", V[:, :, :, dx:dx + 9], U[:, dx])
        out[:, :, 4 * t:4 * t + 4] = torch.einsum("ba,anox->nobx", torch.from_numpy(AT), M)
    assert (out - ref2).abs().max().item() < 1e-12
