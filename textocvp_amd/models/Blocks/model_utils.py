"""
Parameter-initialisation helper and the derived-weight cache used by the module mirrors.
Reference counterpart: models/Blocks/model_utils.py:66-79 (init_xavier_).
"""

import torch
import torch.nn as nn

__all__ = ["init_xavier_", "Derived", "require_inference", "RangeGuard", "refuse_replication", "freeze_params",
           "unfreeze_params", "count_model_params", "structure_epoch", "tracks_structure", "cached_params"]


# ------------------------------------------------------------------------------------------------
# Structure epoch: a process-wide counter that moves whenever ANY nn.Module gets a parameter, buffer or
# sub-module (re)registered -- ``mod.weight = nn.Parameter(...)``, ``block.mlp = ...``, ``load_state_dict(assign=True)``
# (it goes through ``setattr``), ``parametrize.register_parametrization``, pruning / weight_norm (they register
# ``*_orig`` / ``*_g`` parameters) -- or one of this package's modules goes through ``_apply`` (``.to()`` / ``.float()``
# with ``torch.__future__.set_overwrite_module_params_on_conversion(True)`` puts NEW Parameter objects into
# ``_parameters`` without registering them).  ``cached_params`` keeps per-module tuples of Parameter objects for the
# host-bound small-batch path and drops them when the epoch moved: a replaced parameter can never be computed with.
# ------------------------------------------------------------------------------------------------

_EPOCH = [0]


def _bump(*args, **kwargs):
    _EPOCH[0] += 1
    return None                                   # registration hooks: None keeps the registered object


nn.modules.module.register_module_parameter_registration_hook(_bump)
nn.modules.module.register_module_module_registration_hook(_bump)
nn.modules.module.register_module_buffer_registration_hook(_bump)


def structure_epoch():
    return _EPOCH[0]


def tracks_structure(cls):
    """ class decorator: ``_apply`` (``.to`` / ``.cuda`` / ``.float`` ...) moves the structure epoch """
    orig = cls._apply

    def _apply(self, fn, *args, **kwargs):
        _bump()
        return orig(self, fn, *args, **kwargs)
    cls._apply = _apply
    return cls


def _volatile(mod):
    """ modules whose ``weight`` is recomputed per access or per forward (parametrizations; weight_norm / pruning set it
    from a forward pre-hook): their attributes are never cached """
    for m in mod.modules():
        if "parametrizations" in m._modules or m._forward_pre_hooks:
            return True
    return False


def cached_params(mod, build):
    """
    ``build(mod)`` -> tuple of the module's parameters / constants for the hot path, cached in the module's __dict__:
    on a host-bound step (8 sequences: 2400 launches from Python) nn.Module.__getattr__ and nn.Sequential.__getitem__
    are a measurable share of the time per launch (scripts/host_profile.py).  ``load_state_dict`` / ``.to()`` replace the
    parameters' DATA in place (the derived-weight caches follow data_ptr / _version); every way of replacing the
    OBJECTS moves the structure epoch (above), which invalidates the tuple here.
    """
    hit = mod.__dict__.get("_tocvp_params")
    if hit is not None and hit[0] == _EPOCH[0]:
        return hit[1]
    vals = build(mod)
    mod.__dict__["_tocvp_params"] = (-1 if _volatile(mod) else _EPOCH[0], vals)
    return vals


@torch.no_grad()
def init_xavier_(model: nn.Module):
    """ xavier-uniform for >=2-d parameters, zeros for '*.bias' (model_utils.py:66-79). """
    for name, p in model.named_parameters():
        if name.endswith(".bias"):
            p.zero_()
        elif p.dim() > 1:
            nn.init.xavier_uniform_(p)


def freeze_params(model):
    """ requires_grad = False on every parameter (model_utils.py:47-53) """
    for p in model.parameters():
        p.requires_grad = False
    return model


def unfreeze_params(model):
    """ requires_grad = True on every parameter (model_utils.py:56-62) """
    for p in model.parameters():
        p.requires_grad = True
    return model


def count_model_params(model, verbose=False):
    """ number of trainable parameters (model_utils.py:37-44) """
    n = sum(p.numel() for p in model.parameters() if p.requires_grad)
    if verbose:
        print(f"Model has {n} trainable parameters")
    return n


def refuse_replication(self):
    """
    ``_replicate_for_data_parallel`` of the top-level mirrors.  The reference wraps its models in
    ``nn.DataParallel(model, device_ids=range(num_gpus))`` (base/baseEvaluator.py:142-145, :168-171); with
    ONE device that wrapper calls the module directly and works here unchanged.  With several devices
    ``DataParallel.replicate`` would shallow-copy the module per forward: the replicas would share the
    derived-weight caches (``Derived``), the caption K/V cache and the slot-attention workspace of device 0
    -- raw pointers handed to kernels running on other devices.  This path is one process per GPU instead
    (``evaluator.shard_batches`` / ``gather_metrics``, ``bench.py --gpus N``), so replication is refused loudly.
    """
    raise RuntimeError(
        f"{type(self).__name__}: textocvp_amd modules cannot be replicated by nn.DataParallel over several "
        f"devices (derived-weight caches and kernel workspaces are per device). Run one process per visible "
        f"GPU -- torchrun / `bench.py --gpus N` with textocvp_amd.evaluator.shard_batches + gather_metrics -- "
        f"or restrict the wrapper to one device (device_ids=[torch.cuda.current_device()]).")


class Derived:
    """
    Cache of tensors derived from parameters (fused / repacked weights, position tables, the
    collapsed decoder layer 0).  An entry is rebuilt whenever one of its source parameters was
    replaced, moved or modified in place (data_ptr / _version / device signature), so
    ``load_state_dict`` and ``.to(device)`` invalidate it without any hook.
    """

    def __init__(self):
        self._store = {}

    def get(self, key, sources, builder):
        sig = tuple((t.data_ptr(), t._version, t.device.index) for t in sources)
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        with torch.no_grad():
            val = builder()
        self._store[key] = (sig, val)
        return val


class RangeGuard:
    """
    Mixin of the top-level modules (SAVi, ExtendedDINOSAUR, PredictorWrapper).  The default arithmetic
    splits operands into fp16 planes that are valid for |activation| < 255, |weight| < 63 and SATURATE
    beyond.  Loading a state_dict marks the module unchecked; its next forward then runs with every
    fp16-plane kernel verifying its operands (one slow pass) and raises ``kernels.TocvpRangeError``
    naming the knob to change -- a checkpoint outside the range can never produce silently wrong
    frames.  ``evaluator.forward_eval`` resolves the error by itself through
    ``setup_model.calibrate_precision`` (moves the named module to range-free arithmetic).
    """

    def _init_range_guard(self):
        self._range_unchecked = False
        self.register_load_state_dict_post_hook(RangeGuard._mark_unchecked)

    @staticmethod
    def _mark_unchecked(module, incompatible_keys):
        module._range_unchecked = True

    def _guarded(self, fn, *args, **kwargs):
        if not getattr(self, "_range_unchecked", False):
            return fn(*args, **kwargs)
        from ... import kernels as K
        with K.check_range(True):
            out = fn(*args, **kwargs)
        self._range_unchecked = False
        return out


def require_inference(module):
    """ The HIP path has no backward kernels yet (training is a later SURVEY 8f row). """
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise RuntimeError(
            f"{type(module).__name__}: the MI355X kernels are inference-only; call under "
            f"torch.no_grad() (as the reference evaluator does, base/baseEvaluator.py:175)")
