"""
Deterministic synthetic weights / inputs for the slot-rollout hot path.

The reference ships no checkpoints and there is no network, so every
parity test, golden fixture and bench run fills the reference-layout ``state_dict`` from this
generator instead of committing hundreds of MB of weights.  Values depend only on
``(seed, parameter name, shape)`` through numpy's PCG64 stream, which is stable across numpy
versions and machines, so the golden fixtures generated in the survey container and the GPU
box see bit-identical weights.

Nothing here is part of the reference; the scaling rules are only chosen so that activations
stay O(1) through the conv / transformer stacks (He-uniform fan-in scaling for matrices,
LayerNorm gains near 1, small biases).
"""

import zlib
import math

import numpy as np
import torch

__all__ = ["synth_array", "synth_tensor", "fill_module_", "synth_videos", "synth_captions",
           "synth_noise"]


def _rng(name, seed):
    """ Independent PCG64 stream per (seed, name). """
    key = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), key])))


def synth_array(name, shape, kind="uniform", scale=1.0, seed=0):
    """
    float32 array of the given shape.

    kind: 'uniform' -> U(-scale, scale); 'normal' -> N(0, scale^2); 'unit' -> U(0, scale)
    """
    rng = _rng(name, seed)
    shape = tuple(int(s) for s in shape)
    if kind == "uniform":
        out = rng.uniform(-scale, scale, size=shape)
    elif kind == "normal":
        out = rng.standard_normal(size=shape) * scale
    elif kind == "unit":
        out = rng.uniform(0.0, scale, size=shape)
    else:
        raise ValueError(f"unknown {kind = }")
    return out.astype(np.float32)


def synth_tensor(name, shape, kind="uniform", scale=1.0, seed=0):
    """ Same as synth_array but as a CPU torch tensor. """
    return torch.from_numpy(synth_array(name, shape, kind=kind, scale=scale, seed=seed))


FAMILIES = ("damped", "undamped", "xavier")


def _xavier_values(name, shape, seed):
    """
    Family "xavier": the DISTRIBUTION of the reference's own initialisation (models/SAVi.py:279-293
    ``_init_model`` -> ``init_xavier_``, Blocks/model_utils.py:66-79): xavier-uniform weights
    (bound sqrt(6 / (fan_in + fan_out)), conv fans include the kernel area), zero biases, LayerNorm
    gains 1, slot statistics U(+-sqrt(6 / (1 + D))) (initializers.py:80-83).  The reference draws
    these from torch's global generator in module-construction order, which a name-keyed generator
    cannot replay; what matters for parity is the scale of every tensor (O(1) RGB head, zero-centred
    alpha), not the particular draw.  GRU weight_hh is xavier here (orthogonal in the reference).
    """
    last = name.split(".")[-1]
    if "bias" in last:
        return np.zeros(shape, dtype=np.float32)
    if len(shape) == 1:
        return np.ones(shape, dtype=np.float32)
    if len(shape) == 3 and shape[0] == 1:
        return synth_array(name, shape, "uniform", scale=math.sqrt(6.0 / (1 + shape[-1])), seed=seed)
    rf = 1
    for s in shape[2:]:
        rf *= int(s)
    fan_in, fan_out = int(shape[1]) * rf, int(shape[0]) * rf
    return synth_array(name, shape, "uniform", scale=math.sqrt(6.0 / (fan_in + fan_out)), seed=seed)


def _param_values(name, shape, seed, family="damped"):
    """
    Scaling rule per parameter name/shape (see module docstring).  ``family``:
      "damped"    the round-1 weights: RGB rows of the decoder head scaled by 0.05 (pixels inside (0, 1));
      "undamped"  the same weights with an O(1) RGB head (plain He-uniform rows, alpha row x2): pixel
                  errors of the decoder arithmetic are NOT attenuated, rendered values leave [0, 1];
                  ExtendedDINOSAUR: alpha row of the MLPPatchDecoder head x12 (sharp masks), RGB bias 0.5;
      "xavier"    the distribution of the reference's own init (see _xavier_values).
    """
    if family not in FAMILIES:
        raise ValueError(f"unknown weight family {family!r}")
    if family == "xavier":
        return _xavier_values(name, shape, seed)
    last = name.split(".")[-1]
    if name.endswith("pe.pe") or last == "pe":
        # learned temporal positional encoding, reference scale is token_dim ** -0.5
        return synth_array(name, shape, "normal", scale=shape[-1] ** -0.5, seed=seed)
    if "embedding" in name and len(shape) == 2 and "pos_embedding" not in name:
        return synth_array(name, shape, "uniform", scale=1.0, seed=seed)
    if name.endswith("decoder.decoder.4.bias"):
        # rendered RGB should live inside (0, 1) so that the evaluator's clamp is not saturated
        return 0.5 + synth_array(name, shape, "uniform", scale=0.2, seed=seed)
    if name.endswith("decoder.decoder.4.weight"):
        # RGB rows small (stay inside the clamp), alpha row large (sharp, informative masks)
        fan_in = int(shape[1] * shape[2] * shape[3])
        w = synth_array(name, shape, "uniform", scale=math.sqrt(6.0 / fan_in), seed=seed)
        if family == "damped":
            w[:-1] *= 0.05
        w[-1:] *= 2.0
        return w
    if family == "undamped" and ".decoder.mlp." in name and len(shape) == 2 and shape[0] % 2 == 1:
        # MLPPatchDecoder head (features + 1 alpha logit, decoders.py:264-283): a x12 alpha row gives sharp,
        # informative masks (with plain He rows the alpha-softmax over 24 slots stays within 1.5x of uniform and the
        # argmax_K(masks) map hangs on 1e-6 margins); the default family keeps the round-1 values
        w = synth_array(name, shape, "uniform", scale=math.sqrt(6.0 / shape[1]), seed=seed)
        w[-1:] *= 12.0
        return w
    if family == "undamped" and name.endswith("decoder.pos_embed"):
        # learned patch positions of the MLPPatchDecoder at the scale of the slots, so that WHICH slot wins a patch
        # depends on the patch (with +-0.2 positions against +-10 slots one slot takes every patch of a frame)
        return synth_array(name, shape, "normal", scale=2.0, seed=seed)
    if family == "undamped" and "conv_patch_decoder" in name and last == "bias" and tuple(shape) == (3,):
        # RGB bias of the CNN image head: rendered pixels around 0.5 so that the evaluator's clamp rarely saturates
        return 0.5 + synth_array(name, shape, "uniform", scale=0.2, seed=seed)
    if name.endswith("mlp_out.weight"):
        # residual predictor: keep the per-step slot update small so a 19-step rollout stays O(1)
        return synth_array(name, shape, "uniform", scale=0.02 * math.sqrt(6.0 / shape[1]),
                           seed=seed)
    if name.endswith("SelfAttention.q.weight"):
        # T5 attention has no 1/sqrt(d_kv) factor (it lives in the query init): keep logits O(1)
        return synth_array(name, shape, "uniform", scale=math.sqrt(6.0 / shape[1]) / 8.0, seed=seed)
    if "bias" in last:
        return synth_array(name, shape, "uniform", scale=0.1, seed=seed)
    if len(shape) == 1:
        # LayerNorm / BatchNorm gains
        return 1.0 + synth_array(name, shape, "uniform", scale=0.2, seed=seed)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= int(s)
    if len(shape) == 3 and shape[0] == 1:
        # (1, 1, D) / (1, K, D) slot statistics
        fan_in = int(shape[-1])
    return synth_array(name, shape, "uniform", scale=math.sqrt(6.0 / fan_in), seed=seed)


@torch.no_grad()
def fill_module_(module, seed=0, prefix="", family="damped"):
    """
    Overwrite every floating-point entry of ``module.state_dict()`` (parameters and buffers)
    plus known plain-attribute parameters with synthetic values keyed by their state_dict name.
    Works on the reference modules and on this package's mirrors alike, because both expose
    the same key names and shapes.
    """
    sd = module.state_dict()
    for name, t in sd.items():
        if not torch.is_floating_point(t):
            continue
        vals = _param_values(prefix + name, tuple(t.shape), seed, family)
        t.copy_(torch.from_numpy(vals).to(t.dtype))
    return module


def synth_videos(batch, num_frames, channels=3, height=64, width=64, seed=0):
    """ videos in [0, 1], shape (B, L, C, H, W) """
    return synth_tensor("inputs.videos", (batch, num_frames, channels, height, width),
                        "unit", 1.0, seed)


def synth_noise(batch, num_slots, slot_dim, seed=1):
    """ standard-normal slot-initialiser noise, shape (B, K, D) """
    return synth_tensor("inputs.init_noise", (batch, num_slots, slot_dim), "normal", 1.0, seed)


def synth_captions(batch, max_len=12, lengths=None, vocab_size=50, seed=0):
    """
    Token ids in the layout of the reference tokenizer: [CLS]=1, words in [3, vocab), [SEP]=2,
    zero padding after ``lengths[b]`` tokens.  Returns (tokens int64 (B, L), lengths int64 (B,)).
    """
    rng = _rng("inputs.captions", seed)
    if lengths is None:
        lengths = [max_len] * batch
    tokens = np.zeros((batch, max_len), dtype=np.int64)
    for b, n in enumerate(lengths):
        n = int(n)
        assert 2 <= n <= max_len
        tokens[b, 0] = 1
        tokens[b, 1:n - 1] = rng.integers(3, vocab_size, size=n - 2)
        tokens[b, n - 1] = 2
    return torch.from_numpy(tokens), torch.tensor([int(n) for n in lengths], dtype=torch.int64)


def synth_state_dict(manifest, prefix="", seed=0, family="damped"):
    """
    Build a reference-layout weight dict {name: CPU fp32 tensor} from a {name: shape} manifest
    (tests/golden/state_dict_manifest.json) without instantiating any module.
    """
    return {name: torch.from_numpy(_param_values(prefix + name, tuple(shape), seed, family))
            for name, shape in manifest.items()}
