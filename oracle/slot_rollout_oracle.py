"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (fp32, torch-CPU tensor arithmetic) restatement of the reference's autoregressive slot-rollout
hot path: SAVi encode -> slot attention -> transition -> TextOCVP rollout -> spatial-broadcast decode.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this file; the product (``textocvp_amd``) must never route through it.

Every function works on a flat ``dict`` of weights keyed exactly like the reference's
``state_dict``, so it is independent of this repo's module mirrors.
Citations are ``path:line`` relative to the reference root (/root/reference/src/...).

Pinning: checked against golden vectors produced by importing the reference itself in the build
container (tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).
"""

import math

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "layer_norm", "linear", "build_grid", "savi_encode", "slot_attention", "transition_block",
    "savi_decomp", "savi_decode", "text_encoder", "adapted_block", "text_ocvp_step", "rollout",
    "forward_eval", "sub", "uncond_step", "encoder_layer_prenorm", "sinusoid_pe",
    "mlp_patch_decoder", "dinosaur_decomp", "t5_encoder", "vit_encoder", "IMAGENET_MEAN",
    "forward_eval_dinosaur",
]


def sub(sd, prefix):
    """ view of the weight dict below ``prefix`` (with the prefix stripped) """
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


# ------------------------------------------------------------------------------------------------
# elementary ops (torch.nn semantics the reference relies on; SURVEY.md 8c "torch itself")
# ------------------------------------------------------------------------------------------------

def layer_norm(x, w, b, eps):
    """ nn.LayerNorm over the last axis, biased variance. """
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def linear(x, w, b=None):
    """ nn.Linear: y = x W^T + b, W stored (out, in). """
    y = x @ w.t()
    return y if b is None else y + b


def gelu(x):
    """ exact (erf) GELU, the default of nn.TransformerEncoderLayer(activation='gelu') """
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def attention(q, k, v, heads, scale, key_mask=None):
    """
    Multi-head softmax(q k^T * scale) v on (B, T, H*dh) tensors.
    models/Blocks/attention.py:183-215 (attention / split_into_heads / merge_heads).
    key_mask: optional bool (B, Tk), True = key is padding (excluded).
    """
    B, Tq, E = q.shape
    Tk = k.shape[1]
    dh = E // heads
    qh = q.reshape(B, Tq, heads, dh).permute(0, 2, 1, 3)
    kh = k.reshape(B, Tk, heads, dh).permute(0, 2, 1, 3)
    vh = v.reshape(B, Tk, heads, dh).permute(0, 2, 1, 3)
    dots = (qh @ kh.transpose(-1, -2)) * scale                     # (B, H, Tq, Tk)
    if key_mask is not None:
        dots = dots.masked_fill(key_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(dots, dim=-1)
    o = p @ vh                                                     # (B, H, Tq, dh)
    return o.permute(0, 2, 1, 3).reshape(B, Tq, E)


# ------------------------------------------------------------------------------------------------
# SAVi encoder  (models/SAVi.py:226-238)
# ------------------------------------------------------------------------------------------------

def build_grid(resolution):
    """ models/Blocks/model_utils.py:12-34 -> (1, H, W, 4) fp32: [g_y, g_x, 1-g_y, 1-g_x] """
    ranges = [np.linspace(-1.0, 1.0, num=r) for r in resolution]
    g = np.stack(np.meshgrid(*ranges, sparse=False, indexing="ij"), axis=-1)
    g = g.reshape(resolution[0], resolution[1], -1)[None].astype(np.float32)
    return torch.from_numpy(np.concatenate([g, 1.0 - g], axis=-1))


def soft_pos_embed(proj_w, proj_b, resolution):
    """ models/Blocks/model_blocks.py:186-226: Conv1x1(grid) as an (H, W, C) addend. """
    grid = build_grid(resolution)[0]                               # (H, W, 4)
    return grid @ proj_w.reshape(proj_w.shape[0], 4).t() + proj_b  # (H, W, C)


def savi_encode(sd, imgs):
    """
    models/SAVi.py:226-238 + EncodersDecoders/encoders.py:99-159 + SAVi.py:115-120.
    imgs (B, 3, H, W) -> feats (B, H*W, mlp_encoder_dim)
    """
    x = imgs
    i = 0
    while f"encoder.encoder.{i}.block.0.weight" in sd:
        w, b = sd[f"encoder.encoder.{i}.block.0.weight"], sd[f"encoder.encoder.{i}.block.0.bias"]
        x = torch.relu(F.conv2d(x, w, b, padding=w.shape[-1] // 2))
        i += 1
    x = x.permute(0, 2, 3, 1)                                      # (B, H, W, C)
    H, W = x.shape[1:3]
    x = x + soft_pos_embed(sd["encoder_pos_embedding.projection.weight"],
                           sd["encoder_pos_embedding.projection.bias"], (H, W))
    x = x.flatten(1, 2)
    x = layer_norm(x, sd["encoder_mlp.0.weight"], sd["encoder_mlp.0.bias"], 1e-5)
    x = torch.relu(linear(x, sd["encoder_mlp.1.weight"], sd["encoder_mlp.1.bias"]))
    x = linear(x, sd["encoder_mlp.3.weight"], sd["encoder_mlp.3.bias"])
    return x


# ------------------------------------------------------------------------------------------------
# Slot attention + transition  (models/Blocks/attention.py:67-112, 371-396)
# ------------------------------------------------------------------------------------------------

def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """ torch.nn.GRUCell, gate order (r, z, n). """
    gi = linear(x, w_ih, b_ih)
    gh = linear(h, w_hh, b_hh)
    D = h.shape[-1]
    r = torch.sigmoid(gi[..., :D] + gh[..., :D])
    z = torch.sigmoid(gi[..., D:2 * D] + gh[..., D:2 * D])
    n = torch.tanh(gi[..., 2 * D:] + r * gh[..., 2 * D:])
    return (1.0 - z) * n + z * h


def slot_attention(sd, feats, slots, num_iters, eps=1e-8, return_attn=False):
    """
    models/Blocks/attention.py:67-112 with ``sd`` = weights below 'slot_attention.'.
    feats (B, N, Df), slots (B, K, D) -> slots (B, K, D).  scale uses dim_FEATS (:46).
    """
    scale = feats.shape[-1] ** -0.5
    x = layer_norm(feats, sd["norm_input.weight"], sd["norm_input.bias"], 1e-3)
    k = linear(x, sd["to_k.weight"], sd["to_k.bias"])
    v = linear(x, sd["to_v.weight"], sd["to_v.bias"])
    attn = None
    for _ in range(num_iters):
        prev = slots
        s = layer_norm(slots, sd["norm_slot.weight"], sd["norm_slot.bias"], 1e-3)
        q = linear(s, sd["to_q.weight"], sd["to_q.bias"])
        dots = (q @ k.transpose(1, 2)) * scale                     # (B, K, N)
        attn = torch.softmax(dots, dim=1) + eps                    # softmax ACROSS SLOTS (:100)
        w = attn / attn.sum(dim=-1, keepdim=True)
        upd = w @ v                                                # (B, K, D)
        slots = gru_cell(upd, prev, sd["gru.weight_ih"], sd["gru.weight_hh"],
                         sd["gru.bias_ih"], sd["gru.bias_hh"])
        m = layer_norm(slots, sd["norm_mlp.weight"], sd["norm_mlp.bias"], 1e-3)
        m = torch.relu(linear(m, sd["mlp.0.weight"], sd["mlp.0.bias"]))
        slots = slots + linear(m, sd["mlp.2.weight"], sd["mlp.2.bias"])
    return (slots, attn) if return_attn else slots


def mhsa(sd, x, heads):
    """ models/Blocks/attention.py:245-265; q/k/v/out bias-free. """
    q, k, v = linear(x, sd["q.weight"]), linear(x, sd["k.weight"]), linear(x, sd["v.weight"])
    dh = x.shape[-1] // heads
    o = attention(q, k, v, heads, dh ** -0.5)
    return linear(o, sd["out_projection.0.weight"])


def transition_block(sd, x, heads):
    """ post-norm TransformerBlock, models/Blocks/attention.py:387-395 (LN eps 1e-6). """
    if not sd:
        return x                                                   # nn.Identity transition
    y = mhsa(sub(sd, "attn."), x, heads) + x
    y = layer_norm(y, sd["layernorm_query.weight"], sd["layernorm_query.bias"], 1e-6)
    z = linear(torch.relu(linear(y, sd["mlp.0.weight"], sd["mlp.0.bias"])),
               sd["mlp.2.weight"], sd["mlp.2.bias"]) + y
    return layer_norm(z, sd["layernorm_mlp.weight"], sd["layernorm_mlp.bias"], 1e-6)


def savi_decomp(sd, videos, noise, num_imgs, iters_first=3, iters=1, trans_heads=4):
    """
    models/SAVi.py:152-223 with decode=False: videos (B, L, C, H, W) -> slot_history (B, T, K, D).
    ``noise`` (B, K, D) stands for the torch.randn draw of initializers.py:87-94.
    slot_history stores the corrector output (pre-transition), SAVi.py:192-198,212.
    """
    if "initializer.slots_mu" in sd:
        pred = sd["initializer.slots_mu"] + sd["initializer.slots_sigma"] * noise
    else:                                                          # 'Learned' initializer :58-61
        pred = sd["initializer.slots"].expand(videos.shape[0], -1, -1)
    sa, tr = sub(sd, "slot_attention."), sub(sd, "transition_module.")
    hist = []
    for t in range(num_imgs):
        feats = savi_encode(sd, videos[:, t])
        slots = slot_attention(sa, feats, pred, iters_first if t == 0 else iters)
        pred = transition_block(tr, slots, trans_heads)
        hist.append(slots)
    return torch.stack(hist, dim=1)


# ------------------------------------------------------------------------------------------------
# SAVi decoder  (models/SAVi.py:241-275, EncodersDecoders/decoders.py:85-120)
# ------------------------------------------------------------------------------------------------

def savi_decode(sd, slots, resolution=(64, 64), in_channels=3):
    """ slots (B', K, D) -> recons_imgs (B',3,H,W), recons (B',K,3,H,W), masks (B',K,1,H,W) """
    Bp, K, D = slots.shape
    pos = soft_pos_embed(sd["decoder_pos_embedding.projection.weight"],
                         sd["decoder_pos_embedding.projection.bias"], resolution)  # (H, W, D)
    x = slots.reshape(Bp * K, 1, 1, D) + pos[None]                 # broadcast + pos (:264-275)
    x = x.permute(0, 3, 1, 2)
    i = 0
    while f"decoder.decoder.{i}.block.0.weight" in sd:
        w, b = sd[f"decoder.decoder.{i}.block.0.weight"], sd[f"decoder.decoder.{i}.block.0.bias"]
        x = torch.relu(F.conv2d(x, w, b, padding=w.shape[-1] // 2))
        i += 1
    w, b = sd[f"decoder.decoder.{i}.weight"], sd[f"decoder.decoder.{i}.bias"]
    y = F.conv2d(x, w, b, padding=1)                               # (B'K, 4, H, W)
    y = y.reshape(Bp, K, in_channels + 1, y.shape[2], y.shape[3])
    recons, alpha = y[:, :, :in_channels], y[:, :, in_channels:]
    masks = torch.softmax(alpha, dim=1)                            # softmax over slots (:254)
    return (recons * masks).sum(dim=1), recons, masks


def forward_eval_decomp(sd, videos, noise):
    """
    Decomposition-only evaluation, 03_evaluate_decomp_model.py:22-46: SAVi(x=videos, num_imgs=L) with the
    default mode "decomp" / decode=True (models/SAVi.py:139-223: every frame's corrector slots are decoded),
    then recons_imgs.clamp(0, 1).  Returns dict(recons_imgs (B, L, 3, H, W) unclamped, recons_objs
    (B, L, K, 3, H, W), masks (B, L, K, 1, H, W), slot_history (B, L, K, D), recons_clamped).
    """
    B, L = videos.shape[:2]
    hist = savi_decomp(sd, videos, noise, L)
    imgs, recons, masks = savi_decode(sd, hist.reshape(B * L, *hist.shape[2:]), tuple(videos.shape[-2:]))
    imgs = imgs.reshape(B, L, *imgs.shape[1:])
    return {"recons_imgs": imgs, "recons_objs": recons.reshape(B, L, *recons.shape[1:]),
            "masks": masks.reshape(B, L, *masks.shape[1:]), "slot_history": hist,
            "recons_clamped": imgs.clamp(0, 1)}


# ------------------------------------------------------------------------------------------------
# ExtendedDINOSAUR, downstream of the ViT backbone  (models/ExtendedDINOSAUR.py, decoders.py:203-365)
# ------------------------------------------------------------------------------------------------

def mlp_patch_decoder(sd, slots, img_size, patch_size=14, num_layers_cnn=4):
    """
    MLPPatchDecoder.forward (decoders.py:264-307) with ``sd`` = weights below 'decoder.':
    slots (B, K, D) -> recons_imgs (B,3,S,S), recons_feats (B,N,F), masks (B,K,1,g,g).
    CNN head schedule as decoders.py:325-365: ConvBlock(3x3, BatchNorm eval, ReLU) [+ nearest x2],
    final Conv3x3 -> RGB, bilinear resize (align_corners=False) to the image size.
    """
    B, K, D = slots.shape
    pos = sd["pos_embed"]                                          # (1, 1, N, D)
    N = pos.shape[2]
    g = int(N ** 0.5)
    x = slots[:, :, None, :] + pos                                 # broadcast + position (:264-266)
    i = 0
    if "mlp.0.weight" in sd and sd["mlp.0.weight"].dim() == 1:     # initial LayerNorm
        x = layer_norm(x, sd["mlp.0.weight"], sd["mlp.0.bias"], 1e-5)
        i = 1
    lin = [k for k in sorted({int(k.split(".")[1]) for k in sd if k.startswith("mlp.")}) if k >= i
           and sd[f"mlp.{k}.weight"].dim() == 2]
    for j, li in enumerate(lin):
        x = linear(x, sd[f"mlp.{li}.weight"], sd[f"mlp.{li}.bias"])
        if j < len(lin) - 1:
            x = torch.relu(x)
    feats, alpha = x[..., :-1], x[..., -1:]
    alpha = torch.softmax(alpha, dim=1)
    recons_feats = (feats * alpha).sum(dim=1)                      # (B, N, F)
    masks = alpha.reshape(B, K, 1, g, g)

    y = recons_feats.permute(0, 2, 1).reshape(B, -1, g, g)
    size, li = g, 0
    for i in range(num_layers_cnn):
        p = f"conv_patch_decoder.{li}.block."
        y = F.conv2d(y, sd[p + "0.weight"], sd[p + "0.bias"], padding=1)
        y = (y - sd[p + "1.running_mean"][None, :, None, None]) / torch.sqrt(
            sd[p + "1.running_var"][None, :, None, None] + 1e-5) * sd[p + "1.weight"][None, :, None, None] \
            + sd[p + "1.bias"][None, :, None, None]
        y = torch.relu(y)
        li += 1
        if (i + 1) * 2 < patch_size and size < img_size:
            y = F.interpolate(y, scale_factor=2, mode="nearest")
            size *= 2
            li += 1
    y = F.conv2d(y, sd[f"conv_patch_decoder.{li}.weight"], sd[f"conv_patch_decoder.{li}.bias"], padding=1)
    if y.shape[-1] != img_size:
        y = F.interpolate(y, size=(img_size, img_size), mode="bilinear", align_corners=False)
    return y, recons_feats, masks


# ------------------------------------------------------------------------------------------------
# ViT backbone of ExtendedDINOSAUR  (models/EncodersDecoders/timm_encoders.py:18-96)
#
# The arithmetic lives in the third-party package `timm` (un-pinned in the reference's environment.yml:24
# and absent from this image), and the reference holds no test vector for it.  PINNED since round 5 by
# `tests/golden/dinov2_vit.npz` / `e2e_c4.npz`: the reference's own ViTEncoder / ExtendedDINOSAUR code run in
# the build container with `transformers.Dinov2Model` -- an independent implementation of the same network,
# its state dict renamed key by key to timm's (`make_golden.py: HFDinov2AsTimm, hf_to_timm_key`) -- behind
# timm's attribute names (tests/test_oracle_golden.py::test_vit_oracle_vs_dinov2, ::test_e2e_c4_*).  This
# function restates timm's PUBLISHED VisionTransformer algorithm (timm/models/vision_transformer.py,
# 0.9.x: PatchEmbed = Conv2d(k = s = patch) + flatten; _pos_embed = cat(cls_token, x) + pos_embed;
# Block = x + ls1(attn(norm1(x))), x + ls2(mlp(norm2(x))) with LayerScale gamma, fused qkv Linear with
# bias, softmax(q k^T / sqrt(dh)) v, exact-erf GELU MLP; `vit_base_patch14_dinov2`: patch 14, 768 wide,
# 12 blocks, 12 heads, mlp_ratio 4, qkv_bias, LayerNorm eps 1e-6, init_values 1e-5) as called by the
# reference's wrapper: normalise, patch_embed, _pos_embed, patch_drop (identity), norm_pre (identity),
# blocks, drop the class token.  The backbone's final `norm` is NOT applied (timm_encoders.py:64-69).
# ------------------------------------------------------------------------------------------------

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # timm.data.IMAGENET_DEFAULT_MEAN = default_cfg["mean"] of the DINOv2 ViTs


def vit_encoder(sd, imgs, patch=14, heads=12, eps=1e-6, num_blocks=None, keep_cls=False):
    """
    imgs (n, 3, H, W) in [0, 1] -> patch features (n, (H/patch)*(W/patch), E).  ``sd``: timm
    VisionTransformer state_dict (keys below ``encoder.vit_backbone.``).
    The wrapper divides by the MEAN, not the standard deviation (timm_encoders.py:54-56 sets
    ``self.std`` from default_cfg["mean"]): reproduced.
    """
    mean = torch.tensor(IMAGENET_MEAN, dtype=imgs.dtype).view(1, 3, 1, 1)
    x = (imgs - mean) / mean                                               # :88-96, std := mean
    x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    x = x.flatten(2).transpose(1, 2)                                       # (n, N, E)
    n, N, E = x.shape
    x = torch.cat([sd["cls_token"].expand(n, -1, -1), x], dim=1) + sd["pos_embed"]
    i = 0
    while f"blocks.{i}.norm1.weight" in sd and (num_blocks is None or i < num_blocks):
        p = sub(sd, f"blocks.{i}.")
        y = layer_norm(x, p["norm1.weight"], p["norm1.bias"], eps)
        qkv = linear(y, p["attn.qkv.weight"], p["attn.qkv.bias"])
        a = attention(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], heads, (E // heads) ** -0.5)
        a = linear(a, p["attn.proj.weight"], p["attn.proj.bias"])
        x = x + (a * p["ls1.gamma"] if "ls1.gamma" in p else a)
        y = layer_norm(x, p["norm2.weight"], p["norm2.bias"], eps)
        y = linear(gelu(linear(y, p["mlp.fc1.weight"], p["mlp.fc1.bias"])), p["mlp.fc2.weight"], p["mlp.fc2.bias"])
        x = x + (y * p["ls2.gamma"] if "ls2.gamma" in p else y)
        i += 1
    return x if keep_cls else x[:, 1:]


def dinosaur_decomp(sd, feats, noise, iters_first=3, iters=1, trans_heads=4):
    """
    ExtendedDINOSAUR.forward_decomp (:139-208) downstream of the backbone: feats (B, T, N, Dm) patch
    features -> slot_history (B, T, K, D).  linear_feat_proj = LN -> Linear -> ReLU -> Linear (:96-101).
    """
    pred = sd["initializer.slots_mu"] + sd["initializer.slots_sigma"] * noise
    sa, tr = sub(sd, "slot_attention."), sub(sd, "transition_module.")
    hist = []
    for t in range(feats.shape[1]):
        z = layer_norm(feats[:, t], sd["linear_feat_proj.0.weight"], sd["linear_feat_proj.0.bias"], 1e-5)
        z = torch.relu(linear(z, sd["linear_feat_proj.1.weight"], sd["linear_feat_proj.1.bias"]))
        z = linear(z, sd["linear_feat_proj.3.weight"], sd["linear_feat_proj.3.bias"])
        slots = slot_attention(sa, z, pred, iters_first if t == 0 else iters)
        pred = transition_block(tr, slots, trans_heads)
        hist.append(slots)
    return torch.stack(hist, dim=1)


# ------------------------------------------------------------------------------------------------
# Text encoder  (models/EncodersDecoders/text_encoders.py:89-125)
# ------------------------------------------------------------------------------------------------

def text_encoder(sd, tokens, lengths, heads=4):
    """
    tokens (B, L) int64, lengths (B,) -> (B, L, out_dim).  Post-norm nn.TransformerEncoderLayer
    (GELU, eps 1e-5, key-padding mask) x num_layers; padded positions keep finite values and ARE
    consumed downstream (no mask in the predictor's cross-attention).
    """
    B, L = tokens.shape
    x = sd["token_embedding.weight"][tokens] + sd["position_embedding.weight"][:L][None]
    x = layer_norm(x, sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-8)
    x = x * (tokens != 0).unsqueeze(-1).to(x.dtype)
    key_pad = torch.arange(1, L + 1)[None, :] > lengths[:, None]   # (B, L) True = padding
    li = 0
    while f"transformer.layers.{li}.linear1.weight" in sd:
        p = sub(sd, f"transformer.layers.{li}.")
        E = x.shape[-1]
        qkv = linear(x, p["self_attn.in_proj_weight"], p["self_attn.in_proj_bias"])
        q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
        a = attention(q, k, v, heads, (E // heads) ** -0.5, key_mask=key_pad)
        a = linear(a, p["self_attn.out_proj.weight"], p["self_attn.out_proj.bias"])
        x = layer_norm(x + a, p["norm1.weight"], p["norm1.bias"], 1e-5)
        f = linear(gelu(linear(x, p["linear1.weight"], p["linear1.bias"])),
                   p["linear2.weight"], p["linear2.bias"])
        x = layer_norm(x + f, p["norm2.weight"], p["norm2.bias"], 1e-5)
        li += 1
    x = layer_norm(x, sd["text_out_projection.0.weight"], sd["text_out_projection.0.bias"], 1e-5)
    return linear(x, sd["text_out_projection.1.weight"], sd["text_out_projection.1.bias"])


# ------------------------------------------------------------------------------------------------
# T5-small encoder of TextOCVP_T5  (text_cond_OCVP.py:141-151 -> transformers.T5EncoderModel)
# ------------------------------------------------------------------------------------------------

def t5_encoder(sd, ids, mask, heads=8, num_buckets=32, max_distance=128, eps=1e-6):
    """
    Restatement of the published T5 encoder (un-pinned third-party dependency `transformers`,
    environment.yml:22; call site predictor_wrapper.py:101-111): shared embedding, per block
    RMS-norm -> bias-free q/k/v (NO 1/sqrt(d) scaling) + bucketed relative position bias (block 0's
    table, shared by all blocks) + key padding mask -> o-proj residual; RMS-norm -> ReLU FFN
    residual; final RMS-norm.  ``sd`` = weights below 'text_encoder.'.
    """
    def rms(x, w):
        return x * torch.rsqrt((x * x).mean(dim=-1, keepdim=True) + eps) * w
    B, L = ids.shape
    h = sd["encoder.embed_tokens.weight"][ids]
    pos = torch.arange(L)
    rel = pos[None, :] - pos[:, None]
    nb = num_buckets // 2
    bucket = (rel > 0).long() * nb
    a = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(a.float().clamp(min=1) / max_exact)
                         / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    bucket = bucket + torch.where(a < max_exact, a, large)
    bias = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"][bucket]
    bias = bias.permute(2, 0, 1)[None]                              # (1, H, L, L)
    neg = (1.0 - mask[:, None, None, :].to(h.dtype)) * torch.finfo(h.dtype).min
    li = 0
    while f"encoder.block.{li}.layer.0.SelfAttention.q.weight" in sd:
        p = f"encoder.block.{li}.layer."
        n = rms(h, sd[p + "0.layer_norm.weight"])
        E = n.shape[-1]
        dh = E // heads
        def split(t):
            return t.reshape(B, L, heads, dh).permute(0, 2, 1, 3)
        q = split(linear(n, sd[p + "0.SelfAttention.q.weight"]))
        k = split(linear(n, sd[p + "0.SelfAttention.k.weight"]))
        v = split(linear(n, sd[p + "0.SelfAttention.v.weight"]))
        w = torch.softmax(q @ k.transpose(-1, -2) + bias + neg, dim=-1)
        ctx = (w @ v).permute(0, 2, 1, 3).reshape(B, L, E)
        h = h + linear(ctx, sd[p + "0.SelfAttention.o.weight"])
        n = rms(h, sd[p + "1.layer_norm.weight"])
        h = h + linear(torch.relu(linear(n, sd[p + "1.DenseReluDense.wi.weight"])),
                       sd[p + "1.DenseReluDense.wo.weight"])
        li += 1
    return rms(h, sd["encoder.final_layer_norm.weight"])


# ------------------------------------------------------------------------------------------------
# TextOCVP predictor  (models/Predictors/text_cond_OCVP.py:79-105, predictor_wrapper.py:50-87)
# ------------------------------------------------------------------------------------------------

def adapted_block(sd, x, text, heads=8, cross_heads=8, cross_dh=64):
    """
    AdaptedEncoderBlock.forward, models/Blocks/attention.py:504-524 (+ :445-463, :303-319).
    NOTE the final residual is taken from y (post self-attention), not from the cross-attention
    branch output.
    """
    y = x + mhsa(sub(sd, "attn."), layer_norm(x, sd["layernorm_query.weight"],
                                              sd["layernorm_query.bias"], 1e-6), heads)
    c = sub(sd, "cross_attention.")
    qn = layer_norm(y, c["ln_cross_att_q.weight"], c["ln_cross_att_q.bias"], 1e-6)
    kv = layer_norm(text, c["ln_cross_att_kv.weight"], c["ln_cross_att_kv.bias"], 1e-6)
    q = linear(qn, c["cross_attn.q.weight"])
    k = linear(kv, c["cross_attn.k.weight"])
    v = linear(kv, c["cross_attn.v.weight"])
    a = attention(q, k, v, cross_heads, cross_dh ** -0.5)          # no key-padding mask (:314)
    z = linear(a, c["cross_attn.out_projection.weight"], c["cross_attn.out_projection.bias"]) + y
    m = layer_norm(z, c["ln_mlp.weight"], c["ln_mlp.bias"], 1e-6)
    z = linear(torch.relu(linear(m, c["mlp.0.weight"], c["mlp.0.bias"])),
               c["mlp.2.weight"], c["mlp.2.bias"]) + z
    m = layer_norm(z, sd["layernorm_mlp.weight"], sd["layernorm_mlp.bias"], 1e-6)
    return linear(torch.relu(linear(m, sd["mlp.0.weight"], sd["mlp.0.bias"])),
                  sd["mlp.2.weight"], sd["mlp.2.bias"]) + y


def text_ocvp_step(sd, window, text, residual=True):
    """
    BaseTextOCVP.forward (text_cond_OCVP.py:79-105) with ``sd`` = weights below 'predictor.'
    of the PredictorWrapper.  window (B, w, K, D) -> next slots (B, K, D).
    Temporal PE is FLIPPED: newest frame gets pe[0] (model_blocks.py:375-377).
    """
    B, w, K, D = window.shape
    tok = linear(window, sd["mlp_in.weight"], sd["mlp_in.bias"])   # (B, w, K, E)
    pe = sd["pe.pe"][0, :w, 0]                                     # (w, E)
    tok = tok + torch.flip(pe, dims=(0,))[None, :, None, :]
    tok = tok.reshape(B, w * K, -1)
    li = 0
    while f"predictor.{li}.mlp.0.weight" in sd:
        tok = adapted_block(sub(sd, f"predictor.{li}."), tok, text)
        li += 1
    last = tok.reshape(B, w, K, -1)[:, -1]
    out = linear(last, sd["mlp_out.weight"], sd["mlp_out.bias"])
    return out + window[:, -1] if residual else out


# ------------------------------------------------------------------------------------------------
# Unconditioned predictors  (models/Predictors/OCVP.py)
# ------------------------------------------------------------------------------------------------

def sinusoid_pe(max_len, d_model):
    """ SlotPositionalEncoding table, models/Blocks/model_blocks.py:258-266 (NOT flipped, :288) """
    pos = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def encoder_layer_prenorm(p, x, heads):
    """ nn.TransformerEncoderLayer(norm_first=True, batch_first=True, relu, eps 1e-5) in eval mode """
    E = x.shape[-1]
    h = layer_norm(x, p["norm1.weight"], p["norm1.bias"], 1e-5)
    qkv = linear(h, p["self_attn.in_proj_weight"], p["self_attn.in_proj_bias"])
    a = attention(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], heads, (E // heads) ** -0.5)
    x = x + linear(a, p["self_attn.out_proj.weight"], p["self_attn.out_proj.bias"])
    h = layer_norm(x, p["norm2.weight"], p["norm2.bias"], 1e-5)
    return x + linear(torch.relu(linear(h, p["linear1.weight"], p["linear1.bias"])),
                      p["linear2.weight"], p["linear2.bias"])


def uncond_step(sd, window, kind, heads=4, buffer_size=10, residual=True):
    """
    VanillaTransformerPredictor.forward (OCVP.py:100-132) / OCVPSeq.forward (:222-254) with
    OCVPSeqLayer.forward (:301-320); ``sd`` = weights below 'predictor.'.
    """
    B, w, K, D = window.shape
    tok = linear(window, sd["mlp_in.weight"], sd["mlp_in.bias"])
    E = tok.shape[-1]
    tok = tok + sinusoid_pe(buffer_size, E)[:w][None, :, None, :]
    li = 0
    if kind == "VanillaTransformer":
        x = tok.reshape(B, w * K, E)
        while f"transformer_encoders.{li}.linear1.weight" in sd:
            x = encoder_layer_prenorm(sub(sd, f"transformer_encoders.{li}."), x, heads)
            li += 1
        last = x.reshape(B, w, K, E)[:, -1]
    elif kind == "OCVPSeq":
        x = tok
        while f"transformer_encoders.{li}.object_encoder_block.linear1.weight" in sd:
            p = sub(sd, f"transformer_encoders.{li}.")
            x = encoder_layer_prenorm(sub(p, "object_encoder_block."), x.reshape(B * w, K, E), heads)
            x = x.reshape(B, w, K, E).transpose(1, 2).reshape(B * K, w, E)
            x = encoder_layer_prenorm(sub(p, "time_encoder_block."), x, heads)
            x = x.reshape(B, K, w, E).transpose(1, 2)
            li += 1
        last = x[:, -1]
    else:
        raise ValueError(kind)
    out = linear(last, sd["mlp_out.weight"], sd["mlp_out.bias"])
    return out + window[:, -1] if residual else out


def rollout(sd, slot_history, tokens, lengths, num_context, num_preds, buffer_size=10,
            teacher_force=False, kind="TextOCVP_CustomTF", attn_masks=None):
    """ PredictorWrapper.forward, predictor_wrapper.py:50-87 (sd keys start with 'predictor.'). """
    p = sub(sd, "predictor.")
    if kind == "TextOCVP_CustomTF":
        text = text_encoder(sub(p, "text_encoder."), tokens, lengths)
    elif kind == "TextOCVP_T5":                                    # predictor_wrapper.py:101-111 (token_dim == 512)
        text = t5_encoder(sub(p, "text_encoder."), tokens, attn_masks)
    else:
        text = None
    window = slot_history[:, :num_context].clone()
    preds = []
    for t in range(num_preds):
        cur = text_ocvp_step(p, window, text) if text is not None else \
            uncond_step(p, window, kind, buffer_size=buffer_size)
        nxt = slot_history[:, num_context + t] if teacher_force else cur
        window = torch.cat([window, nxt.unsqueeze(1)], dim=1)
        if window.shape[1] > buffer_size:                          # :143-153
            window = window[:, window.shape[1] - buffer_size:]
        preds.append(cur)
    return torch.stack(preds, dim=1)


def forward_eval(savi_sd, pred_sd, videos, tokens, lengths, noise, num_context, num_preds,
                 buffer_size=10, return_decode=False):
    """
    The three calls of Evaluator.forward_eval (05_evaluate_predictor.py:82-96) on CPU.
    Returns slot_history (B,T,K,D), pred_slots (B,P,K,D), pred_imgs (B,P,C,H,W) clamped to [0,1],
    masks (B*P,K,1,H,W); with return_decode also {"recons_imgs" (unclamped), "recons"} of SAVi.decode.
    """
    B, L, C, H, W = videos.shape
    hist = savi_decomp(savi_sd, videos, noise, num_context + num_preds)
    preds = rollout(pred_sd, hist, tokens, lengths, num_context, num_preds, buffer_size)
    K, D = preds.shape[2:]
    imgs, recons, masks = savi_decode(savi_sd, preds.reshape(B * num_preds, K, D), (H, W), C)
    out = (hist, preds, imgs.view(B, num_preds, C, H, W).clamp(0, 1), masks)
    if return_decode:
        out = out + ({"recons_imgs": imgs, "recons": recons},)
    return out


def forward_eval_dinosaur(dino_sd, pred_sd, videos, ids, attn_masks, noise, num_context, num_preds,
                          buffer_size=10, kind="TextOCVP_T5", lengths=None):
    """
    Evaluator.forward_eval (05_evaluate_predictor.py:82-96) on BASELINE configs[3], from PIXELS:
    ExtendedDINOSAUR.forward_decomp(decode=False) (models/ExtendedDINOSAUR.py:139-208: ViT backbone per frame,
    linear_feat_proj, slot attention, transition) -> PredictorWrapper(TextOCVP_T5).forward
    (predictor_wrapper.py:50-87, 101-111) -> ExtendedDINOSAUR.decode (:211-214 -> MLPPatchDecoder,
    decoders.py:264-365) -> view + clamp.  ``dino_sd`` / ``pred_sd``: reference-layout weight dicts (the
    backbone under timm's names below 'encoder.vit_backbone.').
    Returns a dict: encoded_img_feats (B,T,N,768), slot_history (B,T,K,D), pred_slots (B,P,K,D), recons_imgs
    (B*P,3,S,S) unclamped, pred_imgs (B,P,3,S,S) clamped, recons_feats (B*P,N,F), masks (B*P,K,1,g,g).
    """
    B, L, C, H, W = videos.shape
    T = num_context + num_preds
    vit = sub(dino_sd, "encoder.vit_backbone.")
    feats = torch.stack([vit_encoder(vit, videos[:, t]) for t in range(T)], dim=1)
    hist = dinosaur_decomp(dino_sd, feats, noise)
    preds = rollout(pred_sd, hist, ids, lengths, num_context, num_preds, buffer_size, kind=kind,
                    attn_masks=attn_masks)
    K, D = preds.shape[2:]
    imgs, recons_feats, masks = mlp_patch_decoder(sub(dino_sd, "decoder."), preds.reshape(B * num_preds, K, D),
                                                  img_size=H)
    return {"encoded_img_feats": feats, "slot_history": hist, "pred_slots": preds, "recons_imgs": imgs,
            "pred_imgs": imgs.view(B, num_preds, C, H, W).clamp(0, 1), "recons_feats": recons_feats, "masks": masks}
