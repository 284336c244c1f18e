"""
Unconditioned object-centric predictors on the MI355X kernels.
Reference: models/Predictors/OCVP.py (VanillaTransformerPredictor :24-141, OCVPSeq :145-263,
OCVPSeqLayer :267-320).  Both are stacks of pre-norm ``nn.TransformerEncoderLayer`` (ReLU, LayerNorm
eps 1e-5, batch_first) with an un-flipped sinusoidal temporal encoding shared by the slots of a
frame (models/Blocks/model_blocks.py:230-290).

``nn.TransformerEncoderLayer`` objects are parameter containers only (they give the reference's
state_dict keys: self_attn.in_proj_weight, linear1, norm1, ...); the arithmetic runs on the HIP kernels.
"""

import math

import torch
import torch.nn as nn

from ... import kernels as K
from ..Blocks.model_utils import require_inference, tracks_structure

__all__ = ["VanillaTransformerPredictor", "OCVPSeq", "OCVPSeqLayer"]


def _sinusoid_table(max_len, d_model):
    """ (max_len, d_model) table of SlotPositionalEncoding (model_blocks.py:258-266) """
    pos = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def _encoder_layer(layer, x, heads):
    """ pre-norm encoder layer on (B', T, E): x + SA(LN1 x); then + FF(LN2 .) """
    E = x.shape[-1]
    sa = layer.self_attn
    h = K.layer_norm(x, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
    qkv = K.linear(h, sa.in_proj_weight, sa.in_proj_bias)
    a = K.mha(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], heads, (E // heads) ** -0.5)
    x = K.linear(a, sa.out_proj.weight, sa.out_proj.bias, residual=x)
    h = K.layer_norm(x, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
    h = K.linear(h, layer.linear1.weight, layer.linear1.bias, act=K.ACT_RELU)
    return K.linear(h, layer.linear2.weight, layer.linear2.bias, residual=x)


def _make_layer(token_dim, n_heads, hidden_dim):
    return nn.TransformerEncoderLayer(d_model=token_dim, nhead=n_heads, batch_first=True,
                                      norm_first=True, dim_feedforward=hidden_dim)


@tracks_structure
class _SlotPredictorBase(nn.Module):
    def __init__(self, num_slots, slot_dim, token_dim, hidden_dim, num_layers, n_heads, residual,
                 input_buffer_size):
        super().__init__()
        self.num_slots, self.slot_dim, self.token_dim = num_slots, slot_dim, token_dim
        self.hidden_dim, self.num_layers, self.nhead = hidden_dim, num_layers, n_heads
        self.residual, self.input_buffer_size = residual, input_buffer_size
        self.mlp_in = nn.Linear(slot_dim, token_dim)
        self.mlp_out = nn.Linear(token_dim, slot_dim)
        self._pe = _sinusoid_table(input_buffer_size, token_dim)     # plain attribute, like the reference

    def _tokens(self, slots):
        """ mlp_in + temporal encoding (fused as a row-vector epilogue, NOT flipped here) """
        B, w, Ks, _ = slots.shape
        if w > self._pe.shape[0]:
            raise ValueError(f"window of {w} frames exceeds input_buffer_size={self._pe.shape[0]}")
        if self._pe.device != slots.device:
            self._pe = self._pe.to(slots.device)
        return K.linear(slots, self.mlp_in.weight, self.mlp_in.bias,
                        rowvec=self._pe[:w].contiguous(), rv_div=Ks, rv_flip=False)

    def _head(self, last_tokens, slots):
        return K.linear(last_tokens.contiguous(), self.mlp_out.weight, self.mlp_out.bias,
                        residual=slots[:, -1].contiguous() if self.residual else None)


class VanillaTransformerPredictor(_SlotPredictorBase):
    """ joint attention over all (frame, slot) tokens of the window (OCVP.py:24-141) """

    def __init__(self, num_slots, slot_dim, token_dim=128, hidden_dim=256, num_layers=2, n_heads=4,
                 residual=False, input_buffer_size=5):
        super().__init__(num_slots, slot_dim, token_dim, hidden_dim, num_layers, n_heads, residual,
                         input_buffer_size)
        self.transformer_encoders = nn.Sequential(
            *[_make_layer(token_dim, n_heads, hidden_dim) for _ in range(num_layers)])

    def forward(self, slots, **kwargs):
        """ slots (B, w, K, D) -> next slots (B, K, D) """
        require_inference(self)
        B, w, Ks, _ = slots.shape
        slots = slots.contiguous()
        x = self._tokens(slots).reshape(B, w * Ks, self.token_dim)
        for layer in self.transformer_encoders:
            x = _encoder_layer(layer, x, self.nhead)
        return self._head(x.reshape(B, w, Ks, self.token_dim)[:, -1], slots)


@tracks_structure
class OCVPSeqLayer(nn.Module):
    """ object attention within a frame, then time attention per slot (OCVP.py:267-320) """

    def __init__(self, token_dim=128, hidden_dim=256, n_heads=4):
        super().__init__()
        self.token_dim, self.hidden_dim, self.nhead = token_dim, hidden_dim, n_heads
        self.object_encoder_block = _make_layer(token_dim, n_heads, hidden_dim)
        self.time_encoder_block = _make_layer(token_dim, n_heads, hidden_dim)

    def forward(self, inputs, time_mask=None):
        """ inputs (B, w, K, E) -> same shape """
        B, w, Ks, E = inputs.shape
        x = _encoder_layer(self.object_encoder_block, inputs.reshape(B * w, Ks, E), self.nhead)
        x = x.reshape(B, w, Ks, E).transpose(1, 2).reshape(B * Ks, w, E)        # data movement only
        x = _encoder_layer(self.time_encoder_block, x, self.nhead)
        return x.reshape(B, Ks, w, E).transpose(1, 2).contiguous()


class OCVPSeq(_SlotPredictorBase):
    """ decoupled object / time attention applied sequentially (OCVP.py:145-263) """

    def __init__(self, num_slots, slot_dim, token_dim=128, hidden_dim=256, num_layers=2, n_heads=4,
                 residual=False, input_buffer_size=5):
        super().__init__(num_slots, slot_dim, token_dim, hidden_dim, num_layers, n_heads, residual,
                         input_buffer_size)
        self.transformer_encoders = nn.Sequential(
            *[OCVPSeqLayer(token_dim=token_dim, hidden_dim=hidden_dim, n_heads=n_heads)
              for _ in range(num_layers)])

    def forward(self, slots, **kwargs):
        require_inference(self)
        slots = slots.contiguous()
        x = self._tokens(slots)
        for layer in self.transformer_encoders:
            x = layer(x)
        return self._head(x[:, -1], slots)
