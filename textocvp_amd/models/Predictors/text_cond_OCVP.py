"""
Text-conditioned object-centric predictor.  Reference: models/Predictors/text_cond_OCVP.py
(BaseTextOCVP :21-119, forward :79-105; TextOCVP_CustomTF :123-137).
"""

import os

import torch.nn as nn

from ... import kernels as K
from ..Blocks.attention import AdaptedEncoderBlock
from ..Blocks.model_blocks import TemporalPositionalEncoding
from ..Blocks.model_utils import require_inference
from ..EncodersDecoders.text_encoders import TransformerTextEncoder

__all__ = ["TextOCVP_CustomTF"]


class BaseTextOCVP(nn.Module):
    """
    slots of the input window -> ``mlp_in`` (+ flipped learned temporal PE fused in the GEMM
    epilogue) -> N x AdaptedEncoderBlock -> ``mlp_out`` on the LAST frame's tokens (+ residual).

    Step-invariant work is hoisted: per layer, LayerNorm(text) and the fused K/V projection of the
    cross-attention are computed once per caption batch (``prepare_text``) and reused by all
    rollout steps.  Nothing else can be cached across steps because the flipped PE changes every
    token's input at every step.
    """

    def __init__(self, slot_dim, predictor_params, fusion_params, text_encoder_params):
        super().__init__()
        self.predictor_params = predictor_params
        self.fusion_params = fusion_params
        self.text_encoder_params = text_encoder_params
        self.slot_dim = slot_dim
        self.token_dim = predictor_params.get("token_dim")
        self.num_heads = predictor_params.get("n_heads")
        self.hidden_dim = predictor_params.get("hidden_dim")
        self.num_layers = predictor_params.get("num_layers")
        self.residual = predictor_params.get("residual")
        self.input_buffer_size = predictor_params.get("input_buffer_size")

        self.mlp_in = nn.Linear(self.slot_dim, self.token_dim)
        self.mlp_out = nn.Linear(self.token_dim, self.slot_dim)
        self.predictor = nn.ModuleList([
            AdaptedEncoderBlock(embed_dim=self.token_dim, num_heads=self.num_heads,
                                mlp_size=self.hidden_dim, fusion_params=self.fusion_params)
            for _ in range(self.num_layers)])
        self._instantiate_text_encoder()
        self.pe = TemporalPositionalEncoding(d_model=self.token_dim,
                                             max_len=self.input_buffer_size + 1, mode="learned")
        self._text_cache = None
        # arithmetic of the predictor GEMMs: "fp32" | "bf16x3" | "bf16x6" (kernels.gemm_precision)
        self.gemm_precision = os.environ.get("TOCVP_PREDICTOR_PRECISION", "bf16x6")

    def _instantiate_text_encoder(self):
        raise NotImplementedError("'BaseTextOCVP' does not implement '_instantiate_text_encoder'...")

    def prepare_text(self, text_embeddings):
        """ per-layer fused cross-attention K/V of the caption (cached on tensor identity) """
        c = self._text_cache
        if c is not None and c[0] is text_embeddings and c[1] == text_embeddings._version:
            return c[2]
        with K.gemm_precision(self.gemm_precision):
            kv = [blk.cross_attention.project_text(text_embeddings) for blk in self.predictor]
        self._text_cache = (text_embeddings, text_embeddings._version, kv)
        return kv

    def forward(self, slots, text_embeddings, **kwargs):
        """ slots (B, w, K, D) window, text_embeddings (B, Lt, E) -> next slots (B, K, D) """
        require_inference(self)
        B, w, Ks, D = slots.shape
        slots = slots.contiguous()
        text_kv = self.prepare_text(text_embeddings)
        with K.gemm_precision(self.gemm_precision):
            tokens = K.linear(slots, self.mlp_in.weight, self.mlp_in.bias,
                              rowvec=self.pe.rows(w, slots.device), rv_div=Ks, rv_flip=True)
            tokens = tokens.reshape(B, w * Ks, self.token_dim)
            for blk, kv in zip(self.predictor, text_kv):
                tokens = blk(tokens, text_embeddings, text_kv=kv)
            last = tokens.reshape(B, w, Ks, self.token_dim)[:, -1].contiguous()
            return K.linear(last, self.mlp_out.weight, self.mlp_out.bias,
                            residual=slots[:, -1].contiguous() if self.residual else None)


class TextOCVP_CustomTF(BaseTextOCVP):
    """ TextOCVP with the small custom transformer text encoder (text_cond_OCVP.py:123-137). """

    def _instantiate_text_encoder(self):
        p = self.text_encoder_params
        self.text_encoder = TransformerTextEncoder(
            input_dim=p.get("input_dim"), num_layers=p.get("num_layers"),
            num_heads=p.get("num_heads"), output_dim=self.token_dim, vocab_size=p.get("vocab_size"))
