"""
Text-conditioned object-centric predictor.  Reference: models/Predictors/text_cond_OCVP.py
(BaseTextOCVP :21-119, forward :79-105; TextOCVP_CustomTF :123-137).
"""

import math
import os

import torch
import torch.nn as nn

from ... import kernels as K
from ...precision import knob
from ..Blocks.attention import AdaptedEncoderBlock
from ..Blocks.model_blocks import TemporalPositionalEncoding
from ..Blocks.model_utils import require_inference, tracks_structure
from ..EncodersDecoders.text_encoders import TransformerTextEncoder

__all__ = ["TextOCVP_CustomTF", "TextOCVP_T5", "T5TextEncoder"]


@tracks_structure
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
        # arithmetic of the predictor GEMMs: "fp32" | "bf16x3" | "bf16x6" | "f16x3" (kernels.gemm_precision)
        # f16x3 = two fp16 planes per operand, 3 matrix-core products, fp32-class for |activation| < 255
        self.gemm_precision = knob("TOCVP_PREDICTOR_PRECISION", "f16x3")
        self.last_layer_newest_frame_only = os.environ.get("TOCVP_LAST_LAYER_SUBSET", "1") != "0"

    range_fallbacks = {"gemm_precision": {"f16x3": "bf16x6"}}

    def _instantiate_text_encoder(self):
        raise NotImplementedError("'BaseTextOCVP' does not implement '_instantiate_text_encoder'...")

    def _text_kv_sources(self):
        """ every parameter the cached caption K/V depend on: LayerNorm(text) and the k / v projections per layer """
        src = []
        for blk in self.predictor:
            ca = blk.cross_attention
            src += [ca.ln_cross_att_kv.weight, ca.ln_cross_att_kv.bias, ca.cross_attn.k.weight, ca.cross_attn.v.weight,
                    ca.cross_attn.q.weight, ca.cross_attn.out_projection.weight]      # collapsed operands (xattn.hip)
        return src

    def prepare_text(self, text_embeddings):
        """
        per-layer fused cross-attention K/V of the caption.  Cached on the identity + version of the text
        tensor AND on the (data_ptr, version, device) of every weight that enters them, plus the GEMM
        arithmetic: ``load_state_dict`` / ``.to()`` / an optimiser step / a precision fallback between two calls
        with the same text tensor rebuild the cache instead of serving stale K/V.
        """
        sig = (text_embeddings._version, text_embeddings.data_ptr(), self.gemm_precision,
               tuple((t.data_ptr(), t._version, t.device.index) for t in self._text_kv_sources()))
        c = self._text_cache
        if c is not None and c[0] is text_embeddings and c[1] == sig:
            return c[2]
        with K.gemm_precision(self.gemm_precision, owner=(self, "gemm_precision")):
            kv = [blk.cross_attention.project_text(text_embeddings) for blk in self.predictor]
        self._text_cache = (text_embeddings, sig, kv)
        return kv

    def forward(self, slots, text_embeddings, **kwargs):
        """ slots (B, w, K, D) window, text_embeddings (B, Lt, E) -> next slots (B, K, D) """
        require_inference(self)
        B, w, Ks, D = slots.shape
        slots = K.contiguous(slots)
        text_kv = self.prepare_text(text_embeddings)
        with K.gemm_precision(self.gemm_precision, owner=(self, "gemm_precision")):
            tokens = K.linear(slots, self.mlp_in.weight, self.mlp_in.bias,
                              rowvec=self.pe.rows(w, slots.device), rv_div=Ks, rv_flip=True)
            tokens = tokens.reshape(B, w * Ks, self.token_dim)
            nblk = len(self.predictor)
            for i, (blk, kv) in enumerate(zip(self.predictor, text_kv)):
                if i == nblk - 1 and self.last_layer_newest_frame_only:
                    # only tokens[:, -1] (the newest frame) is read below: the final layer computes
                    # just those K rows (keys / values still span the whole window)
                    last = blk.forward_last(tokens, text_embeddings, Ks, text_kv=kv)
                else:
                    tokens = blk(tokens, text_embeddings, text_kv=kv)
            if not self.last_layer_newest_frame_only or nblk == 0:
                last = K.contiguous(tokens.reshape(B, w, Ks, self.token_dim)[:, -1])
            return K.linear(last, self.mlp_out.weight, self.mlp_out.bias,
                            residual=K.contiguous(slots[:, -1]) if self.residual else None)


class TextOCVP_CustomTF(BaseTextOCVP):
    """ TextOCVP with the small custom transformer text encoder (text_cond_OCVP.py:123-137). """

    def _instantiate_text_encoder(self):
        p = self.text_encoder_params
        self.text_encoder = TransformerTextEncoder(
            input_dim=p.get("input_dim"), num_layers=p.get("num_layers"),
            num_heads=p.get("num_heads"), output_dim=self.token_dim, vocab_size=p.get("vocab_size"))


def _t5_relative_buckets(L, num_buckets, max_distance, device):
    """ bidirectional T5 bucket index of (key - query) for an L x L grid (integer bookkeeping) """
    pos = torch.arange(L, device=device)
    rel = pos[None, :] - pos[:, None]                       # key - query
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float().clamp(min=1) / max_exact)
                         / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < max_exact, rel, large)


class T5TextEncoder(nn.Module):
    """
    Encoder of ``t5-small`` as used by TextOCVP_T5 (reference text_cond_OCVP.py:141-151 calls
    ``T5EncoderModel.from_pretrained("t5-small")``, predictor_wrapper.py:101-111 reads
    ``last_hidden_state``).  A randomly initialised ``transformers.T5EncoderModel`` is the PARAMETER
    CONTAINER (identical state_dict keys, so the pretrained / fine-tuned weights of a reference
    checkpoint load unchanged; nothing is downloaded); the arithmetic runs on the HIP kernels:
    embedding gather, RMS LayerNorm, bias-free projections, attention with T5's bucketed
    relative-position bias (scale 1, key-padding from the attention mask), ReLU feed-forward.
    """

    def __init__(self):
        super().__init__()
        from transformers import T5Config, T5EncoderModel
        cfg = T5Config(vocab_size=32128, d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_heads=8,
                       relative_attention_num_buckets=32, relative_attention_max_distance=128,
                       dropout_rate=0.1, layer_norm_epsilon=1e-6, feed_forward_proj="relu")
        self.t5 = T5EncoderModel(cfg)
        self.cfg = cfg

    def forward(self, input_ids, attention_mask):
        """ ids (B, L) int64, mask (B, L) {0,1} -> last_hidden_state (B, L, 512) """
        enc = self.t5.encoder
        dev = enc.embed_tokens.weight.device
        ids = input_ids.to(dev)
        B, L = ids.shape
        key_len = attention_mask.to(dev).sum(dim=1).to(torch.int32).contiguous()
        H, E = self.cfg.num_heads, self.cfg.d_model
        rel = enc.block[0].layer[0].SelfAttention.relative_attention_bias.weight     # (buckets, H)
        buckets = _t5_relative_buckets(L, self.cfg.relative_attention_num_buckets,
                                       self.cfg.relative_attention_max_distance, dev)
        bias = rel.detach()[buckets].permute(2, 0, 1).contiguous()                   # (H, L, L) gather
        eps = self.cfg.layer_norm_epsilon
        h = K.embedding(ids, enc.embed_tokens.weight)
        for blk in enc.block:
            att, ff = blk.layer[0], blk.layer[1]
            sa = att.SelfAttention
            n = K.rms_norm(h, att.layer_norm.weight, eps)
            q, k_, v = K.linear(n, sa.q.weight), K.linear(n, sa.k.weight), K.linear(n, sa.v.weight)
            ctx = K.mha(q, k_, v, H, 1.0, key_len=key_len, bias=bias)
            h = K.linear(ctx, sa.o.weight, residual=h)
            n = K.rms_norm(h, ff.layer_norm.weight, eps)
            mid = K.linear(n, ff.DenseReluDense.wi.weight, act=K.ACT_RELU)
            h = K.linear(mid, ff.DenseReluDense.wo.weight, residual=h)
        return K.rms_norm(h, enc.final_layer_norm.weight, eps)


class TextOCVP_T5(BaseTextOCVP):
    """ TextOCVP with the (frozen) T5-small text encoder (text_cond_OCVP.py:141-151). """

    def _instantiate_text_encoder(self):
        self.text_encoder = T5TextEncoderAdapter()
        self.t5_token_dim = 512


class T5TextEncoderAdapter(T5TextEncoder):
    """
    Exposes the HF module tree directly under ``text_encoder`` so the state_dict keys read
    ``text_encoder.shared.weight`` / ``text_encoder.encoder.block...`` exactly like the reference,
    where ``text_encoder`` IS the T5EncoderModel.
    """

    def __init__(self):
        nn.Module.__init__(self)
        from transformers import T5Config, T5EncoderModel
        cfg = T5Config(vocab_size=32128, d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_heads=8,
                       relative_attention_num_buckets=32, relative_attention_max_distance=128,
                       dropout_rate=0.1, layer_norm_epsilon=1e-6, feed_forward_proj="relu")
        hf = T5EncoderModel(cfg)
        self.cfg = cfg
        self.shared = hf.shared
        self.encoder = hf.encoder
        for p_ in self.parameters():
            p_.requires_grad_(False)                                   # freeze_params (:149)

    @property
    def t5(self):
        return self
