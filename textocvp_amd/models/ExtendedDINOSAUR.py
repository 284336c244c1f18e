"""
ExtendedDINOSAUR video decomposition model on the MI355X kernels.
Reference: models/ExtendedDINOSAUR.py (forward_decomp :139-208, decode :211-214).

The frozen DINOv2 ViT backbone (models/EncodersDecoders/timm_encoders.py, third-party timm arithmetic:
restated from timm's published algorithm; pinned since round 5 by `tests/golden/dinov2_vit.npz`: the
reference's wrapper around `transformers.Dinov2Model`, an independent implementation of the same
network) runs on the same GEMM / attention / LayerNorm kernels as the predictor.  ``forward_decomp`` batches everything that does not depend on the slots over
all frames (backbone, feature projection, k/v projection); only the slot recurrence is sequential.
The extension kwarg ``encoded_img_feats`` (B, T, N, mlp_encoder_dim) bypasses the backbone (features
computed elsewhere, e.g. cached across predictor experiments).
"""

import math

import torch
import torch.nn as nn

from .. import kernels as K
from .Blocks.attention import SlotAttention
from .Blocks.initializers import get_initializer
from .Blocks.model_utils import RangeGuard, init_xavier_, refuse_replication, require_inference, tracks_structure
from .Blocks.transition_models import get_transition_module
from .EncodersDecoders.decoders import get_decoder
from .EncodersDecoders.encoders import get_encoder

__all__ = ["ExtendedDINOSAUR"]


@tracks_structure
class ExtendedDINOSAUR(nn.Module, RangeGuard):
    _replicate_for_data_parallel = refuse_replication      # one process per GPU, never DataParallel replicas

    def __init__(self, img_size, num_slots, slot_dim, num_iterations=1, num_iterations_first=3,
                 in_channels=3, mlp_hidden=128, mlp_encoder_dim=128, initializer=None, encoder=None,
                 decoder=None, transition_module=None, **kwargs):
        super().__init__()
        self.img_size, self.num_slots, self.slot_dim = img_size, num_slots, slot_dim
        self.num_iterations_first, self.num_iterations = num_iterations_first, num_iterations
        self.in_channels, self.mlp_hidden, self.mlp_encoder_dim = in_channels, mlp_hidden, mlp_encoder_dim
        self.initializer = get_initializer(mode=initializer, slot_dim=slot_dim, num_slots=num_slots)
        self.transition_module = get_transition_module(slot_dim=slot_dim, **transition_module)
        if self.img_size is None:
            raise KeyError("'img_size' must be provided in model parameters in order to "
                           "instanciate ViT-based image encoder.")
        if encoder is None or "vit" not in encoder["encoder_name"]:
            raise NameError("Extended-DINOSAUR expects a ViT-Based encoder...")
        encoder = {"encoder_name": encoder["encoder_name"],
                   "encoder_params": dict(encoder.get("encoder_params", {}), img_size=self.img_size)}
        self.encoder = get_encoder(in_channels=in_channels, encoder=encoder)
        self.linear_feat_proj = nn.Sequential(
            nn.LayerNorm(mlp_encoder_dim), nn.Linear(mlp_encoder_dim, mlp_encoder_dim), nn.ReLU(),
            nn.Linear(mlp_encoder_dim, slot_dim))
        if decoder["decoder_name"] != "MLPPatchDecoder":
            raise NameError("Extended-DINOSAUR expects a 'MLPPatchDecoder'...")
        decoder["decoder_params"]["img_size"] = self.img_size
        self.decoder = get_decoder(in_channels=in_channels, decoder=decoder)
        self.slot_attention = SlotAttention(
            dim_feats=slot_dim, dim_slots=slot_dim, num_slots=num_slots,
            num_iters_first=num_iterations_first, num_iters=num_iterations, mlp_hidden=mlp_hidden)
        self._init_model()
        self._init_range_guard()

    def forward(self, mode="decomp", *args, **kwargs):
        if mode == "decomp":
            return self._guarded(self.forward_decomp, *args, **kwargs)
        if mode == "decode":
            return self._guarded(self.decode, *args, **kwargs)
        raise NameError(f"{mode = } not recognized. Use ['decomp', 'decode']")

    def forward_decomp(self, x=None, num_imgs=10, decode=True, encoded_img_feats=None, **kwargs):
        """
        Returns the reference's dict: 'encoded_img_feats', 'slot_history' (+ the decoder outputs
        stacked over time when decode=True).  ``x`` (the video) is only needed by the backbone and
        is ignored when ``encoded_img_feats`` is given.
        """
        require_inference(self)
        if encoded_img_feats is None:
            dev = self.slot_attention.to_q.weight.device
            encoded_img_feats = self.encoder(x[:, :num_imgs].to(dev))      # (B, T, N, Dm), all frames batched
        feats = encoded_img_feats[:, :num_imgs]
        B, T, N, Dm = feats.shape
        predicted = self.initializer(batch_size=B, **kwargs)
        ln, l1, l2 = self.linear_feat_proj[0], self.linear_feat_proj[1], self.linear_feat_proj[3]
        tm = feats.transpose(0, 1).contiguous()                   # (T, B, N, Dm) time-major
        z = K.layer_norm(tm, ln.weight, ln.bias, ln.eps)
        z = K.linear(K.linear(z, l1.weight, l1.bias, act=K.ACT_RELU), l2.weight, l2.bias)
        kv = self.slot_attention.project_kv(z)                    # (T, B, N, 2D), all frames at once
        sa, history = self.slot_attention, []
        for t in range(T):
            slots = sa.iterate(kv[t], predicted, sa.num_iters_first if t == 0 else sa.num_iters)
            predicted = self.transition_module(slots)
            history.append(slots)
        out = {"encoded_img_feats": feats, "slot_history": torch.stack(history, dim=1)}
        if decode:
            dec = self.decode(out["slot_history"].reshape(B * T, self.num_slots, self.slot_dim))
            for k_, v in dec.items():
                out[k_] = v.reshape(B, T, *v.shape[1:]) if v.numel() else v
        return out

    def decode(self, slots):
        require_inference(self)
        return self.decoder(slots.contiguous())

    @torch.no_grad()
    def _init_model(self):
        for m in (self.linear_feat_proj, self.transition_module, self.slot_attention, self.decoder):
            init_xavier_(m)
        nn.init.zeros_(self.slot_attention.gru.bias_ih)
        nn.init.zeros_(self.slot_attention.gru.bias_hh)
        nn.init.orthogonal_(self.slot_attention.gru.weight_hh)
