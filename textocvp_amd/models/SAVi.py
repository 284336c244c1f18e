"""
SAVi video decomposition model on the MI355X kernels.
Reference: models/SAVi.py (forward :139-149, forward_decomp :152-223, encode :226-238,
decode :241-261, broadcast :264-275, _init_model :278-293).
"""

import math
import os

import torch
import torch.nn as nn

from .. import kernels as K
from ..precision import knob
from .Blocks.attention import SlotAttention
from .Blocks.initializers import get_initializer
from .Blocks.model_blocks import SoftPositionEmbed
from .Blocks.model_utils import RangeGuard, init_xavier_, refuse_replication, require_inference, tracks_structure
from .Blocks.transition_models import get_transition_module
from .EncodersDecoders.decoders import get_decoder
from .EncodersDecoders.encoders import get_encoder

__all__ = ["SAVi"]


@tracks_structure
class SAVi(nn.Module, RangeGuard):
    """
    Same constructor kwargs (= keys of configs/models/SAVi.json), ``forward(mode=...)`` contract,
    output dictionaries, ``num_slots`` / ``slot_dim`` attributes and state_dict keys as the
    reference, so ``load_checkpoint`` / ``05_evaluate_predictor.py`` work unchanged.

    Execution differs from the reference's frame-by-frame module calls:
      * everything that does not depend on the slots -- conv encoder, position embedding,
        LayerNorm + MLP, slot-attention input LayerNorm and the fused k/v projection -- is batched
        over frames (time-major chunks), only the slot recurrence stays sequential;
      * the decoder never materialises the (B*K, D, H, W) broadcast (see ConvDecoder).
    """

    _replicate_for_data_parallel = refuse_replication      # one process per GPU, never DataParallel replicas
    decode_accepts_out = True        # decode(slots, out=...) writes into caller-owned result views (evaluator)

    def __init__(self, num_slots, slot_dim, num_iterations=1, num_iterations_first=3,
                 in_channels=3, mlp_hidden=128, mlp_encoder_dim=128,
                 encoder={}, decoder={}, transition_module={}, initializer=None, **kwargs):
        super().__init__()
        self.num_slots = num_slots
        self.slot_dim = slot_dim
        self.in_channels = in_channels
        self.mlp_encoder_dim = mlp_encoder_dim
        # arithmetic of the per-pixel encoder MLP and k/v projection GEMMs (shapes that fit the split kernel)
        self.encoder_gemm_precision = knob("TOCVP_ENCODER_GEMM_PRECISION", "f16x3")

        self.initializer = get_initializer(mode=initializer, slot_dim=slot_dim, num_slots=num_slots)
        self.transition_module = get_transition_module(slot_dim=slot_dim, **transition_module)
        self.build_encoder(encoder_params=encoder)
        self.build_decoder(decoder_params=decoder)
        self.slot_attention = SlotAttention(
            dim_feats=mlp_encoder_dim, dim_slots=slot_dim, num_slots=num_slots,
            num_iters_first=num_iterations_first, num_iters=num_iterations, mlp_hidden=mlp_hidden)
        self._init_model()
        self._init_range_guard()
        self.max_encode_images = 1024        # images encoded per chunk (bounds HBM scratch)

    def build_encoder(self, encoder_params):
        self.encoder = get_encoder(in_channels=self.in_channels, encoder=encoder_params)
        self.out_features = self.encoder.out_features
        self.encoder_pos_embedding = SoftPositionEmbed(
            hidden_size=self.out_features,
            resolution=encoder_params["encoder_params"].get("resolution"))
        self.encoder_mlp = nn.Sequential(
            nn.LayerNorm(self.out_features),
            nn.Linear(self.out_features, self.mlp_encoder_dim),
            nn.ReLU(),
            nn.Linear(self.mlp_encoder_dim, self.mlp_encoder_dim))

    def build_decoder(self, decoder_params):
        self.decoder_resolution = decoder_params["decoder_params"].get("resolution")
        self.decoder_pos_embedding = SoftPositionEmbed(
            hidden_size=self.slot_dim, resolution=self.decoder_resolution)
        self.decoder = get_decoder(in_channels=self.slot_dim, decoder=decoder_params)

    # ------------------------------------------------------------------------------------------
    range_fallbacks = {"encoder_gemm_precision": {"f16x3": "fp32"}}

    def forward(self, mode="decomp", *args, **kwargs):
        if mode == "decomp":
            return self._guarded(self.forward_decomp, *args, **kwargs)
        if mode == "decode":
            return self._guarded(self.decode, *args, **kwargs)
        raise NameError(f"{mode = } not recognized. Use ['decomp', 'decode']")

    def forward_decomp(self, x, num_imgs=10, decode=True, **kwargs):
        """
        x (B, L, C, H, W) -> {'recons_imgs', 'recons_objs', 'masks', 'slot_history'} with
        slot_history (B, num_imgs, K, D) = corrector outputs (pre-transition, SAVi.py:192-198,212).
        With decode=False the three image entries are the reference's stacked empty tensors (0, T).
        ``init_noise`` (B, K, D), if given in kwargs, replaces the initialiser's Gaussian draw.
        """
        require_inference(self)
        B = x.shape[0]
        T = num_imgs
        predicted = self.initializer(batch_size=B, **kwargs)
        history, _ = self.decomp_frames(x, 0, T, predicted)
        slot_history = K.stack1(history)                                # (B, T, K, D)

        if decode:
            out = self.decode(slot_history.reshape(B * T, self.num_slots, self.slot_dim))
            C, H, W = out["recons_imgs"].shape[1:]
            recons_imgs = out["recons_imgs"].reshape(B, T, C, H, W)
            recons_objs = out["recons"].reshape(B, T, self.num_slots, C, H, W)
            masks = out["masks"].reshape(B, T, self.num_slots, 1, H, W)
        else:
            recons_imgs = recons_objs = masks = torch.empty((0, T))
        return {"recons_imgs": recons_imgs, "recons_objs": recons_objs, "masks": masks,
                "slot_history": slot_history}

    def decomp_frames(self, x, t_begin, t_end, predicted):
        """
        The recurrent part of forward_decomp for frames t_begin .. t_end - 1 of x (B, L, C, H, W): encode (batched
        over up to ``max_encode_images`` images), slot-attention iterations (``num_iters_first`` on frame 0), transition.
        ``predicted`` (B, K, D) = the initialiser's slots (t_begin = 0) or the transition output of frame t_begin - 1.
        Returns ([slots of every frame], predicted for frame t_end): a decomposition may be cut at any frame and
        continued later -- evaluator.forward_eval encodes the context frames, starts the rollout and decomposes the
        remaining frames on another stream.  Every image goes through the same kernels whatever the cut.
        """
        require_inference(self)
        B = x.shape[0]
        dev = self.slot_attention.to_q.weight.device
        x = x.to(dev)
        # time-major copy of the frames: (T, B, C, H, W), so that frame t of all samples is one
        # contiguous (B, N, 2D) k/v block for the slot-attention kernel
        frames = K.contiguous(x[:, t_begin:t_end].transpose(0, 1)) if x.dtype == torch.float32 else \
            x[:, t_begin:t_end].transpose(0, 1).contiguous()
        T = t_end - t_begin
        chunk = max(1, self.max_encode_images // max(B, 1))
        sa = self.slot_attention
        history = []
        for t0 in range(0, T, chunk):
            t1 = min(T, t0 + chunk)
            kv = self._encode_kv(frames[t0:t1].reshape((t1 - t0) * B, *frames.shape[2:]))
            if isinstance(kv, K.SplitAct):                      # fp16 operand planes of the (t1 - t0) * B images
                kv = K.SplitAct(kv.planes, (t1 - t0, B) + tuple(kv.shape[1:]))
            else:
                kv = kv.reshape(t1 - t0, B, kv.shape[-2], kv.shape[-1])
            for t in range(t0, t1):
                n_it = sa.num_iters_first if t_begin + t == 0 else sa.num_iters
                slots = sa.iterate(sa.frame_kv(kv, t - t0, t1 - t0), predicted, n_it)
                predicted = self.transition_module(slots)
                history.append(slots)
        return history, predicted

    # ------------------------------------------------------------------------------------------
    def _encode_feats(self, imgs):
        """ conv encoder + position embedding + LayerNorm + MLP: (n,3,H,W) -> (n, H*W, Dm) """
        y = self.encoder.forward_nhwc(imgs)                             # (n, H, W, C) NHWC
        n, H, W, C = y.shape
        ln, l1, l2 = self.encoder_mlp[0], self.encoder_mlp[1], self.encoder_mlp[3]
        z = K.layer_norm(y.reshape(n * H * W, C), ln.weight, ln.bias, ln.eps,
                         add=self.encoder_pos_embedding.table().reshape(H * W, C))
        with K.gemm_precision(self.encoder_gemm_precision, owner=(self, "encoder_gemm_precision")):
            z = K.linear(K.linear(z, l1.weight, l1.bias, act=K.ACT_RELU), l2.weight, l2.bias)
        return z.reshape(n, H * W, self.mlp_encoder_dim)

    def _encode_kv(self, imgs):
        """ image batch -> fused slot-attention keys/values (n, N, 2D) """
        feats = self._encode_feats(imgs)
        with K.gemm_precision(self.encoder_gemm_precision, owner=(self, "encoder_gemm_precision")):
            return self.slot_attention.project_kv(feats)

    def encode(self, x):
        """ x (B, C, H, W) -> features (B, N, mlp_encoder_dim)   (SAVi.py:226-238) """
        require_inference(self)
        return self._encode_feats(x.contiguous())

    def decode(self, slots, out=None):
        """
        slots (B', K, D) -> {'recons_imgs' (B',C,H,W), 'recons' (B',K,C,H,W), 'masks' (B',K,1,H,W)}
        (SAVi.py:241-261; softmax over slots at :254, compositing at :255)
        ``out`` (extension) = (recons_imgs, recons, masks[, clamped_imgs]) views the tail kernel writes into -- every
        frame contiguous, frames any distance apart -- so that per-step decodes land in the evaluator's (B * P, ...)
        results without a stack / copy; ``clamped_imgs`` also receives clamp(recons_imgs, 0, 1).
        """
        require_inference(self)
        imgs, recons, masks = self.decoder.decode_slots(slots.contiguous(),
                                                        self.decoder_pos_embedding.table(), out=out)
        return {"recons_imgs": imgs, "recons": recons, "masks": masks}

    @torch.no_grad()
    def _init_model(self):
        """ xavier init, zero GRU biases, orthogonal hidden-to-hidden (SAVi.py:278-293) """
        init_xavier_(self)
        nn.init.zeros_(self.slot_attention.gru.bias_ih)
        nn.init.zeros_(self.slot_attention.gru.bias_hh)
        nn.init.orthogonal_(self.slot_attention.gru.weight_hh)
        if hasattr(self.slot_attention, "slots_mu"):
            limit = math.sqrt(6.0 / (1 + self.slot_attention.dim_slots))
            nn.init.uniform_(self.slot_attention.slots_mu, -limit, limit)
            nn.init.uniform_(self.slot_attention.slots_sigma, -limit, limit)
