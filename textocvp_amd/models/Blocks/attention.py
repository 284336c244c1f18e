"""
Attention modules of the slot-rollout path on the MI355X kernels.

Mirror of the reference's models/Blocks/attention.py (class names, constructor arguments,
parameter names -> identical state_dict keys).  Forward passes call libtocvp kernels only:
fp32-MFMA GEMMs with fused bias / ReLU / residual epilogues, one-wave-per-row LayerNorm,
flash-style fp32-MFMA attention and the location-streaming slot-attention iteration.
"""

import math
import os

import torch
import torch.nn as nn

from ... import kernels as K
from .model_utils import Derived, cached_params, init_xavier_, require_inference, tracks_structure

__all__ = ["SlotAttention", "MultiHeadSelfAttention", "MultiHeadCrossAttention",
           "TransformerBlock", "TransformerDecoderBlock", "AdaptedEncoderBlock"]


# Activations that only feed GEMMs can leave their producer (LayerNorm, attention and GEMM epilogues) as fp16
# operand planes (the bytes of the fp32 tensor); a GEMM fed with planes runs the persistent all-DMA planes kernel
# (gemm_f16p.hip, planes3: no split in the k-loop, both operands through LDS, 256 x 256 tiles).  TOCVP_PRESPLIT:
#   "0"            never;
#   "wide" (default) only where the consuming GEMM is at least 1536 columns wide (qkv, MLP up-projections);
#   "1"            everywhere the shapes allow.
# Round 3 (planes3), isolated, 38400 rows, dense random operands, planes3 vs the in-loop-split kernel: 2048x512
# 255-280 vs 380-400 us, 1536x512 205-227 vs 239 us, 512x2048 288 vs 290 us, 512x512 107 vs 80 us.  IN the rollout
# the in-loop-split kernel runs 25 % faster than on dense random data (294 us at 2048x512: sparse post-ReLU / small
# LayerNorm'd operands keep the clock up) and planes3 gains 10 % on the wide products (268 us); at B=128 per step:
# "0" 688.3 ms, "wide" 692.0 ms with the decoder overlapped on the second stream (a persistent one-workgroup-per-CU
# kernel with 128 KB of LDS cannot share a CU with the decoder's workgroups, the 36 KB in-loop-split kernel can),
# 3467 vs 3449 frames/s (+0.5 %) without the overlap -> neutral, the in-kernel split stays the default
# (profiles/r03_gemm_planes3.md).
# End of round 3: "wide" with the planes consumed by the TWO-workgroups-per-CU planes kernel of gemm_bf16.hip
# (gemm_f16_planes_kernel: A planes by LDS-DMA, no split instructions in its k-loop; TOCVP_GEMM_P2=0, now the default)
# is the default: same arithmetic (the producer makes the split the consumer would make), +1.5 % on the step with
# the decoder overlapped (3694-3704 -> 3754-3759 frames/s, three alternations on one box) and neutral without the
# overlap (3637-3640 vs 3640-3644): the rollout's GEMMs leave the vector ALUs to the decoder's staging.
# (end of round 4, with the mid-size chunk GEMM as the consumer: planes for EVERY fitting product -- TOCVP_PRESPLIT=all, the
# 512-wide output projections included -- 4034 / 4017 vs 4024 / 4036 frames/s at B = 128, 3669 / 3675 vs 3698 / 3698 at 32,
# 2450 / 2448 vs 2592 / 2571 at 8: "wide" stays)
_PRESPLIT = os.environ.get("TOCVP_PRESPLIT", "wide")
# ... and the MLP's hidden activation leaves the up-projection's epilogue as planes for the down-projection (its 2048-deep
# k-loop then holds no split instructions): another +0.5 % with the decoder overlapped (3799-3807 -> 3819-3827 frames/s,
# two alternations), -0.3 % without; same rows threshold; bit-identical
_PRESPLIT_MLP = os.environ.get("TOCVP_PRESPLIT_MLP", "1") != "0"
_PRESPLIT_MIN_N = 1536
# B=32 (9600 rows) measures 1.5 % slower with planes, B=128 (38400 rows) 2 % faster (round 3, two GEMMs per MLP)
# (round 4, second half: with the mid-size chunk GEMM -- 64 x 256 tiles, A by LDS-DMA -- planes pay from ~2600 rows (B = 8, 2400 rows: 2456-2464 vs 2470-2513 frames/s with them; B = 32: 3674-3678 vs 3554-3559): at 9600
# rows 77.9 / 67.5 / 54.7 us against 87.3 / 78.4 / 62.0 us with fp32 input on the MLP down / up / qkv products; below that the
# skinny split-K kernels on fp32 input stay faster)
_PRESPLIT_MIN_ROWS = int(os.environ.get("TOCVP_PRESPLIT_MIN_ROWS", "2600" if K._GEMM_MID else "16384"))
# the persistent chunk-resident GEMM (csrc/gemm_f16c.hip) for the predictor's plane-input products: 189 vs 222 us isolated on
# the qkv projection at 38400 rows, but its workgroups need a whole CU each and stall the decoder's on the other stream:
# A/B at B = 128 on one box 3997 / 4002 (on) vs 4026 / 4027 frames/s (off) with the decode overlapped, 3894 / 3888 vs
# 3868 / 3872 without -- off by default
_CHUNK_GEMM = os.environ.get("TOCVP_PREDICTOR_CHUNK_GEMM", "0") != "0"
# text cross-attention collapsed over the caption (csrc/xattn.hip): one fused kernel per block instead of
# LayerNorm + q GEMM + attention + output GEMM; TOCVP_XATTN_COLLAPSE=0 keeps the four-kernel path
_XATTN_COLLAPSE = os.environ.get("TOCVP_XATTN_COLLAPSE", "1") != "0"
# longest caption (tokens) the collapsed kernel takes: 32 (captions of 33-50 tokens keep the four-kernel path; a 64-slot form of
# the collapsed kernel was built in round 4, measured equal to it and retired in round 5)
_XATTN_MAX_LT = 32


class TextKV:
    """ step-invariant caption operands of one predictor block: the fused [k | v] projection (B, Lt, 2 inner) and,
    for captions of at most 32 tokens, the collapsed operands (G fragments, HT fragments, Lt) of csrc/xattn.hip """

    __slots__ = ("kv", "collapsed")

    def __init__(self, kv, collapsed=None):
        self.kv, self.collapsed = kv, collapsed


_params = cached_params          # model_utils: per-module tuples, dropped when a parameter object is replaced


def _ln(x, ln, add=None, split=0):
    # planes only for the many-row products: the skinny GEMMs of small batches take fp32 input (split-K over idle CUs,
    # 64-deep k-tiles), mid-size ones measured slower with planes -- and small-batch results stay what they were
    # ... and never in a range-checked pass: a plane-producing epilogue saturates at |x| = 255.9 without a check of
    # its own, so the checked pass hands fp32 tensors to the consuming GEMM, which verifies them (same arithmetic)
    if split and (K._CHECK_RANGE or x.numel() // x.shape[-1] <= _PRESPLIT_MIN_ROWS):
        split = 0
    w, b, eps = _params(ln, lambda m: (m.weight, m.bias, m.eps))
    return K.layer_norm(x, w, b, eps, add=add, split=split)


def _ns(*dims, n_out=None):
    """
    planes for split activations under the active GEMM arithmetic (0 = keep fp32 tensors).
    ``dims``: every dimension that must fit the split kernels; ``n_out``: width of the GEMM that consumes
    the activation (decides in the default "wide" policy).
    """
    ns = K.active_nsplit()
    if _PRESPLIT == "0" or not ns or K._CHECK_RANGE:     # checked pass: fp32 hand-over, verified by the consumer
        return 0
    if _PRESPLIT == "wide" and (ns != 22 or n_out is None or n_out < _PRESPLIT_MIN_N):
        return 0
    return ns if all(d % 64 == 0 for d in dims) else 0


def _mlp(x, seq, residual):
    """ Linear -> ReLU -> Linear (+ residual), both epilogues fused into the GEMMs. """
    w1, b1, w2, b2 = _params(seq, lambda m: (m[0].weight, m[0].bias, m[2].weight, m[2].bias))
    ns = _ns(w1.shape[0], w1.shape[1], w2.shape[0], n_out=w2.shape[0])
    if (_PRESPLIT_MLP and not ns and not K._CHECK_RANGE and K.active_nsplit() == 22 and math.prod(tuple(x.shape[:-1])) > _PRESPLIT_MIN_ROWS
            and all(d % 64 == 0 for d in w1.shape + w2.shape[:1])):
        ns = 22                           # the hidden activation leaves the up-projection's epilogue as planes
    if b1 is not None and b2 is not None and K.mlp_fused_ok(x, w1, w2):
        # the predictor's 512 -> 2048 -> 512 pairs at many rows: ONE kernel, the hidden activation stays on the CU
        # (csrc/mlp_fused.hip; bit-identical to the two GEMMs below on whole 128-row tiles)
        return K.mlp_fused(x, w1, b1, w2, b2, residual=residual)
    h = K.linear(x, w1, b1, act=K.ACT_RELU, out_split=ns, chunk_ok=_CHUNK_GEMM)
    return K.linear(h, w2, b2, residual=residual, chunk_ok=_CHUNK_GEMM)


SD_PLANES = 128          # slot dim of the plane-input slot-attention kernel


@tracks_structure
class SlotAttention(nn.Module):
    """
    Iterative slot attention (reference attention.py:12-128; algorithm :86-110).

    Quirks kept for parity: ``scale = dim_feats ** -0.5`` (:46), softmax ACROSS SLOTS (:100),
    LayerNorm eps 1e-3 (:49-51), 3 iterations on the first frame / 1 afterwards (:90).

    ``forward`` keeps the reference signature.  SAVi uses the two halves separately so that the
    k/v projection of all frames is batched: ``project_kv`` (LayerNorm + ONE fused [to_k; to_v]
    GEMM) and ``iterate`` (the recurrent part).
    ``store_attention_masks`` (default False) makes the kernel also write the (B, K, N)
    attention tensor the reference keeps in ``self.attention_masks`` (:101).
    """

    def __init__(self, dim_feats, dim_slots, num_slots, num_iters_first=2, num_iters=2,
                 mlp_hidden=128, epsilon=1e-8):
        super().__init__()
        self.dim_slots = dim_slots
        self.num_iters_first = num_iters_first
        self.num_iters = num_iters
        self.num_slots = num_slots
        self.epsilon = epsilon
        self.scale = dim_feats ** -0.5

        self.norm_input = nn.LayerNorm(dim_feats, eps=0.001)
        self.norm_slot = nn.LayerNorm(dim_slots, eps=0.001)
        self.norm_mlp = nn.LayerNorm(dim_slots, eps=0.001)
        self.to_q = nn.Linear(dim_slots, dim_slots)
        self.to_k = nn.Linear(dim_feats, dim_slots)
        self.to_v = nn.Linear(dim_feats, dim_slots)
        self.gru = nn.GRUCell(dim_slots, dim_slots)
        self.mlp = nn.Sequential(
            nn.Linear(dim_slots, mlp_hidden), nn.ReLU(), nn.Linear(mlp_hidden, dim_slots))

        self.store_attention_masks = False
        self.attention_masks = None
        self._derived = Derived()
        self._ws = None
        self.kv_planes = os.environ.get("TOCVP_SA_KV_PLANES", "1") != "0"    # k / v as fp16 operand planes

    # -- k/v projection (once per frame; batched over frames by SAVi) --------------------------
    def project_kv(self, inputs):
        """
        inputs (..., N, Df) -> fused kv: k = [..., :D], v = [..., D:].
        Under the f16x3 GEMM arithmetic the projection writes fp16 OPERAND PLANES (a kernels.SplitAct of
        shape (..., N, 2 D): planes (rows, 2, 2 D) of 2^8 * value) that the slot-attention kernel streams
        without converting anything (same HBM bytes as fp32); otherwise an fp32 tensor (..., N, 2 D).
        """
        w = self._derived.get("w_kv", [self.to_k.weight, self.to_v.weight],
                              lambda: torch.cat([self.to_k.weight, self.to_v.weight], 0).contiguous())
        b = self._derived.get("b_kv", [self.to_k.bias, self.to_v.bias],
                              lambda: torch.cat([self.to_k.bias, self.to_v.bias], 0).contiguous())
        # (a range-checked pass keeps fp32 rows: slot_attn_iter verifies them before the kernel splits them itself)
        planes = K.active_nsplit() == 22 and self.kv_planes and w.shape[0] == 2 * SD_PLANES and not K._CHECK_RANGE
        return K.linear(_ln(inputs, self.norm_input), w, b, out_split=22 if planes else 0)

    @staticmethod
    def frame_kv(kv, t, T):
        """ frame ``t`` of a time-major projection of T frames: (B, N, 2 D) fp32 or (B, N, 2, 2 D) fp16 planes """
        if isinstance(kv, K.SplitAct):
            Tn, B, N, D2 = kv.shape
            assert Tn == T
            return kv.planes.view(T, B, N, 2, D2)[t]
        return kv[t]

    # -- recurrent refinement ------------------------------------------------------------------
    def iterate(self, kv, slots, num_iters):
        """ kv from project_kv: (B, N, 2D) fp32 or (B, N, 2, 2D) fp16 planes; slots (B, K, D) -> refined slots """
        B, Ks, D = slots.shape
        if isinstance(kv, K.SplitAct):
            kv = kv.planes.view(*kv.shape[:-1], 2, kv.shape[-1])
        planes = kv.dtype == torch.float16
        N = kv.shape[1]
        # ticket words zeroed once per shape; every shape keeps its workspace for the life of the module -- a captured
        # graph (evaluator.GraphedEval) holds the raw pointer, so a workspace must never be freed behind it
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get((B, N, slots.device))
        if ws is None:
            ws = self._ws[(B, N, slots.device)] = K.slot_attn_workspace(B, N, slots.device)
        attn = None
        for _ in range(num_iters):
            prev = slots
            q = K.linear(_ln(slots, self.norm_slot), self.to_q.weight, self.to_q.bias)
            if self.store_attention_masks:
                attn = torch.empty((B, Ks, N), device=slots.device, dtype=torch.float32)
            if planes:
                upd = K.slot_attn_iter_planes(q, kv, self.scale, self.epsilon, attn_out=attn, ws=ws)
            else:
                upd = K.slot_attn_iter(q, kv[..., :D], kv[..., D:], self.scale, self.epsilon, attn_out=attn,
                                       ws=ws)
            gi = K.linear(upd, self.gru.weight_ih, self.gru.bias_ih)
            gh = K.linear(prev, self.gru.weight_hh, self.gru.bias_hh)
            slots = K.gru_gates(gi, gh, prev)
            slots = _mlp(_ln(slots, self.norm_mlp), self.mlp, residual=slots)
        self.attention_masks = attn
        return slots

    def forward(self, inputs, slots, step=0, **kwargs):
        """ inputs (B, N, Df), slots (B, K, D) -> slots (B, K, D)   (reference :67-112) """
        require_inference(self)
        self.attention_masks = None
        n_it = self.num_iters_first if step == 0 else self.num_iters
        return self.iterate(self.project_kv(inputs.contiguous()), slots.contiguous(), n_it)

    def get_attention_masks(self, shape=None):
        """ last attention (B, K, N[ -> *shape]) -- needs ``store_attention_masks = True`` """
        if self.attention_masks is None:
            raise RuntimeError("set slot_attention.store_attention_masks = True before the forward")
        m = self.attention_masks
        return m if shape is None else m.reshape(m.shape[0], m.shape[1], *shape)


@tracks_structure
class MetaAttention(nn.Module):
    """ q/k/v/out parameter holder shared by self- and cross-attention (reference :136-215). """

    def __init__(self, emb_dim, num_heads=1, dropout=0., out_dim=None, **kwargs):
        assert num_heads >= 1
        if emb_dim % num_heads != 0:
            raise ValueError(f"{emb_dim = } must be divisible by {num_heads}...")
        if dropout != 0.:
            raise NotImplementedError("attention dropout is a training feature (inference-only path)")
        super().__init__()
        out_dim = out_dim if out_dim is not None else emb_dim
        self.emb_dim = emb_dim
        self.num_heads = num_heads
        self.q = nn.Linear(emb_dim, emb_dim, bias=False)
        self.k = nn.Linear(emb_dim, emb_dim, bias=False)
        self.v = nn.Linear(emb_dim, emb_dim, bias=False)
        self.drop = nn.Dropout(dropout)
        self.out_projection = nn.Sequential(nn.Linear(emb_dim, out_dim, bias=False))
        self.attention_masks = None
        self._derived = Derived()

    def forward(self, x):
        raise NotImplementedError("Base-Class does not implement a 'forward' method...")


class MultiHeadSelfAttention(MetaAttention):
    """
    Bias-free multi-head self-attention (reference :219-265).  q, k and v come from ONE GEMM
    against the concatenated (3E, E) weight; the attention kernel reads the three column
    slices of that buffer in place.  ``residual`` is fused into the output-projection epilogue.
    """

    def __init__(self, emb_dim, num_heads=8, dropout=0.):
        super().__init__(emb_dim=emb_dim, num_heads=num_heads, dropout=dropout)

    def forward(self, x, residual=None, **kwargs):
        if kwargs.get("mask", None) is not None:
            raise NotImplementedError("attention masks are not used on the slot-rollout path")
        E = x.shape[-1]                                                   # tensor or SplitAct
        wq, wk, wv, wo, heads = _params(self, lambda m: (m.q.weight, m.k.weight, m.v.weight, m.out_projection[0].weight,
                                                          m.num_heads))
        w = self._derived.get("w_qkv", [wq, wk, wv], lambda: torch.cat([wq, wk, wv], 0).contiguous())
        if K.mha_planes_ok(heads, E):
            # head dim 64 (the predictor blocks): the projection's epilogue writes q / k / v as fp16 operand planes and the
            # attention kernel copies them (csrc/attn_planes.hip) instead of splitting / transposing fp32 rows per query block
            B, T = x.shape[0], x.shape[1]
            qkv = K.linear(x, w, chunk_ok=_CHUNK_GEMM, out_split=22)      # planes (B T, 2, 3E)
            o = K.mha_planes(qkv, 0, qkv, E, qkv, 2 * E, B, T, T, heads, (E // heads) ** -0.5,
                             out_split=_ns(E, n_out=wo.shape[0]))
            return K.linear(o, wo, residual=residual)
        qkv = K.linear(x, w, chunk_ok=_CHUNK_GEMM)                        # (B, T, 3E)
        o = K.mha(qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:], heads, (E // heads) ** -0.5,
                  out_split=_ns(E, n_out=wo.shape[0]))
        return K.linear(o, wo, residual=residual)


    def forward_last(self, x, n_last, residual_last):
        """
        Self-attention evaluated only for the LAST ``n_last`` query tokens of every sequence (keys and
        values still cover all tokens): x (B, T, E) -> (B, n_last, E).  Used by the predictor's final
        layer, whose output is consumed for the newest frame only (text_cond_OCVP.py:101-103), so the
        query / output projections of the other T - n_last tokens are never needed.  Exact.
        """
        B, T, E = x.shape
        w_kv = self._derived.get("w_kv", [self.k.weight, self.v.weight],
                                 lambda: torch.cat([self.k.weight, self.v.weight], 0).contiguous())
        if K.mha_planes_ok(self.num_heads, E):
            kv = K.linear(x, w_kv, chunk_ok=_CHUNK_GEMM, out_split=22)    # planes (B T, 2, 2E)
            q = K.linear(K.contiguous(x[:, T - n_last:]), self.q.weight, out_split=22)
            o = K.mha_planes(q, 0, kv, 0, kv, E, B, n_last, T, self.num_heads, (E // self.num_heads) ** -0.5)
            return K.linear(o, self.out_projection[0].weight, residual=residual_last)
        kv = K.linear(x, w_kv, chunk_ok=_CHUNK_GEMM)                      # (B, T, 2E)
        q = K.linear(K.contiguous(x[:, T - n_last:]), self.q.weight)      # (B, n_last, E)
        o = K.mha(q, kv[..., :E], kv[..., E:], self.num_heads, (E // self.num_heads) ** -0.5)
        return K.linear(o, self.out_projection[0].weight, residual=residual_last)


class MultiHeadCrossAttention(MetaAttention):
    """
    Text-to-slot cross-attention (reference :269-319).  Keys/values depend only on the text and
    are therefore projected once per sequence (``project_kv``) and reused by all rollout steps.
    No key-padding mask: padded text positions participate, as in the reference (:314).
    """

    def __init__(self, emb_dim, dim_head, kv_dim, num_heads=8, dropout=0.):
        super().__init__(emb_dim=emb_dim, num_heads=num_heads, dropout=dropout)
        self.dim_head = dim_head
        inner_dim = dim_head * num_heads
        self.q = nn.Linear(emb_dim, inner_dim, bias=False)
        self.k = nn.Linear(kv_dim, inner_dim, bias=False)
        self.v = nn.Linear(kv_dim, inner_dim, bias=False)
        self.out_projection = nn.Linear(inner_dim, emb_dim)

    def project_kv(self, enc_embs):
        """ normalised text (B, Lt, kv_dim) -> fused (B, Lt, 2*inner) [k | v] """
        w = self._derived.get("w_kv", [self.k.weight, self.v.weight],
                              lambda: torch.cat([self.k.weight, self.v.weight], 0).contiguous())
        return K.linear(enc_embs, w)

    def collapse(self, kv):
        """
        Fold the query and output projections into per-sample operands of the caption (csrc/xattn.hip):
            G[b, h, t, :]  = sum_d Wq[h dh + d, :] K[b, t, h dh + d]      scores = LN(x) G^T (x dim_head ** -0.5)
            HT[b, :, h, t] = sum_d Wo[:, h dh + d] V[b, t, h dh + d]      output = P HT^T
        as fp16 operand planes in MFMA-fragment order, caption slots of a head padded to 16 or 32.  Built once per
        caption batch; exact up to fp32 re-association.  None when the shapes do not fit the fused kernel
        (more than 32 caption tokens, other widths) -- the caller then takes the four-kernel path.
        """
        B, Lt, two_inner = kv.shape
        H, dh, inner = self.num_heads, self.dim_head, two_inner // 2
        E_in, E_out = self.q.weight.shape[1], self.out_projection.weight.shape[0]
        if not (_XATTN_COLLAPSE and Lt <= _XATTN_MAX_LT and H == 8 and dh == 64 and E_in == 512 and E_out == 512 and
                self.out_projection.bias is not None and K.active_nsplit() == 22 and kv.is_contiguous()
                and B * H <= 65535):
            return None
        # caption slots of a head padded to 16 (captions of at most 16 tokens) or 32 (17-32): with 32 the fused kernel
        # still does half the work of the 512 x 512 projections; 33-50 tokens (text_encoders.py:36 admits 50) would need 64
        # slots per head = the uncollapsed width and keep the four-kernel path
        LP = 16 if Lt <= 16 else (32 if Lt <= 32 else 64)
        # both operands from ONE batched launch each over (sample, head), slices addressed in place (round 3: 16 small
        # GEMMs + two strided copies, 351 us per block): exact fp32 MFMA, rows / columns behind Lt stay zero
        G = torch.zeros((B, H, LP, E_in), device=kv.device, dtype=torch.float32)
        HT = torch.zeros((B, E_out, H, LP), device=kv.device, dtype=torch.float32)
        wq, wo = self.q.weight, self.out_projection.weight
        assert wq.is_contiguous() and wo.is_contiguous()
        # G[b, h] (Lt x E_in) = K[b, :, h dh : (h + 1) dh] (Lt x dh) @ Wq[h dh : (h + 1) dh, :] (dh x E_in)
        K.bmm_f32(kv, wq, G, Lt, E_in, dh, lda=two_inner, ldb=E_in, ldc=E_in, batch=(B, H),
                  sA=(Lt * two_inner, dh), sB=(0, dh * E_in), sC=(H * LP * E_in, LP * E_in))
        # HT[b, :, h, :Lt] (E_out x Lt) = Wo[:, h dh : (h + 1) dh] (E_out x dh) @ V[b, :, h dh : (h + 1) dh]^T (dh x Lt)
        K.bmm_f32(wo, kv[:, :, inner:], HT, E_out, Lt, dh, lda=inner, ldb=two_inner, ldc=H * LP, transB=True,
                  batch=(B, H), sA=(0, dh), sB=(Lt * two_inner, dh), sC=(E_out * H * LP, LP))
        return K.xattn_operands(G.view(B * H * LP, E_in), HT.view(B * E_out, H * LP)) + (Lt,)

    def forward(self, enc_embs, query_embs, residual=None, kv=None, **kwargs):
        if kv is None:
            kv = self.project_kv(enc_embs)
        inner = self.q.weight.shape[0]
        q = K.linear(query_embs, self.q.weight)
        o = K.mha(q, kv[..., :inner], kv[..., inner:], self.num_heads, self.dim_head ** -0.5,
                  out_split=_ns(inner, self.out_projection.weight.shape[0],
                                n_out=self.out_projection.weight.shape[0]))
        return K.linear(o, self.out_projection.weight, self.out_projection.bias, residual=residual)


@tracks_structure
class TransformerBlock(nn.Module):
    """
    Transformer encoder block (reference :323-396): pre-norm by default, POST-norm when used as the
    SAVi transition (transition_models.py:24-29).  LayerNorm eps 1e-6.
    """

    def __init__(self, embed_dim, num_heads, mlp_size, pre_norm=True):
        super().__init__()
        assert num_heads >= 1
        self.embed_dim, self.mlp_size, self.num_heads, self.pre_norm = \
            embed_dim, mlp_size, num_heads, pre_norm
        self.attn = MultiHeadSelfAttention(emb_dim=embed_dim, num_heads=num_heads)
        self.mlp = nn.Sequential(
            nn.Linear(embed_dim, mlp_size), nn.ReLU(), nn.Linear(mlp_size, embed_dim))
        self.layernorm_query = nn.LayerNorm(embed_dim, eps=1e-6)
        self.layernorm_mlp = nn.LayerNorm(embed_dim, eps=1e-6)
        init_xavier_(self)

    def forward(self, inputs):
        assert inputs.ndim == 3
        require_inference(self)
        inputs = inputs.contiguous()
        if self.pre_norm:
            E = self.embed_dim
            y = self.attn(_ln(inputs, self.layernorm_query, split=_ns(E, n_out=3 * E)), residual=inputs)
            return _mlp(_ln(y, self.layernorm_mlp, split=_ns(E, self.mlp_size, n_out=self.mlp_size)), self.mlp,
                        residual=y)
        y = _ln(self.attn(inputs, residual=inputs), self.layernorm_query)
        return _ln(_mlp(y, self.mlp, residual=y), self.layernorm_mlp)


@tracks_structure
class TransformerDecoderBlock(nn.Module):
    """ Cross-attention + MLP, both pre-norm with residuals (reference :400-467). """

    def __init__(self, embed_dim, head_dim, kv_dim, num_heads, mlp_size):
        super().__init__()
        self.ln_mlp = nn.LayerNorm(embed_dim, eps=1e-6)
        self.mlp = nn.Sequential(
            nn.Linear(embed_dim, mlp_size), nn.ReLU(), nn.Linear(mlp_size, embed_dim))
        self.ln_cross_att_q = nn.LayerNorm(embed_dim, eps=1e-6)
        self.ln_cross_att_kv = nn.LayerNorm(kv_dim, eps=1e-6)
        self.cross_attn = MultiHeadCrossAttention(
            emb_dim=embed_dim, dim_head=head_dim, num_heads=num_heads, kv_dim=kv_dim)

    def project_text(self, feats):
        """ step-invariant half of the block: LayerNorm(text) -> fused K/V projection (+ the collapsed operands of
        the fused cross-attention kernel when the caption is short enough) """
        kv = self.cross_attn.project_kv(_ln(feats.contiguous(), self.ln_cross_att_kv))
        return TextKV(kv, self.cross_attn.collapse(kv))

    def forward(self, queries, feats, text_kv=None):
        assert queries.ndim == 3
        if text_kv is None:
            text_kv = self.project_text(feats)
        if isinstance(text_kv, TextKV):
            if text_kv.collapsed is not None and K.active_nsplit() == 22 and not isinstance(queries, K.SplitAct):
                # LayerNorm + query projection + attention over the caption + output projection + residual: ONE kernel
                Gf, Hf, Lt = text_kv.collapsed
                qw, qb, qeps, ob, heads, dh, lnm, mlp, hid = _params(self, lambda m: (
                    m.ln_cross_att_q.weight, m.ln_cross_att_q.bias, m.ln_cross_att_q.eps, m.cross_attn.out_projection.bias,
                    m.cross_attn.num_heads, m.cross_attn.dim_head, m.ln_mlp, m.mlp, m.mlp[0].weight.shape[0]))
                z = K.xattn_collapsed(queries, qw, qb, qeps, Gf, Hf, ob, heads, Lt, dh ** -0.5)
                return _mlp(_ln(z, lnm, split=_ns(queries.shape[-1], hid, n_out=hid)), mlp, residual=z)
            text_kv = text_kv.kv
        E = queries.shape[-1]
        z = self.cross_attn(None, query_embs=_ln(queries, self.ln_cross_att_q,
                                                 split=_ns(E, n_out=self.cross_attn.q.weight.shape[0])),
                            residual=queries, kv=text_kv)
        return _mlp(_ln(z, self.ln_mlp, split=_ns(E, self.mlp[0].weight.shape[0], n_out=self.mlp[0].weight.shape[0])),
                    self.mlp, residual=z)


class AdaptedEncoderBlock(TransformerBlock):
    """
    Predictor layer: self-attention, text cross-attention block, MLP (reference :471-534).
    The last residual is taken from ``y`` (output of the self-attention stage), NOT from the
    cross-attention branch -- reference wiring at :521-523, reproduced on purpose.
    """

    def __init__(self, embed_dim, num_heads, mlp_size, fusion_params):
        super().__init__(embed_dim=embed_dim, num_heads=num_heads, mlp_size=mlp_size)
        self.cross_attention = TransformerDecoderBlock(
            embed_dim=embed_dim, kv_dim=embed_dim, head_dim=fusion_params.get("head_dim"),
            num_heads=fusion_params.get("num_heads"), mlp_size=fusion_params.get("mlp_size"))

    def forward(self, x, text_embeddings, text_kv=None):
        assert x.ndim == 3, f"Input 'x' must have 3 dims, but got {x.shape = }..."
        E, hid, lnq, attn, xblk, lnm, mlp = _params(self, lambda m: (
            m.embed_dim, m.mlp_size, m.layernorm_query, m.attn, m.cross_attention, m.layernorm_mlp, m.mlp))
        y = attn(_ln(x, lnq, split=_ns(E, n_out=3 * E)), residual=x)
        z = xblk(queries=y, feats=text_embeddings, text_kv=text_kv)
        return _mlp(_ln(z, lnm, split=_ns(E, hid, n_out=hid)), mlp, residual=y)

    def forward_last(self, x, text_embeddings, n_last, text_kv=None):
        """
        Same block, output for the last ``n_last`` tokens of every sequence only: (B, T, E) ->
        (B, n_last, E).  Only self-attention mixes tokens; everything after it is row-wise, so it is
        evaluated on the n_last rows that are consumed.  Bit-for-bit the rows forward() would give.
        """
        B, T, E = x.shape
        x_last = K.contiguous(x[:, T - n_last:])
        y = self.attn.forward_last(_ln(x, self.layernorm_query), n_last, residual_last=x_last)
        z = self.condition_slots_given_caption(y, text_embeddings, text_kv=text_kv)
        return _mlp(_ln(z, self.layernorm_mlp), self.mlp, residual=y)

    def condition_slots_given_caption(self, slots_to_condition, text_embeddings, text_kv=None):
        return self.cross_attention(queries=slots_to_condition, feats=text_embeddings,
                                    text_kv=text_kv)
