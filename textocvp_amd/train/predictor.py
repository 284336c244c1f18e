"""
Differentiable TextOCVP_CustomTF rollout on the tape autograd (reference forward:
models/Predictors/predictor_wrapper.py:50-87, text_cond_OCVP.py:79-105, Blocks/attention.py:504-524,
EncodersDecoders/text_encoders.py:89-125; differentiated by hand here instead of torch.autograd).

The modules of ``textocvp_amd.models`` are the parameter containers: every ``nn.Parameter`` is wrapped
in a ``Var`` that aliases its storage, so the optimiser updates the model in place and the inference
path sees the trained weights.  Full back-propagation through time: predicted slots are fed back into
the window without detaching, as in the reference.
"""

import os

import torch

from .. import kernels as K
from . import autograd as ag

__all__ = ["TrainablePredictor"]


def _split_last(tape, x, n):
    """ x (..., n * E) -> n Vars (..., E) (contiguous copies); backward concatenates the gradients """
    E = x.data.shape[-1] // n
    parts = [ag.Var(x.data[..., i * E:(i + 1) * E].contiguous(), x.requires_grad) for i in range(n)]
    if x.requires_grad:
        def backward():
            if all(p.grad is None for p in parts):
                return
            g = torch.zeros_like(x.data)
            for i, p in enumerate(parts):
                if p.grad is not None:
                    g[..., i * E:(i + 1) * E] = p.grad
            ag.accumulate(x, g)
        tape.record(backward)
    return parts


class TrainablePredictor:
    def __init__(self, wrapper, precision="f16x3", text_dropout=None, generator=None):
        """
        text_dropout: dropout probability of the caption encoder in training (None = the module's own p,
        0.1 in the reference, text_encoders.py:36); 0 gives the deterministic (eval-mode) gradient.
        generator: torch.Generator on the device for the dropout samples.
        """
        self.wrapper = wrapper
        self.pred = wrapper.predictor
        kind = type(self.pred).__name__
        if kind not in ("TextOCVP_CustomTF", "TextOCVP_T5", "VanillaTransformerPredictor", "OCVPSeq"):
            raise NotImplementedError(f"training step: no differentiable rollout for predictor {kind!r}")
        self.kind = kind
        self.text_conditioned = kind.startswith("TextOCVP")
        # TextOCVP_T5: the pretrained T5 encoder is FROZEN in the reference (freeze_params, text_cond_OCVP.py:
        # 141-151): its embeddings come from the inference path, nothing back-propagates into it
        self.frozen_text = kind == "TextOCVP_T5"
        self.precision = precision
        # drop the LayerNorm outputs and MLP hidden activations after their forward use and rebuild them
        # in the backward pass.  Measured at B=32, K=30, 19 predictions: "0" 63.6 GB, "ln" 55.0 GB at no cost in time,
        # "all" 50.0 GB for +3 % time (every activation is also dropped as soon as its producer has back-propagated)
        self.last_layer_newest_frame_only = os.environ.get("TOCVP_LAST_LAYER_SUBSET", "1") != "0"
        mode = os.environ.get("TOCVP_TRAIN_RECOMPUTE", "ln")           # "ln" | "mlp" | "all" | "0"
        self.recompute_ln = mode in ("all", "1", "ln")
        self.recompute_mlp = mode in ("all", "1", "mlp")
        if self.text_conditioned:
            self.text_dropout = 0.0 if self.frozen_text else float(
                self.pred.text_encoder.dropout.p if text_dropout is None else text_dropout)
        else:           # the unconditioned predictors: dropout of their nn.TransformerEncoderLayer blocks (0.1)
            layer = self.pred.transformer_encoders[0]
            layer = getattr(layer, "object_encoder_block", layer)
            self.text_dropout = float(layer.dropout.p if text_dropout is None else text_dropout)
        self.generator = generator
        self.vars = {}
        self.names = {}
        self.params = {}
        self.all_names = [name for name, _ in wrapper.named_parameters()]   # torch.optim's parameter order
        for name, p in wrapper.named_parameters():
            if not p.requires_grad:                  # frozen (the T5 encoder): no gradient, no optimiser state
                continue
            v = ag.Var(p.data, requires_grad=True, name=name)
            self.vars[id(p)] = v
            self.names[name] = v
            self.params[name] = p

    def V(self, param):
        return self.vars[id(param)]

    def zero_grad(self):
        for v in self.vars.values():
            v.grad = None

    def mark_updated(self):
        """ the optimiser kernel wrote the weights through raw pointers: bump the tensor version counters so
        that every cache derived from a weight (operand planes, fused / packed copies) is rebuilt """
        bump = getattr(torch._C, "_increment_version", None)
        for name, p in self.params.items():
            for t in (p, self.names[name].data):
                if bump is not None:
                    bump(t)
                else:
                    t.add_(0)

    # ---------------------------------------------------------------------------------------
    def _lin(self, tape, x, mod, act=K.ACT_NONE, residual=None):
        return ag.linear(tape, x, self.V(mod.weight), None if mod.bias is None else self.V(mod.bias), act=act,
                         precision=self.precision, residual=residual)

    def _ln(self, tape, x, mod):
        return ag.layer_norm(tape, x, self.V(mod.weight), self.V(mod.bias), mod.eps)

    def encode_text(self, tape, tokens, lengths):
        """ TransformerTextEncoder.forward (text_encoders.py:89-125) """
        te = self.pred.text_encoder
        B, L = tokens.shape
        key_len = lengths.to(device=tokens.device, dtype=torch.int32).contiguous()
        x = ag.embedding(tape, tokens, self.V(te.token_embedding.weight))
        x = ag.add_position_rows(tape, x, self.V(te.position_embedding.weight), list(range(L)))
        pd, gen = self.text_dropout, self.generator

        def drop(t):
            return ag.dropout(tape, t, pd, generator=gen)
        x = drop(self._ln(tape, x, te.layer_norm))
        x = ag.mask_rows(tape, x, tokens != 0)
        E = x.data.shape[-1]
        H = te.num_heads
        for layer in te.transformer.layers:                       # post-norm nn.TransformerEncoderLayer, GELU
            sa = layer.self_attn
            qkv = ag.linear(tape, x, self.V(sa.in_proj_weight), self.V(sa.in_proj_bias), precision=self.precision)
            q, k, v = _split_last(tape, qkv, 3)
            if pd > 0.0:
                a = ag.attention_unfused(tape, q, k, v, H, (E // H) ** -0.5, key_len=key_len, p_drop=pd,
                                         generator=gen)
            else:
                a = ag.attention(tape, q, k, v, H, (E // H) ** -0.5, key_len=key_len)
            x = self._ln(tape, ag.add(tape, drop(self._lin(tape, a, sa.out_proj)), x), layer.norm1)
            h = drop(self._lin(tape, x, layer.linear1, act=K.ACT_GELU))
            x = self._ln(tape, ag.add(tape, drop(self._lin(tape, h, layer.linear2)), x), layer.norm2)
        return self._lin(tape, self._ln(tape, x, te.text_out_projection[0]), te.text_out_projection[1])

    def _self_attention(self, tape, x, attn, residual=None):
        E = x.data.shape[-1]
        q, k, v = (self._lin(tape, x, m) for m in (attn.q, attn.k, attn.v))
        o = ag.attention(tape, q, k, v, attn.num_heads, (E // attn.num_heads) ** -0.5)
        return self._lin(tape, o, attn.out_projection[0], residual=residual)

    def _mlp(self, tape, x, seq, residual=None):
        """ Linear-ReLU-Linear; neither the normalised input nor the hidden activation is kept for the
        backward pass (rebuilt there by one LayerNorm / one GEMM: ``Var.release``) """
        h = self._lin(tape, x, seq[0], act=K.ACT_RELU)
        h.single_use = True                     # the ReLU mask is applied by the GEMM that produces h.grad
        y = self._lin(tape, h, seq[2], residual=residual)
        if self.recompute_mlp:
            h.release()
        if self.recompute_ln:
            x.release()
        return y

    def _prenorm_layer(self, tape, x, layer, heads):
        """ nn.TransformerEncoderLayer(norm_first=True, batch_first=True, relu) in training mode (the blocks of the
        unconditioned predictors, OCVP.py:60-75): x (N, T, E).  Dropout (p of the layer; 0 = eval-mode gradient) on
        the attention probabilities, after both sub-layers and inside the feed-forward. """
        pd, gen = self.text_dropout, self.generator
        E = x.data.shape[-1]
        sa = layer.self_attn
        h = self._ln(tape, x, layer.norm1)
        qkv = ag.linear(tape, h, self.V(sa.in_proj_weight), self.V(sa.in_proj_bias), precision=self.precision)
        q, k, v = _split_last(tape, qkv, 3)
        if pd > 0.0:
            a = ag.attention_unfused(tape, q, k, v, heads, (E // heads) ** -0.5, p_drop=pd, generator=gen)
            x = ag.add(tape, ag.dropout(tape, self._lin(tape, a, sa.out_proj), pd, generator=gen), x)
            h = ag.dropout(tape, self._lin(tape, self._ln(tape, x, layer.norm2), layer.linear1, act=K.ACT_RELU), pd,
                           generator=gen)
            return ag.add(tape, ag.dropout(tape, self._lin(tape, h, layer.linear2), pd, generator=gen), x)
        a = ag.attention(tape, q, k, v, heads, (E // heads) ** -0.5)
        x = self._lin(tape, a, sa.out_proj, residual=x)
        h = self._lin(tape, self._ln(tape, x, layer.norm2), layer.linear1, act=K.ACT_RELU)
        return self._lin(tape, h, layer.linear2, residual=x)

    def step_unconditioned(self, tape, window):
        """ VanillaTransformerPredictor.forward (OCVP.py:100-132) / OCVPSeq.forward (:222-254, layer :301-320) """
        p = self.pred
        w = len(window)
        slots = ag.stack_frames(tape, window)                                   # (B, w, K, D)
        B, _, Ks, _ = slots.data.shape
        tokens = self._lin(tape, slots, p.mlp_in)                               # (B, w, K, E)
        E = tokens.data.shape[-1]
        if p._pe.device != tokens.data.device:
            p._pe = p._pe.to(tokens.data.device)
        tokens = ag.add_position_rows(tape, tokens, ag.Var(p._pe.contiguous()), list(range(w)))   # sinusoid, not flipped
        if self.kind == "VanillaTransformerPredictor":
            x = ag.reshape(tape, tokens, (B, w * Ks, E))
            for layer in p.transformer_encoders:
                x = self._prenorm_layer(tape, x, layer, p.nhead)
            last = ag.take_last_tokens(tape, x, Ks)
        else:
            x = tokens
            for layer in p.transformer_encoders:
                x = self._prenorm_layer(tape, ag.reshape(tape, x, (B * w, Ks, E)), layer.object_encoder_block, p.nhead)
                x = ag.transpose12(tape, ag.reshape(tape, x, (B, w, Ks, E)))            # (B, K, w, E)
                x = self._prenorm_layer(tape, ag.reshape(tape, x, (B * Ks, w, E)), layer.time_encoder_block, p.nhead)
                x = ag.transpose12(tape, ag.reshape(tape, x, (B, Ks, w, E)))            # (B, w, K, E)
            last = ag.take_last_tokens(tape, ag.reshape(tape, x, (B, w * Ks, E)), Ks)
        out = self._lin(tape, last, p.mlp_out)
        return ag.add(tape, out, window[-1]) if p.residual else out

    def text_kv(self, tape, text):
        """ per layer: cross-attention keys / values of the caption.  They do not depend on the rollout step,
        so they are projected ONCE per sequence (as the inference path does, BaseTextOCVP.prepare_text); every
        step's attention adds into the same k / v gradients and the projections back-propagate once. """
        out = []
        for blk in self.pred.predictor:
            cb = blk.cross_attention
            tn = self._ln(tape, text, cb.ln_cross_att_kv)
            out.append((self._lin(tape, tn, cb.cross_attn.k), self._lin(tape, tn, cb.cross_attn.v)))
        return out

    def _after_self_attention(self, tape, y, blk, kv):
        """ the row-wise rest of AdaptedEncoderBlock.forward (attention.py:512-524): text cross-attention block,
        then the MLP whose residual comes from y (the reference's wiring) """
        cb = blk.cross_attention
        ca = cb.cross_attn
        yq = self._ln(tape, y, cb.ln_cross_att_q)
        q = self._lin(tape, yq, ca.q)
        if self.recompute_ln:
            yq.release()
        o = ag.attention(tape, q, kv[0], kv[1], ca.num_heads, ca.dim_head ** -0.5)   # padded text attends too
        z1 = self._lin(tape, o, ca.out_projection, residual=y)
        z = self._mlp(tape, self._ln(tape, z1, cb.ln_mlp), cb.mlp, residual=z1)
        return self._mlp(tape, self._ln(tape, z, blk.layernorm_mlp), blk.mlp, residual=y)

    def _block(self, tape, x, blk, kv):
        """ AdaptedEncoderBlock.forward (attention.py:504-524).  Every residual sum is the epilogue of the
        GEMM that produces the other addend. """
        xq = self._ln(tape, x, blk.layernorm_query)
        y = self._self_attention(tape, xq, blk.attn, residual=x)
        if self.recompute_ln:
            xq.release()
        return self._after_self_attention(tape, y, blk, kv)

    def _block_last(self, tape, x, blk, kv, n):
        """
        The final layer: only the newest frame's n tokens of its output are read (text_cond_OCVP.py:101-103),
        and only self-attention mixes tokens -- keys / values over the whole window, everything else on the
        n rows that are consumed (the inference path's AdaptedEncoderBlock.forward_last).  The dropped rows
        carry exactly zero gradient in the reference as well.
        """
        attn = blk.attn
        E = x.data.shape[-1]
        xq = self._ln(tape, x, blk.layernorm_query)
        k, v = self._lin(tape, xq, attn.k), self._lin(tape, xq, attn.v)
        q = self._lin(tape, ag.take_last_tokens(tape, xq, n), attn.q)
        if self.recompute_ln:
            xq.release()
        o = ag.attention(tape, q, k, v, attn.num_heads, (E // attn.num_heads) ** -0.5)
        y = self._lin(tape, o, attn.out_projection[0], residual=ag.take_last_tokens(tape, x, n))
        return self._after_self_attention(tape, y, blk, kv)

    def step(self, tape, window, text_kv):
        """ BaseTextOCVP.forward: window = list of frame Vars (B, K, D) -> next-slot Var (B, K, D);
        text_kv = self.text_kv(tape, text) """
        p = self.pred
        w = len(window)
        slots = ag.stack_frames(tape, window)                                   # (B, w, K, D)
        B, _, Ks, _ = slots.data.shape
        tokens = self._lin(tape, slots, p.mlp_in)                               # (B, w, K, E)
        pe = self.V(p.pe.pe)
        table = ag.Var(pe.data.reshape(-1, pe.data.shape[-1]), pe.requires_grad)
        if pe.grad is None:
            pe.grad = torch.zeros_like(pe.data)
        table.grad = pe.grad.reshape(table.data.shape)                          # same storage as pe.grad
        tokens = ag.add_position_rows(tape, tokens, table, [w - 1 - i for i in range(w)])
        E = tokens.data.shape[-1]
        x = ag.Var(tokens.data.reshape(B, w * Ks, E), tokens.requires_grad)
        tape.record(lambda xv=x, tv=tokens: ag.accumulate(tv, xv.grad) if xv.grad is not None else None)
        nblk = len(p.predictor)
        for i, (blk, kv) in enumerate(zip(p.predictor, text_kv)):
            if i == nblk - 1 and self.last_layer_newest_frame_only:
                x = self._block_last(tape, x, blk, kv, Ks)                      # (B, K, E)
            else:
                x = self._block(tape, x, blk, kv)
        last = x if nblk and self.last_layer_newest_frame_only else ag.take_last_tokens(tape, x, Ks)
        out = self._lin(tape, last, p.mlp_out)
        return ag.add(tape, out, window[-1]) if p.residual else out

    def rollout(self, tape, slot_history, tokens, lengths, num_preds=None, attn_masks=None):
        """ PredictorWrapper.forward: returns the list of predicted-slot Vars (B, K, D).
        TextOCVP_T5: ``tokens`` are the T5 input ids and ``attn_masks`` their attention mask (lengths unused). """
        wr = self.wrapper
        num_preds = wr.num_preds if num_preds is None else num_preds
        text_kv = None
        if self.text_conditioned:
            if self.frozen_text:
                text = ag.Var(wr.encode_text_caption(caption_tokens=tokens, attn_masks=attn_masks).contiguous())
            else:
                text = self.encode_text(tape, tokens, lengths)
            text_kv = self.text_kv(tape, text)
        window = [ag.Var(slot_history[:, i].contiguous()) for i in range(wr.num_context)]
        preds = []
        teacher = wr.exp_params["prediction_params"]["teacher_force"]
        for t in range(num_preds):
            cur = self.step(tape, window, text_kv) if self.text_conditioned else self.step_unconditioned(tape, window)
            nxt = ag.Var(slot_history[:, wr.num_context + t].contiguous()) if teacher else cur
            window = (window + [nxt])[-wr.input_buffer_size:]
            preds.append(cur)
        return preds
