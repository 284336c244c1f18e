"""
Autoregressive rollout wrapper.  Reference: models/Predictors/predictor_wrapper.py:20-170.
"""

import torch
import torch.nn as nn

from ... import kernels as K
from ..Blocks.model_utils import RangeGuard, refuse_replication, tracks_structure

__all__ = ["PredictorWrapper"]


@tracks_structure
class PredictorWrapper(nn.Module, RangeGuard):

    """
    Rolls a predictor out for ``num_preds`` steps over a sliding window of at most
    ``input_buffer_size`` frames, conditioned on the encoded caption (reference forward :50-87).

    Reference behaviour kept on purpose:
      * teacher forcing follows the experiment config in eval mode as well -- the reference's
        eval check compares a bound method with ``False`` and never fires (:136-139);
      * missing ``caption_tokens`` / ``caption_lengths`` raise KeyError (:97-98, :117-118);
      * the window starts with ``num_context`` frames, grows to the buffer size, then slides (:143-153).
    Host-side logic only: the window bookkeeping is slicing/concatenation of tiny (B, w, K, D)
    tensors; all arithmetic happens inside ``self.predictor`` on the HIP kernels.
    """

    _replicate_for_data_parallel = refuse_replication      # one process per GPU, never DataParallel replicas

    def __init__(self, exp_params, predictor):
        super().__init__()
        self.exp_params = exp_params
        self.predictor = predictor
        self.predictor_name = exp_params["predictor"]["predictor_name"]
        self.predictor_params = exp_params["predictor"]["predictor_params"]
        pp = exp_params["prediction_params"]
        self.num_context = pp["num_context"]
        self.num_preds = pp["num_preds"]
        self.teacher_force = pp["teacher_force"]
        self.input_buffer_size = pp["input_buffer_size"]
        if self.input_buffer_size is None:
            self.input_buffer_size = self.num_context
        self._init_range_guard()

    def forward(self, slot_history, num_preds=None, step_callback=None, **kwargs):
        return self._guarded(self._rollout, slot_history, num_preds=num_preds,
                             step_callback=step_callback, **kwargs)

    def _rollout(self, slot_history, num_preds=None, step_callback=None, **kwargs):
        """
        slot_history (B, T, K, D) -> pred_slots (B, num_preds, K, D).
        ``step_callback(t, pred_t)`` (extension) is invoked right after step t is enqueued, so a
        consumer (the evaluator's decoder on a second HIP stream) can start on frame t while the
        rollout continues.
        """
        self.teacher_force = self.exp_params["prediction_params"]["teacher_force"]
        num_preds = num_preds if num_preds is not None else self.num_preds
        text_embeddings = self.encode_text_caption(**kwargs)

        if not slot_history.is_cuda or slot_history.dtype != torch.float32 or slot_history.shape[-1] % 4:
            raise K.TocvpError("PredictorWrapper: the rollout runs on the GPU in fp32 (no CPU path)")
        # Window bookkeeping without torch kernels: every frame the window can hold lives in ONE buffer
        # (B, num_context + num_preds, K, D); the window of step t is a slice of it, made contiguous by the library's
        # strided copy (the reference concatenates and slices, predictor_wrapper.py:60-69, 143-153)
        B, _, Ks, D = slot_history.shape
        ctx = self.num_context
        buf = torch.empty((B, ctx + num_preds, Ks, D), device=slot_history.device, dtype=torch.float32)
        K.copy_strided(slot_history[:, :ctx], buf[:, :ctx])
        preds = []
        for t in range(num_preds):
            # the first window is ALL context frames (:60); from then on the newest input_buffer_size frames (:143-153)
            lo = max(0, ctx + t - self.input_buffer_size) if t else 0
            window = K.contiguous(buf[:, lo:ctx + t])
            cur = self.predictor(slots=window, time_step=t, text_embeddings=text_embeddings)
            nxt = slot_history[:, ctx + t] if self.teacher_force else cur
            K.copy_strided(nxt, buf[:, ctx + t])
            if self.teacher_force:
                preds.append(cur)
            if step_callback is not None:
                step_callback(t, cur)
        if self.teacher_force:                         # the window holds ground-truth slots: the predictions are apart
            return K.stack1(preds)
        return K.contiguous(buf[:, ctx:])

    def encode_text_caption(self, **kwargs):
        caption = kwargs.get("caption_tokens", None)
        if caption is None:
            raise KeyError("'caption_tokens' must be provided for the text-encoder.")
        if "T5" in self.predictor_name:
            attention_mask = kwargs.get("attn_masks", None)
            if attention_mask is None:
                raise KeyError("'attn_masks' must be provided for T5 Predictor")
            return self.predictor.text_encoder(input_ids=caption, attention_mask=attention_mask)
        if "CustomTF" in self.predictor_name:
            lengths = kwargs.get("caption_lengths", None)
            if lengths is None:
                raise KeyError("'caption_lengths' must be provided for CustomTF Pred.")
            return self.predictor.text_encoder(text=caption, text_length=lengths)
        return None

    def _update_buffer_size(self, inputs):
        extra = inputs.shape[1] - self.input_buffer_size
        return inputs[:, extra:] if extra > 0 else inputs
