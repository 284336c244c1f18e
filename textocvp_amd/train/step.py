"""
One optimisation step of the TextOCVP predictor (reference 04_train_predictor.py:57-108):

  slot_history = SAVi.decomp(videos)                      (frozen, no gradient)
  pred_slots   = autoregressive rollout                    (TrainablePredictor, BPTT)
  pred_imgs    = SAVi.decode(pred_slots)                   (frozen decoder, gradient w.r.t. the slots)
  loss = w_img * MSE(pred_imgs, target_imgs) + w_slot * MSE(pred_slots, target_slots)   (CONFIG.py:42-51)
  clip_grad_norm_(0.05) -> Adam(lr 1e-4) with linear warm-up + cosine annealing (lib/setup_model.py:285-332)

Data-parallel: every rank runs the step on its own batches, gradients are averaged with ONE all-reduce
of a flat buffer (RCCL over xGMI through torch.distributed) before clipping, like DistributedDataParallel.
"""

import math
from collections.abc import Mapping

import torch
import torch.distributed as dist

from .. import kernels as K
from . import autograd as ag
from .decoder import DecoderLoss
from .predictor import TrainablePredictor

__all__ = ["PredictorTrainStep", "StepResult"]

_L = K.lib


def _s():
    return torch.cuda.current_stream().cuda_stream


class StepResult(Mapping):
    """
    Losses / gradient norm / lr of one graph-replayed step, read back from the device on first access: the
    host does not wait for the step it has just queued, so the launch of the next step's graphs (tens of
    milliseconds of host work for ~13 thousand kernel nodes) overlaps the execution of this one.
    Mapping with keys loss, pred_slot_mse, pred_img_mse, grad_norm, lr.
    """
    _KEYS = ("loss", "pred_slot_mse", "pred_img_mse", "grad_norm", "lr")

    def __init__(self, snap, sc_slot, sc_img, lr):
        self._snap, self._sc, self._lr, self._vals = snap, (sc_slot, sc_img), lr, None

    def _get(self):
        if self._vals is None:
            sq_slot, sq_img, norm = (float(v) for v in self._snap.tolist())     # synchronises on the snapshot
            loss_slot, loss_img = sq_slot * self._sc[0], sq_img * self._sc[1]
            self._vals = {"loss": loss_slot + loss_img, "pred_slot_mse": loss_slot, "pred_img_mse": loss_img,
                          "grad_norm": norm, "lr": self._lr}
            self._snap = None
        return self._vals

    def __getitem__(self, key):
        return self._get()[key]

    def __iter__(self):
        return iter(self._KEYS)

    def __len__(self):
        return len(self._KEYS)


class PredictorTrainStep:
    def __init__(self, savi, wrapper, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, clip=0.05,
                 loss_weights=(1.0, 1.0), warmup_steps=2000, scheduler_steps=1e6, eta_min=1e-7,
                 process_group=None, text_dropout=None, generator=None):
        self.savi, self.wrapper = savi.eval(), wrapper
        self.model = TrainablePredictor(wrapper, text_dropout=text_dropout, generator=generator)
        self.decoder = DecoderLoss(savi)
        self.lr, self.betas, self.eps, self.clip = lr, betas, eps, clip
        self.w_img, self.w_slot = loss_weights
        self.warmup_steps, self.scheduler_steps, self.eta_min = warmup_steps, scheduler_steps, eta_min
        self.group = process_group
        self.iteration = 0                            # optimiser steps taken so far
        self.state = {}                               # parameter name -> (exp_avg, exp_avg_sq)

    # ---------------------------------------------------------------------------------------
    def lr_at(self, iter_):
        """
        Learning rate the reference trainer applies to the optimiser step of 0-based iteration ``iter_``
        (base/basePredictorTrainer.py:280-286 calls WarmupVSScehdule BEFORE the step; lib/schedulers.py:
        88-107, 140-157): lr * iter_ / warmup_steps while iter_ <= warmup_steps (so the very first step
        runs at lr 0), lr at iter_ = warmup_steps + 1 (the warm-up object deactivates itself without
        touching the optimiser), then one CosineAnnealingLR.step() per iteration:
        eta_min + (lr - eta_min) (1 + cos(pi k / T_max)) / 2 with k = iter_ - warmup_steps - 1.
        No warm-up is warmup_steps = -1 in the reference (lib/setup_model.py:356); None / 0 mean that here.
        """
        W = self.warmup_steps if self.warmup_steps and self.warmup_steps > 0 else -1
        if iter_ <= W:
            return self.lr * iter_ / W
        k = iter_ - W - 1
        return self.eta_min + (self.lr - self.eta_min) * (1.0 + math.cos(math.pi * k / self.scheduler_steps)) / 2.0

    @torch.no_grad()
    def _forward_backward(self, videos, caption_tokens, caption_lengths, others):
        """ forward + backward on the current stream without any host synchronisation; returns the two
        sums of squares as (1,) device tensors and their scales """
        wr = self.wrapper
        nc, npred = wr.num_context, wr.num_preds
        B, L, C, H, W = videos.shape
        if L < nc + npred:
            raise ValueError(f"Seq. length {L} must be >= {nc + npred = }")
        videos = videos[:, :nc + npred]
        others = dict(others)
        attn_masks = others.pop("attn_masks", None)              # TextOCVP_T5 (predictor_wrapper.py:101-111)
        hist = self.savi(mode="decomp", x=videos, num_imgs=nc + npred, decode=False, **others)["slot_history"]
        self.model.zero_grad()
        tape = ag.Tape()
        preds = self.model.rollout(tape, hist, caption_tokens, caption_lengths, npred, attn_masks=attn_masks)
        stacked = ag.stack_frames(tape, preds)                              # (B, P, K, D)
        Ks, D = stacked.data.shape[2:]
        tgt_slots = hist[:, nc:nc + npred].contiguous()
        tgt_imgs = videos[:, nc:nc + npred].reshape(B * npred, C, H, W).contiguous()
        sq_slot, sc_slot = ag.mse(tape, stacked, tgt_slots, weight=self.w_slot)
        n_img = tgt_imgs.numel()
        sq_img, dslots = self.decoder.loss_and_slot_grad(stacked.data.reshape(B * npred, Ks, D), tgt_imgs,
                                                         grad_scale=2.0 * self.w_img / n_img)
        tape.record(lambda: ag.accumulate(stacked, dslots.reshape(stacked.data.shape)))
        tape.backward()
        self._grads()                                                       # materialise missing gradients
        return sq_slot, sc_slot, sq_img, self.w_img / n_img

    def _range_checked_pass(self, videos, caption_tokens, caption_lengths, others):
        """
        The FIRST eager forward + backward (after construction or ``load_training_state``) verifies every operand of
        the fp16-plane kernels (|activation| < 255, |weight| < 63; they saturate beyond, kernels.TocvpRangeError)
        and moves the arithmetic that trips to its fp32-exponent-range fallback before anything is trained on a
        saturated value: predictor GEMMs f16x3 -> bf16x6, attention products -> exact fp32, frozen decoder convs
        f16x3 -> bf16x3.  Later steps (and the captured graphs) run unchecked, like the inference path after
        ``calibrate_precision``.
        """
        import warnings
        for _ in range(4):
            try:
                with K.check_range(True):
                    return self._forward_backward(videos, caption_tokens, caption_lengths, others)
            except K.TocvpRangeError as err:
                msg = str(err)
                mod, attr = err.owner if isinstance(err.owner, tuple) and len(err.owner) == 2 else (None, None)
                table = getattr(type(mod), "range_fallbacks", {}).get(attr) if mod is not None else None
                if table and getattr(mod, attr, None) in table:          # a module of the inference path (frozen
                    setattr(mod, attr, table[getattr(mod, attr)])        # SAVi, T5 encoder) named its own knob
                elif "GEMM" in msg and self.model.precision == "f16x3":
                    self.model.precision = "bf16x6"
                elif "attention" in msg and K._ATTN_QK16:
                    K._ATTN_QK16 = False
                elif "conv" in msg and self.decoder.dec.conv_precision == "f16x3":
                    self.decoder.dec.conv_precision = "bf16x3"
                else:
                    raise
                warnings.warn(f"training step: {msg} -- switched that arithmetic to its fp32-range fallback")
        raise K.TocvpError("training step: operands out of the fp16-plane range after every fallback")

    def loss_and_grads(self, videos, caption_tokens, caption_lengths, **others):
        """ forward + backward; leaves the gradients in ``self.model.names[*].grad``; returns the losses """
        if not getattr(self, "_range_ok", False):
            sq_slot, sc_slot, sq_img, sc_img = self._range_checked_pass(videos, caption_tokens, caption_lengths, others)
            self._range_ok = True
        else:
            sq_slot, sc_slot, sq_img, sc_img = self._forward_backward(videos, caption_tokens, caption_lengths,
                                                                      others)
        loss_slot, loss_img = float(sq_slot.item()) * sc_slot, float(sq_img.item()) * sc_img
        return {"loss": loss_slot + loss_img, "pred_slot_mse": loss_slot, "pred_img_mse": loss_img}

    # ---------------------------------------------------------------------------------------
    def _grads(self):
        out = []
        for name, v in self.model.names.items():
            if v.grad is None:
                v.grad = torch.zeros_like(v.data)
            out.append(v)
        return out

    def all_reduce_grads(self, force=False):
        """ average the gradients over the ranks with one all-reduce of a flat buffer (``force``: also with a
        single rank, where it is the identity -- exercises the collective itself on a one-GPU box) """
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.group)
        if world == 1 and not force:
            return
        vs = self._grads()
        flat = torch.cat([v.grad.reshape(-1) for v in vs])
        if dist.get_backend(self.group) == "gloo" and flat.is_cuda:
            host = flat.cpu()
            dist.all_reduce(host, group=self.group)
            flat = host.to(flat.device)
        else:
            dist.all_reduce(flat, group=self.group)
        off = 0
        for v in vs:
            n = v.grad.numel()
            ag.axpby(flat[off:off + n].contiguous(), v.grad.reshape(-1), 1.0 / world, 0.0)
            off += n

    def _clip_scale(self):
        """ device side: (2,) tensor {clipping factor, gradient norm} (clip_grad_norm_) """
        vs = self._grads()
        parts = []
        for v in vs:
            n = v.grad.numel()
            nb = min(256, (n + 255) // 256)
            p = torch.empty(nb, device=v.grad.device, dtype=torch.float32)
            K._check(_L().tocvp_sqnorm_partial_f32(v.grad.data_ptr(), p.data_ptr(), nb, n, _s()),
                     "tocvp_sqnorm_partial_f32")
            parts.append(p)
        total = ag.colsum(torch.cat(parts).reshape(-1, 1))
        out = torch.empty(2, device=total.device, dtype=torch.float32)
        K._check(_L().tocvp_clip_scale_f32(total.data_ptr(), float(self.clip or 0.0), out.data_ptr(), _s()),
                 "tocvp_clip_scale_f32")
        return out

    def grad_norm(self):
        return float(self._clip_scale()[1].item())

    def _hyper(self, it):
        """ scalars of optimiser step number ``it`` (1-based): the reference's iter_ is it - 1 """
        b1, b2 = self.betas
        return [self.lr_at(it - 1), b1, b2, self.eps, 1.0 - b1 ** it, 1.0 - b2 ** it]

    def _optimizer_kernels(self):
        """ clip factor + Adam on every parameter, reading the step scalars from ``self._hyper_dev`` """
        clipn = self._clip_scale()
        for name, v in self.model.names.items():
            if v.grad is None:
                v.grad = torch.zeros_like(v.data)
            st = self.state.get(name)
            if st is None:
                st = (torch.zeros_like(v.data), torch.zeros_like(v.data))
                self.state[name] = st
            K._check(_L().tocvp_adam_f32(v.data.data_ptr(), v.grad.data_ptr(), st[0].data_ptr(), st[1].data_ptr(),
                                         v.data.numel(), self._hyper_dev.data_ptr(), clipn.data_ptr(), _s()),
                     "tocvp_adam_f32")
        return clipn

    def _set_hyper(self):
        self.iteration += 1
        vals = self._hyper(self.iteration)
        host = torch.tensor(vals, dtype=torch.float32).pin_memory()     # pinned: the copy below does not make
        if getattr(self, "_hyper_dev", None) is None:                   # the host wait for the queued step
            self._hyper_dev = host.to(next(iter(self.model.names.values())).data.device)
        else:
            self._hyper_dev.copy_(host, non_blocking=True)
        return float(vals[0])

    # ---- checkpoint state, in the reference's formats (lib/setup_model.py:176-184, 228-240) --------------
    def optimizer_state_dict(self):
        """ torch.optim.Adam.state_dict() layout over PredictorWrapper.parameters() order """
        names = list(self.model.all_names)                    # frozen parameters keep their index, without state
        state = {}
        for i, name in enumerate(names):
            if name in self.state:
                m, v = self.state[name]
                state[i] = {"step": torch.tensor(float(self.iteration)), "exp_avg": m.clone(),
                            "exp_avg_sq": v.clone()}
        group = {"lr": self.lr_at(max(self.iteration - 1, 0)), "betas": tuple(self.betas), "eps": self.eps,
                 "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "differentiable": False, "fused": None, "initial_lr": self.lr, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def scheduler_state_dict(self):
        """ the CosineAnnealingLR fields the reference's checkpoint carries (steps taken after the warm-up) """
        W = self.warmup_steps if self.warmup_steps and self.warmup_steps > 0 else -1
        k = max(self.iteration - 1 - W - 1, 0)
        return {"T_max": self.scheduler_steps, "eta_min": self.eta_min, "base_lrs": [self.lr], "last_epoch": k,
                "_step_count": k + 1, "_last_lr": [self.lr_at(max(self.iteration - 1, 0))]}

    def lr_warmup_state_dict(self):
        """ LRWarmUp.state_dict() (lib/schedulers.py:109-114) """
        W = self.warmup_steps if self.warmup_steps and self.warmup_steps > 0 else -1
        active = self.iteration - 1 <= W
        return {"init_lr": self.lr, "warmup_steps": W, "active": active, "final_step": -1 if active else W + 1}

    def state_dict(self, epoch=0):
        """ everything save_checkpoint stores (lib/setup_model.py:178-184) plus the exact step counter """
        return {"epoch": epoch, "model_state_dict": self.wrapper.state_dict(),
                "optimizer_state_dict": self.optimizer_state_dict(),
                "scheduler_state_dict": self.scheduler_state_dict(), "lr_warmup": self.lr_warmup_state_dict(),
                "iteration": self.iteration}

    def load_training_state(self, ckpt):
        """
        Resume from ``state_dict()`` or from a checkpoint written by the reference trainer (its optimizer /
        scheduler / lr_warmup entries; the step count then comes from Adam's per-parameter ``step``).
        The model weights are loaded by setup_model.load_checkpoint.
        """
        opt = ckpt["optimizer_state_dict"]
        names = list(self.model.all_names)
        steps = 0
        for i, st in opt["state"].items():
            name = names[int(i)]
            ref = self.model.names[name].data
            m = st["exp_avg"].to(device=ref.device, dtype=torch.float32).reshape(ref.shape).contiguous().clone()
            v = st["exp_avg_sq"].to(device=ref.device, dtype=torch.float32).reshape(ref.shape).contiguous().clone()
            self.state[name] = (m, v)
            steps = max(steps, int(float(st["step"])))
        self.iteration = int(ckpt.get("iteration", steps))
        self._graphs = None                                   # moments were re-allocated: re-capture
        self._range_ok = False                                # new weights: one range-checked pass again
        return self

    def apply(self):
        """ clip_grad_norm_ + Adam on every predictor parameter; returns (grad norm, lr used) """
        lr = self._set_hyper()
        clipn = self._optimizer_kernels()
        self.model.mark_updated()
        return float(clipn[1].item()), lr

    def step(self, videos, caption_tokens, caption_lengths, **others):
        losses = self.loss_and_grads(videos, caption_tokens, caption_lengths, **others)
        self.all_reduce_grads()
        norm, lr = self.apply()
        losses.update(grad_norm=norm, lr=lr)
        return losses

    # ---------------------------------------------------------------------------------------
    def step_graphed(self, videos, caption_tokens, caption_lengths, **others):
        """
        Same step replayed from two captured HIP graphs (forward + backward; clipping + Adam), with the
        gradient all-reduce between them.  The step issues ~20-35 thousand small launches and is bound by
        the host otherwise.  Shapes must stay fixed; the first call runs one eager step (fills every
        cache and allocation) and captures, later calls copy the batch into the static inputs and replay.
        Dropout samples come from torch's graph-safe generator state, so every replay draws new masks.
        Returns a StepResult: the numbers are fetched when first read, not here.
        """
        if getattr(self, "_graphs", None) is None:
            warm = self.step(videos, caption_tokens, caption_lengths, **others)   # eager step (fills caches)
            self._static = [videos.clone(), None if caption_tokens is None else caption_tokens.clone(),
                            None if caption_lengths is None else caption_lengths.clone(),
                            {k: (v.clone() if torch.is_tensor(v) else v) for k, v in others.items()}]
            torch.cuda.synchronize()
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            gen = self.model.generator
            if gen is not None and hasattr(g1, "register_generator_state"):
                g1.register_generator_state(gen)
            # thread-local capture mode: the RCCL watchdog of a live process group must not invalidate the capture
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                self._static_out = self._forward_backward(*self._static)
            self._set_hyper()
            self.iteration -= 1                                               # capture does not count as a step
            with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
                self._static_clip = self._optimizer_kernels()
            self._graphs = (g1, g2)
            return warm                        # capturing records kernels, it does not run them
        sv, st, sl, so = self._static
        sv.copy_(videos)
        if st is not None:
            st.copy_(caption_tokens)
        if sl is not None:
            sl.copy_(caption_lengths)
        for k, v in others.items():
            if torch.is_tensor(v):
                so[k].copy_(v)
        self._graphs[0].replay()
        self.all_reduce_grads()
        lr = self._set_hyper()
        self._graphs[1].replay()
        self.model.mark_updated()
        sq_slot, sc_slot, sq_img, sc_img = self._static_out
        # the static outputs are overwritten by the next replay: snapshot them (device side, no wait)
        snap = torch.cat([sq_slot.reshape(1), sq_img.reshape(1), self._static_clip[1:2]])
        return StepResult(snap, sc_slot, sc_img, lr)
