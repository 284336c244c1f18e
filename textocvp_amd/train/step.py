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

import torch
import torch.distributed as dist

from .. import kernels as K
from . import autograd as ag
from .decoder import DecoderLoss
from .predictor import TrainablePredictor

__all__ = ["PredictorTrainStep"]

_L = K.lib


def _s():
    return torch.cuda.current_stream().cuda_stream


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
        self.iteration = 0
        self.state = {}                               # id(Var) -> (m, v)

    # ---------------------------------------------------------------------------------------
    def lr_at(self, it):
        """ linear warm-up to lr over warmup_steps, then CosineAnnealingLR(T_max=scheduler_steps, eta_min) """
        if self.warmup_steps and it <= self.warmup_steps:
            return self.lr * it / self.warmup_steps
        t = it - (self.warmup_steps or 0)
        return self.eta_min + (self.lr - self.eta_min) * (1.0 + math.cos(math.pi * t / self.scheduler_steps)) / 2.0

    @torch.no_grad()
    def loss_and_grads(self, videos, caption_tokens, caption_lengths, **others):
        """ forward + backward; leaves the gradients in ``self.model.names[*].grad``; returns the losses """
        wr = self.wrapper
        nc, npred = wr.num_context, wr.num_preds
        B, L, C, H, W = videos.shape
        if L < nc + npred:
            raise ValueError(f"Seq. length {L} must be >= {nc + npred = }")
        videos = videos[:, :nc + npred]
        hist = self.savi(mode="decomp", x=videos, num_imgs=nc + npred, decode=False, **others)["slot_history"]
        self.model.zero_grad()
        tape = ag.Tape()
        preds = self.model.rollout(tape, hist, caption_tokens, caption_lengths, npred)
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
        loss_slot = float(sq_slot.item()) * sc_slot
        loss_img = float(sq_img.item()) * self.w_img / n_img
        return {"loss": loss_slot + loss_img, "pred_slot_mse": loss_slot, "pred_img_mse": loss_img}

    # ---------------------------------------------------------------------------------------
    def _grads(self):
        out = []
        for name, v in self.model.names.items():
            if v.grad is None:
                v.grad = torch.zeros_like(v.data)
            out.append(v)
        return out

    def all_reduce_grads(self):
        """ average the gradients over the ranks with one all-reduce of a flat buffer """
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.group)
        if world == 1:
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

    def grad_norm(self):
        vs = self._grads()
        parts = []
        for v in vs:
            n = v.grad.numel()
            nb = min(256, (n + 255) // 256)
            p = torch.empty(nb, device=v.grad.device, dtype=torch.float32)
            K._check(_L().tocvp_sqnorm_partial_f32(v.grad.data_ptr(), p.data_ptr(), nb, n, _s()),
                     "tocvp_sqnorm_partial_f32")
            parts.append(p)
        allp = torch.cat(parts)
        return math.sqrt(float(ag.colsum(allp.reshape(-1, 1)).item()))

    def apply(self):
        """ clip_grad_norm_ + Adam on every predictor parameter; returns (grad norm, lr used) """
        self.iteration += 1
        norm = self.grad_norm()
        gscale = 1.0
        if self.clip is not None:
            gscale = min(1.0, self.clip / (norm + 1e-6))
        lr = self.lr_at(self.iteration)
        for v in self._grads():
            st = self.state.get(id(v))
            if st is None:
                st = (torch.zeros_like(v.data), torch.zeros_like(v.data))
                self.state[id(v)] = st
            K._check(_L().tocvp_adam_f32(v.data.data_ptr(), v.grad.data_ptr(), st[0].data_ptr(), st[1].data_ptr(),
                                         v.data.numel(), float(lr), float(self.betas[0]), float(self.betas[1]),
                                         float(self.eps), self.iteration, float(gscale), _s()), "tocvp_adam_f32")
        self.model.mark_updated()
        return norm, lr

    def step(self, videos, caption_tokens, caption_lengths, **others):
        losses = self.loss_and_grads(videos, caption_tokens, caption_lengths, **others)
        self.all_reduce_grads()
        norm, lr = self.apply()
        losses.update(grad_norm=norm, lr=lr)
        return losses
