"""
Frozen SAVi decoder for the image-loss term of the predictor training step: forward of SAVi.decode
(SAVi.py:241-275, decoders.py:85-120) that keeps the layer activations of one chunk, the per-pixel MSE
gradient, and the hand-written backward w.r.t. the slots (the decoder's own weights are frozen,
04_train_predictor.py:62-75).  Chunk-local: the MSE gradient of a pixel depends on that pixel only, so
every chunk of frames is decoded, differentiated and dropped before the next one.

  slots --(tap-sum GEMM)--> S --relu(cpos + S[cls])--> conv1+relu --> conv2+relu --> conv3+relu
        --> 3x3 tail --> softmax over slots / compositing --> img --> sum (img - target)^2
Data gradients of the 5x5 convs = the same conv kernels with transposed, flipped weights (bf16x3: the
bf16 planes keep the fp32 exponent range, which matters for small gradients).
"""

import os

import torch

from .. import kernels as K
from . import autograd as ag

__all__ = ["DecoderLoss"]

_L = K.lib
# 0: data gradients of the decoder convs on the direct bf16x3 kernel (rounds 2-4) instead of the Winograd form
_WINO_DGRAD = os.environ.get("TOCVP_TRAIN_WINO_DGRAD", "1") != "0"


def _s():
    return torch.cuda.current_stream().cuda_stream


class DecoderLoss:
    def __init__(self, savi, frames_per_chunk=None):
        dec = savi.decoder
        if type(dec).__name__ != "ConvDecoder" or len(dec.hidden_dims) != 4:
            raise NotImplementedError("training step: SAVi ConvDecoder with 4 conv blocks (reference config)")
        self.savi, self.dec = savi, dec
        self.frames_per_chunk = frames_per_chunk
        self._bwd_w = None

    def _backward_weights(self):
        """ W'[ci][co][dy][dx] = W[co][ci][4-dy][4-dx] of conv blocks 1..3, split for the bf16x3 kernel """
        if self._bwd_w is None:
            out = {}
            for i in (1, 2, 3):
                w = self.dec.decoder[i].conv.weight.detach()
                wt = w.flip(-1, -2).transpose(0, 1).contiguous()
                out[i] = (K.split_conv_weights_bf16(wt), K.split_conv_weights_frag_bf16(wt))
            self._bwd_w = out
            self._zero_bias = torch.zeros(64, device=w.device, dtype=torch.float32)
        return self._bwd_w

    def _conv_fwd(self, i, x, collapsed=None):
        conv = self.dec.decoder[i].conv
        if self.dec.conv_precision == "f16x3" and self.dec.conv_wino:
            return K.conv5x5_dec_wino(x, self.dec._wino(i), conv.bias, relu=True, collapsed=collapsed, in_mode=2)
        if self.dec.conv_precision == "f16x3":
            return K.conv5x5_dec_f16x3(x, self.dec._split16(i), conv.bias, relu=True, collapsed=collapsed)
        if self.dec.conv_precision == "f16f8":
            return K.conv5x5_f16f8(x, self.dec._hybrid(i), conv.bias, relu=True, collapsed=collapsed)
        return K.conv5x5_bf16x3(x, self.dec._split(i), conv.bias, relu=True, collapsed=collapsed,
                                wfrag=self.dec._split_frag(i))

    def _backward_weights_wino(self):
        """ the same transposed, flipped weights as Winograd weight images (split-fp16 planes, per-row scales) """
        if getattr(self, "_bwd_wino", None) is None:
            out = {}
            for i in (1, 2, 3):
                w = self.dec.decoder[i].conv.weight.detach()
                out[i] = K.split_conv_weights_wino_f16x3(w.flip(-1, -2).transpose(0, 1).contiguous())
            self._bwd_wino = out
            self._zero_bias = torch.zeros(64, device=w.device, dtype=torch.float32)
        return self._bwd_wino

    def _conv_bwd(self, i, g, gate=None):
        """ data gradient of conv block i; ``gate`` = the block's input activation (post-ReLU): the gradient
        is masked by that ReLU in the store """
        if self.dec.conv_precision == "f16x3" and self.dec.conv_wino and _WINO_DGRAD:
            # the Winograd form of the same operator (2.5 x fewer matrix products, ~2^-21 per product instead of the
            # bf16 planes' ~2^-16); the gradient's magnitude is measured on the device and sets the operand scale
            wp = self._backward_weights_wino()[i]
            return K.conv5x5_dec_wino(g, wp, self._zero_bias, relu=False, in_mode=2, out_mode=0, auto_scale=True, gate=gate)
        ws, wf = self._backward_weights()[i]
        return K.conv5x5_bf16x3(g, ws, self._zero_bias, relu=False, wfrag=wf, gate=gate)

    @torch.no_grad()
    def loss_and_slot_grad(self, slots, targets, grad_scale):
        """
        slots (F, K, D) fp32, targets (F, 3, H, W).  Returns (sum of squared pixel errors as a (1,) tensor,
        dslots (F, K, D) = grad_scale * d(sum sq err)/d(slots) / 2 ... precisely: the gradient of
        grad_scale/2 * sum (img - target)^2, i.e. pass grad_scale = 2 * weight / numel for weight * MSE).
        """
        dec = self.dec
        F_, Ks, D = slots.shape
        pos = self.savi.decoder_pos_embedding.table()
        H, W, _ = pos.shape
        cpos, tapsum = dec._collapsed_layer0(pos)
        tail = dec.decoder[4]
        fpc = self.frames_per_chunk or max(1, dec.max_slot_images // Ks)
        dslots = torch.empty_like(slots)
        sq = torch.zeros(1, device=slots.device, dtype=torch.float32)
        for f0 in range(0, F_, fpc):
            f1 = min(F_, f0 + fpc)
            nf = f1 - f0
            n = nf * Ks
            sl = slots[f0:f1].reshape(n, D).contiguous()
            S = K.linear(sl, tapsum).reshape(n, 25, 64)
            x1 = self._conv_fwd(1, None, collapsed=(cpos, S))
            x2 = self._conv_fwd(2, x1)
            x3 = self._conv_fwd(3, x2)
            imgs, recons, masks = K.dec_tail(x3, tail.weight, tail.bias, nf, Ks)
            # per-pixel loss gradient: dimg = grad_scale * (img - target)
            tgt = targets[f0:f1].contiguous()
            nel = imgs.numel()
            nblocks = min(1024, (nel + 255) // 256)
            part = torch.empty(nblocks, device=slots.device, dtype=torch.float32)
            dimg = torch.empty_like(imgs)
            K._check(_L().tocvp_mse_f32(imgs.data_ptr(), tgt.data_ptr(), part.data_ptr(), nblocks,
                                        dimg.data_ptr(), nel, float(grad_scale), _s()), "tocvp_mse_f32")
            ag.axpby(ag.colsum(part.reshape(nblocks, 1)), sq, 1.0, 1.0)
            # tail backward
            dy = torch.empty((n, H, W, 4), device=slots.device, dtype=torch.float32)
            K._check(_L().tocvp_dec_tail_grad_f32(dimg.data_ptr(), recons.data_ptr(), masks.data_ptr(),
                                                  dy.data_ptr(), nf, Ks, H, W, _s()), "tocvp_dec_tail_grad_f32")
            g = torch.empty_like(x3)
            K._check(_L().tocvp_conv3x3_t4_f32(dy.data_ptr(), tail.weight.data_ptr(), x3.data_ptr(),
                                               g.data_ptr(), n, H, W, 64, _s()), "tocvp_conv3x3_t4_f32")
            del dy, x3, recons, masks, imgs, dimg
            # conv 3 and 2: data gradient, then the ReLU mask of the layer below
            for i, act in ((3, x2), (2, x1)):
                g = self._conv_bwd(i, g, gate=act)
            del x1, x2
            gin = self._conv_bwd(1, g)
            dS = torch.empty((n, 25, 64), device=slots.device, dtype=torch.float32)
            K._check(_L().tocvp_dec_class_reduce_f32(gin.data_ptr(), cpos.data_ptr(), S.data_ptr(),
                                                     dS.data_ptr(), n, H, W, 64, _s()),
                     "tocvp_dec_class_reduce_f32")
            ds = torch.empty((n, D), device=slots.device, dtype=torch.float32)
            ag.bmm(dS, tapsum, ds, n, D, 25 * 64, 25 * 64, D, D)
            dslots[f0:f1] = ds.reshape(nf, Ks, D)
        return sq, dslots
