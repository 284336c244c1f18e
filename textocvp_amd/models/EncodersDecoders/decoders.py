"""
Slot decoders.  Reference: models/EncodersDecoders/decoders.py (get_decoder :20-48,
ConvDecoder :52-125).
"""

import os

import torch
import torch.nn as nn

from ... import kernels as K
from ..Blocks.model_blocks import ConvBlock
from ..Blocks.model_utils import Derived

__all__ = ["get_decoder", "ConvDecoder"]


def get_decoder(in_channels, decoder, **kwargs):
    """ Factory; pops keys from the params dict like the reference (decoders.py:32-34). """
    name, params = decoder["decoder_name"], decoder["decoder_params"]
    if name == "ConvDecoder":
        return ConvDecoder(in_channels=in_channels, hidden_dims=params.pop("num_channels"),
                           kernel_size=params.pop("kernel_size"), upsample=params.pop("upsample"),
                           out_channels=kwargs.get("out_channels", 4), **params)
    raise NotImplementedError(
        f"decoder {name!r}: only 'ConvDecoder' is built so far (MLPPatchDecoder is SURVEY 8f rank 3)")


class ConvDecoder(nn.Module):
    """
    Spatial-broadcast conv decoder: 4x (Conv5x5 + ReLU) at full resolution, then Conv3x3 -> RGB+alpha.
    Layers are built from hidden_dims[-1] down to hidden_dims[0] (decoders.py:96-117).

    MI355X path (``decode_slots``):
      layer 0   never runs as a conv: its input is broadcast(slot) + pos, so its output is
                cpos[y,x,:] + tapsum[cls(y,x)] @ slot  (exact algebra, 25 border classes);
                cpos = conv0(pos)+b0 is computed ONCE per weight set with the MFMA conv kernel.
      layer 1   tocvp_conv5x5_f32 in collapsed-input mode (synthesises relu(layer 0) on the fly)
      layer 2-3 tocvp_conv5x5_f32
      tail      tocvp_dec_tail_f32: conv3x3 + softmax over slots + compositing, writing the
                reference's three output tensors directly.
    """

    def __init__(self, in_channels, hidden_dims, kernel_size=5, upsample=None, out_channels=4,
                 **kwargs):
        super().__init__()
        if kernel_size != 5 or (upsample is not None and upsample >= 2) or kwargs.get("batch_norm"):
            raise NotImplementedError("ConvDecoder: kernel 5, upsample < 2, no batch-norm")
        if out_channels != 4 or len(hidden_dims) < 2:
            raise NotImplementedError("ConvDecoder: RGB + alpha output, >= 2 hidden layers")
        self.in_channels = self.in_features = in_channels
        self.hidden_dims, self.kernel_size = hidden_dims, kernel_size
        self.out_features, self.out_channels = hidden_dims[0], out_channels
        self.upsample = None
        mods, c = [], in_channels
        for i in range(len(hidden_dims) - 1, -1, -1):
            mods.append(ConvBlock(c, hidden_dims[i], kernel_size, padding=kernel_size // 2))
            c = hidden_dims[i]
        mods.append(nn.Conv2d(self.out_features, out_channels, kernel_size=3, stride=1, padding=1))
        self.decoder = nn.Sequential(*mods)
        self._derived = Derived()
        self.max_slot_images = 2048          # slot images decoded per chunk (bounds HBM scratch)
        # arithmetic of the 64->64 convs: "fp32" (exact fp32 MFMA) or "bf16x3" (split-bf16 operands
        # on the bf16 matrix cores, ~2^-16 per-product error, 5.3x fewer matrix cycles)
        self.conv_precision = os.environ.get("TOCVP_DECODER_PRECISION", "bf16x3")

    # -- derived weights -----------------------------------------------------------------------
    def _packed(self, i):
        conv = self.decoder[i].conv
        return self._derived.get(f"wp{i}", [conv.weight], lambda: K.pack_conv_weights(conv.weight))

    def _split(self, i):
        conv = self.decoder[i].conv
        return self._derived.get(f"ws{i}", [conv.weight],
                                 lambda: K.split_conv_weights_bf16(conv.weight))

    def _collapsed_layer0(self, pos_table):
        """ (cpos (H,W,C0), tapsum (25*C0, D)) for the current weights / position table """
        c0 = self.decoder[0].conv

        def build():
            cpos = K.conv5x5(pos_table[None].contiguous(), K.pack_conv_weights(c0.weight), c0.bias,
                             relu=False)[0].contiguous()
            ts = K.dec_tapsum(c0.weight)
            return cpos, ts.reshape(25 * ts.shape[1], ts.shape[2])
        return self._derived.get("layer0", [c0.weight, c0.bias, pos_table], build)

    # -- forward -------------------------------------------------------------------------------
    def decode_slots(self, slots, pos_table):
        """
        slots (F, K, D), pos_table (H, W, D) -> recons_imgs (F,3,H,W), recons (F,K,3,H,W),
        masks (F,K,1,H,W)   (SAVi.decode, models/SAVi.py:241-261)
        """
        F_, Ks, D = slots.shape
        H, W, _ = pos_table.shape
        dev = slots.device
        n_hidden = len(self.hidden_dims)
        cpos, tapsum = self._collapsed_layer0(pos_table)
        C0 = cpos.shape[-1]
        imgs = torch.empty((F_, 3, H, W), device=dev, dtype=torch.float32)
        recons = torch.empty((F_, Ks, 3, H, W), device=dev, dtype=torch.float32)
        masks = torch.empty((F_, Ks, 1, H, W), device=dev, dtype=torch.float32)
        tail = self.decoder[n_hidden]
        fpc = max(1, self.max_slot_images // Ks)             # frames per chunk
        bufs = [None, None]
        for f0 in range(0, F_, fpc):
            f1 = min(F_, f0 + fpc)
            n = (f1 - f0) * Ks
            S = K.linear(slots[f0:f1].reshape(n, D), tapsum).reshape(n, 25, C0)
            x, which = None, 0
            for i in range(1, n_hidden):
                conv = self.decoder[i].conv
                co = conv.weight.shape[0]
                out = bufs[which]
                if out is None or out.shape != (n, H, W, co):
                    out = torch.empty((n, H, W, co), device=dev, dtype=torch.float32)
                    bufs[which] = out
                split = (self.conv_precision == "bf16x3" and conv.weight.shape[0] == 64
                         and conv.weight.shape[1] == 64)
                if split:
                    x = K.conv5x5_bf16x3(x, self._split(i), conv.bias, relu=True, out=out,
                                         collapsed=(cpos, S) if i == 1 else None)
                elif i == 1:
                    x = K.conv5x5_collapsed(cpos, S, self._packed(1), conv.bias, relu=True, out=out)
                else:
                    x = K.conv5x5(x, self._packed(i), conv.bias, relu=True, out=out)
                which ^= 1
            K.dec_tail(x, tail.weight, tail.bias, f1 - f0, Ks,
                       out=(imgs[f0:f1], recons[f0:f1], masks[f0:f1]))
        return imgs, recons, masks

    def forward(self, x):
        raise NotImplementedError(
            "ConvDecoder.forward on a materialised (B*K, D, H, W) broadcast is deliberately not "
            "provided: use SAVi.decode / decode_slots (the broadcast tensor never exists here)")
