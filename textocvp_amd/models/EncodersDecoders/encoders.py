"""
Image encoders.  Reference: models/EncodersDecoders/encoders.py (get_encoder :27-95,
SimpleConvEncoder :99-159).  Only the conv encoder of the SAVi configs is on the hot path.
"""

import os

import torch.nn as nn

from ... import kernels as K
from ...precision import knob
from ..Blocks.model_blocks import ConvBlock
from ..Blocks.model_utils import Derived

__all__ = ["get_encoder", "SimpleConvEncoder"]


def get_encoder(in_channels, encoder, **kwargs):
    """ Factory; like the reference it POPS keys from the params dict (encoders.py:40-41). """
    name, params = encoder["encoder_name"], encoder["encoder_params"]
    if name == "ConvEncoder":
        return SimpleConvEncoder(in_channels=in_channels, hidden_dims=params.pop("num_channels"),
                                 kernel_size=params.pop("kernel_size"))
    if name in ("vit_base_patch14_dinov2", "vit_small_patch14_dinov2"):
        # reference encoders.py:77-95: timm factory + ViTEncoder wrapper; note that the reference reads
        # 'num_blocks' (ExtendedDINOSAUR.json sets 'encoder_num_blocks', so all 12 blocks run: SURVEY 3.4)
        from . import timm_encoders
        factory = getattr(timm_encoders, name)
        return timm_encoders.ViTEncoder(vit_backbone=factory(img_size=params.get("img_size")),
                                        num_blocks=params.get("num_blocks"))
    raise NotImplementedError(f"Unknown encoder {name}... (built: 'ConvEncoder', DINOv2 ViT-S/14 and ViT-B/14)")


class SimpleConvEncoder(nn.Module):
    """
    Stack of Conv5x5 + ReLU at full resolution.  forward() takes the reference's NCHW input and
    returns NCHW for API parity; the model uses ``forward_nhwc`` which keeps activations NHWC
    (channel-contiguous = coalesced for the implicit-GEMM kernel).
    """

    def __init__(self, in_channels=3, hidden_dims=(64, 64, 64, 64), kernel_size=5, **kwargs):
        super().__init__()
        if kernel_size != 5 or kwargs.get("downsample_encoder", False) or kwargs.get("batch_norm"):
            raise NotImplementedError("SimpleConvEncoder: kernel 5, no downsampling / batch-norm")
        if in_channels != 3:
            raise NotImplementedError("SimpleConvEncoder: RGB input only")
        self.in_channels, self.hidden_dims, self.kernel_size = in_channels, hidden_dims, kernel_size
        self.out_features = hidden_dims[-1]
        blocks, c = [], in_channels
        for h in hidden_dims:
            blocks.append(ConvBlock(c, h, kernel_size, padding=kernel_size // 2, activation=True))
            c = h
        self.encoder = nn.Sequential(*blocks)
        self._derived = Derived()
        # convs 1..: "f16x3" (split fp16 operands on the f16 matrix cores, fp32-class) or "fp32" (exact MFMA)
        self.conv_precision = knob("TOCVP_ENCODER_PRECISION", "f16x3")

    range_fallbacks = {"conv_precision": {"f16x3": "fp32"}}

    def forward_nhwc(self, x):
        """ x: (n, 3, H, W) view of contiguous image planes -> (n, H, W, C) """
        first = self.encoder[0].conv
        y = K.conv5x5_in3(x, first.weight, first.bias)
        with K.range_owner(self, "conv_precision"):
            for i in range(1, len(self.encoder)):
                conv = self.encoder[i].conv
                wp = self._derived.get(f"wp{i}", [conv.weight], lambda c=conv: K.pack_conv_weights(c.weight))
                y = K.conv5x5(y, wp, conv.bias, relu=True, precision=self.conv_precision)
        return y

    def forward(self, x):
        return self.forward_nhwc(x.contiguous()).permute(0, 3, 1, 2)
