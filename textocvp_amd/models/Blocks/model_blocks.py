"""
Building blocks that only hold parameters on this path (their arithmetic is fused into the HIP
kernels of the owning model).  Reference: models/Blocks/model_blocks.py.
"""

import torch
import torch.nn as nn

from ... import kernels as K
from .model_utils import Derived

__all__ = ["ConvBlock", "Upsample", "SoftPositionEmbed", "TemporalPositionalEncoding"]


class Upsample(nn.Module):
    """
    Nearest-neighbour upsampling marker (model_blocks.py:23-45).  It owns no parameters; on the
    MI355X path it is fused into the tile loader of the following convolution.
    """

    def __init__(self, scale_factor):
        super().__init__()
        if scale_factor != 2:
            raise NotImplementedError("only x2 nearest upsampling is fused into the conv kernel")
        self.scale_factor = scale_factor

    def __repr__(self):
        return f"Upsample(scale_factor={self.scale_factor})"


class ConvBlock(nn.Module):
    """
    Conv2d (+ReLU) parameter holder with the reference's key layout ``block.0.{weight,bias}``
    (model_blocks.py:49-108).  BatchNorm / max-pool variants are not on the slot-rollout path.
    The conv itself runs in tocvp_conv5x5_f32 / tocvp_conv5x5_in3_f32, driven by the encoder/decoder.
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=None,
                 batch_norm=False, max_pool=None, activation=True):
        super().__init__()
        if max_pool or stride != 1:
            raise NotImplementedError("ConvBlock: max_pool / stride are not used on the rollout path")
        padding = padding if padding is not None else kernel_size // 2
        layers = [nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding)]
        if batch_norm:
            layers.append(nn.BatchNorm2d(num_features=out_channels))
        if activation:
            layers.append(nn.ReLU())
        self.activation = activation
        self.batch_norm = bool(batch_norm)
        self.block = nn.Sequential(*layers)

    @property
    def conv(self):
        return self.block[0]

    def folded_scale_shift(self):
        """
        Per-channel (scale, shift) such that block(x) = act(conv_nobias(x) * scale + shift):
        eval-mode BatchNorm and the conv bias folded together (BatchNorm only exists in the
        DINOSAUR image head and is affine in eval mode).
        """
        conv = self.conv
        if not self.batch_norm:
            return None, conv.bias.detach()
        bn = self.block[1]
        scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
        shift = (conv.bias.detach() - bn.running_mean) * scale + bn.bias.detach()
        return scale.contiguous(), shift.contiguous()


class SoftPositionEmbed(nn.Module):
    """
    4-channel linear position grid projected by a 1x1 conv (model_blocks.py:186-226).  Only the
    (H, W, C) addend is produced here (tocvp_pos_embed_f32, cached per weights); the add is fused
    into the consumer (LayerNorm prologue in the encoder, collapsed conv in the decoder).
    """

    def __init__(self, hidden_size, resolution, vmin=-1., vmax=1.):
        super().__init__()
        self.projection = nn.Conv2d(4, hidden_size, kernel_size=1)
        self.resolution = tuple(resolution)
        self._derived = Derived()

    def table(self):
        """ (H, W, C) addend on the parameters' device """
        H, W = self.resolution
        return self._derived.get(
            "table", [self.projection.weight, self.projection.bias],
            lambda: K.pos_embed(self.projection.weight, self.projection.bias, H, W))


class TemporalPositionalEncoding(nn.Module):
    """
    Learned temporal encoding shared by all slots of a frame and applied FLIPPED: the newest
    frame receives pe[0] (model_blocks.py:294-379, flip at :376).  The add is fused into the
    ``mlp_in`` GEMM epilogue (row-vector with reversed index), see BaseTextOCVP.forward.
    """

    MODES = ["sinusoid", "learned"]

    def __init__(self, d_model, dropout=0.0, max_len=50, mode="sinusoid"):
        super().__init__()
        if mode not in self.MODES:
            raise ValueError(f"Unknown {mode = }. Use one of {self.MODES}...")
        if dropout != 0.0:
            raise NotImplementedError("dropout is a training feature (inference-only path)")
        self.mode, self.d_model, self.max_len = mode, d_model, max_len
        if mode == "learned":
            self.pe = nn.Parameter(d_model ** -0.5 * torch.randn(1, max_len, 1, d_model))
        else:
            pos = torch.arange(max_len).unsqueeze(1)
            div = torch.exp(torch.arange(0, d_model, 2) * (-torch.log(torch.tensor(10000.0)) / d_model))
            pe = torch.zeros(max_len, d_model)
            pe[:, 0::2] = torch.sin(pos * div)
            pe[:, 1::2] = torch.cos(pos * div)
            self.pe = pe.view(1, max_len, 1, d_model)

    def rows(self, seq_len, device):
        """ (seq_len, d_model) contiguous table for the GEMM epilogue (un-flipped order) """
        if seq_len > self.max_len:
            raise ValueError(f"{seq_len = } exceeds {self.max_len = }")
        pe = self.pe.detach()
        if pe.device != device:
            pe = pe.to(device)
        return pe[0, :seq_len, 0].contiguous()
