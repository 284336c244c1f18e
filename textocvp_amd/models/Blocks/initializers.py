"""
Slot initialisers.  Reference: models/Blocks/initializers.py:12-94.
"""

from math import sqrt

import torch
import torch.nn as nn

from ... import kernels

__all__ = ["get_initializer", "Learned", "LearnedRandom"]


def get_initializer(mode, slot_dim, num_slots):
    """ Factory (initializers.py:12-35). """
    if mode == "Learned":
        return Learned(slot_dim=slot_dim, num_slots=num_slots)
    if mode == "LearnedRandom":
        return LearnedRandom(slot_dim=slot_dim, num_slots=num_slots)
    raise ValueError(f"UPSI, {mode = } is not a recongnized initializer...")


def _uniform_limit(slot_dim):
    return sqrt(6.0 / (1 + slot_dim))


class Learned(nn.Module):
    """ One learned vector per slot, repeated over the batch (initializers.py:39-61). """

    def __init__(self, slot_dim, num_slots):
        super().__init__()
        self.slot_dim, self.num_slots = slot_dim, num_slots
        lim = _uniform_limit(slot_dim)
        self.slots = nn.Parameter(torch.empty(1, num_slots, slot_dim).uniform_(-lim, lim))

    def forward(self, batch_size, **kwargs):
        return self.slots.detach().repeat(batch_size, 1, 1)


class LearnedRandom(nn.Module):
    """
    slots = mu + sigma * N(0, I), redrawn on EVERY forward (initializers.py:65-94, randn at :93).

    RNG parity: the reference's CPU path draws from torch's CPU generator.  The
    draw here is also made on the CPU generator and then moved to the device, so under the same
    ``torch.manual_seed`` the noise is bit-identical to the reference CPU path.  Tests and the
    bench pass the noise explicitly through the extension kwarg ``init_noise`` (B, K, D).
    """

    def __init__(self, slot_dim, num_slots):
        super().__init__()
        self.slot_dim, self.num_slots = slot_dim, num_slots
        lim = _uniform_limit(slot_dim)
        self.slots_mu = nn.Parameter(torch.empty(1, 1, slot_dim).uniform_(-lim, lim))
        self.slots_sigma = nn.Parameter(torch.empty(1, 1, slot_dim).uniform_(-lim, lim))

    def forward(self, batch_size, init_noise=None, **kwargs):
        dev = self.slots_mu.device
        shape = (batch_size, self.num_slots, self.slot_dim)
        if init_noise is None:
            init_noise = torch.randn(shape)           # CPU generator, like the reference CPU path
        if tuple(init_noise.shape) != shape:
            raise ValueError(f"init_noise must have shape {shape}, got {tuple(init_noise.shape)}")
        noise = init_noise.to(device=dev, dtype=torch.float32)
        return kernels.slot_init(self.slots_mu.detach(), self.slots_sigma.detach(), noise)
