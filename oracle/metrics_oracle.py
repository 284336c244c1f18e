"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the image metrics the reference evaluator computes after the rollout
(lib/metrics.py:181-255).  The reference calls piqa==1.2.2 (environment.yml:26), an un-vendored
third-party package that is not installed here, so its published algorithm is restated:
  piqa.psnr.psnr(x, y)        = 10 log10(value_range^2 / (mse + 1e-8)), mse over (C,H,W) per image
  piqa.ssim.SSIM(11, 1.5, 3)  = Gaussian window (normalised, separable, per channel), NO padding,
                                k1 = 0.01, k2 = 0.03, value_range 1, mean over channels and positions.
PARITY UNPINNED against piqa itself (no fixture from piqa can be generated in this container); the
restatement is anchored on the reference's call sites and constructor arguments.
"""

import torch
import torch.nn.functional as F


def psnr(x, y, eps=1e-8):
    mse = ((x - y) ** 2).flatten(1).mean(dim=-1)
    return 10.0 * torch.log10(1.0 / (mse + eps))


def gaussian_window(size=11, sigma=1.5, dtype=torch.float64):
    d = torch.arange(size, dtype=dtype) - (size - 1) / 2
    g = torch.exp(-d ** 2 / (2 * sigma ** 2))
    return g / g.sum()


def ssim(x, y, size=11, sigma=1.5, k1=0.01, k2=0.03):
    """ x, y (N, C, H, W) in [0,1] -> (N,)  (float64 internally: this is the checker) """
    x, y = x.double(), y.double()
    C = x.shape[1]
    g = gaussian_window(size, sigma)

    def blur(t):
        t = F.conv2d(t, g.view(1, 1, -1, 1).repeat(C, 1, 1, 1), groups=C)
        return F.conv2d(t, g.view(1, 1, 1, -1).repeat(C, 1, 1, 1), groups=C)
    mx, my = blur(x), blur(y)
    sxx, syy, sxy = blur(x * x) - mx * mx, blur(y * y) - my * my, blur(x * y) - mx * my
    c1, c2 = k1 ** 2, k2 ** 2
    cs = (2 * sxy + c2) / (sxx + syy + c2)
    ss = (2 * mx * my + c1) / (mx * mx + my * my + c1) * cs
    return ss.flatten(1).mean(dim=-1).float()
