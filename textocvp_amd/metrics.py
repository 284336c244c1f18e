"""
Metric tracker of the evaluation step after the rollout -- mirror of the reference's
lib/metrics.py (MetricTracker :15-144, PSNR :181-212, SSIM :216-255) on the HIP metric kernel
(tocvp_psnr_ssim_f32).  The reference delegates to piqa==1.2.2 (environment.yml:26), which is not
vendored: the kernel restates piqa's published PSNR / SSIM definitions (parity with piqa itself is
unpinned; the test suite checks the kernel against an independent float64 restatement).
LPIPS needs pretrained AlexNet weights from the network and is not built.
"""

import json
import os

import torch

from . import kernels as K

__all__ = ["MetricTracker", "PSNR", "SSIM", "METRICS_DICT"]


class _FrameMetric:
    """ per-(sequence, frame) metric accumulated over batches; aggregate -> (mean, framewise) """

    LOWER_BETTER = False
    which = None

    def __init__(self):
        self.reset()

    def reset(self):
        self.values = []

    def _check(self, t, name):
        if t.dim() != 5:
            raise ValueError(f"{name} with {t.shape = }, but it must be (B, F, C, H, W)")

    def accumulate(self, preds, targets):
        self._check(preds, "Preds"), self._check(targets, "Targets")
        B, F, C, H, W = preds.shape
        p, s = K.psnr_ssim(preds.reshape(B * F, C, H, W), targets.reshape(B * F, C, H, W),
                           clamp01=True, want_psnr=self.which == "psnr",
                           want_ssim=self.which == "ssim")
        cur = (p if self.which == "psnr" else s).view(B, F)
        self.values.append(cur)
        return cur.mean()

    def aggregate(self):
        allv = torch.cat(self.values, dim=0)
        return float(allv.mean()), allv.mean(dim=0)


class PSNR(_FrameMetric):
    which = "psnr"


class SSIM(_FrameMetric):
    which = "ssim"

    def __init__(self, window_size=11, sigma=1.5, n_channels=3):
        if window_size != 11 or sigma != 1.5:
            raise NotImplementedError("the SSIM kernel is built for the reference's 11-tap sigma 1.5 window")
        super().__init__()


METRICS_DICT = {"psnr": PSNR, "ssim": SSIM}


class MetricTracker:
    """ same surface as the reference tracker: accumulate / aggregate / get_results / summary / save """

    def __init__(self, exp_path=None, metrics=["psnr", "ssim"]):
        if not isinstance(metrics, list):
            raise TypeError(f"'metrics' must be a list, not {type(metrics)}")
        for m in metrics:
            if m == "lpips":
                raise NotImplementedError("lpips needs pretrained AlexNet weights (no network here)")
            if m not in METRICS_DICT:
                raise NameError(f"Unknown metric = {m}. Use one of {list(METRICS_DICT)}")
        self.exp_path = exp_path
        self.metric_computers = {m: METRICS_DICT[m]() for m in metrics}
        self.reset_results()

    def reset_results(self):
        self.results = {m: None for m in self.metric_computers}
        for m in self.metric_computers.values():
            m.reset()

    def accumulate(self, preds, targets):
        for mc in self.metric_computers.values():
            mc.accumulate(preds=preds, targets=targets)

    def aggregate(self):
        for name, mc in self.metric_computers.items():
            mean, framewise = mc.aggregate()
            self.results[name] = {"mean": mean, "framewise": framewise}

    def get_results(self):
        return self.results

    def summary(self):
        for name in self.metric_computers:
            print(f"  {name}:  {round(self.results[name]['mean'], 3)}")
        return self.results

    def save_results(self, exp_path, fname):
        results_dir = os.path.join(exp_path, "results", fname)
        os.makedirs(results_dir, exist_ok=True)
        results_file = os.path.join(results_dir, "results.json")
        cur = {}
        for name, res in self.results.items():
            if res is None:
                continue
            cur[name] = {"mean": round(res["mean"], 5),
                         "framewise": [round(r, 5) for r in res["framewise"].cpu().tolist()]}
        if os.path.exists(results_file):
            with open(results_file) as f:
                for k, v in json.load(f).items():
                    cur.setdefault(k, v)
        with open(results_file, "w") as f:
            json.dump(cur, f)
