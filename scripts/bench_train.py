#!/usr/bin/env python
"""
Timing of the predictor TRAINING step (BASELINE configs[4]; reference 04_train_predictor.py:57-108 with the
defaults of CONFIG.py: batch 64, 1 seed + 9 preds, window 10, Adam 1e-4, clip 0.05) on synthetic 64x64
batches: frozen SAVi decomp -> BPTT rollout -> frozen decoder forward/backward -> clipped Adam.
One process per GPU; with torchrun the gradients are averaged by one flat all-reduce per step.

    python scripts/bench_train.py [--batch 64] [--slots 8] [--preds 9] [--steps 5] [--warmup 2]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from textocvp_amd import synth                                                        # noqa: E402
from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor    # noqa: E402
from textocvp_amd.train.step import PredictorTrainStep                                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--slots", type=int, default=8)
    ap.add_argument("--preds", type=int, default=9)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--eager", action="store_true", help="issue every launch from Python instead of replaying "
                                                         "the captured HIP graphs")
    a = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()   # gloo rehearsal: ranks share a GPU
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("TOCVP_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    exp = default_exp_params(num_slots=a.slots, num_context=1, num_preds=a.preds)
    savi, pred = setup_model(exp["model"]).eval(), setup_predictor(exp)
    synth.fill_module_(savi, prefix="savi.")
    synth.fill_module_(pred, prefix="pred.")
    ts = PredictorTrainStep(savi.to(dev), pred.to(dev))
    videos = synth.synth_videos(a.batch, 1 + a.preds, seed=100 + rank).to(dev)
    tokens, lengths = synth.synth_captions(a.batch, max_len=12, seed=100 + rank)
    tokens, lengths = tokens.to(dev), lengths.to(dev)
    noise = synth.synth_noise(a.batch, a.slots, 128, seed=200 + rank).to(dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    out = None
    run = ts.step if a.eager else ts.step_graphed
    for _ in range(a.warmup):
        out = run(videos, tokens, lengths, init_noise=noise)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = run(videos, tokens, lengths, init_noise=noise)
    fence()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({
            "metric": "predictor training steps/s", "value": round(a.steps / dt, 3), "unit": "steps/s",
            "ms_per_step": round(1e3 * dt / a.steps, 1), "n_gpus": world,
            "launch_mode": "eager" if a.eager else "hip graphs (fwd+bwd, optimiser)",
            "sequences_per_s": round(world * a.batch * a.steps / dt, 1),
            "config": {"workload": "configs[4]: TextOCVP_CustomTF training step, frozen SAVi, image + slot MSE, "
                                   "clipped Adam", "batch_per_gpu": a.batch, "num_slots": a.slots,
                       "num_preds": a.preds, "resolution": 64},
            "last": {k: round(float(v), 6) for k, v in out.items()},
            "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
