#!/usr/bin/env python
"""
bench.py -- predicted frames / second of the TextOCVP slot-rollout hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic sequences:
    SAVi encode (20 frames) -> TextOCVP_CustomTF rollout (19 steps) -> decode (19 frames)
at BASELINE.json configs[1]: 30 slots, 64x64x3, 1 seed + 19 predicted frames.  Inputs and weights
are resident in HBM before the timed region.  N > 1: one process per GPU (torchrun), every rank
runs its own batches (weak scaling, no collective inside the rollout) and the per-sequence
metrics are all-gathered ONCE at the end (RCCL).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline": dominant kernel (5x5 conv 64->64 on fp32 MFMA) measured live with HIP events,
  "cpu_baseline": the CPU oracle (oracle/, kind "port") on one sequence of the same workload.
"""

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NUM_SLOTS, NUM_CONTEXT, NUM_PREDS, RES = 30, 1, 19, 64
FP32_MFMA_PEAK_TFLOPS = 157.3            # MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0           # dense bf16 MFMA (same table)
CONV_GFLOP_PER_SLOT_IMAGE = 2 * RES * RES * 64 * 64 * 25 / 1e9   # 0.839: one 64->64 5x5 layer
PATH_GFLOP_PER_FRAME = 149.5             # reference algorithm, K=30


ARITH = {
    ("fp32", "fp32"): "fp32",
    ("f16f8", "f16x3"): "split operands on the matrix cores, fp32 accumulate / storage: decoder convs f16f8 "
                        "(f16 hi product + two e4m3 cross products), predictor GEMMs and encoder convs / "
                        "per-pixel GEMMs f16x3 (hi/lo fp16 planes, 3 products); slot attention, attention "
                        "scores, softmax, LayerNorm exact fp32",
    ("bf16x3", "f16x3"): "split operands on the 16-bit matrix cores (decoder convs bf16x3 = hi/lo bf16 "
                         "planes, predictor GEMMs f16x3 = hi/lo fp16 planes; 3 products each), fp32 "
                         "accumulate / storage; encoder, slot attention, softmax, LayerNorm fp32",
    ("bf16x3", "bf16x6"): "bf16 split operands (decoder convs bf16x3, predictor GEMMs bf16x6), fp32 "
                          "accumulate / storage; encoder, slot attention, softmax, LayerNorm fp32",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("TOCVP_BENCH_BATCH", 128)),
                    help="sequences per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def host_cores():
    """ cores this process may really use: min(affinity mask, cgroup v2/v1 CPU quota) """
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                cores = min(cores, max(1, q // per))
        except (OSError, ValueError):
            pass
    return cores


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(savi, pred):
    """ CPU oracle (port of the reference path) on ONE sequence of the same workload. """
    from oracle import slot_rollout_oracle as O
    from textocvp_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)
    ssd = {k: v.detach().cpu() for k, v in savi.state_dict().items()}
    psd = {k: v.detach().cpu() for k, v in pred.state_dict().items()}
    nseq = int(os.environ.get("TOCVP_CPU_BASELINE_SEQS", 6))
    videos = synth.synth_videos(nseq, NUM_CONTEXT + NUM_PREDS, seed=0)
    tokens, lengths = synth.synth_captions(nseq, max_len=12, seed=0)
    noise = synth.synth_noise(nseq, NUM_SLOTS, 128, seed=1)
    with torch.no_grad():   # untimed warm-up on one sequence (thread pool, MKLDNN primitives)
        O.forward_eval(ssd, psd, videos[:1], tokens[:1], lengths[:1], noise[:1], NUM_CONTEXT,
                       NUM_PREDS)
        t0 = time.perf_counter()
        O.forward_eval(ssd, psd, videos, tokens, lengths, noise, NUM_CONTEXT, NUM_PREDS)
        dt = time.perf_counter() - t0
    return {"value": round(nseq * NUM_PREDS / dt, 3), "unit": "predicted frames/s", "cores": cores,
            "kind": "port",
            "sample": f"one batch of {nseq} sequences (30 slots, 1 seed + 19 preds, 64x64) after a "
                      f"1-sequence warm-up: {dt:.1f} s wall, torch-CPU fp32 oracle, {cores} threads"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # one process per GPU; TOCVP_DIST_BACKEND=gloo lets several ranks share ONE GPU to rehearse the
    # torchrun path on a single-GPU box (the collective then runs on host copies of the metrics)
    backend = os.environ.get("TOCVP_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)     # "nccl" is RCCL on ROCm

    from textocvp_amd import kernels, synth
    from textocvp_amd.evaluator import forward_eval, gather_metrics, psnr_per_frame
    from textocvp_amd.setup_model import default_exp_params, setup_model, setup_predictor

    exp = default_exp_params(num_slots=NUM_SLOTS, num_context=NUM_CONTEXT, num_preds=NUM_PREDS)
    savi = setup_model(exp["model"]).eval()
    pred = setup_predictor(exp).eval()
    synth.fill_module_(savi, prefix="savi.")
    synth.fill_module_(pred, prefix="pred.")
    savi, pred = savi.to(dev), pred.to(dev)

    B = args.batch
    videos = synth.synth_videos(B, NUM_CONTEXT + NUM_PREDS, seed=100 + rank).to(dev)
    tokens, lengths = synth.synth_captions(B, max_len=12, seed=100 + rank)
    tokens, lengths = tokens.to(dev), lengths.to(dev)
    noise = synth.synth_noise(B, NUM_SLOTS, 128, seed=200 + rank).to(dev)

    def step():
        out = forward_eval(savi, pred, videos, NUM_CONTEXT, NUM_PREDS, caption_tokens=tokens,
                           caption_lengths=lengths, init_noise=noise)
        return psnr_per_frame(out["pred_imgs"], out["targets"])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        log(f"world={world} batch/gpu={B} warmup={args.warmup} steps={args.steps}")
    for _ in range(args.warmup):
        step()
    fence()
    if rank == 0:
        log("warmup done, timing")
    kernels.TIMER = kernels.LaunchTimer()
    metrics = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        metrics.append(step())
    all_metrics = gather_metrics(torch.cat(metrics, dim=0))     # the path's ONLY collective
    fence()
    elapsed = time.perf_counter() - t0
    timer, kernels.TIMER = kernels.TIMER, None

    # Outside the timed region: one more pass with the decoder NOT overlapped with the rollout.
    # In the timed steps the conv launches share the GPU with the predictor's kernels (second
    # stream), so their in-situ duration is not a statement about the kernel alone.
    timer_excl = None
    if rank == 0:
        kernels.TIMER = kernels.LaunchTimer()
        forward_eval(savi, pred, videos, NUM_CONTEXT, NUM_PREDS, overlap_decode=False,
                     caption_tokens=tokens, caption_lengths=lengths, init_noise=noise)
        torch.cuda.synchronize()
        timer_excl, kernels.TIMER = kernels.TIMER, None

    t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        frames = world * B * NUM_PREDS * args.steps
        conv = timer.summary().get("conv5x5_64_64")
        roofline = None
        if conv and conv["launches"]:
            avg_ms = conv["total_ms"] / conv["launches"]
            gflop_per_launch = CONV_GFLOP_PER_SLOT_IMAGE * conv["units"] / conv["launches"]
            achieved = gflop_per_launch / avg_ms            # GFLOP/ms == TFLOP/s
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "conv_pmc_summary.json")
            if os.path.exists(pmc):
                with open(pmc) as f:
                    rec = json.load(f)
                # only a summary taken on the kernel that is running now counts
                if rec.get("slot_images_per_launch") and \
                        savi.decoder.conv_precision in rec.get("kernel", ""):
                    traffic = rec["hbm_bytes_per_launch"] * (
                        conv["units"] / conv["launches"]) / rec["slot_images_per_launch"]
            # `achieved` is ALGORITHMIC flops / time; `peak` the dense MFMA peak of the 16-bit operand
            # type the main products run on.  `matrix_units_per_product`: matrix-pipe time per
            # algorithmic product in units of one 16-bit MFMA product (bf16x3: three bf16 products;
            # f16f8: one f16 product + two e4m3 products at twice the f16 rate)
            cp = savi.decoder.conv_precision
            kernel_name, units, peak = {
                "bf16x3": ("conv5x5_bf16x3_kernel (decoder 5x5 conv 64->64, split-bf16 operands, 3 bf16 "
                           "MFMA products per algorithmic product)", 3, BF16_MFMA_PEAK_TFLOPS),
                "f16f8": ("conv5x5_f16f8_kernel (decoder 5x5 conv 64->64, hybrid split: f16 main product "
                          "+ two e4m3 cross products on the 32x32x64 scaled MFMA)", 2,
                          BF16_MFMA_PEAK_TFLOPS),
            }.get(cp, ("conv5x5_mfma_kernel<64,64> (decoder 5x5 conv, exact fp32 MFMA)", 1,
                       FP32_MFMA_PEAK_TFLOPS))
            roofline = {"bound": "mfma",
                        "kernel": kernel_name,
                        "achieved": round(achieved, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                        "matrix_units_per_product": units,
                        "frac_executed_mfma": round(units * achieved / peak, 4),
                        "traffic": traffic, "launches": conv["launches"],
                        "avg_launch_ms": round(avg_ms, 4),
                        "gflop_per_launch": round(gflop_per_launch, 2),
                        "share_of_step_time": round(conv["total_ms"] / 1e3 / elapsed, 3),
                        "note": "achieved / avg_launch_ms are IN SITU (HIP events inside the timed region); "
                                "below 96 sequences per GPU the decoder runs on a second stream "
                                "concurrently with the rollout and shares the CUs; 'exclusive' is the "
                                "same kernel in one extra untimed pass with the GPU to itself"}
            ex = timer_excl.summary().get("conv5x5_64_64") if timer_excl else None
            if ex and ex["launches"]:
                ex_ms = ex["total_ms"] / ex["launches"]
                ex_tf = CONV_GFLOP_PER_SLOT_IMAGE * ex["units"] / ex["launches"] / ex_ms
                roofline["exclusive"] = {"avg_launch_ms": round(ex_ms, 4), "achieved": round(ex_tf, 2),
                                         "frac": round(ex_tf / peak, 4),
                                         "frac_executed_mfma": round(units * ex_tf / peak, 4),
                                         "launches": ex["launches"]}
        line = {
            "metric": "predicted frames/sec (1 seed, 19 preds, 64x64, 30 slots)",
            "value": round(frames / elapsed, 2), "unit": "predicted frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ARITH.get((savi.decoder.conv_precision, pred.predictor.gemm_precision),
                               f"decoder {savi.decoder.conv_precision} / predictor "
                               f"{pred.predictor.gemm_precision}"),
            "data": "synthetic",
            "config": {"workload": "configs[1]: SAVi 30-slot 64x64 + TextOCVP_CustomTF predictor, "
                                   "1 seed + 19 preds (encode 20 frames, 19 rollout steps, "
                                   "decode 19 frames)",
                       "batch_per_gpu": B, "global_batch": B * world, "num_slots": NUM_SLOTS,
                       "num_preds": NUM_PREDS, "resolution": RES, "weights": "synthetic (synth.py)",
                       "sharding": f"sequences x{world}, 1 all-gather of metrics"},
            "path_tflops": round(frames * PATH_GFLOP_PER_FRAME / 1e3 / elapsed, 2),
            "mean_psnr": round(float(all_metrics.mean().item()), 3),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            log(f"GPU: {line['value']} frames/s; timing the CPU oracle on one sequence ...")
            line["cpu_baseline"] = cpu_baseline(savi, pred)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
