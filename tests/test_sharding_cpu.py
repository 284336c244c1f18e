"""
N > 1 path on CPU: two gloo ranks shard whole reference batches round-robin and all-gather their
ragged per-sequence metric tensors once; the result must equal the single-process order-by-rank
concatenation.  (On the GPU node the same code runs over RCCL; no collective exists inside the
rollout itself.)
"""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from textocvp_amd.evaluator import gather_metrics, shard_batches


def psnr_per_frame(preds, targets, eps=1e-8):
    """ (B, P, C, H, W) x2 -> (B, P) PSNR, piqa 1.2.2's 10*log10(1 / (mse + eps)) (lib/metrics.py:181-212): a plain
    torch stand-in for the metric rows, so that the gather has a real payload (the product's metric is
    textocvp_amd.metrics / tocvp_psnr_ssim_f32 on the GPU) """
    mse = ((preds - targets) ** 2).flatten(2).mean(dim=-1)
    return 10.0 * torch.log10(1.0 / (mse + eps))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _metrics_for_batch(j, P=5):
    """ deterministic stand-in for the (B_j, P) PSNR rows of reference batch j (ragged B_j) """
    g = torch.Generator().manual_seed(1000 + j)
    bj = 2 + (j % 3)
    preds = torch.rand(bj, P, 3, 8, 8, generator=g)
    targets = torch.rand(bj, P, 3, 8, 8, generator=g)
    return psnr_per_frame(preds, targets)


def _worker(rank, world, port, num_batches, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_batches(num_batches, rank, world)
    local = torch.cat([_metrics_for_batch(j) for j in mine], dim=0) if mine else torch.zeros(0, 5)
    full = gather_metrics(local)
    torch.save({"full": full, "mine": mine}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process(tmp_path):
    world, num_batches = 2, 7
    mp.spawn(_worker, args=(world, _free_port(), num_batches, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    # batch-preserving round-robin: every batch owned exactly once
    owned = sorted(j for o in outs for j in o["mine"])
    assert owned == list(range(num_batches))
    assert outs[0]["mine"] == [0, 2, 4, 6] and outs[1]["mine"] == [1, 3, 5]
    expect = torch.cat([_metrics_for_batch(j) for r in range(world)
                        for j in shard_batches(num_batches, r, world)], dim=0)
    for o in outs:
        assert torch.equal(o["full"], expect)
    # the aggregate is independent of the sharding
    single = torch.cat([_metrics_for_batch(j) for j in range(num_batches)], dim=0)
    assert torch.allclose(expect.mean(), single.mean(), atol=1e-6)
    assert torch.allclose(expect.sum(0).sort().values, single.sum(0).sort().values, atol=1e-4)


def test_single_process_gather_is_identity():
    x = torch.arange(12.0).reshape(4, 3)
    assert torch.equal(gather_metrics(x), x)


def test_psnr_definition():
    a = torch.zeros(1, 1, 3, 4, 4)
    b = torch.full((1, 1, 3, 4, 4), 0.1)
    assert abs(psnr_per_frame(a, b).item() - 10 * torch.log10(torch.tensor(1 / (0.01 + 1e-8))).item()) < 1e-4


# ---------------------------------------------------------------------------------------------------
# bench.py --gpus N without torchrun: the launcher that replaces nn.DataParallel (baseEvaluator.py:142-145)
# ---------------------------------------------------------------------------------------------------
def _bench_module():
    import importlib.util
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_launcher_argument_to_children_env():
    """ --gpus N -> N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, same argv, no exec """
    bench = _bench_module()
    env = bench.worker_env(3, 8, 29512, base={"PATH": "/bin", "WORLD_SIZE": "junk"})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29512"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"

    started = []

    class FakeProc:
        def __init__(self, cmd, env):
            started.append((cmd, env))
            self.rc = 0 if env["RANK"] != "2" else 7

        def wait(self):
            return self.rc

    rc = bench.launch_workers(4, argv=["--gpus", "4", "--steps", "2"], device_count=8,
                              popen=lambda cmd, env: FakeProc(cmd, env))
    assert rc == 7                                                 # a failing rank fails the launcher
    assert [e["RANK"] for _, e in started] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" for _, e in started)
    assert len({e["MASTER_PORT"] for _, e in started}) == 1
    assert all(cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "2"] for cmd, _ in started)
    # fewer devices than ranks: refuse, start nothing
    started.clear()
    assert bench.launch_workers(8, argv=[], device_count=1, popen=lambda cmd, env: FakeProc(cmd, env)) != 0
    assert started == []


def _run_launcher(extra_env, *argv, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k_, None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, cwd=root,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_launcher_end_to_end_with_eight_stub_ranks():
    """
    ``python bench.py --gpus 8`` for real: the launcher starts eight fresh worker processes (nothing is mocked), they
    rendezvous over gloo on 127.0.0.1, run the timed-region protocol of the real worker (bench.run_timed: warm-up,
    barrier, K steps, the ONE gather of the metric rows, barrier, MAX of the elapsed time) on a CPU stand-in of the hot
    path (TOCVP_BENCH_STUB=1) and rank 0 prints the line.  Checks: n_gpus, every rank's rows present and in rank order,
    a failing rank's exit code reaches the launcher's, and the launcher refuses (starting nothing) when fewer devices
    than ranks are visible and the backend is RCCL.
    """
    import json
    stub = {"TOCVP_BENCH_STUB": "1", "TOCVP_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"}
    res = _run_launcher(stub, "--gpus", "8", "--batch", "2", "--steps", "2", "--warmup", "1")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout                              # rank 0 alone prints
    line = lines[0]
    assert line["n_gpus"] == 8 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["gathered_rows"] == 8 * 2 * 2                       # ranks x steps x sequences per step
    assert line["row_owner"] == [r for r in range(8) for _ in range(4)]     # rows in rank order
    assert line["value"] > 0 and line["ms_per_step"] >= 2.0         # the slowest rank (6 ms per step) sets the time
    # --strong: --batch is the GLOBAL batch, split over the ranks (the reference's nn.DataParallel scatters one batch)
    res = _run_launcher(stub, "--gpus", "4", "--batch", "8", "--steps", "1", "--warmup", "0", "--strong")
    assert res.returncode == 0, res.stderr[-2000:]
    line = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")][0]
    assert line["scaling"] == "strong" and line["n_gpus"] == 4 and line["global_batch"] == 8 and line["gathered_rows"] == 8
    res = _run_launcher(stub, "--gpus", "3", "--batch", "8", "--steps", "1", "--warmup", "0", "--strong")
    assert res.returncode != 0 and "not a multiple" in res.stderr
    # a rank that fails AFTER the collectives: every other rank ends cleanly, the launcher returns the worst code
    res = _run_launcher(dict(stub, TOCVP_BENCH_STUB_FAIL_RANK="5"), "--gpus", "8", "--batch", "1", "--steps", "1",
                        "--warmup", "0")
    assert res.returncode == 7, (res.returncode, res.stderr[-1000:])
    # RCCL backend with fewer visible devices than ranks (this container has none): refuse, start nothing
    res = _run_launcher({"TOCVP_BENCH_STUB": "1"}, "--gpus", "8", "--batch", "1", "--steps", "1")
    assert res.returncode == 2 and "GPU(s) are visible" in res.stderr and not res.stdout.strip()
