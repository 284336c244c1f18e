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
