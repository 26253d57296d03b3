"""world_size-2 gloo tests (CPU) of the data-parallel gradient all-reduce wrapper:
with distinct per-rank batches the reduced gradient equals the mean of the per-rank gradients; with identical
batches it equals the single-process gradient; replicas start from rank 0's weights."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy(seed):
    torch.manual_seed(seed)
    # one "big" tensor (own all-reduce, launched mid-backward) and several small ones (bucketed)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.BatchNorm1d(64), torch.nn.ReLU(), torch.nn.Linear(64, 2048),
                               torch.nn.Tanh(), torch.nn.Linear(2048, 8))


def _worker(rank, world, port, out_dir, same_data):
    sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lpdnet_hip.parallel import GradAllReduce
    model = _toy(100 + rank)                          # different init per rank: the wrapper must broadcast rank 0's
    ddp = GradAllReduce(model, big_bytes=64 * 2048 * 4)
    g = torch.Generator().manual_seed(7 if same_data else 7 + rank)
    x = torch.randn(12, 16, generator=g)
    for _ in range(2):                                # two steps: hook / bucket state must reset between backward passes
        model.zero_grad()
        ddp(x).pow(2).mean().backward()
    torch.save({"grads": [p.grad.clone() for p in model.parameters()], "w0": model[0].weight.detach().clone(),
                "x": x}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("same_data", [True, False])
def test_grad_allreduce_gloo_world2(tmp_path, same_data):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), same_data), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rank{i}.pt")) for i in range(world)]
    # replicas hold identical weights (rank 0's) and identical reduced gradients
    assert torch.equal(r[0]["w0"], r[1]["w0"])
    assert torch.equal(r[0]["w0"], _toy(100)[0].weight)
    for a, b in zip(r[0]["grads"], r[1]["grads"]):
        assert torch.equal(a, b)
    # expected: mean over ranks of the single-process gradient on that rank's batch
    expect = None
    for i in range(world):
        m = _toy(100)
        m.zero_grad()
        m(r[i]["x"]).pow(2).mean().backward()
        gs = [p.grad for p in m.parameters()]
        expect = gs if expect is None else [e + g for e, g in zip(expect, gs)]
    expect = [e / world for e in expect]
    for got, want in zip(r[0]["grads"], expect):
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-7)


def _pnv_stub_forward(model, x):
    """A parameter-touching stand-in for PointNetVlad.forward (the real one needs the GPU): every parameter enters the
    output, scaled by the data, so that every gradient is non-zero and differs between ranks with different data."""
    acc = x.new_zeros(())
    for i, p in enumerate(model.parameters()):
        acc = acc + (p * p).mean() * (1.0 + 0.1 * (i % 7)) * x.mean()
    return acc


def _pnv_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import types
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lpdnet_hip.parallel import GradAllReduce
    from util.PointNetVlad import PointNetVlad
    torch.manual_seed(50 + rank)
    model = PointNetVlad(num_points=64, featnet="lpdnet")             # the real module tree: 55 state_dict keys, 17.6 M parameters
    model.forward = types.MethodType(_pnv_stub_forward, model)
    ddp = GradAllReduce(model)                                        # default thresholds: hidden1_weights (67 MB) is the one big tensor
    x = torch.full((4,), 1.0 + rank)
    ddp(x).backward()
    big = [n for n, p in model.named_parameters() if p.numel() * 4 >= ddp.big_bytes]
    torch.save({"big": big, "stats": dict(ddp.stats),
                "gnorm": {n: p.grad.double().norm().item() for n, p in model.named_parameters()},
                "probe": {n: p.grad.reshape(-1)[:4].clone() for n, p in model.named_parameters()},
                "w": {n: p.detach().reshape(-1)[:4].clone() for n, p in model.named_parameters()}},
               os.path.join(out_dir, f"pnv{rank}.pt"))
    dist.destroy_process_group()


def test_grad_allreduce_wraps_the_real_pointnetvlad(tmp_path):
    """GradAllReduce on PointNetVlad itself (module tree on CPU, forward replaced by a parameter-touching stub): rank 0's
    weights everywhere, hidden1_weights reduced from its hook as the one big tensor, everything else in one bucket, and the
    result is the mean of the per-rank gradients."""
    world, port = 2, _free_port()
    mp.spawn(_pnv_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"pnv{i}.pt")) for i in range(world)]
    assert r[0]["big"] == ["net_vlad.hidden1_weights"]
    assert r[0]["stats"]["steps"] == 1 and r[0]["stats"]["big_reduced"] == 1
    assert r[0]["stats"]["bucket_elems"] == 17605184 - 65536 * 256
    sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
    from util.PointNetVlad import PointNetVlad
    torch.manual_seed(50)
    ref = PointNetVlad(num_points=64, featnet="lpdnet")
    for n, p in ref.named_parameters():
        assert torch.equal(r[1]["w"][n], p.detach().reshape(-1)[:4]), n        # rank 1 holds rank 0's weights
    # gradient of the stub: d/dp = 2 p / numel * c_i * mean(x); mean over ranks of mean(x) = 1.5
    for i, (n, p) in enumerate(ref.named_parameters()):
        want = 2 * p.detach().reshape(-1)[:4] / p.numel() * (1.0 + 0.1 * (i % 7)) * 1.5
        assert torch.allclose(r[0]["probe"][n], want, rtol=1e-5, atol=1e-12), n
        assert torch.equal(r[0]["probe"][n], r[1]["probe"][n]), n


def test_stale_gradients_are_not_rescaled():
    """A parameter that takes no part in this backward keeps its earlier .grad untouched."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
    from lpdnet_hip.parallel import GradAllReduce
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        a, b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
        both = torch.nn.ModuleList([a, b])
        ddp = GradAllReduce(both, big_bytes=1, reduce_when_single=True)
        b.weight.grad = torch.full_like(b.weight, 3.0)
        a(torch.ones(2, 4)).sum().backward()
        assert torch.equal(b.weight.grad, torch.full_like(b.weight, 3.0))
        assert ddp.stats["steps"] == 1 and ddp.stats["big_reduced"] == 2
    finally:
        dist.destroy_process_group()


def test_bench_launcher_builds_the_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` without RANK is a launcher: no GPU call, one child process tree of N ranks on 127.0.0.1."""
    import importlib
    import subprocess
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("RANK", raising=False)
    args = bench.parse()
    assert bench.launch_ranks(args) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(bench, "visible_gpus", lambda: 1)
    with pytest.raises(SystemExit):
        bench.launch_ranks(args)
