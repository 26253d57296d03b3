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
