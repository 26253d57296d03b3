"""The data-parallel training wrapper on the REAL model with a REAL process group on the GPU box: world size 1 over RCCL (the
collectives are issued and must leave the gradients of the single-process step unchanged) and world size 2 over gloo with both
ranks on the box's one GPU (each rank its own tuples: reduced gradients = mean of the two single-process gradients).  -m gpu only.
The toy-model arithmetic is covered by the gloo world-2 tests in tests/test_parallel_cpu.py; the N-rank launch by
`python bench.py --gpus N` (the launcher test there)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from oracle import lpd_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu


def _step(net, x, bq, P, Ng):
    import loss.pointnetvlad_loss as L
    out = net(x).view(bq, -1, 256)
    q, p, n, o = torch.split(out, [1, P, Ng, 1], dim=1)
    loss = L.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
    loss.backward()
    return loss


def test_grad_allreduce_on_pointnetvlad_over_rccl_world1(cuda):
    import torch.distributed as dist
    from lpdnet_hip.parallel import GradAllReduce
    from util.PointNetVlad import PointNetVlad
    N, bq, P, Ng = 512, 1, 2, 2
    B = bq * (1 + P + Ng + 1)
    sd = orc.synthetic_state("lpdnet", num_points=N)
    x = torch.from_numpy(synth.cloud(21, B, N)).unsqueeze(1).to(cuda)

    def fresh():
        m = PointNetVlad(num_points=N, featnet="lpdnet")
        m.load_state_dict(sd, strict=True)
        return m.to(cuda).train()
    plain = fresh()
    l0 = _step(plain, x, bq, P, Ng)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=cuda)
    try:
        wrapped = fresh()
        ddp = GradAllReduce(wrapped, reduce_when_single=True)
        for _ in range(2):                                    # twice: bucket / hook state resets between backward passes
            wrapped.zero_grad(set_to_none=True)
            l1 = _step(ddp, x, bq, P, Ng)
        torch.cuda.synchronize()
        assert dist.get_backend() == "nccl"
        assert ddp.stats["steps"] == 2 and ddp.stats["big_reduced"] == 2      # hidden1_weights, once per step
        assert ddp.stats["bucket_elems"] == 17605184 - 65536 * 256
    finally:
        dist.destroy_process_group()
    assert abs(l0.item() - l1.item()) < 1e-4 * abs(l0.item())     # run-to-run: fp64-atomic summation order
    for (n, a), (_, b) in zip(plain.named_parameters(), wrapped.named_parameters()):
        err = (a.grad - b.grad).norm().item() / max(a.grad.norm().item(), 1e-20)
        # not bitwise: BatchNorm sums use fp64 atomics whose order varies run to run; behind the max over k a last-bit
        # difference can re-route a gradient entry to another arg-max edge (tests/test_train_gpu.py header), so the trunk
        # gets the trunk tolerance and the head (in front of every max) the tight one
        assert err < (1e-4 if n.startswith("net_vlad.") else 1e-2), (n, err)


def test_grad_allreduce_world2_real_model_one_gpu(cuda, tmp_path):
    """configs[3]'s exchange on the real model with more than one rank: two processes (gloo; both on cuda:0 -- the box has one GPU),
    each with its OWN tuples, run the real PointNetVlad forward / backward (custom autograd.Functions, `hidden1_weights` reduced
    from its post-accumulate hook while the trunk's backward still runs, the 3.3 MB bucket at the engine callback) under
    GradAllReduce.  Reduced gradients must equal the mean of the two single-process gradients (head 1e-4; trunk 1e-2: behind the
    max over k a last-bit difference re-routes single gradient entries, tests/test_train_gpu.py header), both replicas must hold
    rank 0's weights, and hidden1_weights must be the first gradient to become ready."""
    from util.PointNetVlad import PointNetVlad
    import _world2_worker as w2
    N, bq, P, Ng = 1024, 1, 2, 2
    B = bq * (1 + P + Ng + 1)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_world2_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(tmp_path), str(N), str(bq), str(P), str(Ng)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r = [torch.load(os.path.join(tmp_path, f"rank{i}.pt")) for i in range(2)]
    sd = orc.synthetic_state("lpdnet", num_points=N)
    singles = []
    for rank in range(2):
        m = PointNetVlad(num_points=N, featnet="lpdnet")
        m.load_state_dict(sd, strict=True)
        m = m.to(cuda).train()
        _step(m, w2.tuples_of_rank(rank, B, N).to(cuda), bq, P, Ng)
        singles.append({n: p.grad.detach().cpu() for n, p in m.named_parameters()})
    assert torch.equal(r[0]["w_probe"], r[1]["w_probe"]) and torch.equal(r[0]["w_probe"].view(4, 4), sd["emb_nn.conv3_lpd.weight"][:4, :4].view(4, 4))
    worst = {}
    for n in singles[0]:
        assert torch.equal(r[0]["grads"][n], r[1]["grads"][n]), n            # one reduced gradient on both ranks
        want = 0.5 * (singles[0][n] + singles[1][n])
        if want.norm().item() < 1e-9:
            continue
        err = (r[0]["grads"][n] - want).norm().item() / max(want.norm().item(), 1e-20)
        worst[n] = err
        assert err < (1e-4 if n.startswith("net_vlad.") else 1e-2), (n, err)
        # ... and it is NOT rank 0's own gradient (the ranks really had different tuples)
    assert (singles[0]["net_vlad.hidden1_weights"] - singles[1]["net_vlad.hidden1_weights"]).norm() > 0.1 * singles[0]["net_vlad.hidden1_weights"].norm()
    for rr in r:
        assert rr["stats"]["steps"] == 2 and rr["stats"]["big_reduced"] == 2
        assert rr["stats"]["bucket_elems"] == 17605184 - 65536 * 256
        assert rr["first_grad"] == "net_vlad.hidden1_weights" or rr["first_grad"].startswith("net_vlad."), rr["first_grad"]
    print("world-2 reduced-vs-mean rel-L2: head max %.2e, trunk max %.2e" % (
        max(v for k, v in worst.items() if k.startswith("net_vlad.")), max(v for k, v in worst.items() if not k.startswith("net_vlad."))))


def test_bench_two_ranks_dry_run_over_gloo_on_one_gpu(cuda):
    """`python bench.py --gpus 2 --dist-backend gloo`: the whole multi-rank code path of the bench on the box's one GPU (RCCL
    refuses two ranks on one device, so gloo carries the collectives) -- the launcher starts the ranks as a child process tree
    without touching the GPU itself, every rank embeds its own shard, the train leg runs under GradAllReduce and fills the
    `exchange` block, rank 0 prints ONE JSON line with n_gpus = 2, a per-rank list of two and the dry-run mark.  No timing in it is
    asserted or means anything (two ranks share one GPU)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
           "--batch", "4", "--points", "1024", "--train-steps", "1", "--no-train-bf16", "--no-cpu-baseline", "--no-secondary",
           "--settle-seconds", "0", "--settle-max-seconds", "0"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and "dry_run" in rec and rec["scaling"] == "weak" and rec["unit"] == "descriptors/s"
    assert len(rec["descriptors_per_s_per_rank"]) == 2 and rec["value"] > 0
    assert rec["steps"] == 2 and rec["config"]["clouds_per_step_per_gpu"] == 4
    assert len(lines[0]) < 4096 and rec["detail"] == "bench_detail_n2.json"      # the parsed line is the short one ...
    for key in ("allreduce_ms_per_step_isolated", "allreduce_busbw_GBps", "allreduce_exposed_ms_per_step"):
        assert key in rec["train"]["exchange"]
    full = json.load(open(os.path.join(root, rec["detail"])))                      # ... everything else is in the side file
    ex = full["train"]["exchange"]
    assert ex is not None and ex["gradient_bytes"] > 4 * 17_000_000      # 17.6 M fp32 parameters
    for key in ("allreduce_ms_per_step_isolated", "allreduce_busbw_GBps", "ms_per_step_without_exchange", "allreduce_exposed_ms_per_step"):
        assert key in ex
    assert "gloo" in ex["backend"] and full["train"]["tuples_per_s"] > 0 and full["value"] == rec["value"]
