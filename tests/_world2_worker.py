"""Helper of tests/test_parallel_gpu.py::test_grad_allreduce_world2_real_model_one_gpu: ONE rank of a world-2 gloo job whose two
ranks share cuda:0.  Each rank owns its own tuples (train_pointnetvlad.py:79-81 is what GradAllReduce replaces), runs the REAL
PointNetVlad forward / lazy quadruplet loss / backward under the wrapper twice and dumps its reduced gradients.

    python tests/_world2_worker.py <rank> <world> <port> <out_dir> <N> <bq> <P> <Ng>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lpd-net-pytorch_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def tuples_of_rank(rank, B, N):
    from oracle import synth
    return torch.from_numpy(synth.scene_cloud(40 + rank, B, N)).unsqueeze(1)


def main():
    rank, world, port, out_dir, N, bq, P, Ng = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], *map(int, sys.argv[5:9])
    from oracle import lpd_oracle as orc
    from lpdnet_hip.parallel import GradAllReduce
    from util.PointNetVlad import PointNetVlad
    import loss.pointnetvlad_loss as L
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    B = bq * (1 + P + Ng + 1)
    m = PointNetVlad(num_points=N, featnet="lpdnet")
    m.load_state_dict(orc.synthetic_state("lpdnet", num_points=N), strict=True)
    if rank != 0:                    # a replica that starts elsewhere: the wrapper must broadcast rank 0's weights
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(1.25)
    m = m.to(dev).train()
    ddp = GradAllReduce(m)
    order = []
    for name, p in m.named_parameters():
        p.register_post_accumulate_grad_hook(lambda q, name=name: order.append(name))
    x = tuples_of_rank(rank, B, N).to(dev)
    for _ in range(2):               # twice: hook / bucket state resets between backward passes
        m.zero_grad(set_to_none=True)
        order.clear()
        out = ddp(x).view(bq, -1, 256)
        q, p, n, o = torch.split(out, [1, P, Ng, 1], dim=1)
        loss = L.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
        loss.backward()
    torch.cuda.synchronize()
    torch.save({"grads": {n: p.grad.detach().cpu() for n, p in m.named_parameters()}, "loss": float(loss.item()),
                "stats": dict(ddp.stats), "first_grad": order[0] if order else None,
                "w_probe": m.emb_nn.conv3_lpd.weight.detach().cpu()[:4, :4].clone()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
