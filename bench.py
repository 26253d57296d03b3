#!/usr/bin/env python3
"""bench.py -- headline benchmark of the LPD-Net global-descriptor hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): LPD-Net (featnet=lpdnet, emb_dims=1024, k=20, no T-Nets) eval
forward, N=4096 points, eval_batch_size=32 clouds per step per GPU; synthetic U[-1,1)^3 clouds
already resident in HBM; random-init weights of the architecture.  One "step" = one forward of one
32-cloud batch -> 32 global descriptors.  Multi-GPU = one process per GPU, each rank embeds its own
shard of clouds (the path shards by cloud, no data-path collective): weak scaling.

Prints ONE JSON line on rank 0.  `roofline` is for the kNN-aggregation kernel BASELINE.json names
(edge_gather_max on the SN1 stage): algorithmic bytes = 3152 B/point / average launch
time measured with HIP events on the launch stream inside the timed region.  `cpu_baseline` is the
oracle (torch-CPU restatement of the reference path) timed on this box's host cores on a bounded
sample; it is a reported baseline, not the optimisation target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
KAGG_BYTES_PER_POINT = 3152  # C=256, k=20, fp32, split form: P row + Q row + out row + idx


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clouds per step per GPU (eval_batch_size)")
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(model, points, seconds_target=15.0):
    """Oracle (CPU port of the reference path) on a bounded sample of the same workload."""
    from oracle import lpd_oracle as orc  # checker-only import, cpu_baseline leg
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(4321)
    Bs = 2
    x = torch.rand((Bs, 1, points, 3), generator=g) * 2 - 1
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    with torch.no_grad():
        t0 = time.time()
        ref = orc.pointnetvlad_forward(sd, x, featnet="lpdnet", train=False)   # warm-up (also builds the C oracle)
        first = time.time() - t0
        reps = max(1, min(8, int(seconds_target / max(first, 1e-3)) - 1))
        times = []
        for _ in range(reps):
            t0 = time.time()
            orc.pointnetvlad_forward(sd, x, featnet="lpdnet", train=False)
            times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(Bs / med, 3), "unit": "descriptors/s", "cores": cores, "kind": "port",
            "sample": f"{reps} timed eval forwards of {Bs} clouds x {points} pts (median), torch-CPU oracle, {cores} threads"}, x, ref


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", device_id=dev)

    from lpdnet_hip import ops
    from util.PointNetVlad import PointNetVlad

    torch.manual_seed(1234)  # reference util/initPara.py:86
    model = PointNetVlad(num_points=args.points, featnet="lpdnet", emb_dims=1024, output_dim=256)
    # non-trivial BatchNorm statistics so that eval-mode BN is a real affine (random-init keeps 0/1)
    g = torch.Generator().manual_seed(99)
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
            m.weight.data.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
            m.bias.data.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    model = model.to(dev).eval()

    gen = torch.Generator().manual_seed(1234 + rank)
    nbuf = 2
    clouds = [(torch.rand((args.batch, 1, args.points, 3), generator=gen) * 2 - 1).to(dev) for _ in range(nbuf)]

    def step(i):
        with torch.no_grad():
            return model(clouds[i % nbuf])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    # per-kernel HIP-event timing of the K-agg launches, on the launch stream, inside the timed region
    ops.PROFILE = {}
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = ops.PROFILE
    ops.PROFILE = None
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_desc = args.batch * args.steps * world
    value = total_desc / elapsed

    kern = {}
    for name, evs in prof.items():
        ms = [a.elapsed_time(b) for a, b in evs]
        kern[name] = {"launches": len(ms), "avg_us": round(1e3 * sum(ms) / len(ms), 2)}
    roof = None
    key = "edge_gather_max[C=256]"
    if key in kern:
        t_s = kern[key]["avg_us"] * 1e-6
        alg = KAGG_BYTES_PER_POINT * args.batch * args.points
        ach = alg / t_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "kagg_pmc.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"kernel": "edge_gather_max_kernel<64> (SN1 stage, C=256, k=20)", "bound": "hbm", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg, "avg_launch_us": kern[key]["avg_us"]}

    if rank == 0:
        line = {
            "metric": "global descriptors/sec (4096-pt clouds)", "value": round(value, 2), "unit": "descriptors/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (U[-1,1)^3 clouds resident in HBM, seed 1234+rank; random-init weights, randomised BN statistics)",
            "config": {"workload": "BASELINE configs[1]: LPD-Net (featnet=lpdnet, emb_dims=1024, k=20, no T-Nets) eval forward, "
                                   f"N={args.points}, eval_batch_size={args.batch} clouds/step/GPU",
                       "clouds_per_step_per_gpu": args.batch, "num_points": args.points, "parallelism": f"shard-by-cloud x{world}"},
            "roofline": roof, "kernels": kern,
        }
        if world == 1 and not args.no_cpu_baseline:
            base, xs, ref = cpu_baseline(model, args.points)
            with torch.no_grad():
                got = model(xs.to(dev)).cpu()
            rel = ((got - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max().item()
            line["cpu_baseline"] = base
            line["parity_norm_rel_vs_oracle"] = float(f"{rel:.3e}")
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
