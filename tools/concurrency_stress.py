"""Two (or three) eval forwards in flight on different HIP streams, every result checked against the single-stream forward of the same
input: the regression test-by-volume for kernels whose results depend on what else runs on the chip (round 6 found one: a
compiler-packed fp32 chain in lpd_front.hip, profiles/r06_concurrency_packed_f32.txt).   python tools/concurrency_stress.py [iterations]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from oracle import lpd_oracle as orc, synth
from util.PointNetVlad import PointNetVlad
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().amax(dim=1) / b.abs().amax(dim=1)).max().item()


for featnet in ("lpdnet", "lpdnetorigin", "pointnet"):
    m = PointNetVlad(num_points=4096, featnet=featnet)
    m.load_state_dict(orc.synthetic_state(featnet, num_points=4096))
    m = m.to(dev).eval()
    xb = torch.from_numpy(synth.cloud(77, 70, 4096)).unsqueeze(1).to(dev)
    batches = [xb[:32], xb[32:64], xb[64:70], xb[3:4]]
    streams = [torch.cuda.Stream() for _ in range(3)]
    with torch.no_grad():
        refs = [m(x).clone() for x in batches]
        torch.cuda.synchronize()
        bad, worst = 0, 0.0
        for it in range(iters):
            outs = []
            for j, x in enumerate(batches):
                with torch.cuda.stream(streams[(it + j) % 3]):
                    outs.append(m(x))
            torch.cuda.synchronize()
            e = max(rel(o, r) for o, r in zip(outs, refs))
            worst = max(worst, e)
            bad += e > 2e-6
    print(f"{featnet}: {bad} of {iters} iterations off (4 batches of 32 / 32 / 6 / 1 clouds on three streams), worst {worst:.2e}", flush=True)
