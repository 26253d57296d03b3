"""Per-op HIP-event timing of the eval forward on one stream: python tools/eval_profile.py [clouds per step ...]   (default 32)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import engine, ops
from util.PointNetVlad import PointNetVlad
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = PointNetVlad(num_points=4096, featnet="lpdnet").to(dev).eval()
engine._SIDE_FORCE.mode = False
for B in [int(a) for a in sys.argv[1:]] or [32]:
    x = torch.rand(B, 1, 4096, 3, device=dev) * 2 - 1
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            m(x)
        e1.record(); torch.cuda.synchronize()
        ops.PROFILE = {}
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
    rows = sorted(((sum(a.elapsed_time(b) for a, b in evs) / 5, len(evs) // 5, name) for name, evs in prof.items()), reverse=True)
    print(f"B = {B}: {e0.elapsed_time(e1) / 20:.3f} ms/step on one stream; sum of the ops {sum(r[0] for r in rows):.3f} ms, {sum(r[1] for r in rows)} launches")
    for ms, n, name in rows[:28]:
        print(f"{ms * 1e3:9.1f} us  x{n:<2d} {name}")
