"""Eval forward with / without the second HIP stream at small batches: python tools/side_small.py [clouds per step ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import engine
from util.PointNetVlad import PointNetVlad
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = PointNetVlad(num_points=4096, featnet="lpdnet").to(dev).eval()


def timeit(fn, n=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in [int(a) for a in sys.argv[1:]] or [1, 6, 10, 24, 32]:
    x = torch.rand(B, 1, 4096, 3, device=dev) * 2 - 1
    res = []
    with torch.no_grad():
        for rep in range(2):
            for mode in (False, True):
                engine._SIDE_FORCE.mode = mode
                res.append((mode, timeit(lambda: m(x))))
    engine._SIDE_FORCE.mode = None
    one = min(t for md, t in res if not md); two = min(t for md, t in res if md)
    print(f"B = {B:3d}: one stream {one:.3f} ms, two streams {two:.3f} ms ({100 * (two / one - 1):+.1f} %)")
