"""Micro-benchmark of lpd_knn on the model's real feature-space input (F0) and on xyz, all impls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import engine, ops
from util.PointNetVlad import PointNetVlad
dev = torch.device("cuda:0")
B, N = 32, 4096
torch.manual_seed(1234)
m = PointNetVlad(num_points=N, featnet="lpdnet")
g = torch.Generator().manual_seed(99)
for mod in m.modules():
    if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
        mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
        mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
        mod.weight.data.copy_(0.5 + torch.rand(mod.weight.shape, generator=g))
        mod.bias.data.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
m = m.to(dev).eval()
x = (torch.rand((B, 1, N, 3), generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
engine.DEBUG_AUX = {}
with torch.no_grad():
    m(x)
f0 = engine.DEBUG_AUX["F0"]; engine.DEBUG_AUX = None
f0_cm = ops.transpose(f0.view(B, N, 64))
xyz_sorted = ops.morton_sort(x)
xyz_cm = ops.transpose(xyz_sorted.view(B, N, 3))
xyz_raw_cm = ops.transpose(x.view(B, N, 3))
impls = [int(a) for a in sys.argv[1:]] or [0, 2]
for name, t in (("F0 (C=64, z-ordered)", f0_cm), ("xyz (C=3, z-ordered)", xyz_cm), ("xyz (C=3, file order)", xyz_raw_cm)):
    for impl in impls:
        for _ in range(2): ops.knn(t, 20, impl=impl)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): r = ops.knn(t, 20, impl=impl)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"{name:26s} impl {impl}: {dt*1e6:8.1f} us")
