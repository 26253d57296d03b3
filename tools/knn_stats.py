"""GPU diagnostic: admission / drain statistics of the product kNN kernel (impl 74 = dbg 64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
import torch, numpy as np
from lpdnet_hip import engine, ops
from util.PointNetVlad import PointNetVlad
dev = torch.device("cuda:0")
B, N = 8, 4096
torch.manual_seed(1234)
m = PointNetVlad(num_points=N, featnet="lpdnet").to(dev).eval()
x = (torch.rand((B, 1, N, 3), generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
engine.DEBUG_AUX = {}
with torch.no_grad(): m(x)
f0 = engine.DEBUG_AUX["F0"]; engine.DEBUG_AUX = None
xs = ops.morton_sort(x)
for name, t in (("F0 z-ordered", ops.transpose(f0.view(B, N, 64))), ("xyz z-ordered", ops.transpose(xs.view(B, N, 3))), ("xyz file order", ops.transpose(x.view(B, N, 3)))):
    st = ops.knn(t, 20, impl=74).cpu().numpy()
    adm = st[..., 0] + st[..., 4]; it = st[..., 1]; ps = st[..., 2]
    wave_it = it.reshape(B, N // 32, 32)[..., 0]
    print(f"{name:16s} admitted/query: mean {adm.mean():.1f} median {np.median(adm):.0f} p90 {np.percentile(adm,90):.0f} max {adm.max()} | drain iterations/wave: mean {wave_it.mean():.1f} max {wave_it.max()} | passing tiles/wave (of 128): mean {ps.reshape(B,N//32,32)[...,0].mean():.1f}")
    wadm = adm.reshape(B, N // 32, 32)
    print(f"{'':16s} per-wave max admitted/query: mean {wadm.max(-1).mean():.1f}; per-wave mean: {wadm.mean(-1).mean():.1f}")
