"""Timing experiment: how much of the large-cloud best-first kNN is the dependent global read of the tabulated tile bound per tested tile?
impl 8 = the kernel as it is (diagnostic build), impl 7 = every bound read from ONE cache line (results meaningless)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd")); sys.path.insert(0, ROOT)
import torch
from lpdnet_hip import ops
dev = torch.device("cuda:0")
B, N, k = 16, 16384, 64
g = torch.Generator().manual_seed(1)
W1 = (torch.randn((3, 64), generator=g) * 0.8).to(dev)
W2 = (torch.randn((64, 64), generator=g) * 0.2).to(dev)
for C in (64, 3):
    x = torch.rand((B, N, 3), generator=g) * 2 - 1
    xs = ops.morton_sort(x.contiguous().view(B, 1, N, 3).to(dev)).view(B, N, 3)          # Z-ordered clouds, as in the pipeline
    f = torch.nn.functional.leaky_relu(torch.nn.functional.leaky_relu(xs @ W1, 0.01) @ W2, 0.01)   # smooth 64-d features of xyz (like F0)
    xc = (xs if C == 3 else f).transpose(1, 2).contiguous()
    for impl in (0, 8, 7):
        for _ in range(2):
            ops.knn(xc, k, impl=impl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ops.knn(xc, k, impl=impl)
        torch.cuda.synchronize()
        print(f"C={C} impl {impl}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per {B} clouds")
