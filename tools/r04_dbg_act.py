import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"), os.path.join(os.path.dirname(__file__), "..")]
import torch
from lpdnet_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
M, K, N = 4096, 1024, 64
x = torch.randn(M, K, generator=g).to(dev)
w = torch.nn.Parameter((torch.randn(K, N, generator=g) / K ** 0.5).to(dev))
sc, sh = (0.5 + torch.rand(K, generator=g)).to(dev), (0.3 * torch.randn(K, generator=g)).to(dev)
sc[::5] *= -1
xa, c = ops.gemm_act(x, w.data, sc, sh, 2, 0.01)
ref = ops.affine_act(x, sc, sh, 2, 0.01)
ne = xa != ref
print("mismatch", int(ne.sum()), "of", ne.numel(), "nan", int(torch.isnan(xa).sum()))
idx = ne.nonzero()
print("rows", idx[:, 0].unique()[:20].tolist(), "cols", idx[:, 1].unique()[:40].tolist())
for r, cc in idx[:8].tolist():
    print(r, cc, xa[r, cc].item(), ref[r, cc].item(), x[r, cc].item(), sc[cc].item(), sh[cc].item())
print("c rel", ((c - ops.gemm(ref, w.data, b_kmajor=True)).abs().max() / c.abs().max()).item())
