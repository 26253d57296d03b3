import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
import torch
from util.PointNetVlad import PointNetVlad
import loss.pointnetvlad_loss as L
dev = torch.device("cuda:0")
bq, P, Ng, N = 2, 2, 18, 4096
B = bq * (P + Ng + 2)
torch.manual_seed(1234)
m = PointNetVlad(num_points=N, featnet="lpdnet").to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-5)
gen = torch.Generator().manual_seed(777)
for i in range(6):
    x = (torch.rand((B, 1, N, 3), generator=gen) * 2 - 1).to(dev)
    opt.zero_grad()
    out = m(x).view(bq, -1, 256)
    q, p, n, o = torch.split(out, [1, P, Ng, 1], dim=1)
    dp = ((p - q) ** 2).sum(2); dn = ((n - q) ** 2).sum(2); d2 = ((n - o) ** 2).sum(2)
    loss = L.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
    loss.backward(); opt.step()
    print(i, 'loss', loss.item(), 'dpos min', dp.min(1)[0].tolist(), 'dneg min', dn.min(1)[0].tolist(), 'd2 min', d2.min(1)[0].tolist(), 'out absmax', out.abs().max().item())
