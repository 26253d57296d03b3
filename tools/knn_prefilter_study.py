"""Would a cheap low-precision distance tile (one bf16 MFMA product, 1/16 of the exact tile's matrix cycles) prove most of the
visited-but-useless candidate tiles of the feature-space kNN irrelevant?  Per (query tile W, candidate tile T) of the bench
model's F0 features: 'needed' = T holds a candidate that beats some query's final k-th best; 'bound' = the centroid / radius
bound the kernel uses cannot exclude T; 'pre' = neither can the bf16 tile with a rigorous error bound (7.9e-3 |q||c|).
python tools/knn_prefilter_study.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from lpdnet_hip import engine
from util.PointNetVlad import PointNetVlad
dev = torch.device("cuda:0")
B, N, k = 4, 4096, 20
torch.manual_seed(1234)
model = PointNetVlad(num_points=N, featnet="lpdnet", emb_dims=1024, output_dim=256)
g = torch.Generator().manual_seed(99)
for m in model.modules():
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
        m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g)); m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
        m.weight.data.copy_(0.5 + torch.rand(m.weight.shape, generator=g)); m.bias.data.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
model = model.to(dev).eval()
gen = torch.Generator().manual_seed(1234)
x = (torch.rand((B, 1, N, 3), generator=gen) * 2 - 1).to(dev)
engine.DEBUG_AUX = {}
with torch.no_grad():
    model(x)
F0 = engine.DEBUG_AUX["F0"].view(B, N, 64).double()
engine.DEBUG_AUX = None
nt = N // 32
for center in (False, True):
    tot = {"needed": 0.0, "bound": 0.0, "pre": 0.0, "pre_and_bound": 0.0, "pre2": 0.0}
    for b in range(B):
        f = F0[b]
        fc = f - f.mean(0, keepdim=True) if center else f
        d2 = torch.cdist(f, f) ** 2
        pd = -d2
        thr = pd.topk(k, dim=1).values[:, -1]                         # final k-th best of every query
        beats = pd >= thr[:, None]                                     # [q, c]
        needed = beats.view(nt, 32, nt, 32).any(3).any(1)               # [W, T]
        t = f.view(nt, 32, 64)
        cen = t.mean(1)
        rad = (t - cen[:, None]).norm(dim=-1).amax(1)
        lb = (torch.cdist(f, cen) - rad[None]).clamp_min(0)            # [q, T]
        bound = (-(lb ** 2) >= thr[:, None]).view(nt, 32, nt).any(1)
        fb = fc.float().bfloat16().double()
        nrm = fc.norm(dim=1)
        dot_b = fb @ fb.t()
        n2 = (fc * fc).sum(1)
        pd_b = 2 * dot_b - n2[:, None] - n2[None]
        eps = 2.0 * nrm[:, None] * nrm[None] * 7.9e-3           # (1 + u)^2 - 1 with u = 2^-8, + accumulation
        pre = ((pd_b + eps) >= thr[:, None]).view(nt, 32, nt, 32).any(3).any(1)
        # two-product variant: q exact to 16 bits (hi + lo) against bf16 candidates: error only from the candidate side
        eps2 = 2.0 * nrm[:, None] * nrm[None] * 3.95e-3
        pd_2 = 2 * (fc @ fb.t()) - n2[:, None] - n2[None]
        pre2 = ((pd_2 + eps2) >= thr[:, None]).view(nt, 32, nt, 32).any(3).any(1)
        tot["needed"] += needed.float().sum(1).mean().item(); tot["bound"] += bound.float().sum(1).mean().item()
        tot["pre"] += pre.float().sum(1).mean().item(); tot["pre_and_bound"] += (pre & bound).float().sum(1).mean().item()
        tot["pre2"] += (pre2 & bound).float().sum(1).mean().item()
        if b == 0:
            print("  |f| mean %.3f  |f - mean| mean %.3f  kth-neighbour distance mean %.4f" % (f.norm(dim=1).mean(), (f - f.mean(0)).norm(dim=1).mean(), (-thr).sqrt().mean()))
    print("centred" if center else "raw", {k_: round(v / B, 1) for k_, v in tot.items()}, "tiles per query tile (of %d)" % nt)
