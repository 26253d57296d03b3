"""GPU diagnostic: trunk backward vs the fp64 oracle, next to the fp32 oracle's own distance from fp64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
import torch
from oracle import lpd_oracle as orc, synth
from util.PointNetVlad import PointNetVlad
from lpdnet_hip import engine, autograd as ag
dev = torch.device("cuda:0")
def rel(a, b): return ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
for (B, N) in [(6, 256), (6, 1024)]:
    M = B * N
    m = PointNetVlad(num_points=N, featnet="lpdnet")
    sd0 = orc.synthetic_state("lpdnet", num_points=N)
    m.load_state_dict(sd0); m = m.to(dev).train()
    g = torch.Generator().manual_seed(1)
    xc = torch.from_numpy(synth.cloud(21, B, N)).unsqueeze(1)
    cot = torch.randn(M, 1024, generator=g)
    engine.DEBUG_AUX = {}
    f, _, _ = ag.lpdnet_features_train(m.emb_nn, xc.to(dev))
    aux = engine.DEBUG_AUX; engine.DEBUG_AUX = None
    (f * cot.to(dev)).sum().backward()
    gidx = [aux["idx_feat"].cpu().long(), aux["idx_xyz"].cpu().long()]
    grads = {}
    for dt in (torch.float32, torch.float64):
        sd = {k: (v.to(dt).clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) else (v.to(dt) if v.dtype == torch.float32 else v.clone())) for k, v in sd0.items()}
        it = iter(gidx); orig = orc.knn; orc.knn = lambda xx, k: next(it)
        of = orc.lpdnet_features(sd, xc.to(dt), train=True)
        orc.knn = orig
        (of.squeeze(-1).permute(0, 2, 1).reshape(M, 1024) * cot.to(dt)).sum().backward()
        grads[dt] = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    print(f"[trunk B={B} N={N}]   GPU-vs-fp64   oracle32-vs-fp64")
    for name, prm in m.emb_nn.named_parameters():
        k = 'emb_nn.' + name
        print(f"    {name:28s} {rel(prm.grad, grads[torch.float64][k]):.2e}   {rel(grads[torch.float32][k], grads[torch.float64][k]):.2e}")
