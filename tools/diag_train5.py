"""GPU diagnostic: stage-wise backward intermediates of the trunk vs the fp64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
import torch
from oracle import lpd_oracle as orc, synth
from util.PointNetVlad import PointNetVlad
from lpdnet_hip import engine, autograd as ag
dev = torch.device("cuda:0")
def rel(a, b): return ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
for (B, N) in [(12, 512), (6, 512), (12, 256)]:
    M = B * N
    m = PointNetVlad(num_points=N, featnet="lpdnet")
    sd0 = orc.synthetic_state("lpdnet", num_points=N)
    m.load_state_dict(sd0); m = m.to(dev).train()
    g = torch.Generator().manual_seed(1)
    xc = torch.from_numpy(synth.cloud(21, B, N)).unsqueeze(1)
    cot = torch.randn(M, 1024, generator=g)
    engine.DEBUG_AUX = {}
    f, _, _ = ag.lpdnet_features_train(m.emb_nn, xc.to(dev))
    (f * cot.to(dev)).sum().backward()
    aux = engine.DEBUG_AUX; engine.DEBUG_AUX = None
    gidx = [aux["idx_feat"].cpu().long(), aux["idx_xyz"].cpu().long()]
    res = {}
    for dt in (torch.float32, torch.float64):
        sd = {k: (v.to(dt).clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) else (v.to(dt) if v.dtype == torch.float32 else v.clone())) for k, v in sd0.items()}
        it = iter(gidx); orig = orc.knn; orc.knn = lambda xx, k: next(it)
        oaux = {}
        of = orc.lpdnet_features(sd, xc.to(dt), train=True, aux=oaux)
        orc.knn = orig
        for t in ("x1", "x2", "x3", "F0"): oaux[t].retain_grad()
        (of.squeeze(-1).permute(0, 2, 1).reshape(M, 1024) * cot.to(dt)).sum().backward()
        pm = lambda t: t.grad.reshape(B, t.shape[1], N).permute(0, 2, 1).reshape(M, -1)
        res[dt] = dict(dx1=pm(oaux["x1"]), dx2=pm(oaux["x2"]), dx3=pm(oaux["x3"]), df0=pm(oaux["F0"]),
                       g={k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None})
    r64, r32 = res[torch.float64], res[torch.float32]
    dcat = aux["dcat"]
    print(f"[B={B} N={N}]              GPU-vs-64   o32-vs-64")
    print(f"   dx1 (dcat[:, :128])      {rel(dcat[:, :128], r64['dx1']):.2e}   {rel(r32['dx1'], r64['dx1']):.2e}")
    print(f"   dx2 (accumulated)        {rel(dcat[:, 128:256], r64['dx2']):.2e}   {rel(r32['dx2'], r64['dx2']):.2e}")
    print(f"   dx3                      {rel(dcat[:, 256:], r64['dx3']):.2e}   {rel(r32['dx3'], r64['dx3']):.2e}")
    print(f"   dF0                      {rel(aux['df0'], r64['df0']):.2e}   {rel(r32['df0'], r64['df0']):.2e}")
    for name in ("convSN1.0.weight", "convDG2.1.weight", "convDG2.0.weight", "convDG1.1.weight", "convDG1.0.weight", "conv2_lpd.weight"):
        prm = dict(m.emb_nn.named_parameters())[name]
        print(f"   {name:24s} {rel(prm.grad, r64['g']['emb_nn.' + name]):.2e}   {rel(r32['g']['emb_nn.' + name], r64['g']['emb_nn.' + name]):.2e}")
