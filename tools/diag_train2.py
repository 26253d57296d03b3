"""GPU diagnostic: NetVLAD head and LPDNet trunk backward in isolation against the oracle's autograd."""
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
    # ---- head alone
    feat = torch.randn(M, 1024, generator=g)
    w = torch.randn(B, 256, generator=g)
    fg = feat.to(dev).requires_grad_(True)
    out = ag.netvlad_train(m.net_vlad, fg, B, N)
    (out * w.to(dev)).sum().backward()
    sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) else v.clone()) for k, v in sd0.items()}
    fc = feat.clone().requires_grad_(True)
    f4 = fc.view(B, N, 1024).permute(0, 2, 1).unsqueeze(-1)
    oo = orc.netvlad(sd, f4, train=True)
    (oo * w).sum().backward()
    print(f"[head B={B} N={N}] out {rel(out.detach(), oo.detach()):.2e} dfeat {rel(fg.grad, fc.grad):.2e}")
    # ---- trunk alone with a fixed cotangent
    m.zero_grad()
    xc = torch.from_numpy(synth.cloud(21, B, N)).unsqueeze(1)
    cot = torch.randn(M, 1024, generator=g)
    engine.DEBUG_AUX = {}
    f, _, _ = ag.lpdnet_features_train(m.emb_nn, xc.to(dev))
    aux = engine.DEBUG_AUX; engine.DEBUG_AUX = None
    (f * cot.to(dev)).sum().backward()
    gidx = [aux["idx_feat"].cpu().long(), aux["idx_xyz"].cpu().long()]
    sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) else v.clone()) for k, v in sd0.items()}
    it = iter(gidx); orig = orc.knn; orc.knn = lambda xx, k: next(it)
    oaux = {}
    of = orc.lpdnet_features(sd, xc, train=True, aux=oaux)
    orc.knn = orig
    ofp = of.squeeze(-1).permute(0, 2, 1).reshape(M, 1024)
    (ofp * cot).sum().backward()
    print(f"[trunk B={B} N={N}] feat {rel(f.detach(), ofp.detach()):.2e}")
    cat_o = torch.cat((oaux['x1'], oaux['x2'], oaux['x3']), dim=1).squeeze(-1).permute(0, 2, 1).reshape(M, 512)
    for nm, sl in (("x1", slice(0, 128)), ("x2", slice(128, 256)), ("x3", slice(256, 512))):
        print(f"    {nm} fwd {rel(aux['cat'][:, sl], cat_o[:, sl].detach()):.2e}")
    for name, prm in m.emb_nn.named_parameters():
        print(f"    {name:28s} {rel(prm.grad, sd['emb_nn.' + name].grad):.2e}   |g| {prm.grad.norm().item():.3e}")
