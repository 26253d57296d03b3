"""GPU diagnostic: full train step vs fp64 oracle; stage-wise gradient checks (d out, d feat)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
import torch
from oracle import lpd_oracle as orc, synth
from util.PointNetVlad import PointNetVlad
from lpdnet_hip import engine, autograd as ag
import loss.pointnetvlad_loss as L
dev = torch.device("cuda:0")
def rel(a, b): return ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
for (bq, P, Ng, N) in [(1, 2, 2, 256), (2, 1, 3, 512)]:
    B = bq * (P + Ng + 2); M = B * N
    m = PointNetVlad(num_points=N, featnet="lpdnet")
    sd0 = orc.synthetic_state("lpdnet", num_points=N)
    m.load_state_dict(sd0); m = m.to(dev).train()
    xc = torch.from_numpy(synth.cloud(21, B, N)).unsqueeze(1)
    engine.DEBUG_AUX = {}
    feat, _, _ = m.emb_nn._features(xc.to(dev))
    aux = engine.DEBUG_AUX; engine.DEBUG_AUX = None
    feat.retain_grad()
    out = m.net_vlad._pool(feat, B, N); out.retain_grad()
    q, p, n, o = torch.split(out.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
    loss = L.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False); loss.backward()
    gidx = [aux["idx_feat"].cpu().long(), aux["idx_xyz"].cpu().long()]
    R = {}
    for dt in (torch.float32, torch.float64):
        sd = {k: (v.to(dt).clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) else (v.to(dt) if v.dtype == torch.float32 else v.clone())) for k, v in sd0.items()}
        it = iter(gidx); orig = orc.knn; orc.knn = lambda xx, k: next(it)
        of = orc.lpdnet_features(sd, xc.to(dt), train=True); of.retain_grad()
        orc.knn = orig
        oo = orc.netvlad(sd, of, train=True); oo.retain_grad()
        a, b, c, d = torch.split(oo.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
        ol = orc.quadruplet_loss(a, b, c, d, 0.5, 0.2, True, True, False); ol.backward()
        R[dt] = dict(loss=ol.item(), dout=oo.grad, dfeat=of.grad.squeeze(-1).permute(0, 2, 1).reshape(M, 1024), out=oo.detach(),
                     g={k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None})
    r64, r32 = R[torch.float64], R[torch.float32]
    print(f"cfg {(bq,P,Ng,N)} loss gpu {loss.item():.7f} o32 {r32['loss']:.7f} o64 {r64['loss']:.7f}")
    print(f"   out   gpu-vs-64 {rel(out.detach(), r64['out']):.2e}  o32-vs-64 {rel(r32['out'], r64['out']):.2e}")
    print(f"   dout  gpu-vs-64 {rel(out.grad, r64['dout']):.2e}  o32-vs-64 {rel(r32['dout'], r64['dout']):.2e}")
    print(f"   dfeat gpu-vs-64 {rel(feat.grad, r64['dfeat']):.2e}  o32-vs-64 {rel(r32['dfeat'], r64['dfeat']):.2e}")
    for name, prm in m.named_parameters():
        print(f"   {name:42s} gpu-vs-64 {rel(prm.grad, r64['g'][name]):.2e}  o32-vs-64 {rel(r32['g'][name], r64['g'][name]):.2e}")
