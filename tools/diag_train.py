"""GPU diagnostic: per-parameter gradient error of the HIP training path vs the oracle, with the oracle
using (a) its own kNN graphs, (b) the GPU's graphs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
import torch
from oracle import lpd_oracle as orc, synth
from util.PointNetVlad import PointNetVlad
from lpdnet_hip import engine
import loss.pointnetvlad_loss as L

dev = torch.device("cuda:0")
for (bq, P, Ng, N) in [(1, 2, 2, 256), (2, 1, 3, 512), (1, 2, 2, 1024)]:
    B = bq * (P + Ng + 2)
    m = PointNetVlad(num_points=N, featnet="lpdnet")
    sd0 = orc.synthetic_state("lpdnet", num_points=N)
    m.load_state_dict(sd0); m = m.to(dev).train()
    xc = torch.from_numpy(synth.cloud(21, B, N)).unsqueeze(1)
    engine.DEBUG_AUX = {}
    out = m(xc.to(dev))
    aux = engine.DEBUG_AUX; engine.DEBUG_AUX = None
    q, p, n, o = torch.split(out.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
    loss = L.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False); loss.backward()
    gidx = [aux["idx_feat"].cpu().long(), aux["idx_xyz"].cpu().long()]
    res = {}
    for mode in ("own", "gpu_graph"):
        sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) else v.clone()) for k, v in sd0.items()}
        orig = orc.knn
        if mode == "gpu_graph":
            it = iter(gidx); orc.knn = lambda xx, k: next(it)
        oaux = {}
        od = orc.pointnetvlad_forward(sd, xc, featnet="lpdnet", train=True, aux=oaux)
        orc.knn = orig
        a, b, c, d = torch.split(od.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
        ol = orc.quadruplet_loss(a, b, c, d, 0.5, 0.2, True, True, False); ol.backward()
        res[mode] = (ol.item(), {k: v.grad for k, v in sd.items() if v.requires_grad}, oaux)
    same = (res["own"][2]["idx_feat"] == gidx[0]).all(-1).float().mean().item()
    print(f"cfg {(bq,P,Ng,N)} loss gpu {loss.item():.6f} oracle {res['own'][0]:.6f} / {res['gpu_graph'][0]:.6f}; feature-kNN rows equal {same:.4f}")
    for name, prm in m.named_parameters():
        g = prm.grad.cpu()
        e1 = ((g - res['own'][1][name]).norm() / res['own'][1][name].norm()).item()
        e2 = ((g - res['gpu_graph'][1][name]).norm() / res['gpu_graph'][1][name].norm()).item()
        print(f"   {name:45s} own-graph {e1:.2e}   gpu-graph {e2:.2e}")
