"""Descriptor error budget for the T-Net training variants: MI355X vs fp32 oracle vs fp64 oracle (same kNN graphs)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import lpd_oracle as orc, synth
from lpdnet_hip import engine
from util.PointNetVlad import PointNetVlad

def rel(a, b):
    return ((a - b).abs().amax(dim=1) / b.abs().amax(dim=1)).max().item()

for featnet, variant in (("lpdnet", dict(xyz_trans=True, feature_transform=True)), ("lpdnetorigin", dict(xyz_trans=True, feature_transform=True)),
                         ("lpdnet", dict(feature_transform=True)), ("lpdnet", dict(xyz_trans=True)), ("lpdnet", {})):
    for seed in (21, 22, 23):
        N, B = 256, 6
        m = PointNetVlad(num_points=N, featnet=featnet, **variant)
        sd0 = orc.synthetic_state(featnet, num_points=N, **variant)
        m.load_state_dict(sd0, strict=True)
        m = m.cuda().train()
        xc = torch.from_numpy(synth.cloud(seed, B, N)).unsqueeze(1)
        engine.DEBUG_AUX = {}
        engine.MORTON_ORDER = False
        with torch.no_grad():
            pass
        out = m(xc.cuda()).detach().cpu()
        aux = engine.DEBUG_AUX
        res = {}
        for dt in (torch.float32, torch.float64):
            graphs = iter([aux["idx_feat"].cpu().long(), aux["idx_xyz"].cpu().long()])
            sd = {k: (v.to(dt) if v.dtype == torch.float32 else v.clone()) for k, v in sd0.items()}
            orig = orc.knn
            orc.knn = lambda xx, k: next(graphs)
            try:
                with torch.no_grad():
                    res[dt] = orc.pointnetvlad_forward(sd, xc.to(dt), featnet=featnet, train=True, new_stats={}, **variant).double()
            finally:
                orc.knn = orig
        print(featnet, variant, seed, "gpu-vs-64 %.2e  gpu-vs-32 %.2e  32-vs-64 %.2e" % (rel(out.double(), res[torch.float64]),
              rel(out.double(), res[torch.float32]), rel(res[torch.float32], res[torch.float64])), flush=True)
