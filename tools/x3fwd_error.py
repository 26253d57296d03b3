import os, sys
sys.path.insert(0, "/root/repo/lpd-net-pytorch_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import lpd_oracle as orc, synth
from lpdnet_hip import ops, engine
from util.PointNetVlad import PointNetVlad
import loss.pointnetvlad_loss as L
dev = torch.device("cuda:0")
for (bq, P, Ng, N) in ((1, 2, 2, 512), (2, 2, 4, 1024), (2, 2, 6, 1024), (3, 2, 6, 512)):
    B = bq * (P + Ng + 2)
    sd0 = orc.synthetic_state("lpdnet", num_points=N)
    xc = torch.from_numpy(synth.cloud(21, B, N)).unsqueeze(1)
    res = {}
    for x3 in (False, True):
        ops.TRAIN_FWD_BF16X3 = x3
        m = PointNetVlad(num_points=N, featnet="lpdnet"); m.load_state_dict(sd0); m = m.to(dev).train()
        engine.DEBUG_AUX = {}; engine.MORTON_ORDER = False
        with torch.no_grad():
            out = m(xc.to(dev))
        aux = engine.DEBUG_AUX; engine.DEBUG_AUX = None; engine.MORTON_ORDER = True
        graphs = iter([aux["idx_feat"].cpu().long(), aux["idx_xyz"].cpu().long()])
        sd = {k: (v.double() if v.dtype == torch.float32 else v.clone()) for k, v in sd0.items()}
        orig = orc.knn; orc.knn = lambda xx, kk: next(graphs)
        with torch.no_grad():
            od = orc.pointnetvlad_forward(sd, xc.double(), featnet="lpdnet", train=True)
        orc.knn = orig
        res[x3] = ((out.cpu().double() - od).abs().amax(1) / od.abs().amax(1)).max().item()
    print(f"B={B} N={N}: exact fwd {res[False]:.2e}  x3 fwd {res[True]:.2e}")
