import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from oracle import lpd_oracle as orc, synth
from lpdnet_hip import ops
B, C, N, k, impl = [int(v) for v in sys.argv[1:6]]
x = torch.from_numpy(synth.cloud(7, B, N, C)).float()      # [B,N,C]
xc = x.transpose(1, 2).contiguous()                        # [B,C,N]
print("input", tuple(xc.shape), flush=True)
idx = ops.knn(xc.cuda(), k, impl=impl)
torch.cuda.synchronize()
print("done kernel", flush=True)
want, _ = orc.knn_np(x.numpy(), k)
got = idx.cpu().numpy()
print("mismatching rows", int((got != want).any(-1).sum()), "of", B * N, flush=True)
