"""All kernels of the training step under rocprofv3 (ours and torch's): run as
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o p -- python3 tools/train_kernels.py [f32|bf16] [steps]
and read <dir>/**/p_kernel_stats.csv (calls / steps = launches per step)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import synth
from lpdnet_hip import harness, autograd
from util.PointNetVlad import PointNetVlad
autograd.set_train_storage(sys.argv[1] if len(sys.argv) > 1 else "bf16")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
N, bq, P, Ng = 4096, 2, 2, 18
m = PointNetVlad(num_points=N, featnet="lpdnet").to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-5)
tups = [torch.from_numpy(synth.cloud(s, bq * (2 + P + Ng), N)).view(bq, 2 + P + Ng, N, 3).to(dev) for s in range(steps)]
for tup in tups:
    harness.train_step(m, opt, tup[:, :1], tup[:, 1:1 + P], tup[:, 1 + P:1 + P + Ng], tup[:, 1 + P + Ng:])
torch.cuda.synchronize()
