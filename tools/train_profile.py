"""Per-op HIP-event timing of one training step (B = 44 clouds, N = 4096): python tools/train_profile.py [featnet]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import synth
from lpdnet_hip import ops, harness
from util.PointNetVlad import PointNetVlad
featnet = sys.argv[1] if len(sys.argv) > 1 else "lpdnet"
from lpdnet_hip import autograd
autograd.set_train_storage(sys.argv[2] if len(sys.argv) > 2 else "f32")
dev = torch.device("cuda:0")
N, bq, P, Ng = 4096, 2, 2, 18
m = PointNetVlad(num_points=N, featnet=featnet).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
def step(seed):
    tup = torch.from_numpy(synth.cloud(seed, bq * (2 + P + Ng), N)).view(bq, 2 + P + Ng, N, 3).to(dev)
    return harness.train_step(m, opt, tup[:, :1], tup[:, 1:1 + P], tup[:, 1 + P:1 + P + Ng], tup[:, 1 + P + Ng:])
for s in range(2):
    step(s)
torch.cuda.synchronize()
ops.PROFILE = {}
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); step(5); t1.record(); torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
rows = sorted(((sum(a.elapsed_time(b) for a, b in evs), len(evs), name) for name, evs in prof.items()), reverse=True)
tot = sum(r[0] for r in rows)
print(f"step {t0.elapsed_time(t1):.2f} ms (with event overhead); kernels {tot:.2f} ms")
for ms, n, name in rows[:60]:
    print(f"{ms*1e3:9.1f} us  x{n:<3d} {name}")
