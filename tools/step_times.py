"""Per-step wall times of the eval forward (bench workload): distribution, to tell a slow kernel from a hiccup."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from util.PointNetVlad import PointNetVlad
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = PointNetVlad(num_points=4096, featnet="lpdnet", emb_dims=1024, output_dim=256).to(dev).eval()
gen = torch.Generator().manual_seed(1234)
clouds = [(torch.rand((32, 1, 4096, 3), generator=gen) * 2 - 1).to(dev) for _ in range(2)]
with torch.no_grad():
    for i in range(5): model(clouds[i % 2])
    torch.cuda.synchronize()
    ts = []
    for i in range(60):
        t0 = time.perf_counter(); model(clouds[i % 2]); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ts_sorted = sorted(ts)
print("per-step ms (synchronised each step): min %.3f median %.3f p90 %.3f max %.3f" % (ts_sorted[0], ts_sorted[30], ts_sorted[54], ts_sorted[-1]))
print("first 12:", [round(t, 2) for t in ts[:12]])
# host-side issue time: how long the CPU needs to enqueue one step (no synchronisation in between)
from lpdnet_hip import ops
for prof in (False, True):
    ops.PROFILE = {} if prof else None
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40): model(clouds[i % 2])
        t_issue = (time.perf_counter() - t0) / 40 * 1e3
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / 40 * 1e3
    ops.PROFILE = None
    print(f"events {'on ' if prof else 'off'}: host issue {t_issue:.3f} ms/step, wall {t_all:.3f} ms/step")
