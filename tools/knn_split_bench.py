"""Small-batch eval forward with and without the multi-wave kNN (LPD_KNN_SPLIT is read once per process: one child process per mode):
    python tools/knn_split_bench.py [clouds per step ...]          (default 1 6 10 16 24 32)
Prints ms per step (wall, 40 forwards) and the HIP-event time of the two kNN searches (one profiled forward sequence on one stream)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.join(%r, "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import engine, ops
from util.PointNetVlad import PointNetVlad
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = PointNetVlad(num_points=4096, featnet="lpdnet").to(dev).eval()
for B in [int(a) for a in sys.argv[1:]]:
    x = (torch.rand(B, 1, 4096, 3, device=dev) * 2 - 1)
    with torch.no_grad():
        for _ in range(8): m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40): m(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
        engine._SIDE_FORCE.mode = False
        ops.PROFILE, ops.PROFILE_ONLY = {}, ("knn",)
        for _ in range(6): m(x)
        torch.cuda.synchronize()
        k = {n: sum(a.elapsed_time(b) for a, b in ev[1:]) / (len(ev) - 1) * 1e3 for n, ev in ops.PROFILE.items()}
        ops.PROFILE, ops.PROFILE_ONLY = None, None
        engine._SIDE_FORCE.mode = None
    print(f"  B={B:3d}: {ms:.3f} ms/step   " + "  ".join(f"{n} {v:.0f} us" for n, v in sorted(k.items())), flush=True)
''' % HERE
args = sys.argv[1:] or ["1", "6", "10", "16", "24", "32"]
for mode in ("0", "default", "1"):
    env = dict(os.environ)
    env.pop("LPD_KNN_SPLIT", None)
    if mode != "default":
        env["LPD_KNN_SPLIT"] = mode
    print(f"LPD_KNN_SPLIT={mode}", flush=True)
    subprocess.run([sys.executable, "-c", CHILD] + args, env=env, check=False)
