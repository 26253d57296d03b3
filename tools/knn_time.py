"""HIP-event time of the two kNN searches inside the eval forward:  python tools/knn_time.py [B N k ...]   (default 32 4096 20  16 16384 64)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import engine, ops
from util.PointNetVlad import PointNetVlad
dev = torch.device("cuda:0")
nums = [int(a) for a in sys.argv[1:]] or [32, 4096, 20, 16, 16384, 64]
engine._SIDE_FORCE.mode = False
for B, N, k in zip(nums[0::3], nums[1::3], nums[2::3]):
    torch.manual_seed(0)
    m = PointNetVlad(num_points=N, featnet="lpdnet")
    m.emb_nn.k = k
    m = m.to(dev).eval()
    x = torch.rand(B, 1, N, 3, device=dev) * 2 - 1
    with torch.no_grad():
        for _ in range(4):
            m(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            m(x)
        e1.record(); torch.cuda.synchronize()
        ops.PROFILE, ops.PROFILE_ONLY = {}, ("knn",)
        for _ in range(6):
            m(x)
        torch.cuda.synchronize()
        kk = {n: sum(a.elapsed_time(b) for a, b in ev[1:]) / (len(ev) - 1) * 1e3 for n, ev in ops.PROFILE.items()}
        ops.PROFILE, ops.PROFILE_ONLY = None, None
    print(f"B={B} N={N} k={k}: {e0.elapsed_time(e1) / 10:.3f} ms/step  " + "  ".join(f"{n} {v:.0f} us" for n, v in sorted(kk.items())), flush=True)
    del m, x
    torch.cuda.empty_cache()
