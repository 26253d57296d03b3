"""Where does the HOST spend its time per eval forward?  python tools/host_profile.py"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from util.PointNetVlad import PointNetVlad
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = PointNetVlad(num_points=4096, featnet="lpdnet").to(dev).eval()
x = (torch.rand(32, 1, 4096, 3, device=dev) * 2 - 1)
with torch.no_grad():
    for _ in range(8):
        m(x)
    torch.cuda.synchronize()
    # host-only time: enqueue 50 forwards without waiting for the GPU in between (the queue absorbs them) vs. wall with sync
    t0 = time.perf_counter()
    for _ in range(50):
        m(x)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"host enqueue {t_enq / 50 * 1e3:.3f} ms/forward; with the GPU {t_all / 50 * 1e3:.3f} ms/forward")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        m(x)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)

# is the host throttled by the runtime's queue (it would then run at the GPU's pace, a fixed number of launches ahead)?
with torch.no_grad():
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for _ in range(40):
        m(x)
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    d = [(b - a) * 1e3 for a, b in zip(ts, ts[1:])]
    print("host time of forwards 1..40 (ms):", " ".join(f"{v:.2f}" for v in d))
