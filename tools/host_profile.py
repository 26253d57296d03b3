"""Where the HOST time of a small-batch eval forward goes (the one-cloud forward is host-bound: 0.55 ms eager against 0.41 ms of GPU
critical path): cProfile over N forwards, top functions by own time and by cumulative time.   python tools/host_profile.py [clouds] [N]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from util.PointNetVlad import PointNetVlad
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = PointNetVlad(num_points=4096, featnet="lpdnet").to(dev).eval()
x = torch.rand(B, 1, 4096, 3, device=dev) * 2 - 1
with torch.no_grad():
    for _ in range(20):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m(x)
    t_enq = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B}: host enqueue {t_enq:.3f} ms per forward, wall {t_all:.3f} ms per forward")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        m(x)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumtime").print_stats(28)
