"""Does capturing the eval forward in a HIP graph pay?  python tools/graph_try.py [clouds per step ...]   (default 32)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from util.PointNetVlad import PointNetVlad
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = PointNetVlad(num_points=4096, featnet="lpdnet").to(dev).eval()
BS = [int(a) for a in sys.argv[1:]] or [32]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in BS:
  x = (torch.rand(B, 1, 4096, 3, device=dev) * 2 - 1)
  print("clouds per step", B)
  with torch.no_grad():
      eager = timeit(lambda: m(x))
      print(f"eager {eager:.3f} ms/step")
      s = torch.cuda.Stream()
      s.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(s):
          for _ in range(3):
              y = m(x)
      torch.cuda.current_stream().wait_stream(s)
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g):
          y = m(x)
      ref = m(x)
      g.replay()
      torch.cuda.synchronize()
      print("graph == eager:", (y - ref).abs().max().item())
      print(f"graph {timeit(g.replay):.3f} ms/step")
