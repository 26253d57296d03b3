"""kNN timing at the pipeline's two shapes (B=32, N=4096, k=20; C=64 features / C=3 xyz), point-major entry, all launches
of the op included (operand image, tile statistics, launch order, search).  LPD_KNN_ORDER=0 disables the longest-first order."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import synth
from lpdnet_hip import ops
B, N, k = 32, 4096, 20
x = torch.from_numpy(synth.cloud(1234, B, N)).unsqueeze(1).cuda()
xs = ops.morton_sort(x)
g = torch.Generator().manual_seed(0)
W1 = torch.randn(64, 3, generator=g).cuda(); W2 = (torch.randn(64, 64, generator=g) / 8).cuda()
f = torch.nn.functional.leaky_relu(torch.nn.functional.leaky_relu(xs.view(B * N, 3) @ W1.t(), 0.01) @ W2.t(), 0.01)
for name, rows in (("feat64", f.contiguous()), ("xyz", xs.view(B * N, 3).contiguous())):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3): ops.knn_pm(rows, B, N, k)
    ev[0].record()
    for _ in range(10): ops.knn_pm(rows, B, N, k)
    ev[1].record(); torch.cuda.synchronize()
    print("order", os.environ.get("LPD_KNN_ORDER", "1"), name, "%.1f us" % (ev[0].elapsed_time(ev[1]) * 100), flush=True)
