import os, sys
sys.path.insert(0, "lpd-net-pytorch_amd"); sys.path.insert(0, ".")
import torch
from oracle import synth
from lpdnet_hip import ops
B, N, k = 32, 4096, 20
x = torch.from_numpy(synth.cloud(1234, B, N)).unsqueeze(1).cuda()
xs = ops.morton_sort(x)
g = torch.Generator().manual_seed(0)
W1 = torch.randn(64, 3, generator=g).cuda(); W2 = (torch.randn(64, 64, generator=g) / 8).cuda()
f = torch.nn.functional.leaky_relu(torch.nn.functional.leaky_relu(xs.view(B * N, 3) @ W1.t(), 0.01) @ W2.t(), 0.01)
for name, feat in (("xyz", ops.transpose(xs.view(B, N, 3))), ("feat64", ops.transpose(f.view(B, N, 64).contiguous()))):
    st = ops.knn(feat, k, impl=5).view(B * N, k)[:, :5].float()
    w = st.view(-1, 32, 5)[:, 0, :]
    t = w[:, 0]
    q = torch.quantile(t, torch.tensor([0.5, 0.9, 0.99, 1.0]).cuda())
    it = w[:, 1]
    qi = torch.quantile(it, torch.tensor([0.5, 0.9, 0.99, 1.0]).cuda())
    print(name, "tiles mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f | drain-iters mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (t.mean(), *q.tolist(), it.mean(), *qi.tolist()))
    # per-cloud position profile: mean tiles by tile index
    tw = t.view(B, -1).mean(0)
    print("   by tile position (16 bins):", [round(v, 0) for v in tw.view(16, -1).mean(1).tolist()])
