import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import synth
from lpdnet_hip import ops
B, N, k = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 4096, 20)))
x = torch.from_numpy(synth.cloud(1234, B, N)).unsqueeze(1).cuda()
xs = ops.morton_sort(x)
for name, feat in (("xyz", ops.transpose(xs.view(B, N, 3))), ("feat64", None)):
    if feat is None:
        g = torch.Generator().manual_seed(0)
        W1 = torch.randn(64, 3, generator=g).cuda(); W2 = (torch.randn(64, 64, generator=g) / 8).cuda()
        f = torch.nn.functional.leaky_relu(torch.nn.functional.leaky_relu(xs.view(B * N, 3) @ W1.t(), 0.01) @ W2.t(), 0.01)
        feat = ops.transpose(f.view(B, N, 64).contiguous())
    for simpl, label in ((5, "z-walk"),):
        raw = ops.knn(feat, k, impl=simpl).view(B * N, k)
        st = raw[:, :5].float()
        w = st.view(-1, 32, 5)[:, 0, :]     # per wave (first query of each tile)
        if k >= 12:
            ph = raw.view(-1, 32, k)[:, 0, 5:12].float()
            print(name, "low-precision tests/wave %.1f; cycles/wave: advance %.0f  exact tile %.0f  select+queue %.0f  drain %.0f  total %.0f; bound tests %.1f" % tuple(ph.mean(0).tolist()), flush=True)
        print(name, label, "tiles visited/wave %.1f  drain iterations %.1f  drains %.1f  admitted per half-lane %.1f / %.1f" % (
            w[:, 0].mean(), w[:, 1].mean(), w[:, 3].mean(), st[:, 2].mean(), st[:, 4].mean()), flush=True)
    for impl in (4, 6):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(2): ops.knn(feat, k, impl=impl)
        ev[0].record()
        for _ in range(5): ops.knn(feat, k, impl=impl)
        ev[1].record(); torch.cuda.synchronize()
        print("   impl", impl, "%.1f us (B=%d)" % (ev[0].elapsed_time(ev[1]) * 200, B), flush=True)
