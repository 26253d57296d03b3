"""How well does a tile-level estimate predict the number of candidate tiles a wave of the best-first kNN visits?
(input to a longest-first launch order; tools/knn_dist.py shows the distribution itself)"""
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
for name, rows, C in (("xyz", xs.view(B, N, 3), 3), ("feat64", f.view(B, N, 64), 64)):
    st = ops.knn(ops.transpose(rows.contiguous()), k, impl=5).view(B * N, k)[:, :5].float()
    actual = st.view(B, N // 32, 32, 5)[:, :, 0, 0]                      # [B, nt] tiles visited
    t = rows.view(B, N // 32, 32, C)
    cen = t.mean(2)                                                       # [B, nt, C]
    rad = (t - cen.unsqueeze(2)).norm(dim=-1).amax(2)                     # [B, nt]
    d = torch.cdist(cen, cen)                                             # [B, nt, nt]
    gap = (d - rad.unsqueeze(1) - rad.unsqueeze(2)).clamp_min(0)          # lower bound of point distances between tiles
    for nm, pr in (("radius only", rad), ("mean gap to the 16 curve neighbours", torch.stack([gap[:, torch.arange(gap.shape[1]), (torch.arange(gap.shape[1]) + o).clamp(0, gap.shape[1] - 1)] for o in range(-8, 9) if o], 0).mean(0))):
        a, p_ = actual.flatten(), pr.flatten()
        order = torch.argsort(p_, descending=True)
        print(f"{name} predictor '{nm}': corr {torch.corrcoef(torch.stack([a, p_]))[0, 1].item():.3f}  top-half {a[order[: len(a) // 2]].mean().item():.1f} bottom-half {a[order[len(a) // 2:]].mean().item():.1f}")
    for alpha in (0.5, 0.75, 1.0, 1.5):
        pred = (gap <= alpha * rad.unsqueeze(2)).float().sum(2)           # tiles within alpha * own radius
        a, p = actual.flatten(), pred.flatten()
        corr = torch.corrcoef(torch.stack([a, p]))[0, 1].item()
        # quality of a longest-first order: mean actual length of the predicted top half vs the bottom half
        order = torch.argsort(p, descending=True)
        top, bot = a[order[: len(a) // 2]].mean().item(), a[order[len(a) // 2:]].mean().item()
        top5 = a[order[: len(a) // 20]].mean().item()
        print(f"{name} alpha {alpha}: corr {corr:.3f}  pred mean {p.mean():.1f}  actual mean {a.mean():.1f}  top-half {top:.1f} bottom-half {bot:.1f} top-5% {top5:.1f} (actual p95 {torch.quantile(a, 0.95).item():.0f})")

    # list-scheduling simulation: 8 XCDs x 256 wave slots, each XCD owns 4 clouds (512 waves); wave time ~ tiles + c0
    import heapq
    def makespan(lengths):                      # lengths in dispatch order for one XCD
        slots = [0.0] * 256
        heapq.heapify(slots)
        end = 0.0
        for L in lengths:
            t = heapq.heappop(slots) + L
            end = max(end, t)
            heapq.heappush(slots, t)
        return end
    c0 = 8.0                                    # fixed part of a wave (bound table, merge) in tile units
    pred = (gap <= 0.5 * rad.unsqueeze(2)).float().sum(2)
    res = {"current": [], "predicted-longest-first": [], "oracle-longest-first": [], "ideal": []}
    for xcd in range(8):
        a = (actual[xcd * 4:(xcd + 1) * 4] + c0).cpu()          # [4, nt]
        p = pred[xcd * 4:(xcd + 1) * 4].cpu()
        res["current"].append(makespan(a.flatten().tolist()))
        o = torch.argsort(p.flatten(), descending=True)
        res["predicted-longest-first"].append(makespan(a.flatten()[o].tolist()))
        o = torch.argsort(a.flatten(), descending=True)
        res["oracle-longest-first"].append(makespan(a.flatten()[o].tolist()))
        res["ideal"].append(a.sum().item() / 256)
    print("  ", name, "makespan (tile units, max over XCDs):", {k_: round(max(v), 1) for k_, v in res.items()})
