"""Windowed K-agg (lpd_edge_gather_maxw) against the direct gather at the stress shape: python tools/kaggw_bench.py [B] [N] [k]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
C = 256
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = (torch.rand(B, 1, N, 3, generator=g) * 2 - 1).to(dev)
xs = ops.morton_sort(x)
idx = ops.knn_pm(xs.view(B * N, 3), B, N, k)
M = B * N
d = (idx.view(B, N, k) - torch.arange(N, device=dev).view(1, N, 1)).abs()
w = torch.arange(N, device=dev).view(1, N, 1) // 4095
miss = (idx.view(B, N, k) // 4095 != w).float().mean().item()
print(f"B={B} N={N} k={k}: out-of-window neighbours {miss:.4f}; |j - i| median {d[:1].float().median().item():.0f} p90 {d[:1].float().flatten()[:8000000].quantile(0.9).item():.0f}")
PQ = torch.randn(M, 2 * C, generator=g).to(dev)
scale, shift = (torch.rand(C, generator=g) - 0.3).to(dev), torch.randn(C, generator=g).to(dev)
pq = ops.panels_empty(B, N, 2 * C, dev)
pq[:, :C // 8] = ops.rows_to_panels(PQ[:, :C].contiguous(), B)
pq[:, C // 8:] = ops.rows_to_panels(PQ[:, C:].contiguous(), B)
outp = ops.panels_empty(B, N, C, dev)
out = torch.empty(M, C, device=dev)
i16 = ops.pack_idx16w(idx)                       # [B, N, k]: out-of-window neighbours first (the product setting)
i16_plain = ops.pack_idx16w(idx.view(-1, k))     # kNN order
near = (torch.arange(N, device=dev).view(1, N, 1) // 4095 * 4095 + torch.randint(0, 4000, (B, N, k), device=dev)).clamp_(max=N - 1).to(torch.int32)
i16n = ops.pack_idx16w(near)
i16n_plain = ops.pack_idx16w(near.view(-1, k))
# all-hit AND conflict-free: the 8 points of a ds_read_b128 lane group read rows of 8 different residues mod 8 at every step
# (point i, step t -> row i + 8 t inside its window): what the LDS gathers cost with no bank conflict at all
ar = torch.arange(N, device=dev).view(1, N, 1)
wbase = ar // 4095 * 4095
wlen = (N - wbase).clamp(max=4095)
seq = (wbase + (ar - wbase + 8 * torch.arange(k, device=dev).view(1, 1, k)) % wlen).expand(B, N, k).contiguous().to(torch.int32)
i16s = ops.pack_idx16w(seq.view(-1, k))
alg = (3 * C * 4 + 4 * k) * M


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


kw = dict(scale=scale, shift=shift, act=ops.ACT_LEAKY, slope=0.01)
for name, fn in [("direct, row-major", lambda: ops.edge_gather_max(PQ[:, :C], PQ[:, C:], idx, N, out=out, **kw)),
                 ("window, row-major", lambda: ops.edge_gather_maxw(PQ[:, :C], PQ[:, C:], i16, N, out=out, **kw)),
                 ("window, panels", lambda: ops.edge_gather_maxw(pq[:, :C // 8], pq[:, C // 8:], i16, N, out=outp, **kw)),
                 ("window, panels, kNN-order lists", lambda: ops.edge_gather_maxw(pq[:, :C // 8], pq[:, C // 8:], i16_plain, N, out=outp, **kw)),
                 ("window, panels, all-hit, kNN order", lambda: ops.edge_gather_maxw(pq[:, :C // 8], pq[:, C // 8:], i16n_plain, N, out=outp, **kw)),
                 ("pack_idx16w (permuting)", lambda: ops.pack_idx16w(idx)),
                 ("pack_idx16w (plain)", lambda: ops.pack_idx16w(idx.view(-1, k))),
                 ("window, panels, all-hit graph", lambda: ops.edge_gather_maxw(pq[:, :C // 8], pq[:, C // 8:], i16n, N, out=outp, **kw)),
                 ("window, panels, all-hit, conflict-free", lambda: ops.edge_gather_maxw(pq[:, :C // 8], pq[:, C // 8:], i16s, N, out=outp, **kw)),
                 ("window, row-major, all-hit graph", lambda: ops.edge_gather_maxw(PQ[:, :C], PQ[:, C:], i16n, N, out=out, **kw))]:
    us = timeit(fn)
    print(f"{name:36s} {us:9.1f} us  {alg / us / 1e3:7.1f} GB/s algorithmic = {alg / us / 1e3 / 8000:.3f} of 8 TB/s")
