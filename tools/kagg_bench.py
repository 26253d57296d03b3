"""K-agg timing on the bench workload (B=32, N=4096, k=20, Z-ordered clouds, real kNN graphs) + exactness check.
usage: LPD_KAGG=<T*1000+CS | 0> python tools/kagg_bench.py [B] [N] [k]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import synth
from lpdnet_hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.cloud(1234, B, N)).unsqueeze(1).to(dev)
xs = ops.morton_sort(x)
idx = ops.knn(ops.transpose(xs.view(B, N, 3)), k)
M = B * N
g = torch.Generator(device="cpu").manual_seed(1)
for C in (256, 128):
    pq = torch.randn((M, 2 * C), generator=g).to(dev)
    scale = (torch.rand(C, generator=g) - 0.3).to(dev)
    shift = torch.randn(C, generator=g).to(dev)
    out = torch.empty((M, C), device=dev)
    ops.edge_gather_max(pq[:, :C], pq[:, C:], idx, N, scale=scale, shift=shift, act=ops.ACT_LEAKY, slope=0.01, out=out)
    # exactness against a plain gather (chunked)
    bad = 0
    for b in range(0, B, 4):
        P = pq[b * N:(b + 4) * N, :C].view(-1, N, C)
        ii = idx.view(B, N, k)[b:b + 4].long()
        gath = torch.stack([P[c][ii[c]] for c in range(P.shape[0])])          # [4,N,k,C]
        sel = torch.where(scale >= 0, gath.amax(2), gath.amin(2))
        ref = torch.nn.functional.leaky_relu(scale * (sel + pq[b * N:(b + 4) * N, C:].view(-1, N, C)) + shift, 0.01)
        bad += (ref.view(-1, C) != out[b * N:(b + 4) * N]).sum().item()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        ops.edge_gather_max(pq[:, :C], pq[:, C:], idx, N, scale=scale, shift=shift, act=ops.ACT_LEAKY, slope=0.01, out=out)
    ev[0].record()
    R = 20
    for _ in range(R):
        ops.edge_gather_max(pq[:, :C], pq[:, C:], idx, N, scale=scale, shift=shift, act=ops.ACT_LEAKY, slope=0.01, out=out)
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / R
    alg = M * (3 * C * 4 + 4 * k)
    # cloud-resident kernel
    if k == 20 and N <= 4096:
        idx16 = ops.pack_idx16(idx)
        out3 = torch.empty((M, C), device=dev)
        def run16():
            ops.edge_gather_max16(pq[:, :C], pq[:, C:], idx16, N, scale=scale, shift=shift, act=ops.ACT_LEAKY, slope=0.01, out=out3)
        run16()
        bad3 = (out3 != out).sum().item()
        for _ in range(3): run16()
        ev[0].record()
        for _ in range(R): run16()
        ev[1].record(); torch.cuda.synchronize()
        us3 = ev[0].elapsed_time(ev[1]) * 1e3 / R
        print(f"   idx16 row-major C={C}: {us3:.1f} us -> {alg/us3/1e6:.2f} TB/s ({alg/us3/1e6/8*100:.1f}%)  mismatches {bad3}", flush=True)
    print(f"LPD_KAGG={os.environ.get('LPD_KAGG','default')} C={C} B={B} N={N} k={k}: {us:.1f} us  alg {alg/1e6:.0f} MB -> {alg/us/1e6:.2f} TB/s "
          f"({alg/us/1e6/8*100:.1f}% of 8 TB/s)  mismatches {bad}", flush=True)
