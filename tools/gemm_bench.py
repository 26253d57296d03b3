"""GEMM timing at the pipeline's shapes: f32-input MFMA vs split-bf16 (bf16x3), with the error of each against fp64."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [("conv3 512->1024", 131072, 1024, 512, False, False, 1, 1), ("SN1 proj 128->512", 131072, 512, 128, False, False, 1, 1),
          ("DG1 proj 64->256", 131072, 256, 64, False, False, 1, 1), ("assign 1024->64", 131072, 64, 1024, False, True, 1, 1),
          ("vlad pool (batched, A k-major)", 1024, 64, 4096, True, True, 32, 1), ("hidden 65536->256 split-K", 32, 256, 65536, False, True, 1, 64),
          ("dW conv3 (both k-major, split-K)", 1024, 512, 131072, True, True, 1, 16)]
for name, M, N, K, ak, bk, nb, splits in shapes:
    A = torch.randn((nb, K, M) if ak else (nb, M, K), generator=g).to(dev)
    B = (torch.randn((nb, K, N) if bk else (nb, N, K), generator=g) / K ** 0.5).to(dev)
    if nb == 1:
        A, B = A[0], B[0]
    Al = A.transpose(-1, -2) if ak else A
    Bl = B if bk else B.transpose(-1, -2)
    rows = slice(0, min(M, 512))
    ref = (Al[..., rows, :].double() @ Bl.double())
    line = f"{name:36s} M={M} N={N} K={K} batch={nb}:"
    for exact in (True, False):
        out = ops.gemm(A, B, a_kmajor=ak, b_kmajor=bk, splits=splits, exact=exact)
        err = ((out[..., rows, :].double() - ref).abs().max() / ref.abs().max()).item()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(2):
            ops.gemm(A, B, a_kmajor=ak, b_kmajor=bk, splits=splits, exact=exact, out=out)
        ev[0].record()
        for _ in range(10):
            ops.gemm(A, B, a_kmajor=ak, b_kmajor=bk, splits=splits, exact=exact, out=out)
        ev[1].record(); torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 100
        line += f"  {'f32' if exact else 'x3 '} {us:8.1f} us {2*M*N*K*nb/us/1e6:7.1f} TF err {err:.1e}"
    print(line, flush=True)
