"""Prepared-fragment split-bf16 GEMM (lpd_gemm_x3w) variants against the generic split-bf16 kernel at the pipeline's shapes.
impl 2 = 128x128 blocks, 3 = 128x256 blocks (wave strip 128x64)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
Bc, Np = 32, 4096
# name, N, K, b_kmajor, a_panels, out_panels
shapes = [("conv3 512->1024 rows", 1024, 512, False, False, False), ("conv3 512->1024 panels A", 1024, 512, False, True, False),
          ("SN1 proj 128->512 panels A,C", 512, 128, False, True, True), ("SN1 proj 128->512 rows", 512, 128, False, False, False),
          ("DG1 proj 64->256 rows", 256, 64, False, False, False),
          ("dX conv3 1024->512 (W k-major)", 512, 1024, True, False, False), ("dX 256->128 (W k-major)", 128, 256, True, False, False)]
M = Bc * Np
for name, N, K, bk, ap, cp in shapes:
    X = torch.randn((M, K), generator=g).to(dev)
    W = (torch.randn((K, N) if bk else (N, K), generator=g) / K ** 0.5).to(dev)
    A = ops.rows_to_panels(X, Bc) if ap else X
    ref = X[:512].double() @ (W.double() if bk else W.double().t())
    line = f"{name:34s} M={M} N={N} K={K}:"
    for label, fwd, impl in (("generic", False, 0), ("x3w-2", True, 2), ("x3w-3", True, 3)):
        if bk and not fwd:
            continue
        ops.X3W_FORWARD, ops.X3W_IMPL = fwd, impl
        call = lambda out=None: ops.gemm(A, W, b_kmajor=bk, a_panels=ap, out_panels=cp, out=out)
        try:
            out = call()
        except Exception as e:
            line += f"  {label} FAILED {e}"
            continue
        rows = ops.panels_to_rows(out)[:512] if cp else out[:512]
        err = ((rows.double() - ref).abs().max() / ref.abs().max()).item()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(2):
            call(out)
        ev[0].record()
        for _ in range(10):
            call(out)
        ev[1].record(); torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 100
        line += f"  {label} {us:7.1f} us {6*M*N*K/us/1e6:6.0f} TF err {err:.1e}"
    print(line, flush=True)
