"""HIP-event time of the NetVLAD hidden projection [M, 65536] x [65536, 256] (+ slab reduction) by row count:  python tools/head_time.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import ops
dev = torch.device("cuda:0")
K, N = 65536, 256
W = torch.randn(K, N, device=dev) / 256
sc, sh = torch.rand(N, device=dev), torch.rand(N, device=dev)
scratch = torch.zeros(128 << 20, device=dev)     # 512 MB READ between repetitions (clean lines): the weights come from HBM, not from the caches
MODE = os.environ.get("HEAD_FLUSH", "read")      # read | write (a 256-MB fill leaves the caches full of dirty lines) | none (weights cache-resident)
for M in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 32, 44]:
    A = torch.randn(M, K, device=dev)
    ref = (A.double() @ W.double()) * sc.double() + sh.double()
    out = ops.gemm(A, W, b_kmajor=True, scale=sc, shift=sh, splits=K // 128, exact=True)
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    ts = []
    for _ in range(10):
        if MODE == "read":
            scratch.sum()
        elif MODE == "write":
            scratch[: 64 << 20].zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(A, W, b_kmajor=True, scale=sc, shift=sh, splits=K // 128, exact=True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    t = ts[len(ts) // 2]
    print(f"M={M:3d}: {t:6.1f} us  = {4 * K * N / t / 1e6:5.2f} TB/s of weights, {0.5 * K * N * 4 / t / 1e6 / 4:.2f} of 8 TB/s ... rel err {err:.1e}", flush=True)
