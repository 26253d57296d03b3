"""lpd_gemm_tn (weight-gradient products dW = A^T B, split-bf16) at the training step's shapes: time and error against fp64.
python tools/tn_bench.py [substring of the shape name | all] [bf16]      (bf16: A as bfloat16 rows -- the bf16-storage mode's conv3 map)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [("dW conv3", 180224, 1024, 512, 1), ("dWc assignment", 180224, 1024, 64, 1), ("dW SN1", 180224, 512, 128, 1),
          ("dW DG1", 180224, 256, 64, 1), ("pooling x44", 4096, 1024, 64, 44), ("pooling x32", 4096, 1024, 64, 32),
          ("dW DG2 (edges)", 3604480, 128, 128, 1)]
only = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "all" else None
a16 = "bf16" in sys.argv[2:]
for name, M, KA, KB, nb in shapes:
    if only and only not in name:
        continue
    A = torch.randn((nb, M, KA) if nb > 1 else (M, KA), generator=g).to(dev)
    if a16:
        A = A.to(torch.bfloat16)
    B = torch.randn((nb, M, KB) if nb > 1 else (M, KB), generator=g).to(dev)
    out = ops.gemm_tn(A, B)
    rows = slice(0, 16384)
    if nb > 1:
        ref = A[0].double().t() @ B[0].double(); got = out[0]
    else:
        ref = None; got = out
        ref = torch.zeros((KA, KB), dtype=torch.float64, device=dev)
        for r0 in range(0, M, 65536):
            ref += A[r0:r0 + 65536].double().t() @ B[r0:r0 + 65536].double()
    err = ((got.double() - ref).abs().max() / ref.abs().max()).item()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        ops.gemm_tn(A, B)
    ev[0].record()
    for _ in range(10):
        ops.gemm_tn(A, B)
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 100
    gb = (A.numel() * A.element_size() + B.numel() * 4) / 1e9
    print(f"{name:18s} M={M} {KA}x{KB} x{nb}: {us:8.1f} us  {(4 if a16 else 6)*M*KA*KB*nb/us/1e6:7.1f} TF(bf16)  {gb/us*1e3:5.2f} TB/s  err {err:.1e}", flush=True)
