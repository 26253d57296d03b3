"""One GEMM shape in a loop (for rocprofv3 --pmc): python tools/gemm_one.py M N K ak bk exact [batch] [splits]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import ops
M, N, K, ak, bk, exact = [int(v) for v in sys.argv[1:7]]
nb = int(sys.argv[7]) if len(sys.argv) > 7 else 1
splits = int(sys.argv[8]) if len(sys.argv) > 8 else 1
dev = torch.device("cuda:0")
A = torch.randn((nb, K, M) if ak else (nb, M, K)).to(dev)
B = torch.randn((nb, K, N) if bk else (nb, N, K)).to(dev)
if nb == 1:
    A, B = A[0], B[0]
for _ in range(5):
    out = ops.gemm(A, B, a_kmajor=bool(ak), b_kmajor=bool(bk), splits=splits, exact=bool(exact))
torch.cuda.synchronize()
