"""conv3-shaped prepared-fragment GEMM in a loop (for rocprofv3 --pmc): python tools/x3w_one.py [panels]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import ops
dev = torch.device("cuda:0")
Bc, Np, K, N = 32, 4096, 512, 1024
X = torch.randn(Bc * Np, K, device=dev)
W = torch.nn.Parameter(torch.randn(N, K, device=dev) / K ** 0.5)
A = ops.rows_to_panels(X, Bc) if len(sys.argv) > 1 else X
with torch.no_grad():
    for _ in range(6):
        out = ops.gemm(A, W, b_kmajor=False, a_panels=len(sys.argv) > 1)
torch.cuda.synchronize()
