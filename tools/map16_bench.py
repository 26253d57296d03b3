"""The elementwise / short-product kernels of the bf16 conv3 map at the training step's shape (180224 x 1024), fp32 against bf16 tensors:
python tools/map16_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, C = 180224, 1024
x = torch.randn(M, C, generator=g).to(dev)
dy = torch.randn(M, C, generator=g).to(dev)
bn = torch.nn.BatchNorm1d(C).to(dev)
st = ops.bn_train_stats(x, bn)
x16, dy16 = x.to(torch.bfloat16), dy.to(torch.bfloat16)


def timeit(name, fn, gb):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 100
    print(f"{name:34s} {us:8.1f} us  {gb / us * 1e3:5.2f} TB/s", flush=True)


e = M * C / 1e9
timeit("bn_act_bwd fp32", lambda: ops.bn_act_bwd(dy, x, st, ops.ACT_LEAKY, 0.2), e * 4 * 5)
timeit("bn_act_bwd bf16", lambda: ops.bn_act_bwd_bf16(dy16, x16, st, ops.ACT_LEAKY, 0.2), e * 2 * 5)
w = (torch.randn(C, 64, generator=g) / 32).to(dev)
timeit("assignment + act fp32", lambda: ops.gemm_act(x, w, st.scale, st.shift, ops.ACT_LEAKY, 0.2), e * 8)
timeit("assignment + act bf16", lambda: ops.gemm_act(x16, w, st.scale, st.shift, ops.ACT_LEAKY, 0.2, out_bf16=True), e * 4)
w3 = (torch.randn(C, 512, generator=g) / 32).to(dev)
timeit("dX = dY W fp32", lambda: ops.gemm(dy, w3, b_kmajor=True), e * 4 + M * 512 * 4 / 1e9)
timeit("dX = dY W bf16", lambda: ops.gemm_bf16a(dy16, w3, b_kmajor=True), e * 2 + M * 512 * 4 / 1e9)
ada = torch.randn(44, 4096, 128, generator=g).to(dev)
rhs = torch.randn(44, C, 128, generator=g).to(dev)
timeit("dfeat (x3t rows) fp32", lambda: ops.gemm(ada, rhs, a_kmajor=False, b_kmajor=False), e * 4)
timeit("dfeat (x3t rows) bf16", lambda: ops.gemm(ada, rhs, a_kmajor=False, b_kmajor=False, out_bf16=True), e * 2)
dv = torch.randn(44, C, 64, generator=g).to(dev)
timeit("dA batched fp32", lambda: ops.gemm(x.view(44, 4096, C), dv, a_kmajor=False, b_kmajor=True), e * 4)
timeit("dA batched bf16", lambda: ops.gemm(x16.view(44, 4096, C), dv, a_kmajor=False, b_kmajor=True), e * 2)
cat = torch.randn(M, 512, generator=g).to(dev)
wc3 = (torch.randn(C, 512, generator=g) / 22).to(dev)
timeit("conv3 + stats fp32", lambda: ops.linear_bn_stats(cat, wc3, bn), e * 4 + M * 512 * 4 / 1e9)
timeit("conv3 + stats bf16", lambda: ops.linear_bn_stats(cat, wc3, bn, out_bf16=True), e * 2 + M * 512 * 4 / 1e9)
