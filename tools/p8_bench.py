"""lpd_gemm_p8 (conv3 on pre-split cloud panels) against fp64 and against the prepared-fragment kernel it replaces."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
import torch
from lpdnet_hip import ops

dev = torch.device("cuda:0")
B, N, K, NO = int(os.environ.get("B", 32)), int(os.environ.get("N", 4096)), 512, 1024
g = torch.Generator(device="cpu").manual_seed(3)
X = (torch.randn((B * N, K), generator=g) * 1.5).to(dev)
W = (torch.randn((NO, K), generator=g) * 0.05).to(dev)
sc = (0.5 + torch.rand(NO, generator=g)).to(dev)
sh = (0.1 * torch.randn(NO, generator=g)).to(dev)
P = ops.rows_to_panels(X, B)
S = ops.split_panels(P)
torch.cuda.synchronize()
# split planes reproduce x to 2^-17
hi, lo = S[0].float(), S[1].float()
rec = ops.panels_to_rows((hi + lo).contiguous()) if False else (hi + lo).permute(0, 2, 1, 3).reshape(B * N, K)
print("split residual", ((rec - X).abs().max() / X.abs().max()).item())
ref_rows = torch.arange(0, B * N, 997, device=dev)
ref = torch.nn.functional.leaky_relu((X[ref_rows].double() @ W.double().t()) * sc.double() + sh.double(), 0.01)
for panels in (False, True):
    out = ops.gemm_p8(S, W, scale=sc, shift=sh, act=ops.ACT_LEAKY, slope=0.01, out_panels=panels)
    torch.cuda.synchronize()
    rows = ops.panels_to_rows(out) if panels else out
    err = ((rows[ref_rows].double() - ref).abs().max() / ref.abs().max()).item()
    print("p8 out_panels=%s: max err vs fp64 (norm-rel) %.2e" % (panels, err))
old = ops.gemm(P, W, b_kmajor=False, a_panels=True, scale=sc, shift=sh, act=ops.ACT_LEAKY, slope=0.01)
torch.cuda.synchronize()
print("x3w err %.2e; p8 vs x3w %.2e" % (((old[ref_rows].double() - ref).abs().max() / ref.abs().max()).item(),
                                         ((old - ops.gemm_p8(S, W, scale=sc, shift=sh, act=ops.ACT_LEAKY, slope=0.01)).abs().max() / old.abs().max()).item()))

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]

outr = torch.empty((B * N, NO), device=dev)
outp = ops.panels_empty(B, N, NO, dev)
W2 = torch.nn.Parameter((torch.randn((NO, 64), generator=g) * 0.03).to(dev))
for impl in (0, 5, 6):
    ops.P8_IMPL = impl
    if impl == 0:
        print("fused assignment: rows median/min us", timeit(lambda: ops.gemm_p8(S, W, shift=sh, act=ops.ACT_LEAKY, out=outr, assign_w=W2.data)))
    print("impl", impl, "p8 rows   median/min us", timeit(lambda: ops.gemm_p8(S, W, scale=sc, shift=sh, act=ops.ACT_LEAKY, out=outr)))
    print("impl", impl, "p8 panels median/min us", timeit(lambda: ops.gemm_p8(S, W, scale=sc, shift=sh, act=ops.ACT_LEAKY, out=outp, out_panels=True)))
for impl, name in ((8, "no LDS-DMA in the loop"), (16, "no MFMAs (6 units ahead)"), (16 + 5, "no MFMAs (5 units ahead)"), (24, "neither (reads + barriers + stores)"), (32, "full, nt loads"), (64, "full, every phase's 12 MFMAs issued twice"), (96, "full, WITH s_setprio around the MFMAs"), (128, "full, NO output stores")):
    ops.P8_IMPL = impl
    try:
        print("timing-only:", name, timeit(lambda: ops.gemm_p8(S, W, shift=sh, act=ops.ACT_LEAKY, out=outr)))
    except Exception as exc:      # the product library rejects them: rebuild with LPD_EXTRA_FLAGS=-DLPD_P8_BENCH python -m lpdnet_hip._build
        print("timing-only variants are not in this build (LPD_EXTRA_FLAGS=-DLPD_P8_BENCH):", str(exc).splitlines()[0])
        break
ops.P8_IMPL = 0
print("x3w (current conv3)    median/min us", timeit(lambda: ops.gemm(P, W, b_kmajor=False, a_panels=True, scale=sc, shift=sh, act=ops.ACT_LEAKY, out=outr)))
print("split_panels           median/min us", timeit(lambda: ops.split_panels(P, out=S)))
# repeatability / race screen: 30 runs must be bit-identical
base = ops.gemm_p8(S, W, scale=sc, shift=sh, act=ops.ACT_LEAKY).clone()
bad = 0
for i in range(30):
    o = ops.gemm_p8(S, W, scale=sc, shift=sh, act=ops.ACT_LEAKY)
    bad += int(not torch.equal(o, base))
print("race screen: %d of 30 runs differ" % bad)
