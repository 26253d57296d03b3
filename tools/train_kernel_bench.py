"""Kernels of the bf16 training step timed in isolation at configs[2]'s shapes (44 clouds x 4096 points):

    python tools/train_kernel_bench.py [conv3] [map] [bnbwd]      (default: all three)

  conv3  conv3 of the step (180224 x 1024 x 512) in the forms that show what its epilogue costs: bf16 rows -> bf16 map + statistics (the
         step's launch), fp32 rows, fp32 map, the same product without statistics and with fp32 stores, the library's bf16 GEMM (yardstick)
  map    the four launches that stream the [180224, 1024] bf16 conv3 map (assignment, pooling, dA, assignment dW): TB/s of map bytes
  bnbwd  bn3 backward on the bf16 map (two passes); LPD_DEBUG=reduce-grid=4096 restores round 5's 4096-block reduction launches
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lpd-net-pytorch_amd")]
import torch
from lpdnet_hip import ops

dev = torch.device("cuda:0")
B, N, E, K = 44, 4096, 1024, 64
M = B * N
torch.manual_seed(0)


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


which = [a for a in sys.argv[1:] if a in ("conv3", "map", "bnbwd")] or ["conv3", "map", "bnbwd"]
tag = os.environ.get("LPD_DEBUG", "")
with ops.bf16_gemm():
    if "conv3" in which:
        x = torch.randn(M, 512, device=dev)
        x16 = x.bfloat16()
        w = torch.randn(E, 512, device=dev) * 0.05
        bn = torch.nn.BatchNorm2d(E).to(dev).train()
        out = torch.empty(M, E, device=dev)
        y16 = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
        for name, fn in [("bf16 rows -> bf16 map + statistics (the step's conv3)", lambda: ops.linear_bn_stats(x16, w, bn, out_bf16=True)),
                         ("fp32 rows -> bf16 map + statistics", lambda: ops.linear_bn_stats(x, w, bn, out_bf16=True)),
                         ("fp32 rows -> fp32 map + statistics", lambda: ops.linear_bn_stats(x, w, bn)),
                         ("bf16 rows -> fp32 map, no statistics (gemm_bf16a)", lambda: ops.gemm_bf16a(x16, w, b_kmajor=False, out=out)),
                         ("torch bf16 matmul (library yardstick, ONE product)", lambda: torch.matmul(x16, w.bfloat16().t(), out=y16))]:
            print(f"conv3 {tag:24s} {name:58s} {t(fn):7.1f} us", flush=True)
        del x, x16, out, y16
    if "map" in which or "bnbwd" in which:
        feat = torch.randn(M, E, device=dev).bfloat16()
    if "map" in which:
        wkn = torch.randn(E, K, device=dev) * 0.05
        sc, sh = torch.rand(E, device=dev) + 0.5, torch.randn(E, device=dev) * 0.1
        a = torch.softmax(torch.randn(M, K, device=dev), 1)
        dv = torch.randn(B, E, K, device=dev) * 0.01
        da0 = torch.randn(M, K, device=dev) * 0.01
        aff = (sc, sh, ops.ACT_LEAKY, 0.01)
        gb = M * E * 2 / 1e9
        for name, fn in [("assignment + bn3 / act in the loader (gemm_act)", lambda: ops.gemm_act(feat, wkn, sc, sh, ops.ACT_LEAKY, 0.01, out_bf16=True, store=False)),
                         ("pooling on the raw map (gemm_tn, batched)", lambda: ops.gemm_tn(feat.view(B, N, E), a.view(B, N, K), a_affine=aff)),
                         ("dA (batched, per-cloud weights)", lambda: ops.gemm(feat.view(B, N, E), dv, a_kmajor=False, b_kmajor=True, a_affine=aff)),
                         ("assignment dW (gemm_tn)", lambda: ops.gemm_tn(feat, da0, a_affine=aff))]:
            us = t(fn)
            print(f"map   {tag:24s} {name:58s} {us:7.1f} us  {gb / us * 1e3:5.2f} TB/s of map bytes", flush=True)
    if "bnbwd" in which:
        g = (torch.randn(M, E, device=dev) * 0.01).bfloat16()
        bn = torch.nn.BatchNorm2d(E).to(dev).train()
        st = ops.bn_train_stats(feat.float(), bn)
        us = t(lambda: ops.bn_act_bwd_bf16(g, feat, st, ops.ACT_LEAKY, 0.01))
        print(f"bnbwd {tag:24s} {'bn3 backward on the bf16 map (reduce + apply)':58s} {us:7.1f} us", flush=True)
