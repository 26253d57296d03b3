"""What the library GEMM reaches on this box at the conv3 shape (a yardstick for the hand-written kernels, not a code path):
python tools/blas_probe.py"""
import torch
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M, N = 180224, 1024
for K in (512, 1536):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.mm(a, b.t()))
    print(f"bf16 mm {M}x{N}x{K}: {us:.0f} us  {2*M*N*K/us/1e6:.0f} TF/s", flush=True)
a = torch.randn(M, 512, device=dev); b = torch.randn(N, 512, device=dev)
us = t(lambda: torch.mm(a, b.t()))
print(f"fp32 mm {M}x{N}x512: {us:.0f} us  {2*M*N*512/us/1e6:.0f} TF/s")
a = torch.randn(131072, 512, device=dev, dtype=torch.bfloat16)
us = t(lambda: torch.mm(a, torch.randn(1024, 512, device=dev, dtype=torch.bfloat16).t()))
print(f"bf16 mm 131072x1024x512 (incl. randn): {us:.0f} us")
