"""Cloud-panel K-agg (SN1 stage: C=256) timing as the pipeline runs it, against buffer placement: the three streams (P, Q
panels of one buffer, out panels of another) are offset against each other by a pad allocated in front."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import synth
from lpdnet_hip import ops
B, N, k, C = 32, 4096, 20, 256
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.cloud(1234, B, N)).unsqueeze(1).to(dev)
xs = ops.morton_sort(x)
idx = ops.knn_pm(xs.view(B * N, 3).contiguous(), B, N, k)
i16 = ops.pack_idx16(idx)
g = torch.Generator().manual_seed(1)
scale = (torch.rand(C, generator=g) - 0.3).to(dev); shift = torch.randn(C, generator=g).to(dev)
alg = B * N * (3 * C * 4 + 4 * k)
big = torch.empty(3 * 1024 * 1024 * 1024 // 4, device=dev)      # one arena: explicit placement
PAD_ROWS = int(os.environ.get("PAD_ROWS", ops.PANEL_PAD_ROWS))
def panels(off_floats, ch):
    ld = N + PAD_ROWS
    n = B * (ch // 8) * ld * 8
    t = big[off_floats:off_floats + n].view(B, ch // 8, ld, 8)[:, :, :N]
    return t, off_floats + n
for pad_kb in (0, 4100, 16384 + 36):
    off = 0
    pq3, off = panels(off, 512)
    off += pad_kb * 256
    cat, off = panels(off, 512)
    pq3.normal_()
    def run():
        ops.edge_gather_max16(pq3[:, 0:32], pq3[:, 32:64], i16, N, scale=scale, shift=shift, act=ops.ACT_LEAKY, slope=0.01, out=cat[:, 32:64])
    for _ in range(3): run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20): run()
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 50
    print(f"rows+{PAD_ROWS} pad {pad_kb:6d} KiB: {us:6.1f} us  {alg/us/1e6:.2f} TB/s ({alg/us/1e6/8*100:.1f} %)   pq3 @ {pq3.data_ptr() % (1<<21):#x} cat @ {cat.data_ptr() % (1<<21):#x}", flush=True)
