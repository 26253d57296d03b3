"""What a 16-bit key plane in the K-agg would meet: on the SN1 stage of a real forward (configs[1] shape by
default), for every (point, channel): does the maximum over the k neighbours tie on the HIGH 16 bits of the order-preserving integer
image of the fp32 projection (i.e. would the low halves be needed to name the maximum)?  And how many DISTINCT neighbours hold the
arg-max of the 8 channels of a K-agg slice (each needs its own low-half read)?
    python tools/kagg_keyplane_study.py [B N k]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from oracle import synth, lpd_oracle as orc
from lpdnet_hip import engine
from util.PointNetVlad import PointNetVlad
B, N, k = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 4096, 20)))
dev = torch.device("cuda:0")
m = PointNetVlad(num_points=N, featnet="lpdnet")
m.load_state_dict(orc.synthetic_state("lpdnet", num_points=N))
m.emb_nn.k = k
m = m.to(dev).eval()
x = torch.from_numpy(synth.scene_cloud(7, B, N)).unsqueeze(1).to(dev)
engine.DEBUG_AUX = {}
with torch.no_grad():
    m(x)
aux, engine.DEBUG_AUX = engine.DEBUG_AUX, None
x2 = aux["cat"][:, 128:256].float()                      # [M, 128] rows (in the engine's Z-order)
idx = aux["idx_xyz"].long()                              # [B, N, k]
W = m.emb_nn.convSN1[0].weight.reshape(256, 256)[:, :128]           # neighbour half of the split weight
P = x2 @ W.t()                                           # [M, 256]
bits = P.view(torch.int32)
key = torch.where(bits < 0, ~bits, bits | (-2 ** 31)).to(torch.int64) & 0xffffffff      # order-preserving unsigned image
hi = key >> 16
tie = torch.zeros((), device=dev, dtype=torch.float64)
distinct = torch.zeros((), device=dev, dtype=torch.float64)
cnt = 0
for b in range(B):
    nb = idx[b] + b * N                                  # [N, k]
    for c0 in range(0, 256, 64):
        h = hi[nb][:, :, c0:c0 + 64]                     # [N, k, 64]
        kk = key[nb][:, :, c0:c0 + 64]
        mh = h.max(dim=1, keepdim=True).values
        tie += ((h == mh).sum(dim=1) > 1).double().sum()
        am = kk.argmax(dim=1).view(N, 8, 8)              # arg-max neighbour slot per channel, grouped by 8-channel slice
        distinct += torch.stack([(am == s).any(dim=2) for s in range(k)], 0).sum(0).double().sum()
        cnt += N * 64
print(f"B={B} N={N} k={k}: (point, channel) pairs whose maximum ties on the high 16 bits: {100 * tie.item() / cnt:.1f} %")
print(f"distinct arg-max neighbours per (point, 8-channel slice): {distinct.item() / (cnt / 8):.2f} of 8 channels")
