import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"), os.path.join(os.path.dirname(__file__), "..")]
import torch
from lpdnet_hip import ops
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import test_ops_gpu as T
dev = torch.device("cuda:0")
B, N, k, C = 2, 320, 20, 128
P, Q, idx, _, _ = T._edge_inputs(B, N, C, k, 900 + N + k)
P, Q, idx = P.to(dev), Q.to(dev), idx.to(dev)
g = torch.Generator().manual_seed(N)
W2 = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev)
bn1, bn2a, bn2b = T._bn_for(C, 3).to(dev).train(), T._bn_for(C, 4).to(dev).train(), T._bn_for(C, 4).to(dev).train()
_, _, _, st1 = ops.edge_split_fwd(P, Q, idx, N, bn=bn1)
Y1, Z32, zs1, a1, _ = ops.edge_mlp_train(P, Q, idx, N, st1.scale, st1.shift, W2, bn2a, 2, 0.01, False, z_bf16=False)
Y2, Z16, zs2, a2, _ = ops.edge_mlp_train(P, Q, idx, N, st1.scale, st1.shift, W2, bn2b, 2, 0.01, False, z_bf16=True)
ref = Z32.to(torch.bfloat16)
ne = (Z16 != ref)
print("Y equal", torch.equal(Y1, Y2), "zsel equal", torch.equal(zs1, zs2), "mismatch", int(ne.sum()), "of", ne.numel())
if ne.any():
    ii = ne.nonzero()[:10]
    for r, c in ii.tolist():
        print(r, c, Z32[r, c].item(), Z16[r, c].float().item(), ref[r, c].float().item())
    print("cols of mismatches mod 2:", (ne.nonzero()[:, 1] % 2).float().mean().item(), "rows mod k", (ne.nonzero()[:, 0] % k).float().mean().item())
    d = (Z16.float() - Z32).abs() / Z32.abs().clamp_min(1e-20)
    print("max rel", d.max().item())
