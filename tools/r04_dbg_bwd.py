import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(__file__), "..", "lpd-net-pytorch_amd"), os.path.join(os.path.dirname(__file__), ".."),
                os.path.join(os.path.dirname(__file__), "..", "tests")]
import torch
from lpdnet_hip import ops
import test_ops_gpu as T
dev = torch.device("cuda:0")
for bf16 in (False, True):
    B, N, k, C = 2, 320, 20, 128
    M = B * N
    P, Q, idx, _, _ = T._edge_inputs(B, N, C, k, 1900 + N + k)
    P, Q, idx = P.to(dev), Q.to(dev), idx.to(dev)
    g = torch.Generator().manual_seed(N + 1)
    W2 = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev)
    bn1, bn2 = T._bn_for(C, 3).to(dev).train(), T._bn_for(C, 4).to(dev).train()
    s1sum, usel, arg1, st1 = ops.edge_split_fwd(P, Q, idx, N, bn=bn1)
    Y, Z, zsel, arg2, st2 = ops.edge_mlp_train(P, Q, idx, N, st1.scale, st1.shift, W2, bn2, 2, 0.01, bf16, z_bf16=bf16)
    dcat = torch.randn(M, 512, generator=g).to(dev)
    dx1, dx2 = dcat[:, 0:128], dcat[:, 128:256]
    dt = torch.bfloat16 if bf16 else torch.float32
    dpre2, red2 = ops.bn_sel_bwd_reduce(dx2, zsel, st2, 2, 0.01, dtype=dt)
    graph = ops.GraphT(idx, N)
    dY = (ops.gemm_bf16s_bnbwd if bf16 else ops.gemm_f32s_bnbwd)(Z, arg2, dpre2, k, W2, st2, red2)
    dq_a, dp_a = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev)
    fn = ops.edge_bn_bwd_bf16 if bf16 else ops.edge_bn_bwd
    dU, dg_a, db_a = fn(dx1, arg1, k, Y, st1, 2, 0.01, dense=dY.clone(), dQ=dq_a, post_bn=bn1)
    (ops.gather_sum_rows_bf16 if bf16 else ops.gather_sum_rows)(dU, graph, dp_a)
    G, gsum, red1 = ops.edge_mlp_train_bwd(Z, arg2, dpre2, W2, st2, red2, Y, arg1, dx1, bn1, k, 2, 0.01)
    dP, dQ = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev)
    ops.edge_dense_bwd_apply(G, gsum, s1sum, P, Q, graph, st1, red1, k, dP=dP, dQ=dQ)
    rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()
    print("bf16" if bf16 else "f32", "dP", rel(dP, dp_a), "dQ", rel(dQ, dq_a), "gsum vs G", rel(gsum, G.float().view(M, k, C).sum(1)))
    print("  col means: dQ new", dQ.mean(0)[:4].tolist(), "chain", dq_a.mean(0)[:4].tolist(), "dU-sum", dU.float().view(M, k, C).sum(1).mean(0)[:4].tolist())
    print("  col means: dP new", dP.mean(0)[:4].tolist(), "chain", dp_a.mean(0)[:4].tolist())
    Sref = P[(idx.long() + (torch.arange(B, device=dev) * N).view(B, 1, 1)).view(M, k)].sum(1)
    print("  S vs gather", rel(s1sum, Sref), "dq chain vs dU-sum", rel(dq_a, dU.float().view(M, k, C).sum(1)))
    # closed form in torch from G
    E = M * k
    m1 = red1[0] / E; m2 = red1[1] / E
    sc, mu, isd = st1.scale.double(), st1.mean.double(), st1.invstd.double()
    dq_t = sc * (G.double().view(M, k, C).sum(1) - k * m1 - m2 * isd * (Sref.double() + k * (Q.double() - mu)))
    print("  dQ new vs torch closed form", rel(dQ, dq_t), " chain vs torch closed form", rel(dq_a, dq_t))
