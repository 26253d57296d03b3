"""CPU tests: pin the oracle (oracle/) to the golden vectors captured from the reference
(tests/golden/make_golden.py).  These run without a GPU and never touch the product kernels."""
import os

import numpy as np
import pytest
import torch

from oracle import lpd_oracle as orc
from oracle import synth


def _load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, tag + ".npz"))


@pytest.mark.parametrize("tag", ["knn_c3_n4096_k20", "knn_c64_n4096_k20", "knn_c3_n16384_k64", "knn_c3_n100_k7"])
def test_oracle_knn_bit_exact_vs_reference(golden_dir, tag):
    g = _load(golden_dir, tag)
    C, N, k, stride = int(g["C"]), int(g["N"]), int(g["k"]), int(g["row_stride"])
    x_pm = synth.cloud(int(g["cloud_seed"]), 1, N, C)
    idx, pd = orc.knn_np(x_pm, k)
    ok = (idx[0][::stride] == g["idx"].astype(np.int32)).all(-1)
    assert (~ok & ~g["tie"]).sum() == 0          # bit-exact wherever torch's topk is well defined
    assert (np.diff(pd[0], axis=1) <= 0).all()   # sorted descending
    assert (idx[0][:, 0] == np.arange(N)).mean() > 0.95  # self is (almost always) the nearest


def test_oracle_knn_edge_cases():
    # duplicates: equal distances -> lower index first (documented tie rule)
    pts = np.zeros((1, 8, 3), np.float32)
    pts[0, 4:] = 1.0
    idx, _ = orc.knn_np(pts, 4)
    assert idx[0, 0].tolist() == [0, 1, 2, 3] and idx[0, 5].tolist() == [4, 5, 6, 7]
    assert orc.knn_tie_rows(pts, 4).all()
    # k == N returns a permutation of all points
    pts = synth.cloud(1, 1, 16)
    idx, _ = orc.knn_np(pts, 16)
    assert (np.sort(idx[0], axis=1) == np.arange(16)).all()


EVAL_CASES = [("eval_lpdnet_b2_n4096", "lpdnet", {}),
              ("eval_lpdnet_tnets_b2_n1024", "lpdnet", dict(feature_transform=True, xyz_trans=True)),
              ("eval_lpdnetorigin_b2_n1024", "lpdnetorigin", {}),
              ("eval_pointnet_b2_n4096", "pointnet", {}),
              ("eval_pointnet_ft_b2_n1024", "pointnet", dict(feature_transform=True))]


@pytest.mark.parametrize("tag,featnet,kw", EVAL_CASES)
def test_oracle_eval_forward_vs_reference(golden_dir, tag, featnet, kw):
    g = _load(golden_dir, tag)
    B, N = int(g["B"]), int(g["N"])
    sd = orc.synthetic_state(featnet, num_points=N, **kw)
    x = torch.from_numpy(synth.cloud(int(g["seed"]), B, N)).unsqueeze(1)
    aux = {}
    with torch.no_grad():
        desc = orc.pointnetvlad_forward(sd, x, featnet=featnet, train=False, aux=aux, **kw)
    ref = torch.from_numpy(g["desc"])
    rel = ((desc - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max().item()
    assert rel < 1e-5, rel
    if featnet != "pointnet":
        ok = (aux["idx_xyz"].numpy() == g["idx_xyz"].astype(np.int64)).all(-1)
        assert (~ok & ~g["tie_xyz"]).sum() == 0
        assert (aux["idx_feat"].numpy() == g["idx_feat"].astype(np.int64)).all(-1).mean() > 0.99


def test_c_restatement_of_the_whole_path_vs_reference(golden_dir):
    """oracle/lpd_forward.c (plain C, the reference's own formulation, the timed CPU baseline of bench.py) against the
    reference's golden descriptors: LPD-Net eval at N = 4096 / k = 20 and N = 2048 / k = 64."""
    for tag, k in (("eval_lpdnet_b2_n4096", 20), ("eval_lpdnet_k64_b2_n2048", 64)):
        g = _load(golden_dir, tag)
        B, N = int(g["B"]), int(g["N"])
        sd = orc.synthetic_state("lpdnet", num_points=N)
        x = torch.from_numpy(synth.cloud(int(g["seed"]), B, N)).unsqueeze(1)
        d, used = orc.forward_lpdnet_c(sd, x, k=k)
        assert used >= 1
        rel = (np.abs(d - g["desc"]).max(1) / np.abs(g["desc"]).max(1)).max()
        assert rel < 1e-5, (tag, rel)


def test_oracle_eval_k64_vs_reference(golden_dir):
    """The reference run with emb_nn.k = 64 (tests/golden/make_golden_r2.py): the stress configuration's neighbourhood size."""
    g = _load(golden_dir, "eval_lpdnet_k64_b2_n2048")
    B, N, k, st = int(g["B"]), int(g["N"]), int(g["k"]), int(g["row_stride"])
    sd = orc.synthetic_state("lpdnet", num_points=N)
    x = torch.from_numpy(synth.cloud(int(g["seed"]), B, N)).unsqueeze(1)
    aux = {}
    with torch.no_grad():
        desc = orc.pointnetvlad_forward(sd, x, featnet="lpdnet", train=False, k=k, aux=aux)
    ref = torch.from_numpy(g["desc"])
    assert ((desc - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max().item() < 1e-5
    ok = (aux["idx_xyz"].numpy()[:, ::st] == g["idx_xyz"].astype(np.int64)).all(-1)
    assert (~ok & ~g["tie_xyz"]).sum() == 0
    assert (aux["idx_feat"].numpy()[:, ::st] == g["idx_feat"].astype(np.int64)).all(-1).mean() > 0.99


def test_oracle_max_over_k_override_is_the_max_on_its_own_choices():
    """argsel (oracle._max_k) fed with the oracle's own arg-max reproduces the plain forward bit for bit."""
    N, B = 128, 2
    sd = orc.synthetic_state("lpdnet", num_points=N)
    x = torch.from_numpy(synth.cloud(3, B, N)).unsqueeze(1)
    with torch.no_grad():
        aux = {}
        f = orc.lpdnet_features(sd, x, aux=aux)
        calls = {}
        orig = orc._max_k

        def rec(e, name, argsel):
            calls[name] = e.argmax(dim=-1)
            return orig(e, name, argsel)
        orc._max_k = rec
        try:
            orc.lpdnet_features(sd, x)
        finally:
            orc._max_k = orig
        f2 = orc.lpdnet_features(sd, x, argsel=calls)
    assert set(calls) == {"x1", "x2", "x3"} and torch.equal(f, f2)


TRAIN_CASES = [("train_lpdnet_bq1_p2_n2_n1024", "lpdnet", {}), ("train_pointnet_bq1_p2_n2_n4096", "pointnet", {}),
               ("train_lpdnetorigin_bq1_p2_n2_n1024", "lpdnetorigin", {}),
               ("train_lpdnet_t3d_bq1_p2_n2_n1024", "lpdnet", dict(xyz_trans=True))]


@pytest.mark.parametrize("tag,featnet,kw", TRAIN_CASES)
def test_oracle_train_step0_vs_reference(golden_dir, tag, featnet, kw):
    g = _load(golden_dir, tag)
    bq, P, Ng, N = [int(v) for v in g["dims"]]
    B = bq * (1 + P + Ng + 1)
    sd0 = orc.synthetic_state(featnet, num_points=N, **kw)
    sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) else v.clone())
          for k, v in sd0.items()}
    x = torch.from_numpy(synth.cloud(int(g["seed"]), B, N)).unsqueeze(1)
    new_stats = {}
    desc = orc.pointnetvlad_forward(sd, x, featnet=featnet, train=True, new_stats=new_stats, **kw)
    q, p, n, o = torch.split(desc.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
    loss = orc.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
    loss.backward()
    ref = torch.from_numpy(g["desc"])
    assert ((desc.detach() - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max().item() < 1e-4
    assert abs(loss.item() - float(g["loss"])) < 5e-4 * max(1.0, abs(float(g["loss"])))
    checked = 0
    for key in g.files:
        if key.startswith("grad/"):
            name = key[5:]
            got, want = sd[name].grad.numpy(), g[key]
            err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
            assert err < (1e-2 if kw else 5e-3), (name, err)      # the T-Net case: 7e-3 between the two fp32 evaluations
            checked += 1
        elif key.startswith("gsum/"):
            name = key[5:]
            l2 = sd[name].grad.double().pow(2).sum().sqrt().item()
            assert abs(l2 - g[key][2]) < (1e-2 if kw else 5e-3) * max(g[key][2], 1e-6), (name, l2, g[key][2])
            checked += 1
        elif key.startswith("buf/"):
            name = key[4:]
            assert np.allclose(new_stats[name].numpy(), g[key], rtol=1e-4, atol=1e-5), name
    assert checked > 20


def test_oracle_losses_vs_reference(golden_dir):
    g = _load(golden_dir, "loss_kat.npz"[:-4])
    bq, P, Ng, D = [int(v) for v in g["dims"]]
    q = 0.5 * torch.from_numpy(synth.uniform("loss/q", bq * D).astype(np.float32).reshape(bq, 1, D))
    pos = 0.5 * torch.from_numpy(synth.uniform("loss/pos", bq * P * D).astype(np.float32).reshape(bq, P, D))
    neg = 0.5 * torch.from_numpy(synth.uniform("loss/neg", bq * Ng * D).astype(np.float32).reshape(bq, Ng, D))
    oth = 0.5 * torch.from_numpy(synth.uniform("loss/oth", bq * D).astype(np.float32).reshape(bq, 1, D))
    for row in g["table"]:
        use_min, lazy, ign = bool(row[0]), bool(row[1]), bool(row[2])
        assert abs(orc.quadruplet_loss(q, pos, neg, oth, 0.5, 0.2, use_min, lazy, ign).item() - row[3]) < 1e-5
        assert abs(orc.triplet_loss(q, pos, neg, 0.5, use_min, lazy, ign).item() - row[4]) < 1e-5
        assert abs(orc.triplet_loss_wrapper(q, pos, neg, oth, 0.5, 0.2, use_min, lazy, ign).item() - row[5]) < 1e-5
    mn, mx = orc.best_pos_distance(q, pos)
    assert np.allclose(mn.numpy(), g["min_pos"]) and np.allclose(mx.numpy(), g["max_pos"])
    assert np.allclose(g["hand"], [4.7, 2.5, 0.0], atol=1e-6)


def test_synth_is_deterministic():
    a = synth.tensor_for("emb_nn.conv1_lpd.weight", (64, 3, 1))
    b = synth.tensor_for("emb_nn.conv1_lpd.weight", (64, 3, 1))
    assert a.dtype == np.float32 and np.array_equal(a, b)
    # pinned values: change detection for the closed-form generator (fixtures depend on it)
    assert np.allclose(synth.uniform("a", 3), [-0.25653807, 0.05090679, 0.15482962], atol=1e-8)
    assert synth.tensor_for("x.running_var", (8,)).min() >= 0.5
