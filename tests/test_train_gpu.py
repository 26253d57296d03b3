"""Training-path parity (forward in train mode, lazy quadruplet loss, backward) of the HIP path against the
reference golden vectors (train step 0) and the oracle's autograd.  -m gpu only.

Tolerances: descriptors 1e-4 norm-relative; loss 5e-4 relative; gradients 1e-2 relative L2 per tensor with a
median over tensors below 2.5e-3.  Why not tighter: the backward of `max over k` routes each gradient entry
to ONE arg-max edge, and among the ~3 M maxima of a step a handful are decided by the last fp32 bit, so any
two fp32 evaluations (reference vs oracle: up to 9e-4; fp32 oracle vs fp64 oracle: up to 7e-3 on these
inputs; MI355X vs fp64 oracle: 1e-4..5e-3; measured in round 1) differ by O(1e-3) in relative L2 of the
trunk gradients, while everything before the first max (NetVLAD head, loss) agrees to 1e-5.  The oracle
comparison therefore runs the oracle in fp64 on the kNN graphs the GPU produced (the kNN op has its own
bit-exact test)."""
import os

import numpy as np
import pytest
import torch

from oracle import lpd_oracle as orc
from oracle import synth

pytestmark = pytest.mark.gpu
GRAD_TOL = 1e-2
GRAD_MEDIAN_TOL = 2.5e-3


# gradients vs the fp64 oracle evaluated on the GPU's own kNN graphs AND arg-max choices (no flips left).  What remains is the
# conditioning of the step itself: the head's BatchNorms run over B = 6..16 descriptor rows and amplify the fp32 rounding of
# the forward ~100x into dOut, which then scales EVERY trunk gradient by a common factor (measured: all trunk tensors of one
# case sit at the same error -- 9e-4 at B = 6 / N = 512, 4.5e-4 at B = 16 / N = 1024, 2.2e-4 for lpdnetorigin; worst single
# tensor 2.1e-3, a BatchNorm bias whose sum cancels).  3e-3 per tensor with a 1e-3 median is 3-10x below a 1 % error.
FLIP_FREE_TOL = 3e-3
FLIP_FREE_MEDIAN = 1e-3


def _train_model(N, cuda, featnet="lpdnet", **variant):
    from util.PointNetVlad import PointNetVlad
    m = PointNetVlad(num_points=N, featnet=featnet, **variant)
    sd = orc.synthetic_state(featnet, num_points=N, **variant)
    m.load_state_dict(sd, strict=True)
    return m.to(cuda).train(), sd


def _step(m, x, bq, P, Ng):
    import loss.pointnetvlad_loss as L
    out = m(x)
    q, p, n, o = torch.split(out.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
    loss = L.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
    loss.backward()
    return out, loss


@pytest.mark.parametrize("tag,featnet,kw", [("train_lpdnet_bq1_p2_n2_n1024", "lpdnet", {}),
                                            ("train_pointnet_bq1_p2_n2_n4096", "pointnet", {}),
                                            ("train_lpdnetorigin_bq1_p2_n2_n1024", "lpdnetorigin", {}),
                                            ("train_lpdnet_t3d_bq1_p2_n2_n1024", "lpdnet", dict(xyz_trans=True))],
                         ids=["lpdnet", "pointnet-forward+FD-pinned", "lpdnetorigin", "lpdnet+t3d"])
def test_train_step0_vs_reference_golden(cuda, golden_dir, tag, featnet, kw):
    """Step 0 of the reference's training loop.  lpdnet, lpdnetorigin and lpdnet + coordinate T-Net: every gradient in the
    fixture is the reference's own autograd.  PointNet trunk ("pinned on forward + FD"): the fixture's point_net.* gradients
    are the oracle's, validated by fp64 finite differences of the reference forward (the reference's torch-CPU BatchNorm2d
    backward is inconsistent there, tests/golden/make_golden.py); test_pointnet_gradients_vs_finite_differences_of_the_
    reference compares the GPU with those finite differences directly."""
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    bq, P, Ng, N = [int(v) for v in g["dims"]]
    B = bq * (1 + P + Ng + 1)
    m, _ = _train_model(N, cuda, featnet, **kw)
    x = torch.from_numpy(synth.cloud(int(g["seed"]), B, N)).unsqueeze(1).to(cuda)
    out, loss = _step(m, x, bq, P, Ng)
    ref = torch.from_numpy(g["desc"])
    rel = ((out.detach().cpu() - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max().item()
    assert rel < 1e-4, rel
    assert abs(loss.item() - float(g["loss"])) < 5e-4 * abs(float(g["loss"]))
    params = dict(m.named_parameters())
    wnorm = {n: (p.grad.norm().item() if p.grad is not None else 0.0) for n, p in params.items()}
    worst = 0.0
    for key in g.files:
        if key.startswith("grad/"):
            name = key[5:]
            got, want = params[name].grad.cpu().numpy(), g[key]
            err = np.linalg.norm(got - want) / np.linalg.norm(want)
            worst = max(worst, err)
            assert err < GRAD_TOL, (name, err)
        elif key.startswith("gprobe/"):
            name = key[7:]
            got = params[name].grad.detach().cpu().reshape(-1)[torch.from_numpy(g["gpos/" + name])].numpy()
            scale = g["gsum/" + name][2] / np.sqrt(params[name].numel())      # rms of the tensor
            assert np.abs(got - g[key]).max() < 0.05 * scale + GRAD_TOL * np.abs(g[key]).max(), name
        elif key.startswith("gsum/"):
            name = key[5:]
            l2 = params[name].grad.double().pow(2).sum().sqrt().item()
            assert abs(l2 - g[key][2]) < GRAD_TOL * g[key][2], (name, l2, g[key][2])
        elif key.startswith("buf/"):
            name = key[4:]
            buf = dict(m.named_buffers())[name].cpu().numpy()
            assert np.allclose(buf, g[key], rtol=2e-4, atol=2e-5), name
        elif key.startswith("nograd/"):       # parameters the forward never touches (unused feature_trans)
            assert params[key[7:]].grad is None, key
        elif key.startswith("zerograd/"):     # bias in front of a BatchNorm: analytically zero, rounding noise only
            name = key[9:]
            assert params[name].grad.norm().item() < 1e-3 * max(1.0, wnorm.get(name[:-5] + ".weight", 0.0)), key
    for name, b in m.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(b) == 1


VARIANTS = [("lpdnet", 1, 2, 2, 256, {}), ("lpdnet", 2, 1, 3, 512, {}), ("lpdnetorigin", 1, 2, 2, 256, {}),
            ("lpdnet", 1, 2, 2, 256, dict(xyz_trans=True)), ("lpdnet", 1, 2, 2, 256, dict(xyz_trans=True, feature_transform=True)),
            ("lpdnetorigin", 1, 2, 2, 256, dict(xyz_trans=True, feature_transform=True)),
            ("pointnet", 1, 2, 2, 256, {}), ("pointnet", 2, 1, 3, 512, dict(feature_transform=True))]


@pytest.mark.parametrize("featnet,bq,P,Ng,N,variant", VARIANTS, ids=lambda v: "+".join(v) if isinstance(v, dict) else str(v))
def test_train_grads_vs_oracle(cuda, featnet, bq, P, Ng, N, variant):
    from lpdnet_hip import engine
    B = bq * (1 + P + Ng + 1)
    m, sd0 = _train_model(N, cuda, featnet, **variant)
    xc = torch.from_numpy(synth.cloud(21, B, N)).unsqueeze(1)
    engine.DEBUG_AUX = {}
    engine.MORTON_ORDER = False   # compare index tensors in the caller's point order
    try:
        out, loss = _step(m, xc.to(cuda), bq, P, Ng)
        aux = engine.DEBUG_AUX
    finally:
        engine.DEBUG_AUX = None
        engine.MORTON_ORDER = True
    graphs = iter([aux["idx_feat"].cpu().long(), aux["idx_xyz"].cpu().long()] if featnet != "pointnet" else [])
    dt = torch.float64
    sd = {k: (v.to(dt).requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var"))
              else (v.to(dt) if v.dtype == torch.float32 else v.clone())) for k, v in sd0.items()}
    new_stats = {}
    orig = orc.knn
    orc.knn = lambda xx, k: next(graphs)          # the GPU's graphs (kNN parity is tested bit-exactly elsewhere)
    try:
        od = orc.pointnetvlad_forward(sd, xc.to(dt), featnet=featnet, train=True, new_stats=new_stats, **variant)
    finally:
        orc.knn = orig
    q, p, n, o = torch.split(od.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
    ol = orc.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
    assert ol.item() > 0, "vacuous fixture: hinge inactive"
    ol.backward()
    unused = {n for n, v in sd.items() if v.is_floating_point() and v.requires_grad and v.grad is None}
    assert {n for n, p in m.named_parameters() if p.grad is None} == unused, "a parameter the oracle reaches has no HIP gradient"
    rel = ((out.detach().cpu().double() - od.detach()).abs().amax(dim=1) / od.detach().abs().amax(dim=1)).max().item()
    # Both T-Nets chained in train mode (BatchNorm over B = 6 rows inside each) is ill-conditioned: the fp32 oracle itself
    # sits 0.4e-4..1.2e-4 from the fp64 oracle there, the MI355X path 0.4e-4..1.2e-4 (measured in round 1), so the
    # comparison against fp64 gets 3e-4 for that variant; every other variant keeps the 1e-4 bar.
    desc_tol = 3e-4 if len(variant) == 2 else 1e-4
    assert rel < desc_tol, rel
    assert abs(loss.item() - ol.item()) < (5e-4 if len(variant) < 2 else 2e-3) * abs(ol.item())
    errs = {}
    for name, prm in m.named_parameters():
        want = sd[name].grad
        if want is None:
            continue
        if want.norm().item() < 1e-6 * max(1.0, sd[name].detach().norm().item()):   # bias in front of a BatchNorm: analytically 0
            wgrad = sd[name.replace(".bias", ".weight")].grad.norm().item()      # rounding noise scales with the layer's gradient
            assert prm.grad.norm().item() < 1e-3 * max(1.0, wgrad), name
            continue
        errs[name] = ((prm.grad.cpu().double() - want).norm() / want.norm()).item()
        assert errs[name] < GRAD_TOL, (name, errs[name])
    assert float(np.median(list(errs.values()))) < (GRAD_MEDIAN_TOL if len(variant) < 2 else 2 * GRAD_MEDIAN_TOL), errs
    # everything before the first max-over-k (head) is tight: no arg-max flips, only rounding (the backward products
    # run as split-bf16, ~5e-6 each, amplified by the B-row BatchNorms of the head: measured <= 2.7e-4)
    for name, e in errs.items():
        if name.startswith("net_vlad."):
            assert e < 5e-4, (name, e)
    for name, b in m.named_buffers():
        if name.endswith(("running_mean", "running_var")):
            assert torch.allclose(b.cpu().double(), new_stats[name], rtol=2e-4, atol=2e-5), name


def test_pointnet_gradients_vs_finite_differences_of_the_reference(cuda, golden_dir):
    """point_net.* gradients are the tensors the reference's CPU autograd cannot pin (BatchNorm2d backward bug): the fixture
    holds fp64 central differences of the REFERENCE's own forward + loss at 21 entries of 7 such tensors
    (tests/golden/make_golden_r2.py); the GPU gradient is compared with them directly, 2e-3 relative."""
    g = np.load(os.path.join(golden_dir, "train_pointnet_fd_probes.npz"))
    bq, P, Ng, N = [int(v) for v in g["dims"]]
    m, _ = _train_model(N, cuda, "pointnet")
    x = torch.from_numpy(synth.cloud(int(g["seed"]), bq * (P + Ng + 2), N)).unsqueeze(1).to(cuda)
    _step(m, x, bq, P, Ng)
    params = dict(m.named_parameters())
    n = 0
    for key in g.files:
        if key.startswith("fd/"):
            name = key[3:]
            got = params[name].grad.detach().cpu().reshape(-1)[torch.from_numpy(g["fdpos/" + name])].double().numpy()
            want = g[key]
            assert np.all(np.abs(got - want) <= 2e-3 * np.maximum(np.abs(want), 1.0)), (name, got, want)
            n += len(want)
    assert n >= 21


@pytest.mark.parametrize("featnet,bq,P,Ng,N", [("lpdnet", 1, 2, 2, 512), ("lpdnetorigin", 1, 2, 2, 512), ("lpdnet", 2, 2, 4, 1024)])
def test_train_grads_flip_free_vs_oracle(cuda, featnet, bq, P, Ng, N):
    """The 1e-2 gradient gate above has to absorb arg-max flips behind the max over k.  Here the fp64 oracle is evaluated on
    the GPU's kNN graphs AND on the GPU's arg-max choices (oracle argsel), so no flip is left and EVERY tensor -- trunk
    included -- must agree to FLIP_FREE_TOL relative L2: a 1 % bug in a trunk weight gradient cannot hide here."""
    from lpdnet_hip import engine
    B = bq * (1 + P + Ng + 1)
    m, sd0 = _train_model(N, cuda, featnet)
    xc = torch.from_numpy(synth.cloud(21, B, N)).unsqueeze(1)
    engine.DEBUG_AUX = {}
    engine.MORTON_ORDER = False
    try:
        out, loss = _step(m, xc.to(cuda), bq, P, Ng)
        aux = engine.DEBUG_AUX
    finally:
        engine.DEBUG_AUX = None
        engine.MORTON_ORDER = True
    k = m.emb_nn.k
    argsel = {n: a.view(B, N, -1).permute(0, 2, 1).contiguous().cpu().long() for n, a in aux["argsel"].items()}   # [B,C,N]
    assert all(int(a.max()) < k for a in argsel.values())
    graphs = iter([aux["idx_feat"].cpu().long(), aux["idx_xyz"].cpu().long()])
    dt = torch.float64
    sd = {kk: (v.to(dt).requires_grad_(True) if v.dtype == torch.float32 and not kk.endswith(("running_mean", "running_var"))
               else (v.to(dt) if v.dtype == torch.float32 else v.clone())) for kk, v in sd0.items()}
    orig = orc.knn
    orc.knn = lambda xx, kk: next(graphs)
    try:
        od = orc.pointnetvlad_forward(sd, xc.to(dt), featnet=featnet, train=True, argsel=argsel)
    finally:
        orc.knn = orig
    q, p, n, o = torch.split(od.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
    ol = orc.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
    assert ol.item() > 0
    ol.backward()
    rel = ((out.detach().cpu().double() - od.detach()).abs().amax(dim=1) / od.detach().abs().amax(dim=1)).max().item()
    assert rel < 1e-4, rel
    errs = {}
    for name, prm in m.named_parameters():
        want = sd[name].grad
        if want is None or want.norm().item() < 1e-6 * max(1.0, sd[name].detach().norm().item()):
            continue
        errs[name] = ((prm.grad.cpu().double() - want).norm() / want.norm()).item()
    assert len(errs) >= 18
    if os.environ.get("LPD_TEST_VERBOSE"):
        print(featnet, N, sorted(((round(e, 7), n) for n, e in errs.items()), reverse=True)[:12])
    bad = {n: e for n, e in errs.items() if e >= FLIP_FREE_TOL}
    assert not bad, bad
    assert float(np.median(list(errs.values()))) < FLIP_FREE_MEDIAN, errs


def _mem_available_gib():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


CFG2_FLIP_FREE_HOST_GIB = 150      # the fp64 oracle with autograd at 44 x 4096 points keeps ~110 GiB of [B, C, N, k] tensors alive


def test_train_cfg2_full_size_flip_free_vs_fp64_oracle(cuda, golden_dir):
    """BASELINE configs[2] at its stated size (bq = 2, P = 2, Ng = 18 -> 44 clouds x 4096 points, the fixture's clouds), fp32 storage,
    WITHOUT the two sources of legitimate disagreement: the fp64 oracle runs on the GPU's own kNN graphs (bit-exactness of the kNN is
    tested separately) and on the GPU's own arg-max choices behind every max over k.  What is left is rounding, so north_star's bar
    applies undiluted: every descriptor within 1e-4 (norm-relative) and every gradient tensor within FLIP_FREE_TOL = 3e-3 relative L2
    (median 1e-3).  test_train_step0_cfg2_full_size_vs_reference gates the same step against the REFERENCE's runs at 1e-3 because two
    evaluations of it differ through feature-space kNN near-ties; this test is the evidence that near-ties and arg-max flips are the
    whole of that excess (train_pointnetvlad.py:202-217)."""
    from lpdnet_hip import engine
    if _mem_available_gib() < CFG2_FLIP_FREE_HOST_GIB:
        pytest.skip(f"host has {_mem_available_gib():.0f} GiB available; the fp64 autograd oracle at 44 x 4096 needs ~{CFG2_FLIP_FREE_HOST_GIB}")
    g = np.load(os.path.join(golden_dir, "train_lpdnet_bq2_p2_n18_n4096.npz"))
    bq, P, Ng, N = [int(v) for v in g["dims"]]
    B = bq * (1 + P + Ng + 1)
    assert (B, N) == (44, 4096)
    m, sd0 = _train_model(N, cuda, "lpdnet")
    xc = torch.from_numpy(synth.scene_cloud(int(g["seed"]), B, N)).unsqueeze(1)
    engine.DEBUG_AUX = {}
    engine.MORTON_ORDER = False
    try:
        out, loss = _step(m, xc.to(cuda), bq, P, Ng)
        aux = engine.DEBUG_AUX
    finally:
        engine.DEBUG_AUX = None
        engine.MORTON_ORDER = True
    k = m.emb_nn.k
    argsel = {n: a.view(B, N, -1).permute(0, 2, 1).contiguous().cpu().long() for n, a in aux["argsel"].items()}   # [B,C,N]
    assert all(int(a.max()) < k for a in argsel.values())
    graphs = iter([aux["idx_feat"].cpu().long(), aux["idx_xyz"].cpu().long()])
    got = {name: prm.grad.detach().cpu().double() for name, prm in m.named_parameters() if prm.grad is not None}
    outc, lossv = out.detach().cpu().double(), loss.item()
    del aux, m, out, loss
    torch.cuda.empty_cache()
    dt = torch.float64
    sd = {kk: (v.to(dt).requires_grad_(True) if v.dtype == torch.float32 and not kk.endswith(("running_mean", "running_var"))
               else (v.to(dt) if v.dtype == torch.float32 else v.clone())) for kk, v in sd0.items()}
    orig = orc.knn
    orc.knn = lambda xx, kk: next(graphs)
    try:
        od = orc.pointnetvlad_forward(sd, xc.to(dt), featnet="lpdnet", train=True, argsel=argsel)
    finally:
        orc.knn = orig
    q, p, n, o = torch.split(od.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
    ol = orc.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
    assert ol.item() > 0
    ol.backward()
    rel = ((outc - od.detach()).abs().amax(dim=1) / od.detach().abs().amax(dim=1))
    errs = {}
    for name, have in got.items():
        want = sd[name].grad
        if want is None or want.norm().item() < 1e-6 * max(1.0, sd[name].detach().norm().item()):
            continue
        errs[name] = ((have - want).norm() / want.norm()).item()
    _cfg2_report("flip-free f32", desc_max=rel.max().item(), desc_median=rel.median().item(), loss=lossv, loss64=ol.item(),
                 grad_max=max(errs.values()), grad_median=float(np.median(list(errs.values()))),
                 worst=sorted(((round(e, 6), n_) for n_, e in errs.items()), reverse=True)[:4])
    assert rel.max().item() < 1e-4, rel.max().item()
    assert abs(lossv - ol.item()) < 5e-4 * abs(ol.item())
    assert len(errs) >= 18
    bad = {n_: e for n_, e in errs.items() if e >= FLIP_FREE_TOL}
    assert not bad, bad
    assert float(np.median(list(errs.values()))) < FLIP_FREE_MEDIAN, errs


# bf16 storage of the DG-chain edge tensors: every stored value carries a 2^-9 relative rounding; BatchNorm turns that into
# 2^-9 |z| / sigma per normalised value and the head's BatchNorms over B = 6..16 descriptor rows amplify it once more (~100x,
# as for the fp32 rounding, which lands at 1e-4 there).  Measured on these fixtures: descriptors 1.3e-2 / 1.8e-2 norm-relative,
# loss 1.5 %, gradients (oracle on the GPU's own arg-max choices) 7-12 % relative L2 per tensor, median 6 %.
BF16_DESC_TOL = 3e-2
BF16_LOSS_TOL = 5e-2
BF16_GRAD_TOL = 0.2
BF16_GRAD_MEDIAN = 0.1


@pytest.mark.parametrize("bq,P,Ng,N", [(1, 2, 2, 512), (2, 2, 4, 1024)])
def test_train_bf16_storage_vs_oracle(cuda, bq, P, Ng, N):
    """BASELINE configs[2] as stated (bf16): autograd.set_train_storage("bf16") keeps the DG1 -> DG2 edge tensors and their
    gradients in bf16 and runs the products on them on the bf16 MFMA; statistics, reductions, the split-form SN1 stage and the
    kNN stay fp32 / fp64.  Compared with the fp64 oracle on the GPU's kNN graphs and arg-max choices at the stated (looser) bf16
    tolerances; the feature-space graph itself must be the one the fp32 mode builds (bit-identical indices)."""
    from lpdnet_hip import autograd, engine
    B = bq * (1 + P + Ng + 1)
    xc = torch.from_numpy(synth.cloud(21, B, N)).unsqueeze(1)
    runs = {}
    for storage in ("f32", "bf16"):
        m, sd0 = _train_model(N, cuda, "lpdnet")
        prev = autograd.set_train_storage(storage)
        engine.DEBUG_AUX = {}
        engine.MORTON_ORDER = False
        try:
            out, loss = _step(m, xc.to(cuda), bq, P, Ng)
            runs[storage] = (m, out, loss, engine.DEBUG_AUX)
        finally:
            engine.DEBUG_AUX = None
            engine.MORTON_ORDER = True
            autograd.set_train_storage(prev)
    m, out, loss, aux = runs["bf16"]
    assert torch.equal(aux["idx_feat"], runs["f32"][3]["idx_feat"]) and torch.equal(aux["idx_xyz"], runs["f32"][3]["idx_xyz"])
    graphs = iter([aux["idx_feat"].cpu().long(), aux["idx_xyz"].cpu().long()])
    dt = torch.float64
    sd = {kk: (v.to(dt).requires_grad_(True) if v.dtype == torch.float32 and not kk.endswith(("running_mean", "running_var"))
               else (v.to(dt) if v.dtype == torch.float32 else v.clone())) for kk, v in sd0.items()}
    k = m.emb_nn.k
    argsel = {n_: a.view(B, N, -1).permute(0, 2, 1).contiguous().cpu().long() for n_, a in aux["argsel"].items()}
    orig = orc.knn
    orc.knn = lambda xx, kk: next(graphs)
    try:
        od = orc.pointnetvlad_forward(sd, xc.to(dt), featnet="lpdnet", train=True, argsel=argsel)
    finally:
        orc.knn = orig
    q, p, n, o = torch.split(od.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
    ol = orc.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
    ol.backward()
    rel = ((out.detach().cpu().double() - od.detach()).abs().amax(dim=1) / od.detach().abs().amax(dim=1)).max().item()
    assert rel < BF16_DESC_TOL, rel
    assert abs(loss.item() - ol.item()) < BF16_LOSS_TOL * abs(ol.item())
    errs = {}
    for name, prm in m.named_parameters():
        want = sd[name].grad
        if want is None or want.norm().item() < 1e-6 * max(1.0, sd[name].detach().norm().item()):
            continue
        errs[name] = ((prm.grad.cpu().double() - want).norm() / want.norm()).item()
    if os.environ.get("LPD_TEST_VERBOSE"):
        print("bf16", N, "desc", rel, sorted(((round(e, 5), n_) for n_, e in errs.items()), reverse=True)[:8])
    assert max(errs.values()) < BF16_GRAD_TOL and float(np.median(list(errs.values()))) < BF16_GRAD_MEDIAN, errs
    # and the two storage modes agree with each other at the bf16 tolerance
    d32 = runs["f32"][1].detach()
    assert ((out.detach() - d32).abs().amax(dim=1) / d32.abs().amax(dim=1)).max().item() < BF16_DESC_TOL


def test_train_then_eval_roundtrip_and_adam_step(cuda):
    """model.train() step + optimizer.step(), then model.eval() forward uses the updated running statistics."""
    N, bq, P, Ng = 256, 1, 2, 2
    m, _ = _train_model(N, cuda)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    x = torch.from_numpy(synth.cloud(33, bq * (P + Ng + 2), N)).unsqueeze(1).to(cuda)
    before = m.emb_nn.bn3_lpd.running_mean.clone()
    opt.zero_grad()
    _, loss = _step(m, x, bq, P, Ng)
    opt.step()
    assert torch.isfinite(loss)
    assert not torch.equal(before, m.emb_nn.bn3_lpd.running_mean)
    m.eval()
    with torch.no_grad():
        d = m(x)
    assert torch.isfinite(d).all() and d.shape == (bq * (P + Ng + 2), 256)


@pytest.mark.parametrize("gating,add_bn", [(True, False), (False, True), (False, False)])
def test_netvlad_constructor_variants_train_and_eval(cuda, gating, add_bn):
    """NetVLADLoupe(gating=..., add_batch_norm=...) (PointNetVlad.py:33-36,55-56,80-81,94-96,108-109): the variants
    PointNetVlad itself never constructs -- train-mode forward + backward and eval forward against the oracle."""
    from util.PointNetVlad import NetVLADLoupe
    E, N, K, O, B = 128, 256, 64, 32, 6      # cluster_size 64: what lpd_vlad_finalize is built for
    g = torch.Generator().manual_seed(5)
    head = NetVLADLoupe(feature_size=E, max_samples=N, cluster_size=K, output_dim=O, gating=gating, add_batch_norm=add_bn)
    sd = {"net_vlad." + k: v.detach().clone() for k, v in head.state_dict().items()}
    x = torch.randn(B, E, N, 1, generator=g)
    for train in (True, False):
        head = head.to(cuda).train(train)
        dt = torch.float64
        if not train:      # the train-mode pass moved the running statistics: the eval oracle takes the module's current state
            sd = {"net_vlad." + k: v.detach().cpu().clone() for k, v in head.state_dict().items()}
        osd = {k: (v.to(dt).requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v.clone())
               for k, v in sd.items()}
        xo = x.to(dt).requires_grad_(True)
        want = orc.netvlad(osd, xo, train=train)
        xg = x.to(cuda).requires_grad_(True)
        got = head(xg)
        assert got.shape == (B, O)
        rel = ((got.detach().cpu().double() - want.detach()).abs().amax(dim=1) / want.detach().abs().amax(dim=1)).max().item()
        assert rel < 1e-4, (train, rel)
        if train:
            w = torch.randn(B, O, generator=g)
            (got * w.to(cuda)).sum().backward()
            (want * w.to(dt)).sum().backward()
            assert ((xg.grad.cpu().double() - xo.grad).norm() / xo.grad.norm()).item() < 5e-4
            n = 0
            for name, prm in head.named_parameters():
                ref = osd["net_vlad." + name].grad
                err = ((prm.grad.cpu().double() - ref).norm() / ref.norm().clamp_min(1e-30)).item()
                assert err < 1e-3, (name, err)
                n += 1
            assert n == 3 + (2 if add_bn else 1) + 2 + ((1 + (2 if add_bn else 1)) if gating else 0)


def test_pointnetfeat_max_pool_and_best_pos_distance_autograd(cuda):
    """PointNetfeat(max_pool=True) (PointNetVlad.py:234-239: global max over the points + the alignment matrix), eval and
    train with gradients through the max; best_pos_distance carries gradients like the reference's (pointnetvlad_loss.py:6-12)."""
    import loss.pointnetvlad_loss as L
    from util.PointNetVlad import PointNetfeat
    N, B = 256, 4
    net = PointNetfeat(num_points=N, max_pool=True, emb_dims=1024)
    full = orc.synthetic_state("pointnet", num_points=N)
    net.load_state_dict({k[len("point_net."):]: v for k, v in full.items() if k.startswith("point_net.")}, strict=True)
    net = net.to(cuda)
    xc = torch.from_numpy(synth.cloud(8, B, N)).unsqueeze(1)
    for train in (False, True):
        net.train(train)
        dt = torch.float64
        osd = {k: (v.to(dt).requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var"))
                   else (v.to(dt) if v.dtype == torch.float32 else v.clone())) for k, v in full.items()}
        want = orc.pointnet_features(osd, xc.to(dt), train=train).squeeze(-1).max(dim=2)[0]          # [B, E]
        got, trans = net(xc.to(cuda))
        assert got.shape == (B, 1024) and trans.shape == (B, 3, 3)
        assert _desc_rel(got, want) < 1e-4
        if train:
            g = torch.Generator().manual_seed(1)
            w = torch.randn(B, 1024, generator=g)
            (got * w.to(cuda)).sum().backward()
            (want * w.to(dt)).sum().backward()
            for name in ("conv5.weight", "conv1.weight", "bn3.weight", "stn.fc1.weight"):
                a, b = dict(net.named_parameters())[name].grad.cpu().double(), osd["point_net." + name].grad
                assert ((a - b).norm() / b.norm()).item() < 2e-3, name
    with pytest.raises(NotImplementedError):
        PointNetfeat(num_points=N, max_pool=True, global_feat=False).to(cuda).eval()(xc.to(cuda))
    # best_pos_distance with autograd
    g = torch.Generator().manual_seed(3)
    q = torch.randn(3, 1, 256, generator=g)
    p = torch.randn(3, 4, 256, generator=g)
    qg, pg = q.to(cuda).requires_grad_(True), p.to(cuda).requires_grad_(True)
    qo, po = q.double().requires_grad_(True), p.double().requires_grad_(True)
    mn, mx = L.best_pos_distance(qg, pg)
    omn, omx = orc.best_pos_distance(qo, po)
    assert torch.allclose(mn.detach().cpu().double(), omn.detach(), rtol=1e-5) and torch.allclose(mx.detach().cpu().double(), omx.detach(), rtol=1e-5)
    (2.0 * mn.sum() - 0.5 * mx.sum()).backward()
    (2.0 * omn.sum() - 0.5 * omx.sum()).backward()
    assert torch.allclose(qg.grad.cpu().double(), qo.grad, rtol=1e-4, atol=1e-5) and torch.allclose(pg.grad.cpu().double(), po.grad, rtol=1e-4, atol=1e-5)


def _desc_rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().amax(dim=1) / b.abs().amax(dim=1)).max().item()


@pytest.mark.parametrize("cls,t3d", [("LPDNet", False), ("LPDNet", True), ("LPDNetOrign", True)])
def test_use_mfea_trunks_eval_and_train(cuda, cls, t3d):
    """LPDNet / LPDNetOrign(use_mFea=True) (lpdnet_model.py:26-30,183-186,215-224): [B,1,N,8] inputs = xyz + 5 handcrafted
    features; conv1 takes 8 columns, the coordinate T-Net and the second graph see the xyz part only.  PointNetVlad never
    enables it; the trunks are checked on their own against the oracle, eval and train (with gradients)."""
    from util import lpdnet_model as lm
    featnet = "lpdnet" if cls == "LPDNet" else "lpdnetorigin"
    N, B = 256, 4
    net = getattr(lm, cls)(emb_dims=1024, use_mFea=True, t3d=t3d, tfea=False)
    full = orc.synthetic_state(featnet, num_points=N, xyz_trans=t3d, use_mFea=True)
    net.load_state_dict({k[len("emb_nn."):]: v for k, v in full.items() if k.startswith("emb_nn.")}, strict=True)
    net = net.to(cuda)
    g = torch.Generator().manual_seed(2)
    x8 = torch.cat((torch.from_numpy(synth.cloud(5, B, N)), torch.rand(B, N, 5, generator=g) - 0.5), dim=2).unsqueeze(1)     # [B,1,N,8]
    fn = orc.lpdnet_features if cls == "LPDNet" else orc.lpdnet_origin_features
    for train in (False, True):
        net.train(train)
        dt = torch.float64
        osd = {k: (v.to(dt).requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var"))
                   else (v.to(dt) if v.dtype == torch.float32 else v.clone())) for k, v in full.items()}
        got = net(x8.to(cuda))                                                    # [B,E,N,1]
        assert got.shape == (B, 1024, N, 1)
        with torch.enable_grad():
            want = fn(osd, x8.to(dt), train=train, t3d=t3d)
        err = (got.detach().cpu().double() - want.detach()).abs().max().item() / want.detach().abs().max().item()
        assert err < 2e-4, (train, err)      # per-point features (feature-space kNN flips move single entries)
        if train:
            w = torch.randn(B, 1024, N, 1, generator=g)
            (got * w.to(cuda)).sum().backward()
            (want * w.to(dt)).sum().backward()
            name = "conv1_lpd.weight" if cls == "LPDNet" else "conv1_lpd.0.weight"
            a, b = dict(net.named_parameters())[name].grad.cpu().double(), osd["emb_nn." + name].grad
            assert a.shape == b.shape and a.shape[1] == 8
            assert ((a - b).norm() / b.norm()).item() < 2e-2, name
    with pytest.raises(ValueError):
        net.eval()(x8[..., :3].to(cuda))


# ---- BASELINE configs[2] at its stated size: batch_num_queries 2, positives 2, negatives 18 -> B = 44 clouds of 4096 points ----
def _cfg2_report(tag, **kv):
    if os.environ.get("LPD_TEST_VERBOSE"):
        print("cfg2", tag, {k: (float(f"{v:.3e}") if isinstance(v, float) else v) for k, v in kv.items()})


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_train_step0_cfg2_full_size_vs_reference(cuda, golden_dir, storage):
    """Step 0 of the REFERENCE's training loop at the size configs[2] states (train_pointnetvlad.py:202-217, config.py:15-17;
    fixture: tests/golden/make_golden_r3.py, structured clouds `synth.scene_cloud`), in both storage modes of the HIP path.

    Yardstick for the descriptors: the fixture also holds the reference's OWN fp64 forward (`desc64`).  Two evaluations of this
    step differ through a handful of feature-space kNN near-ties and the head's BatchNorms over the 44 batch rows whatever the
    precision: the reference's fp32 run sits 3.3e-4 (max over the 44 descriptors, norm-relative) / 7.3e-5 (median) from its fp64
    run, the fp32 oracle 6.8e-4 / 5.1e-5.  The fp32-storage path must be as close to fp64 as those are (max <= 1e-3, median
    <= 1.5e-4), and -- the direct figure -- as close to the reference's fp32 descriptors as the reference's two precisions are to
    each other (same gates; printed as `vs_ref32_*`); gradients against the reference's fp32 autograd at CFG2_GATES (fp32 storage:
    5e-3 per tensor, median 2e-3; measured 1.7e-3 / 8e-4)."""
    from lpdnet_hip import autograd
    g = np.load(os.path.join(golden_dir, "train_lpdnet_bq2_p2_n18_n4096.npz"))
    bq, P, Ng, N = [int(v) for v in g["dims"]]
    B = bq * (1 + P + Ng + 1)
    assert (B, N) == (44, 4096)
    m, _ = _train_model(N, cuda, "lpdnet")
    x = torch.from_numpy(synth.scene_cloud(int(g["seed"]), B, N)).unsqueeze(1).to(cuda)
    prev = autograd.set_train_storage(storage)
    try:
        out, loss = _step(m, x, bq, P, Ng)
    finally:
        autograd.set_train_storage(prev)
    d64 = torch.from_numpy(g["desc64"])

    def nr(a):
        return ((a.double() - d64).abs().amax(dim=1) / d64.abs().amax(dim=1))
    e_ref, e_gpu = nr(torch.from_numpy(g["desc"])), nr(out.detach().cpu())
    d32 = torch.from_numpy(g["desc"]).double()             # the reference's own fp32 run: GPU fp32 vs reference fp32, directly
    e_32 = (out.detach().cpu().double() - d32).abs().amax(dim=1) / d32.abs().amax(dim=1)
    q, p, n, o = torch.split(d64.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
    loss64 = orc.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False).item()
    loss_err = abs(loss.item() - loss64) / abs(loss64)
    ref_loss_err = abs(float(g["loss"]) - loss64) / abs(loss64)
    params = dict(m.named_parameters())
    errs, l2errs = {}, {}
    for key in g.files:
        if key.startswith("grad/"):
            name = key[5:]
            errs[name] = float(np.linalg.norm(params[name].grad.cpu().numpy() - g[key]) / np.linalg.norm(g[key]))
        elif key.startswith("gsum/"):
            name = key[5:]
            l2errs[name] = abs(params[name].grad.double().pow(2).sum().sqrt().item() - g[key][2]) / g[key][2]
    perr = {}
    for key in g.files:
        if key.startswith("gprobe/"):
            name = key[7:]
            got = params[name].grad.detach().cpu().reshape(-1)[torch.from_numpy(g["gpos/" + name])].numpy()
            perr[name] = float(np.linalg.norm(got - g[key]) / max(np.linalg.norm(g[key]), 1e-30))
    berr = 0.0
    for key in g.files:
        if key.startswith("buf/"):
            buf = dict(m.named_buffers())[key[4:]].cpu().numpy()
            berr = max(berr, float(np.abs(buf - g[key]).max() / (np.abs(g[key]).max() + 1e-6)))
    _cfg2_report(storage, vs_ref32_max=e_32.max().item(), vs_ref32_median=e_32.median().item(),
                 desc_max=e_gpu.max().item(), desc_median=e_gpu.median().item(), ref_max=e_ref.max().item(),
                 ref_median=e_ref.median().item(), loss=loss.item(), loss64=loss64, loss_err=loss_err, ref_loss_err=ref_loss_err,
                 grad_max=max(errs.values()), grad_median=float(np.median(list(errs.values()))), l2_max=max(l2errs.values()),
                 probe_max=max(perr.values()), running_stats=berr,
                 worst=sorted(((round(e, 5), n_) for n_, e in errs.items()), reverse=True)[:4])
    desc_max, desc_med, loss_tol, g_tol, g_med = CFG2_GATES[storage]
    assert e_gpu.max().item() < desc_max and e_gpu.median().item() < desc_med, (e_gpu.max().item(), e_gpu.median().item())
    assert e_32.max().item() < desc_max and e_32.median().item() < desc_med, ("vs reference fp32", e_32.max().item(), e_32.median().item())
    assert loss_err < loss_tol, (loss.item(), loss64)
    assert max(errs.values()) < g_tol and float(np.median(list(errs.values()))) < g_med, errs
    assert max(l2errs.values()) < g_tol, l2errs
    assert berr < (2e-4 if storage == "f32" else 2e-2), berr
    for name, b in m.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(b) == 1


# storage -> (descriptor max, descriptor median [norm-relative to the reference's fp64 forward], loss, gradient max, gradient median)
# Measured on MI355X (round 3): fp32 storage 2.3e-4 / 2.8e-5 (closer to fp64 than the reference's own fp32 run: 3.2e-4 / 7.3e-5),
# loss 2e-6, gradients vs the reference's fp32 autograd 1.7e-3 max / 7.7e-4 median, running statistics 5e-5;
# bf16 storage 2.7e-3 / 1.4e-3, loss 4.9e-4, gradients 3.2e-2 max / 9.2e-3 median -- 5-10x tighter than on the B = 6 / B = 16
# fixtures (test_train_bf16_storage_vs_oracle: 1.8e-2, 1.5 %, 7-12 %), as expected when the head's BatchNorms run over 44 rows
# instead of 6, and that is what the gates below state.
# Round 5.  fp32 storage: the 1e-3 / 1.5e-4 descriptor gate stands on test_train_cfg2_full_size_flip_free_vs_fp64_oracle -- with the GPU's
# own graphs and arg-max choices the same step is within 1.9e-5 of fp64, so the excess against the reference's runs IS near-ties.
# bf16 storage, derived from the format: every stored edge value / map element carries a relative rounding of u = 2^-9 (uniform in +-u/2:
# rms u / sqrt(12) = 5.6e-4); a BatchNorm turns it into that fraction of |z| / sigma ~ 1-3 normalised units, the pooling over 4096 points
# and the max over k average it down, and the head's two BatchNorms over the B = 44 descriptor rows amplify what is left ~10x (the fp32
# rounding of the same step, 6e-8, lands at 2.3e-4 / 3e-5: 500x at the maximum, 50x at the median; 5.6e-4 / sqrt(4096 * 20) * 500 = 1e-3).
# Measured: descriptors 1.5e-3 max / 8.9e-4 median, loss 3.1e-4, gradients 1.9e-2 max / 7.7e-3 median (rounds 4 and 5, unchanged by the
# round-5 kernels).  Gates at 1.6-2x of that: one more rounding of a stored tensor (a factor sqrt(2)) still passes, a second one does not.
CFG2_GATES = {"f32": (1e-3, 1.5e-4, 5e-4, 5e-3, 2e-3),
              "bf16": (3e-3, 1.5e-3, 1e-3, 0.035, 0.013)}


def test_bf16_storage_converges_like_fp32(cuda):
    """32 Adam steps on a FIXED set of 8 tuples (bq = 2 -> four batches, cycled), fp32 storage against bf16 storage from the same
    initial weights: both losses must fall, and the bf16 run must end within 10 % of the fp32 run (mean loss over the last cycle
    of four batches).  The margins are 40 / 20 instead of the reference's 0.5 / 0.2 so that the hinges stay active over the whole
    run: with 0.5 / 0.2 these tuples are separated after ONE update (measured: losses 10.0, 0, 0, 21.8, then exactly 0 in both
    modes), which compares nothing."""
    import loss.pointnetvlad_loss as L
    from lpdnet_hip import autograd
    N, bq, P, Ng, steps = 1024, 2, 2, 2, 32
    per = 1 + P + Ng + 1
    tuples = torch.from_numpy(synth.scene_cloud(51, 8 * per, N)).view(4, bq * per, 1, N, 3).to(cuda)
    curves = {}
    for storage in ("f32", "bf16"):
        m, _ = _train_model(N, cuda, "lpdnet")
        opt = torch.optim.Adam(m.parameters(), lr=CONVERGENCE_LR)
        prev = autograd.set_train_storage(storage)
        try:
            losses = []
            for it in range(steps):
                opt.zero_grad(set_to_none=True)
                q, p, n, o = torch.split(m(tuples[it % 4]).view(bq, -1, 256), [1, P, Ng, 1], dim=1)
                loss = L.quadruplet_loss(q, p, n, o, 40.0, 20.0, use_min=True, lazy=True, ignore_zero_loss=False)
                loss.backward()
                opt.step()
                losses.append(loss.item())
        finally:
            autograd.set_train_storage(prev)
        curves[storage] = losses
    first = {s: float(np.mean(c[:4])) for s, c in curves.items()}
    last = {s: float(np.mean(c[-4:])) for s, c in curves.items()}
    if os.environ.get("LPD_TEST_VERBOSE"):
        print("convergence", {s: [round(v, 3) for v in c] for s, c in curves.items()}, first, last)
    for s in curves:
        assert np.isfinite(curves[s]).all()
        assert last[s] < 0.7 * first[s], (s, first[s], last[s])
    assert abs(last["bf16"] - last["f32"]) < 0.10 * max(last["f32"], 0.05 * first["f32"]), (last, first)
    # ... and the two runs track each other step by step (measured: 62.88 / 62.89, 46.32 / 46.28, 25.48 / 25.49, ... until both
    # reach zero after a dozen steps): every step within 10 % of the fp32 loss + 1 % of the initial loss
    for a, b in zip(curves["f32"], curves["bf16"]):
        assert abs(a - b) <= 0.10 * a + 0.01 * first["f32"], (curves["f32"], curves["bf16"])


CONVERGENCE_LR = 1e-4


def test_standalone_train_mode_forwards_of_the_submodules(cuda):
    """TranformNet, STN3d (with and without BatchNorm, k = 3 and k = 64) and GatingContext (with and without BatchNorm) called
    on their own in train mode -- what the reference's modules do when used outside PointNetVlad (lpdnet_model.py:295-313,
    PointNetVlad.py:103-115,152-179): batch statistics, running-stat update, gradients to the input and every parameter,
    against the fp64 oracle."""
    from util.lpdnet_model import TranformNet
    from util.PointNetVlad import STN3d, GatingContext
    dt = torch.float64
    g = torch.Generator().manual_seed(5)

    def run(mod, pre, x, ofn, names):
        mod = mod.to(cuda).train()
        with torch.no_grad():
            for p in mod.parameters():
                p.copy_((torch.rand(p.shape, generator=g) - 0.5) * (0.4 if p.dim() > 1 else 1.0) + (0.8 if p.dim() == 1 else 0.0))
        sd = {pre + k: (v.detach().cpu().to(dt).requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var"))
                        else v.detach().cpu().to(dt) if v.dtype == torch.float32 else v.detach().cpu().clone()) for k, v in mod.state_dict().items()}
        xg = x.to(cuda).requires_grad_(True)
        xo = x.to(dt).requires_grad_(True)
        got = mod(xg)
        new_stats = {}
        want = ofn(sd, xo, new_stats)
        assert got.shape == want.shape
        assert _desc_rel(got.reshape(got.shape[0], -1), want.reshape(want.shape[0], -1)) < 2e-4
        w = torch.randn(want.shape, generator=g)
        (got * w.to(cuda)).sum().backward()
        (want * w.to(dt)).sum().backward()
        assert ((xg.grad.cpu().double() - xo.grad).norm() / xo.grad.norm()).item() < 5e-3
        for n in names:
            a, b = dict(mod.named_parameters())[n].grad.cpu().double(), sd[pre + n].grad
            assert ((a - b).norm() / b.norm()).item() < 5e-3, n
        for n, b in mod.named_buffers():
            if n.endswith("running_mean"):
                assert torch.allclose(b.cpu().double(), new_stats[pre + n], rtol=1e-4, atol=1e-5), n
            if n.endswith("num_batches_tracked"):
                assert int(b) == 1

    B, N = 8, 256
    x3 = torch.randn(B, 3, N, generator=g)
    run(TranformNet(3), "t.", x3, lambda sd, x, ns: orc.transform_net(sd, "t.", x, True, ns), ["conv1.weight", "conv3.weight", "bn3.weight", "fc2.weight", "fc3.bias"])
    x64 = torch.randn(B, 64, N, generator=g)
    run(TranformNet(64), "t.", x64, lambda sd, x, ns: orc.transform_net(sd, "t.", x, True, ns), ["conv1.weight", "bn5.bias", "fc3.weight"])
    xs = torch.randn(B, 1, N, 3, generator=g)
    run(STN3d(num_points=N, k=3, use_bn=True), "s.", xs, lambda sd, x, ns: orc.stn3d(sd, "s.", x, 3, True, ns, True), ["conv1.weight", "bn4.weight", "fc3.weight"])
    run(STN3d(num_points=N, k=3, use_bn=False), "s.", xs, lambda sd, x, ns: orc.stn3d(sd, "s.", x, 3, True, ns, False), ["conv2.weight", "fc1.bias", "fc3.weight"])
    xf = torch.randn(B, 64, N, 1, generator=g)
    run(STN3d(num_points=N, k=64, use_bn=True), "s.", xf, lambda sd, x, ns: orc.stn3d(sd, "s.", x, 64, True, ns, True), ["conv1.weight", "bn1.bias", "fc3.bias"])

    def gating_oracle(bn):
        def f(sd, x, ns):
            gts = torch.matmul(x, sd["g.gating_weights"])
            gts = orc._bn(sd, "g.bn1", gts, True, ns) if bn else gts + sd["g.gating_biases"]
            return x * torch.sigmoid(gts)
        return f
    xh = torch.randn(12, 256, generator=g)
    run(GatingContext(256, add_batch_norm=True), "g.", xh, gating_oracle(True), ["gating_weights", "bn1.weight", "bn1.bias"])
    run(GatingContext(256, add_batch_norm=False), "g.", xh, gating_oracle(False), ["gating_weights", "gating_biases"])


@pytest.mark.parametrize("env", [{"LPD_DEBUG": "no-dg2-bwd-fused,no-tn256,no-gemm-stats,no-split-lds,no-x3t-rows"},
                                 {"LPD_DEBUG": "no-gemm-tn"},
                                 {"LPD_DEBUG": "no-edge-mlp-train"},                                  # round 4: the DG1 -> DG2 stage on the round-3 chain
                                 {"LPD_DEBUG": "no-edge-mlp-train-bwd,no-assign-act"},       # ... its backward on the round-3 chain, bn3 as its own pass
                                 {"LPD_DEBUG": "no-z-bf16"},
                                 {"LPD_DEBUG": "no-map-bf16,no-tn-tr,no-dw-sel-tr"},  # fp32 conv3 map in the bf16 mode, register-transposing dW
                                 {"LPD_DEBUG": "no-split-bwd-bf16,no-feat-in-loader,no-x3w-batched,no-edge-noz,no-cat-bf16,no-pq3-bf16"},
                                 {"LPD_DEBUG": "reduce-grid=4096,x3t-rb=128,tn-blocks=1024"}],     # round 5's launch shapes of the reductions / map gradient / A^T B
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_training_switches_are_live(env):
    """The training-path switches README.md documents are read at import (or at the first launch): the bf16-storage oracle
    comparison and the reference's train step 0 run in a fresh interpreter with the round-3 kernels switched off together (the paths
    they replaced stay tested), and with the register-transposing weight-gradient kernels off."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_train_gpu.py"), "-k",
           "test_train_bf16_storage_vs_oracle or (test_train_step0_vs_reference_golden and lpdnet_bq1) or test_train_step0_cfg2_full_size"]
    r = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
