"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header declares,
the module mirror carries the reference's names/shapes, and the product path refuses to run
without a GPU (no fallback).  No compute calls."""
import ctypes
import inspect
import os
import re

import pytest
import torch

from oracle import lpd_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "lpd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lpd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from lpdnet_hip import _lib
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"liblpd_hip.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.lpd_version() >= 100
    # no torch symbols in the ABI library's dependencies: pure HIP runtime
    deps = os.popen(f"ldd {_lib.LIB_PATH}").read()
    assert "libtorch" not in deps and "libc10" not in deps


@pytest.mark.parametrize("featnet,kw", [("lpdnet", {}), ("lpdnet", dict(feature_transform=True, xyz_trans=True)),
                                        ("pointnet", {}), ("lpdnetorigin", {}), ("lpdnetorigin", dict(xyz_trans=True))])
def test_state_dict_matches_reference_names_and_shapes(featnet, kw):
    from util.PointNetVlad import PointNetVlad
    m = PointNetVlad(num_points=256, featnet=featnet, **kw)
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    want = {k: tuple(v) for k, v in orc.state_shapes(featnet, num_points=256, **kw).items()}  # verified against the reference
    assert mine == want
    m.load_state_dict(orc.synthetic_state(featnet, num_points=256, **kw), strict=True)
    n_params = sum(p.numel() for p in m.parameters())
    if featnet == "lpdnet" and not kw:
        assert n_params == 17605184 and len(mine) == 55


def test_constructor_signatures_match_reference():
    from util import PointNetVlad as P
    from util import lpdnet_model as L
    import loss.pointnetvlad_loss as Lo

    def params(f):
        return [(n, p.default) for n, p in inspect.signature(f).parameters.items() if n != "self"]
    assert params(P.PointNetVlad.__init__) == [("num_points", 4096), ("global_feat", True), ("feature_transform", False),
                                               ("max_pool", False), ("output_dim", 256), ("emb_dims", 1024),
                                               ("featnet", "lpdnet"), ("xyz_trans", False)]
    assert params(P.NetVLADLoupe.__init__) == [("feature_size", inspect._empty), ("max_samples", inspect._empty),
                                               ("cluster_size", inspect._empty), ("output_dim", inspect._empty),
                                               ("gating", True), ("add_batch_norm", True), ("is_training", True)]
    assert params(P.STN3d.__init__) == [("num_points", 2500), ("k", 3), ("use_bn", True)]
    assert params(P.PointNetfeat.__init__)[:4] == [("num_points", 2500), ("global_feat", True), ("feature_transform", False), ("max_pool", True)]
    assert params(L.LPDNet.__init__)[:5] == [("emb_dims", 512), ("use_mFea", False), ("t3d", True), ("tfea", False), ("use_relu", False)]
    assert params(L.TranformNet.__init__) == [("k", 3), ("negative_slope", 1e-2), ("use_relu", True)]
    assert [n for n, _ in params(L.knn)] == ["x", "k"]
    assert params(L.get_graph_feature) == [("x", inspect._empty), ("k", 20), ("idx", None)]
    assert params(L.get_graph_feature_Origin) == [("x", inspect._empty), ("k", 20), ("idx", None), ("cat", True)]
    assert [n for n, _ in params(Lo.quadruplet_loss)] == ["q_vec", "pos_vecs", "neg_vecs", "other_neg", "m1", "m2", "use_min", "lazy", "ignore_zero_loss"]
    assert [n for n, _ in params(Lo.triplet_loss)] == ["q_vec", "pos_vecs", "neg_vecs", "margin", "use_min", "lazy", "ignore_zero_loss"]
    assert [n for n, _ in params(Lo.best_pos_distance)] == ["query", "pos_vecs"]


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU, never route through a CPU implementation."""
    from lpdnet_hip import LpdHipError
    from util.PointNetVlad import PointNetVlad
    from util import lpdnet_model as L
    import loss.pointnetvlad_loss as Lo
    m = PointNetVlad(num_points=64, featnet="lpdnet").eval()
    with pytest.raises(LpdHipError):
        m(torch.zeros(1, 1, 64, 3))
    with pytest.raises(LpdHipError):
        L.knn(torch.zeros(1, 3, 64), 4)
    with pytest.raises(LpdHipError):
        Lo.quadruplet_loss(torch.zeros(1, 1, 8), torch.zeros(1, 2, 8), torch.zeros(1, 2, 8), torch.zeros(1, 1, 8), 0.5, 0.2)
    with pytest.raises(ValueError):
        PointNetVlad(featnet="bogus")


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lpd-net-pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "lpd_oracle.so" not in text, f
