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


def test_retrieval_oracle_matches_the_kdtree_form():
    """The numpy brute-force restatement of evaluate.py:162-206 (what the GPU path is compared with) gives exactly what the
    reference's own library call (sklearn KDTree) gives, on synthetic runs with the reference's data structure."""
    import numpy as np
    from oracle import retrieval_oracle as ro
    db, qv, qsets = ro.synthetic_runs(seed=3)
    for m in range(3):
        for n in range(3):
            if m == n:
                continue
            a = ro.get_recall_kdtree(m, n, db, qv, qsets)
            b = ro.get_recall_bruteforce(m, n, db, qv, qsets)
            assert np.allclose(a[0], b[0]) and a[2] == b[2] and np.allclose(a[1], b[1])
            assert 0 < b[0][0] <= b[0][-1] <= 100      # not vacuous: some but not all queries are answered at rank 1


def test_checkpoint_interop_with_the_reference_format(tmp_path):
    """save_checkpoint / load_pretrained (train_pointnetvlad.py:64-77,172-199): the .ckpt dictionary keys, strict load,
    optimizer state, the `module.` prefix of DataParallel checkpoints, and bare .t7 state_dicts (strict=False)."""
    from lpdnet_hip import harness
    from util.PointNetVlad import PointNetVlad
    m = PointNetVlad(num_points=256, featnet="lpdnet")
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    ck = tmp_path / "3-model.ckpt"
    harness.save_checkpoint(ck, m, opt, epoch=3, total_iterations=1234, recall=81.5)
    # ... and from an nn.DataParallel wrapper, whose `.module` the reference unwraps before saving (train_pointnetvlad.py:174-177)
    harness.save_checkpoint(tmp_path / "wrapped.ckpt", torch.nn.DataParallel(m), opt, epoch=3, total_iterations=1234, recall=81.5)
    assert list(torch.load(tmp_path / "wrapped.ckpt")["state_dict"]) == list(sd0)
    blob = torch.load(ck)
    assert set(blob) == {"epoch", "iter", "state_dict", "optimizer", "recall"} and blob["iter"] == 1234
    assert list(blob["state_dict"]) == list(sd0)
    m2 = PointNetVlad(num_points=256, featnet="lpdnet")
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    assert harness.load_pretrained(m2, ck, opt2) == (4, 1234)
    assert all(torch.equal(m2.state_dict()[k], v) for k, v in sd0.items())
    # the reference stores `recall` as numpy.float64 (np.mean of the per-run recalls, train_pointnetvlad.py:172-199): a
    # checkpoint written by the reference itself must load (torch >= 2.6 refuses numpy scalars under weights_only=True)
    import numpy as np
    ref_blob = dict(blob, recall=np.float64(81.5))
    ck_ref = tmp_path / "ref.ckpt"
    torch.save(ref_blob, ck_ref)
    m_ref = PointNetVlad(num_points=256, featnet="lpdnet")
    assert harness.load_pretrained(m_ref, ck_ref) == (4, 1234)
    harness.save_checkpoint(tmp_path / "np.ckpt", m, opt, epoch=0, total_iterations=1, recall=np.float64(12.5))
    assert type(torch.load(tmp_path / "np.ckpt")["recall"]) is float
    # a checkpoint written from an nn.DataParallel wrapper ("module." keys), as the reference's script.py handles
    blob["state_dict"] = {"module." + k: v for k, v in blob["state_dict"].items()}
    ck2 = tmp_path / "dp.ckpt"
    torch.save(blob, ck2)
    m3 = PointNetVlad(num_points=256, featnet="lpdnet")
    harness.load_pretrained(m3, ck2)
    assert all(torch.equal(m3.state_dict()[k], v) for k, v in sd0.items())
    # bare state_dict (.t7): non-strict, keys that are missing stay at their initial values
    t7 = tmp_path / "weights.t7"
    partial = {k: v for k, v in sd0.items() if k.startswith("net_vlad.")}
    torch.save(partial, t7)
    m4 = PointNetVlad(num_points=256, featnet="lpdnet")
    assert harness.load_pretrained(m4, t7) == (0, 0)
    assert torch.equal(m4.state_dict()["net_vlad.cluster_weights"], sd0["net_vlad.cluster_weights"])


def test_load_pc_file_contract(tmp_path):
    """ingest.load_pc_file / load_pc_files (loading_pointclouds.py:26-47): float64 [4096,3] from a header-less little-endian
    file, an empty array for a file of the wrong size, which load_pc_files skips."""
    import numpy as np
    from lpdnet_hip import ingest
    g = np.random.default_rng(0)
    good = g.standard_normal((2, 4096, 3))
    good[0].tofile(tmp_path / "a.bin")
    good[1].tofile(tmp_path / "c.bin")
    g.standard_normal(100).tofile(tmp_path / "b.bin")
    pc = ingest.load_pc_file("a.bin", str(tmp_path))
    assert pc.dtype == np.float64 and pc.shape == (4096, 3) and np.array_equal(pc, good[0])
    assert ingest.load_pc_file("b.bin", str(tmp_path)).shape == (0,)
    pcs = ingest.load_pc_files(["a.bin", "b.bin", "c.bin"], str(tmp_path))
    assert pcs.shape == (2, 4096, 3) and np.array_equal(pcs, good)


def test_morton_order_set_on_the_main_thread_reaches_worker_threads():
    """engine.MORTON_ORDER is configuration as well as a test hook: set once on the main thread it must reach threads that never
    set it themselves (nn.DataParallel replica threads, train_pointnetvlad.py:80); a worker's own setting stays its own."""
    import threading
    from lpdnet_hip import engine
    seen = {}

    def worker():
        seen["inherited"] = engine.MORTON_ORDER
        engine.MORTON_ORDER = True           # this thread's override
        seen["own"] = engine.MORTON_ORDER
    assert engine.MORTON_ORDER is True
    engine.MORTON_ORDER = False
    try:
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert seen == {"inherited": False, "own": True}
        assert engine.MORTON_ORDER is False  # the worker's override did not leak back
    finally:
        engine.MORTON_ORDER = True
    assert engine.MORTON_ORDER is True


def test_fused_statistics_gates_stop_at_the_workspace_width():
    """The statistics workspace holds LPD_STAT_CMAX = 1024 columns: wider layers (emb_dims = 2048) must fall back to the separate
    statistics pass / the fp32 map instead of reaching lpd_gemm_x3w_stats, which refuses them."""
    from lpdnet_hip import ops
    assert ops.STAT_CMAX == 1024
    assert ops.linear_bn_stats_fused_applies(44 * 4096, 1024, 512)
    assert not ops.linear_bn_stats_fused_applies(44 * 4096, 2048, 512)
    src = open(os.path.join(ROOT, "lpd-net-pytorch_amd", "csrc", "lpd_common.h")).read()
    assert re.search(r"#define\s+LPD_STAT_CMAX\s+1024\b", src)


def test_bench_kernel_rooflines_from_a_kernel_table():
    """bench.py's per-kernel roofline entries (conv3, edge MLP, both kNN searches): SURVEY 8(d)'s algorithmic FLOPs over the op's event
    time over the peak, the executed fraction (three bf16 products) beside it.  Pure arithmetic on a kernel table: checked here on CPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    kern = {"gemm_p8+assign[131072x1024x512]": {"launches": 5, "avg_us": 450.0}, "edge_mlpx3[128->128]": {"launches": 5, "avg_us": 270.0},
            "knn[C=64,k=20]": {"launches": 5, "avg_us": 400.0}, "knn[C=3,k=20]": {"launches": 5, "avg_us": 200.0}}
    r = bench.kernel_rooflines(kern, 32, 4096, 20, True)
    conv3 = r["conv3 + NetVLAD assignment"]
    assert conv3["algorithmic_flops_per_launch"] == 2 * 131072 * 512 * 1024 + 2 * 131072 * 1024 * 64
    assert abs(conv3["frac"] - conv3["algorithmic_flops_per_launch"] / 450e-6 / 1e12 / 2500.0) < 1e-4
    assert abs(conv3["executed_frac"] - 3 * conv3["frac"]) < 2e-4
    mlp = r["edge MLP (DG1 act -> DG2 conv -> max over k)"]
    assert mlp["algorithmic_flops_per_launch"] == 2 * 131072 * 20 * 128 * 128 and mlp["peak"] == 2500.0
    # kNN: no visit rate -> no roofline fraction at all (only the speed against a perfect brute-force kernel, as its own field)
    knn = r["feature-space kNN"]
    assert knn["frac"] is None and knn["peak"] == 157.3
    assert knn["bruteforce_flops_per_launch"] == 32 * (2 * 4096 * 4096 * 64 + 3 * 4096 * 4096)
    assert abs(knn["speedup_vs_bruteforce_at_peak"] - knn["bruteforce_flops_per_launch"] / 400e-6 / 1e12 / 157.3) < 1e-3
    assert r["xyz kNN"]["bruteforce_flops_per_launch"] == 32 * (2 * 4096 * 4096 * 3 + 3 * 4096 * 4096)
    # with the walk's measured visit rate: executed FLOPs = query tiles x visited tiles x 32 x 32 x 2 C; never above 1
    r = bench.kernel_rooflines(kern, 32, 4096, 20, True, {64: 23.3, 3: 40.0})
    knn = r["feature-space kNN"]
    assert knn["executed_flops_per_launch"] == int(32 * 128 * 23.3 * 32 * 32 * 2 * 64)
    assert abs(knn["frac"] - knn["executed_flops_per_launch"] / 400e-6 / 1e12 / 157.3) < 1e-4 and knn["frac"] < 0.3
    assert r["xyz kNN"]["executed_flops_per_launch"] == int(32 * 128 * 40.0 * 32 * 32 * 2 * 4)
    crazy = bench.kernel_rooflines({"knn[C=64,k=20]": {"launches": 1, "avg_us": 1.0}}, 32, 4096, 20, True, {64: 1e9})
    assert crazy["feature-space kNN"]["frac"] <= 1.0 and crazy["feature-space kNN"]["tiles_visited_per_query_tile"] == 128
    for e in r.values():
        assert e["frac"] is None or e["frac"] <= 1.0
    # the K-agg numerator counts the uint16 indices the kernel reads
    assert bench.KAGG_IDX_BYTES == 2 and bench.KAGG_ROW_BYTES == 3 * 256 * 4
    exact = bench.kernel_rooflines(kern, 32, 4096, 20, False)
    assert "executed_frac" not in exact["conv3 + NetVLAD assignment"] and exact["conv3 + NetVLAD assignment"]["peak"] == 157.3


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_stdout_line_stays_parseable():
    """The driver parses bench.py's final stdout line; round 5's 20 KB line (kernel tables, notes, five secondary records) came back
    unparsed.  compact_line() must keep the contract's keys and stay below 4 KB whatever the full record holds."""
    import json
    bench = _load_bench()
    kern = {f"op{i}[{131072}x{i}x512]": {"launches": 5, "avg_us": 10.0 + i} for i in range(60)}
    kern.update({"gemm_p8+assign[131072x1024x512]": {"launches": 5, "avg_us": 450.0}, "edge_mlpx3[128->128]": {"launches": 5, "avg_us": 270.0},
                 "knn[C=64,k=20]": {"launches": 5, "avg_us": 400.0}, "knn[C=3,k=20]": {"launches": 5, "avg_us": 200.0}})
    rk = bench.kernel_rooflines(kern, 32, 4096, 20, True, {64: 23.3, 3: 40.0})
    note = "a paragraph-long note " * 40
    train = {"metric": "quadruplet train-steps/sec", "value": 122.0, "unit": "steps/s", "ms_per_step": 8.2, "dtype": "bf16", "steps": 10,
             "clouds_per_s": 5368.3, "config": note, "losses": [0.1] * 10, "per_step": {"gpu_ms": [8.2] * 10, "host_enqueue_ms": [3.0] * 10},
             "cpu_baseline": {"value": 0.16, "clouds_per_s": 0.99, "sample": note},
             "exchange": {"allreduce_ms_per_step_isolated": 0.5, "allreduce_busbw_GBps": 250.0, "allreduce_exposed_ms_per_step": 0.1, "buckets": note}}
    sec = {f"secondary record number {i} with a long descriptive name " * 2: {"value": 1.0, "ms_per_step": 2.0, "kernels": kern, "roofline_kernels": rk,
                                                                             "roofline": {"frac": 0.2, "kernel": note}} for i in range(6)}
    sec["operating points"] = {f"{b} clouds/step": {"value": 1.0, "ms_per_step": 0.4} for b in (1, 2, 6, 10, 24)}
    full = {"metric": "global descriptors/sec (4096-pt clouds)", "value": 16468.35, "unit": "descriptors/s", "n_gpus": 8, "steps": 20, "warmup": 3,
            "ms_per_step": 1.943, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": note,
            "config": {"workload": "BASELINE configs[1]: " + note, "parallelism": note, "arithmetic": note, "hip_streams": note, "settle": note},
            "descriptors_per_s_per_rank": [16000.0] * 8,
            "roofline": {"kernel": note, "bound": "hbm", "achieved": 4290.5, "peak": 8000.0, "unit": "GB/s", "frac": 0.5363, "traffic": 423756192,
                         "traffic_source": note, "algorithmic_bytes_per_launch": 407896064, "avg_launch_us": 95.07, "numerator": note,
                         "frac_direct_form": 0.2785, "stage": {"kernels": ["a", "b"], "us": 168.0, "frac_direct_form": 0.158}},
            "roofline_kernels": rk, "kernels": kern, "train": dict(train, dtype="f32"), "train_bf16": train, "secondary": sec,
            "cpu_baseline": {"value": 22.1, "unit": "descriptors/s", "cores": 128, "kind": "port", "sample": note, "sample_short": "128 clouds x 4096 pts, plain-C port",
                             "torch_cpu_cross_check": {"value": 3.6, "sample": note}, "host_cpu": "EPYC"},
            "parity_norm_rel_vs_oracle": 4.0e-7}
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 4096 and "\n" not in text
    back = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "parity_norm_rel_vs_oracle", "train", "train_bf16", "detail"):
        assert key in back, key
    assert back["config"]["workload"].startswith("BASELINE configs[1]")
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "frac_direct_form", "algorithmic_bytes_per_launch", "avg_launch_us"):
        assert key in back["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in back["cpu_baseline"], key
    assert len(back["cpu_baseline"]["sample"]) <= 120
    assert back["train_bf16"]["ms_per_step"] == 8.2 and back["train_bf16"]["dtype"] == "bf16"
    assert "kernels" not in back and back["detail"].endswith(".json")
    # a record without the optional blocks (--no-train --no-secondary --no-cpu-baseline, N > 1) still yields a line
    small = bench.compact_line({k_: full[k_] for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline")})
    assert small["value"] == full["value"] and small["roofline"]["frac"] == 0.5363


def test_header_marks_superseded_entry_points_and_debug_switches_are_one_variable():
    """include/lpd_hip.h lists the entry points that are not on a default path (measured by tools/abi_coverage.py): every name it lists
    is a declared entry point.  The A/B ablation switches are ONE environment variable (LPD_DEBUG, lpdnet_hip/_debug.py + lpd_debug()
    in csrc/lpd_abi.hip): no other LPD_* variable is read anywhere but the documented product switches."""
    hdr = open(os.path.join(ROOT, "include", "lpd_hip.h")).read()
    block = hdr[hdr.index("NOT on a default path"):hdr.index("#ifndef LPD_HIP_H")]
    declared = set(re.findall(r"\b(lpd_[a-z0-9_]+)\s*\(", hdr[hdr.index("#ifndef LPD_HIP_H"):]))
    listed = set(re.findall(r"\b(lpd_[a-z0-9_]+)\b", block)) - {"lpd_knn", "lpd_debug"}
    listed = {n + "_ws_floats" if n == "_ws_floats" else n for n in listed}
    assert len(listed) >= 17 and listed <= declared | {"lpd_hip"}, sorted(listed - declared)
    allowed = {"LPD_DEBUG", "LPD_HIP_LIB", "LPD_GEMM_FP32", "LPD_EVAL_CHUNK", "LPD_SIDE_STREAM", "LPD_REPLAY", "LPD_EXTRA_FLAGS"}
    read = set()
    pkg = os.path.join(ROOT, "lpd-net-pytorch_amd")
    for d, _, files in os.walk(pkg):
        if os.sep + "build" in d:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                read |= set(re.findall(r"(?:getenv\(|environ\.get\(|environ\[)\s*[\"'](LPD_[A-Z0-9_]+)", src))
    assert read <= allowed, sorted(read - allowed)
    from lpdnet_hip import _debug
    assert _debug.on("anything-unset") and not _debug.on("anything-unset", False) and _debug.value("unset-int", 7) == 7
