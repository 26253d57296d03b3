"""Which entry points of include/lpd_hip.h does a DEFAULT run call?  Wraps every function of the loaded library in a counter and runs, with
no LPD_DEBUG token set: eval forwards of the three trunks (lpdnet with and without T-Nets, lpdnetorigin, pointnet; small and large batch,
N = 4096 and a large cloud with k = 64), a quadruplet train step of each trunk in both storage modes, the public ops
(knn / get_graph_feature), ingest, retrieval and hard-negative mining.  Prints the entry points that were never called: those are
reachable only through an LPD_DEBUG token or from tests (on-device cross-checks) -- the list include/lpd_hip.h marks as such.
    python tools/abi_coverage.py > profiles/<tag>_abi_coverage.txt"""
import collections, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lpdnet_hip import _lib
assert not os.environ.get("LPD_DEBUG"), "run without LPD_DEBUG"
lib = _lib.load()
counts = collections.Counter()


def wrap(name, fn):
    def w(*a):
        counts[name] += 1
        return fn(*a)
    w.__name__ = name
    w.argtypes, w.restype = fn.argtypes, fn.restype
    return w


for name in list(_lib.SIGNATURES):
    setattr(lib, name, wrap(name, getattr(lib, name)))

from lpdnet_hip import autograd, harness, ingest, ops
from util.PointNetVlad import PointNetVlad
import util.lpdnet_model as lm
import loss.pointnetvlad_loss as L
dev = torch.device("cuda:0")
torch.manual_seed(0)


def clouds(B, N):
    return (torch.rand(B, 1, N, 3, device=dev) * 2 - 1)


def train_step(m, B, N, bq, P, Ng):
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-5)
    out = m(clouds(B, N)).view(bq, -1, 256)
    q, p, n, o = torch.split(out, [1, P, Ng, 1], dim=1)
    L.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False).backward()
    opt.step()
    L.triplet_loss_wrapper(q.detach(), p.detach(), n.detach(), o.detach(), 0.5, 0.2, False, False, False)


for featnet, kw in (("lpdnet", {}), ("lpdnet", dict(feature_transform=True, xyz_trans=True)), ("lpdnetorigin", {}), ("pointnet", {}),
                    ("pointnet", dict(feature_transform=True))):
    m = PointNetVlad(num_points=4096, featnet=featnet, **kw).to(dev).eval()
    with torch.no_grad():
        for B in (1, 1, 1, 6, 32, 40):
            m(clouds(B, 4096))
    for storage in autograd.TRAIN_STORAGES:
        prev = autograd.set_train_storage(storage)
        try:
            train_step(m, 44, 4096, 2, 2, 18)
            train_step(m, 10, 4096, 2, 1, 2)
        finally:
            autograd.set_train_storage(prev)
    del m
    torch.cuda.empty_cache()
# shapes outside the tuned classes (N not a multiple of 128: generic kernels), the exact-fp32 product mode (LPD_GEMM_FP32=1), the loss helpers
for N_ in (1000, 300):
    m = PointNetVlad(num_points=N_, featnet="lpdnet").to(dev).eval()
    with torch.no_grad():
        m(clouds(3, N_))
    train_step(m, 10, N_, 2, 1, 2)
    del m
was = ops.GEMM_BF16X3
ops.GEMM_BF16X3 = False
try:
    for featnet in ("lpdnet", "lpdnetorigin"):
        m = PointNetVlad(num_points=4096, featnet=featnet).to(dev).eval()
        with torch.no_grad():
            m(clouds(2, 4096)), m(clouds(32, 4096))
        train_step(m, 10, 4096, 2, 1, 2)
        del m
finally:
    ops.GEMM_BF16X3 = was
qv = torch.rand(2, 1, 256, device=dev, requires_grad=True)
pv = torch.rand(2, 3, 256, device=dev, requires_grad=True)
mn, mx = L.best_pos_distance(qv, pv)
(mn.sum() + mx.sum()).backward()
L.triplet_loss(qv, pv, torch.rand(2, 5, 256, device=dev), 0.5, use_min=True, lazy=True).backward()
torch.cuda.empty_cache()
m = PointNetVlad(num_points=16384, featnet="lpdnet").to(dev).eval()
m.emb_nn.k = 64
with torch.no_grad():
    m(clouds(4, 16384))
del m
x = torch.rand(2, 64, 1024, device=dev)
lm.knn(x, 20)
lm.get_graph_feature(x.unsqueeze(-1) if False else x, k=20)
lm.get_graph_feature_Origin(x, k=20)
net = lm.LPDNet(emb_dims=1024, t3d=False).to(dev).eval()
with torch.no_grad():
    net(clouds(2, 1024))
# callers: latent vectors, ingest (f64 -> f32 on the device), retrieval, hard-negative mining
m = PointNetVlad(num_points=1024, featnet="lpdnet").to(dev).eval()
data = np.random.default_rng(0).uniform(-1, 1, (20, 1024, 3))
harness.get_latent_vectors(m, data, 8)
import tempfile
with tempfile.TemporaryDirectory() as td:
    names = []
    for i in range(5):
        data[i].astype(np.float64).tofile(os.path.join(td, f"{i}.bin"))
        names.append(f"{i}.bin")
    ingest.get_latent_vectors_from_files(m, names, 2, dataset_folder=td, num_points=1024)
d = torch.rand(50, 256, device=dev)
ops.retrieval_topk(d[:10], d, 5)
cand = torch.randint(0, 50, (4, 20), dtype=torch.int32, device=dev)
ops.hard_negatives(d, d[:4].contiguous(), cand, 3)
torch.cuda.synchronize()
never = [n for n in _lib.SIGNATURES if counts[n] == 0]
print(f"{len(_lib.SIGNATURES)} entry points bound; {len(_lib.SIGNATURES) - len(never)} called by the default runs; never called ({len(never)}):")
for n in never:
    print("  ", n)
print("calls:", dict(sorted(counts.items(), key=lambda kv: -kv[1])))
