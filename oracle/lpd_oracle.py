"""oracle/lpd_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the LPD-Net / PointNetVLAD global-descriptor path and its losses, written as
pure functions over a state_dict (`sd`: name -> fp32 torch CPU tensor with the reference's key
names).  Floating-point path => a plain torch-CPU fp32 restatement (explicit
matmul / mean / var formulas; autograd gives the backward used by the training parity tests); the
bit-critical kNN goes through the C restatement oracle/lpd_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; it is the
checker, never the product path (lpd-net-pytorch_amd/ never imports it).

Parity status: PINNED -- checked in tests/test_oracle_golden.py against tests/golden/*.npz,
which hold outputs of the reference itself run in the build container
(tests/golden/make_golden.py).  The reference has no tests of its own.

Citations are file:line relative to /root/reference.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblpd_oracle.so")
_lib = None

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


_C_SOURCES = ("lpd_oracle.c", "lpd_forward.c", "lpd_oracle_knn.h")


def build_c_oracle(force=False):
    """gcc-build oracle/liblpd_oracle.so (recipe = oracle/Makefile)."""
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in _C_SOURCES)
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < newest:
        subprocess.run(["make", "-C", _HERE, "-B", "liblpd_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def build_c_oracle_native(out_path):
    """The same library compiled for THIS host's CPU (-march=native) into out_path: what bench.py's cpu_baseline leg times."""
    subprocess.run(["make", "-C", _HERE, "-B", "liblpd_oracle.so", "ARCH=native", f"OUT={out_path}"], check=True, capture_output=True)
    return out_path


_W_ORDER = (["emb_nn.conv1_lpd.weight"] + [f"emb_nn.bn1_lpd.{s}" for s in ("weight", "bias", "running_mean", "running_var")] +
            ["emb_nn.conv2_lpd.weight"] + [f"emb_nn.bn2_lpd.{s}" for s in ("weight", "bias", "running_mean", "running_var")] +
            ["emb_nn.convDG1.0.weight"] + [f"emb_nn.convDG1.1.{s}" for s in ("weight", "bias", "running_mean", "running_var")] +
            ["emb_nn.convDG2.0.weight"] + [f"emb_nn.convDG2.1.{s}" for s in ("weight", "bias", "running_mean", "running_var")] +
            ["emb_nn.convSN1.0.weight"] + [f"emb_nn.convSN1.1.{s}" for s in ("weight", "bias", "running_mean", "running_var")] +
            ["emb_nn.conv3_lpd.weight"] + [f"emb_nn.bn3_lpd.{s}" for s in ("weight", "bias", "running_mean", "running_var")] +
            ["net_vlad.cluster_weights"] + [f"net_vlad.bn1.{s}" for s in ("weight", "bias", "running_mean", "running_var")] +
            ["net_vlad.cluster_weights2", "net_vlad.hidden1_weights"] +
            [f"net_vlad.bn2.{s}" for s in ("weight", "bias", "running_mean", "running_var")] +
            ["net_vlad.context_gating.gating_weights"] +
            [f"net_vlad.context_gating.bn1.{s}" for s in ("weight", "bias", "running_mean", "running_var")])


def forward_lpdnet_c(sd, x, k=20, threads=0, lib=None):
    """The plain-C restatement of the whole eval path (oracle/lpd_forward.c: the reference's formulation, one cloud per OpenMP
    thread): sd = state_dict of PointNetVlad(featnet='lpdnet') without T-Nets, x [B,1,N,3] -> (descriptors [B,256] float32
    numpy, threads used)."""
    lib = lib if lib is not None else _clib()
    xs = np.ascontiguousarray(x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x, dtype=np.float32).reshape(-1, x.shape[-2], 3)
    B, N = xs.shape[0], xs.shape[1]
    keep = [np.ascontiguousarray((sd[n].detach().cpu().numpy() if isinstance(sd[n], torch.Tensor) else sd[n]), dtype=np.float32) for n in _W_ORDER]
    E = keep[25].shape[0]
    ptrs = (ctypes.c_void_p * len(keep))(*[a.ctypes.data_as(ctypes.c_void_p) for a in keep])
    desc = np.empty((B, 256), np.float32)
    fn = lib.lpd_oracle_forward_lpdnet
    fn.restype = ctypes.c_int
    used = fn(xs.ctypes.data_as(ctypes.c_void_p), B, N, int(k), int(E), ptrs, desc.ctypes.data_as(ctypes.c_void_p), int(threads))
    if used < 0:
        raise MemoryError("lpd_oracle_forward_lpdnet: allocation failed")
    return desc, used


def _clib():
    global _lib
    if _lib is None:
        build_c_oracle()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def num_threads():
    return int(_clib().lpd_oracle_num_threads())


# --------------------------------------------------------------------------------------------
# kNN  (util/lpdnet_model.py:317-326)
# --------------------------------------------------------------------------------------------
def knn_np(x_pm: np.ndarray, k: int):
    """x_pm [B,N,C] point-major fp32 -> (idx int32 [B,N,k], pd fp32 [B,N,k]) by the C restatement."""
    x_pm = np.ascontiguousarray(x_pm, dtype=np.float32)
    B, N, C = x_pm.shape
    idx = np.empty((B, N, k), np.int32)
    pd = np.empty((B, N, k), np.float32)
    _clib().lpd_oracle_knn(x_pm.ctypes.data_as(ctypes.c_void_p), B, N, C, k,
                           idx.ctypes.data_as(ctypes.c_void_p), pd.ctypes.data_as(ctypes.c_void_p))
    return idx, pd


def knn_tie_rows(x_pm: np.ndarray, k: int) -> np.ndarray:
    """[B,N] bool: rows whose top-k depends on a tie rule (excluded from bit-exact comparisons)."""
    x_pm = np.ascontiguousarray(x_pm, dtype=np.float32)
    B, N, C = x_pm.shape
    m = np.zeros((B, N), np.uint8)
    _clib().lpd_oracle_knn_tie_rows(x_pm.ctypes.data_as(ctypes.c_void_p), B, N, C, k,
                                    m.ctypes.data_as(ctypes.c_void_p))
    return m.astype(bool)


def knn(x_bcn: torch.Tensor, k: int) -> torch.Tensor:
    """Reference signature: x [B,C,N] -> idx int64 [B,N,k] (lpdnet_model.py:317)."""
    x_pm = x_bcn.detach().transpose(1, 2).contiguous().numpy()
    idx, _ = knn_np(x_pm, k)
    return torch.from_numpy(idx.astype(np.int64))


# --------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------
def _act_leaky(x, slope=0.01):
    return torch.where(x > 0, x, x * slope)


def _relu(x):
    return torch.clamp(x, min=0.0)


def _conv1x1(sd, key, x):
    """x [B,Cin,*] ; weight sd[key+'.weight'] [Cout,Cin,...ones]; optional bias."""
    w = sd[key + ".weight"]
    w2 = w.reshape(w.shape[0], -1)
    shp = x.shape
    y = torch.matmul(w2, x.reshape(shp[0], shp[1], -1))
    if key + ".bias" in sd:
        y = y + sd[key + ".bias"].reshape(1, -1, 1)
    return y.reshape(shp[0], w2.shape[0], *shp[2:])


def _bn(sd, key, x, train, new_stats):
    """BatchNorm over every dim except 1 (nn.BatchNorm1d/2d semantics: eps 1e-5, momentum 0.1,
    biased var for normalisation, unbiased var into running_var)."""
    C = x.shape[1]
    red = [d for d in range(x.dim()) if d != 1]
    view = [1, C] + [1] * (x.dim() - 2)
    if train:
        mean = x.mean(dim=red)
        var = ((x - mean.reshape(view)) ** 2).mean(dim=red)
        n = x.numel() // C
        if new_stats is not None:
            with torch.no_grad():
                new_stats[key + ".running_mean"] = (1 - BN_MOMENTUM) * sd[key + ".running_mean"] + BN_MOMENTUM * mean
                new_stats[key + ".running_var"] = (1 - BN_MOMENTUM) * sd[key + ".running_var"] + BN_MOMENTUM * var * (n / max(n - 1, 1))
                new_stats[key + ".num_batches_tracked"] = sd[key + ".num_batches_tracked"] + 1
    else:
        mean, var = sd[key + ".running_mean"], sd[key + ".running_var"]
    xhat = (x - mean.reshape(view)) / torch.sqrt(var.reshape(view) + BN_EPS)
    return xhat * sd[key + ".weight"].reshape(view) + sd[key + ".bias"].reshape(view)


def _linear(sd, key, x):
    return torch.matmul(x, sd[key + ".weight"].t()) + sd[key + ".bias"]


def graph_feature(x_bcn, k=20, idx=None):
    """cat(neighbour, centre) edge tensor [B,2C,N,k] (lpdnet_model.py:331-363, cat_or_stack=True)."""
    B, C, N = x_bcn.shape[0], x_bcn.shape[1], x_bcn.shape[2]
    x_bcn = x_bcn.reshape(B, C, N)
    if idx is None:
        idx = knn(x_bcn, k)
    pts = x_bcn.transpose(1, 2)                                   # [B,N,C]
    nbr = torch.stack([pts[b][idx[b]] for b in range(B)])          # [B,N,k,C]
    ctr = pts.unsqueeze(2).expand(B, N, idx.shape[-1], C)
    return torch.cat((nbr, ctr), dim=3).permute(0, 3, 1, 2)


def graph_feature_origin(x_bcn, k=20, idx=None, cat=True):
    """DGCNN-style cat(centre, neighbour - centre) or neighbours only (lpdnet_model.py:116-145)."""
    B, C, N = x_bcn.shape[0], x_bcn.shape[1], x_bcn.shape[2]
    x_bcn = x_bcn.reshape(B, C, N)
    if idx is None:
        idx = knn(x_bcn, k)
    pts = x_bcn.transpose(1, 2)
    nbr = torch.stack([pts[b][idx[b]] for b in range(B)])
    if cat:
        ctr = pts.unsqueeze(2).expand(B, N, idx.shape[-1], C)
        return torch.cat((ctr, nbr - ctr), dim=3).permute(0, 3, 1, 2)
    return nbr.permute(0, 3, 1, 2)


def transform_net(sd, pre, x_bcn, train, new_stats):
    """T-Net (lpdnet_model.py:273-313): conv k->64->128->1024 + BN + ReLU, max over N, fc 512, 256, k*k, + I."""
    kdim = x_bcn.shape[1]
    h = _relu(_bn(sd, pre + "bn1", _conv1x1(sd, pre + "conv1", x_bcn), train, new_stats))
    h = _relu(_bn(sd, pre + "bn2", _conv1x1(sd, pre + "conv2", h), train, new_stats))
    h = _relu(_bn(sd, pre + "bn3", _conv1x1(sd, pre + "conv3", h), train, new_stats))
    h = h.max(dim=2)[0]                                           # [B,1024]
    h = _relu(_bn(sd, pre + "bn4", _linear(sd, pre + "fc1", h), train, new_stats))
    h = _relu(_bn(sd, pre + "bn5", _linear(sd, pre + "fc2", h), train, new_stats))
    h = _linear(sd, pre + "fc3", h)
    h = h + torch.eye(kdim, dtype=h.dtype).reshape(1, kdim * kdim)
    return h.reshape(-1, kdim, kdim)


def _max_k(e, name, argsel):
    """max over the k neighbours (lpdnet_model.py:250,252,258).  argsel[name] ([B,C,N] int64), when given, prescribes WHICH
    neighbour slot is taken for every (cloud, channel, point): tests use it to evaluate the oracle on the arg-max choices the
    GPU made, which removes the last-bit arg-max flips from a gradient comparison (tests/test_train_gpu.py)."""
    if argsel is not None and name in argsel:
        return torch.gather(e, -1, argsel[name].unsqueeze(-1))
    return e.max(dim=-1, keepdim=True)[0]


def _front_input(sd, x, pre, t3d, train, new_stats):
    """lpdnet_model.py:212-229 / :69-86: x [B,1,N,3] or, with use_mFea, [B,1,N,8] (xyz + 5 handcrafted features) ->
    (conv1 input [B,3|8,N], raw xyz [B,3,N]); the coordinate T-Net acts on the xyz part only."""
    p = x.squeeze(1).transpose(1, 2)                               # [B,dims,N]
    if p.shape[1] > 3:
        xyz, feature = p[:, :3], p[:, 3:]
        if t3d:
            trans = transform_net(sd, pre + "t_net3d.", xyz, train, new_stats)
            return torch.cat([torch.bmm(xyz.transpose(1, 2), trans).transpose(1, 2), feature], dim=1), xyz
        return p, xyz
    if t3d:
        trans = transform_net(sd, pre + "t_net3d.", p, train, new_stats)
        return torch.bmm(p.transpose(1, 2), trans).transpose(1, 2), p
    return p, p


def lpdnet_features(sd, x, *, train=False, t3d=False, tfea=False, k=20, new_stats=None, aux=None, pre="emb_nn.", argsel=None):
    """LPDNet.forward (lpdnet_model.py:211-268), useBN=True, cat_or_stack=True.  x [B,1,N,3|8] -> [B,E,N,1]."""
    p, xyz = _front_input(sd, x, pre, t3d, train, new_stats)       # raw xyz feeds the second graph even when t3d (:226)
    f = _act_leaky(_bn(sd, pre + "bn1_lpd", _conv1x1(sd, pre + "conv1_lpd", p), train, new_stats))
    f = _act_leaky(_bn(sd, pre + "bn2_lpd", _conv1x1(sd, pre + "conv2_lpd", f), train, new_stats))
    if tfea:
        tf = transform_net(sd, pre + "t_net_fea.", f, train, new_stats)
        f = torch.bmm(f.transpose(1, 2), tf).transpose(1, 2)
    idx_feat = knn(f, k)
    e = graph_feature(f, k, idx_feat)                              # [B,128,N,k]
    e = _act_leaky(_bn(sd, pre + "convDG1.1", _conv1x1(sd, pre + "convDG1.0", e), train, new_stats))
    x1 = _max_k(e, "x1", argsel)
    e = _act_leaky(_bn(sd, pre + "convDG2.1", _conv1x1(sd, pre + "convDG2.0", e), train, new_stats))
    x2 = _max_k(e, "x2", argsel)
    idx_xyz = knn(xyz, k)
    e = graph_feature(x2, k, idx_xyz)                              # [B,256,N,k]
    e = _act_leaky(_bn(sd, pre + "convSN1.1", _conv1x1(sd, pre + "convSN1.0", e), train, new_stats))
    x3 = _max_k(e, "x3", argsel)
    cat = torch.cat((x1, x2, x3), dim=1).squeeze(-1)               # [B,512,N]
    out = _act_leaky(_bn(sd, pre + "bn3_lpd", _conv1x1(sd, pre + "conv3_lpd", cat), train, new_stats))
    if aux is not None:
        aux.update(F0=f, idx_feat=idx_feat, idx_xyz=idx_xyz, x1=x1, x2=x2, x3=x3)
    return out.unsqueeze(-1)


def lpdnet_origin_features(sd, x, *, train=False, t3d=False, tfea=False, k=20, new_stats=None, aux=None, pre="emb_nn.", argsel=None):
    """LPDNetOrign.forward (lpdnet_model.py:68-114), useBN=True."""
    def seq(name, h):
        return _act_leaky(_bn(sd, pre + name + ".1", _conv1x1(sd, pre + name + ".0", h), train, new_stats))
    p, xyz = _front_input(sd, x, pre, t3d, train, new_stats)
    f = seq("conv1_lpd", p)
    f = seq("conv2_lpd", f)
    if tfea:
        tf = transform_net(sd, pre + "t_net_fea.", f, train, new_stats)
        f = torch.bmm(f.transpose(1, 2), tf).transpose(1, 2)
    idx_feat = knn(f, k)
    e = graph_feature_origin(f, k, idx_feat)                       # [B,128,N,k]
    e = seq("convDG1", e)
    e = seq("convDG2", e)
    g = _max_k(e, "dg", argsel)                                    # [B,64,N,1]
    idx_xyz = knn(xyz, k)
    e = graph_feature_origin(g, k, idx_xyz, cat=False)             # [B,64,N,k]
    e = seq("convSN1", e)
    e = seq("convSN2", e)
    h = _max_k(e, "sn", argsel).squeeze(-1)                        # [B,64,N]
    h = seq("conv3_lpd", h)
    h = seq("conv4_lpd", h)
    h = seq("conv5_lpd", h)
    if aux is not None:
        aux.update(F0=f, idx_feat=idx_feat, idx_xyz=idx_xyz)
    return h.unsqueeze(-1)


def stn3d(sd, pre, x, kdim, train, new_stats, use_bn):
    """STN3d (PointNetVlad.py:126-179). x: [B,1,N,3] for k=3, [B,64,N,1] for k=64."""
    B = x.shape[0]
    if kdim == 3:
        h = x.squeeze(1).transpose(1, 2)                           # Conv2d(1,64,(1,3)) == 3->64 per point
    else:
        h = x.squeeze(-1)

    def blk(conv, bn, h):
        y = _conv1x1(sd, pre + conv, h)
        if use_bn:
            y = _bn(sd, pre + bn, y, train, new_stats)
        return _relu(y)
    h = blk("conv1", "bn1", h)
    h = blk("conv2", "bn2", h)
    h = blk("conv3", "bn3", h)
    h = h.max(dim=2)[0]                                            # MaxPool2d((num_points,1))
    h = _linear(sd, pre + "fc1", h)
    if use_bn:
        h = _bn(sd, pre + "bn4", h, train, new_stats)
    h = _relu(h)
    h = _linear(sd, pre + "fc2", h)
    if use_bn:
        h = _bn(sd, pre + "bn5", h, train, new_stats)
    h = _relu(h)
    h = _linear(sd, pre + "fc3", h)
    h = h + torch.eye(kdim, dtype=h.dtype).reshape(1, kdim * kdim)
    return h.reshape(B, kdim, kdim)


def pointnet_features(sd, x, *, train=False, feature_transform=False, new_stats=None, pre="point_net."):
    """PointNetfeat.forward with max_pool=False (PointNetVlad.py:204-233). x [B,1,N,3] -> [B,E,N,1]."""
    trans = stn3d(sd, pre + "stn.", x, 3, train, new_stats, use_bn=False)
    p = torch.matmul(x.squeeze(1), trans).transpose(1, 2)          # [B,3,N]
    h = _relu(_bn(sd, pre + "bn1", _conv1x1(sd, pre + "conv1", p), train, new_stats))
    h = _relu(_bn(sd, pre + "bn2", _conv1x1(sd, pre + "conv2", h), train, new_stats))
    if feature_transform:
        ft = stn3d(sd, pre + "feature_trans.", h.unsqueeze(-1), 64, train, new_stats, use_bn=False)
        h = torch.matmul(h.transpose(1, 2), ft).transpose(1, 2)
    h = _relu(_bn(sd, pre + "bn3", _conv1x1(sd, pre + "conv3", h), train, new_stats))
    h = _relu(_bn(sd, pre + "bn4", _conv1x1(sd, pre + "conv4", h), train, new_stats))
    h = _bn(sd, pre + "bn5", _conv1x1(sd, pre + "conv5", h), train, new_stats)   # no ReLU after bn5 (:230)
    return h.unsqueeze(-1)


def netvlad(sd, feat, *, train=False, new_stats=None, aux=None, pre="net_vlad."):
    """NetVLADLoupe.forward + GatingContext (PointNetVlad.py:45-83,103-115). feat [B,E,N,1] -> [B,out]."""
    B, E, N = feat.shape[0], feat.shape[1], feat.shape[2]
    x = feat.squeeze(-1).transpose(1, 2)                           # [B,N,E]
    a = torch.matmul(x, sd[pre + "cluster_weights"])               # [B,N,K]
    K = a.shape[-1]
    if pre + "cluster_biases" in sd:                               # add_batch_norm=False (PointNetVlad.py:33-36,55-56)
        a = a + sd[pre + "cluster_biases"]
    else:
        a = _bn(sd, pre + "bn1", a.reshape(-1, K), train, new_stats).reshape(B, N, K)
    a = torch.softmax(a, dim=-1)
    a_sum = a.sum(dim=1, keepdim=True)                             # [B,1,K]
    res = a_sum * sd[pre + "cluster_weights2"]                     # [B,E,K]
    v = torch.matmul(a.transpose(1, 2), x).transpose(1, 2) - res   # [B,E,K]
    v = v / torch.clamp(torch.sqrt((v * v).sum(dim=1, keepdim=True)), min=1e-12)
    v = v.reshape(B, E * K)
    v = v / torch.clamp(torch.sqrt((v * v).sum(dim=1, keepdim=True)), min=1e-12)
    if aux is not None:
        aux.update(vlad=v)
    h = torch.matmul(v, sd[pre + "hidden1_weights"])
    h = _bn(sd, pre + "bn2", h, train, new_stats)
    if pre + "context_gating.gating_weights" not in sd:            # gating=False (PointNetVlad.py:80-81)
        return h
    g = torch.matmul(h, sd[pre + "context_gating.gating_weights"])
    if pre + "context_gating.gating_biases" in sd:                 # GatingContext(add_batch_norm=False) (:94-96,108-109)
        g = g + sd[pre + "context_gating.gating_biases"]
    else:
        g = _bn(sd, pre + "context_gating.bn1", g, train, new_stats)
    return h * torch.sigmoid(g)


def pointnetvlad_forward(sd, x, *, featnet="lpdnet", train=False, feature_transform=False, xyz_trans=False, k=20,
                         new_stats=None, aux=None, argsel=None):
    """PointNetVlad.forward (PointNetVlad.py:261-270). x [B,1,N,3] fp32 -> [B,output_dim]."""
    if featnet == "lpdnet":
        f = lpdnet_features(sd, x, train=train, t3d=xyz_trans, tfea=feature_transform, k=k, new_stats=new_stats, aux=aux,
                            argsel=argsel)
    elif featnet == "lpdnetorigin":
        f = lpdnet_origin_features(sd, x, train=train, t3d=xyz_trans, tfea=feature_transform, k=k, new_stats=new_stats, aux=aux,
                                   argsel=argsel)
    elif featnet == "pointnet":
        f = pointnet_features(sd, x, train=train, feature_transform=feature_transform, new_stats=new_stats)
    else:
        raise ValueError("featnet error")
    return netvlad(sd, f, train=train, new_stats=new_stats, aux=aux)


# --------------------------------------------------------------------------------------------
# losses  (loss/pointnetvlad_loss.py)
# --------------------------------------------------------------------------------------------
def best_pos_distance(q, pos):
    """(:6-12) q [bq,1,D], pos [bq,P,D] -> (min, max) squared distance over the positives."""
    d = ((pos - q) ** 2).sum(dim=2)
    return d.min(dim=1)[0], d.max(dim=1)[0]


def _hinge_reduce(term, lazy, ignore_zero_loss):
    per_q = term.max(dim=1)[0] if lazy else term.sum(dim=1)
    if ignore_zero_loss:
        hard = (per_q > 1e-16).to(per_q.dtype).sum()
        return per_q.sum() / (hard + 1e-16)
    return per_q.mean()


def triplet_loss(q, pos, neg, margin, use_min=False, lazy=False, ignore_zero_loss=False):
    """(:15-42)"""
    mn, mx = best_pos_distance(q, pos)
    positive = (mn if use_min else mx).reshape(-1, 1)
    term = torch.clamp(margin + positive - ((neg - q) ** 2).sum(dim=2), min=0.0)
    return _hinge_reduce(term, lazy, ignore_zero_loss)


def triplet_loss_wrapper(q, pos, neg, other_neg, m1, m2, use_min=False, lazy=False, ignore_zero_loss=False):
    """(:45-46) ignores other_neg and m2."""
    return triplet_loss(q, pos, neg, m1, use_min, lazy, ignore_zero_loss)


def quadruplet_loss(q, pos, neg, other_neg, m1, m2, use_min=False, lazy=False, ignore_zero_loss=False):
    """(:49-97)"""
    mn, mx = best_pos_distance(q, pos)
    positive = (mn if use_min else mx).reshape(-1, 1)
    first = torch.clamp(m1 + positive - ((neg - q) ** 2).sum(dim=2), min=0.0)
    second = torch.clamp(m2 + positive - ((neg - other_neg) ** 2).sum(dim=2), min=0.0)
    return _hinge_reduce(first, lazy, ignore_zero_loss) + _hinge_reduce(second, lazy, ignore_zero_loss)


# --------------------------------------------------------------------------------------------
# helpers for tests
# --------------------------------------------------------------------------------------------
def state_shapes(featnet="lpdnet", *, emb_dims=1024, num_points=4096, output_dim=256, feature_transform=False,
                 xyz_trans=False, use_mFea=False):
    """{key: shape} of the reference state_dict for a configuration."""
    E = emb_dims
    shapes = {}

    def bn(prefix, c):
        shapes[prefix + ".weight"] = (c,)
        shapes[prefix + ".bias"] = (c,)
        shapes[prefix + ".running_mean"] = (c,)
        shapes[prefix + ".running_var"] = (c,)
        shapes[prefix + ".num_batches_tracked"] = ()

    def tnet(prefix, kd):  # TranformNet: Conv1d with bias
        for name, (o, i) in (("conv1", (64, kd)), ("conv2", (128, 64)), ("conv3", (1024, 128))):
            shapes[f"{prefix}.{name}.weight"] = (o, i, 1)
            shapes[f"{prefix}.{name}.bias"] = (o,)
        for name, (o, i) in (("fc1", (512, 1024)), ("fc2", (256, 512)), ("fc3", (kd * kd, 256))):
            shapes[f"{prefix}.{name}.weight"] = (o, i)
            shapes[f"{prefix}.{name}.bias"] = (o,)
        for name, c in (("bn1", 64), ("bn2", 128), ("bn3", 1024), ("bn4", 512), ("bn5", 256)):
            bn(f"{prefix}.{name}", c)

    if featnet == "lpdnet":
        if xyz_trans:
            tnet("emb_nn.t_net3d", 3)
        if feature_transform:
            tnet("emb_nn.t_net_fea", 64)
        for name, (o, i) in (("convDG1", (128, 128)), ("convDG2", (128, 128)), ("convSN1", (256, 256))):
            shapes[f"emb_nn.{name}.0.weight"] = (o, i, 1, 1)
            bn(f"emb_nn.{name}.1", o)
        shapes["emb_nn.conv1_lpd.weight"] = (64, 8 if use_mFea else 3, 1)
        shapes["emb_nn.conv2_lpd.weight"] = (64, 64, 1)
        shapes["emb_nn.conv3_lpd.weight"] = (E, 512, 1)
        bn("emb_nn.bn1_lpd", 64)
        bn("emb_nn.bn2_lpd", 64)
        bn("emb_nn.bn3_lpd", E)
    elif featnet == "lpdnetorigin":
        if xyz_trans:
            tnet("emb_nn.t_net3d", 3)
        if feature_transform:
            tnet("emb_nn.t_net_fea", 64)
        for name, (o, i) in (("convDG1", (64, 128)), ("convDG2", (64, 64)), ("convSN1", (64, 64)), ("convSN2", (64, 64))):
            shapes[f"emb_nn.{name}.0.weight"] = (o, i, 1, 1)
            bn(f"emb_nn.{name}.1", o)
        for name, (o, i) in (("conv1_lpd", (64, 8 if use_mFea else 3)), ("conv2_lpd", (64, 64)), ("conv3_lpd", (64, 64)),
                             ("conv4_lpd", (128, 64)), ("conv5_lpd", (E, 128))):
            shapes[f"emb_nn.{name}.0.weight"] = (o, i, 1)
            bn(f"emb_nn.{name}.1", o)
    elif featnet == "pointnet":
        def stn(prefix, kd):  # use_bn=False inside PointNetfeat: no bn keys
            ch, ks = (1, 3) if kd == 3 else (kd, 1)
            shapes[f"{prefix}.conv1.weight"] = (64, ch, 1, ks)
            shapes[f"{prefix}.conv1.bias"] = (64,)
            shapes[f"{prefix}.conv2.weight"] = (128, 64, 1, 1)
            shapes[f"{prefix}.conv2.bias"] = (128,)
            shapes[f"{prefix}.conv3.weight"] = (1024, 128, 1, 1)
            shapes[f"{prefix}.conv3.bias"] = (1024,)
            for name, (o, i) in (("fc1", (512, 1024)), ("fc2", (256, 512)), ("fc3", (kd * kd, 256))):
                shapes[f"{prefix}.{name}.weight"] = (o, i)
                shapes[f"{prefix}.{name}.bias"] = (o,)
        stn("point_net.stn", 3)
        stn("point_net.feature_trans", 64)
        for name, shp in (("conv1", (64, 1, 1, 3)), ("conv2", (64, 64, 1, 1)), ("conv3", (64, 64, 1, 1)),
                          ("conv4", (128, 64, 1, 1)), ("conv5", (E, 128, 1, 1))):
            shapes[f"point_net.{name}.weight"] = shp
            shapes[f"point_net.{name}.bias"] = (shp[0],)
        for name, c in (("bn1", 64), ("bn2", 64), ("bn3", 64), ("bn4", 128), ("bn5", E)):
            bn(f"point_net.{name}", c)
    else:
        raise ValueError("featnet error")
    K = 64
    shapes["net_vlad.cluster_weights"] = (E, K)
    shapes["net_vlad.cluster_weights2"] = (1, E, K)
    shapes["net_vlad.hidden1_weights"] = (K * E, output_dim)
    bn("net_vlad.bn1", K)
    bn("net_vlad.bn2", output_dim)
    shapes["net_vlad.context_gating.gating_weights"] = (output_dim, output_dim)
    bn("net_vlad.context_gating.bn1", output_dim)
    return shapes


def synthetic_state(featnet="lpdnet", **cfg):
    """Closed-form synthetic state_dict (torch CPU tensors) for a configuration; see oracle/synth.py."""
    from . import synth
    gain = cfg.pop("gain", 1.0)
    shapes = state_shapes(featnet, **cfg)
    return {k: torch.from_numpy(v) for k, v in synth.state_dict_like(shapes, gain=gain).items()}
