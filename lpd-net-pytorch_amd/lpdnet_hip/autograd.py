"""Training path: torch.autograd.Function wrappers whose forward AND backward run on the HIP kernels.

torch supplies the tape, the parameter/gradient tensors and the optimizer; all arithmetic on
activations happens in liblpd_hip.so.
"""
import ctypes

import torch

from . import _debug, _lib, ops
from .ops import _ptr, _req, _stream


# ------------------------------------------------------------------------------------------------
# The measurement / test hooks (ops.PROFILE, ops.PROFILE_ONLY, engine.DEBUG_AUX) are per host thread, and autograd runs a CUDA
# node's backward on ITS OWN worker thread: every Function remembers the hooks of the thread that ran its forward and its backward
# runs under them, so that a profile or a debug dict sees the backward's launches of ITS forward -- and nobody else's.  The hooks
# are captured when the forward runs: open `ops.PROFILE` / `engine.DEBUG_AUX` BEFORE the forward whose backward is to be recorded
# (a dict opened only around `loss.backward()` records nothing: the worker thread cannot know which thread called backward()).
# ------------------------------------------------------------------------------------------------
def _hooked_forward(fn):
    def forward(ctx, *args):
        from . import engine
        ctx._lpd_hooks = (ops._TLS.PROFILE, ops._TLS.PROFILE_ONLY, engine._TLS.DEBUG_AUX)
        return fn(ctx, *args)
    forward.__doc__ = fn.__doc__
    return forward


def _hooked_backward(fn):
    def backward(ctx, *grads):
        from . import engine
        hooks = getattr(ctx, "_lpd_hooks", None)
        if hooks is None or hooks == (None, None, None):
            return fn(ctx, *grads)
        prev = (ops._TLS.PROFILE, ops._TLS.PROFILE_ONLY, engine._TLS.DEBUG_AUX)
        ops._TLS.PROFILE, ops._TLS.PROFILE_ONLY, engine._TLS.DEBUG_AUX = hooks
        try:
            return fn(ctx, *grads)
        finally:
            ops._TLS.PROFILE, ops._TLS.PROFILE_ONLY, engine._TLS.DEBUG_AUX = prev
    backward.__doc__ = fn.__doc__
    return backward


# ------------------------------------------------------------------------------------------------
# losses (loss/pointnetvlad_loss.py)
# ------------------------------------------------------------------------------------------------
def _rows3(t, name):
    _req(t, name)
    if t.dim() != 3:
        raise ValueError(f"{name}: expected [bq, n, D], got {tuple(t.shape)}")
    if t.stride(2) != 1:
        t = t.contiguous()
    return t


class _MetricLoss(torch.autograd.Function):
    @staticmethod
    @_hooked_forward
    def forward(ctx, q, pos, neg, other, m1, m2, use_min, lazy, ignore_zero, quad):
        q, pos, neg = _rows3(q, "q_vec"), _rows3(pos, "pos_vecs"), _rows3(neg, "neg_vecs")
        if quad:
            other = _rows3(other, "other_neg")
        bq, P, D = pos.shape
        Ng = neg.shape[1]
        if q.shape != (bq, 1, D) or neg.shape[0] != bq or neg.shape[2] != D or (quad and other.shape != (bq, 1, D)):
            raise ValueError("metric loss: inconsistent shapes")
        dev = q.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        minmax = torch.empty((2, bq), dtype=torch.float32, device=dev)
        gq = torch.empty((bq, 1, D), dtype=torch.float32, device=dev)
        gpos = torch.empty((bq, P, D), dtype=torch.float32, device=dev)
        gneg = torch.empty((bq, Ng, D), dtype=torch.float32, device=dev)
        gother = torch.empty((bq, 1, D), dtype=torch.float32, device=dev) if quad else None
        lib = _lib.load()
        _lib.check(lib.lpd_metric_loss(
            _ptr(q), q.stride(0), _ptr(pos), pos.stride(0), pos.stride(1), _ptr(neg), neg.stride(0), neg.stride(1),
            _ptr(other) if quad else None, other.stride(0) if quad else 0, bq, P, Ng, D, float(m1), float(m2),
            int(bool(use_min)), int(bool(lazy)), int(bool(ignore_zero)), int(bool(quad)), _ptr(loss), _ptr(minmax),
            _ptr(gq), _ptr(gpos), _ptr(gneg), _ptr(gother), _stream()), "lpd_metric_loss")
        ctx.save_for_backward(gq, gpos, gneg, gother if quad else gq)
        ctx.quad = quad
        return loss

    @staticmethod
    @_hooked_backward
    def backward(ctx, g):
        gq, gpos, gneg, gother = ctx.saved_tensors
        return (g * gq, g * gpos, g * gneg, (g * gother) if ctx.quad else None, None, None, None, None, None, None)


def metric_loss(q, pos, neg, other, m1, m2, use_min, lazy, ignore_zero, quad):
    return _MetricLoss.apply(q, pos, neg, other, m1, m2, use_min, lazy, ignore_zero, quad)


class _BestPos(torch.autograd.Function):
    """(min_pos, max_pos) of loss/pointnetvlad_loss.py:6-12 with their gradients: d min_pos / dq = -2 (pos_argmin - q), etc."""

    @staticmethod
    @_hooked_forward
    def forward(ctx, query, pos_vecs):
        mn, mx = _best_pos_values(query, pos_vecs)
        ctx.save_for_backward(query, pos_vecs)
        return mn, mx

    @staticmethod
    @_hooked_backward
    def backward(ctx, gmin, gmax):
        query, pos = ctx.saved_tensors
        q, p = _rows3(query, "query"), _rows3(pos, "pos_vecs")
        bq, P, D = p.shape
        gq = torch.empty((bq, 1, D), dtype=torch.float32, device=q.device)
        gp = torch.empty((bq, P, D), dtype=torch.float32, device=q.device)
        lib = _lib.load()
        gmin, gmax = gmin.contiguous().float(), gmax.contiguous().float()      # named: the buffers must outlive the launch call
        _lib.check(lib.lpd_best_pos_bwd(_ptr(q), q.stride(0), _ptr(p), p.stride(0), p.stride(1), _ptr(gmin), _ptr(gmax), bq, P, D,
                                        _ptr(gq), _ptr(gp), _stream()), "lpd_best_pos_bwd")
        return gq, gp


def best_pos_distance(query, pos_vecs):
    """min / max squared distance to the positives (loss/pointnetvlad_loss.py:6-12), differentiable like the reference's."""
    if torch.is_grad_enabled() and (query.requires_grad or pos_vecs.requires_grad):
        return _BestPos.apply(query, pos_vecs)
    return _best_pos_values(query, pos_vecs)


def _best_pos_values(query, pos_vecs):
    q, pos = _rows3(query, "query"), _rows3(pos_vecs, "pos_vecs")
    bq, P, D = pos.shape
    dev = q.device
    scratch = torch.empty((), dtype=torch.float32, device=dev)
    minmax = torch.empty((2, bq), dtype=torch.float32, device=dev)
    gq = torch.empty((bq, 1, D), dtype=torch.float32, device=dev)
    gpos = torch.empty((bq, P, D), dtype=torch.float32, device=dev)
    gneg = torch.empty_like(gpos)
    lib = _lib.load()
    # reuse the fused kernel with the positives standing in as negatives (triplet form)
    _lib.check(lib.lpd_metric_loss(_ptr(q), q.stride(0), _ptr(pos), pos.stride(0), pos.stride(1), _ptr(pos), pos.stride(0),
                                   pos.stride(1), None, 0, bq, P, P, D, 0.0, 0.0, 0, 0, 0, 0, _ptr(scratch), _ptr(minmax),
                                   _ptr(gq), _ptr(gpos), _ptr(gneg), None, _stream()), "lpd_metric_loss")
    return minmax[0], minmax[1]


# ------------------------------------------------------------------------------------------------
# model training forward / backward
#
# Train-mode BatchNorm needs batch statistics over ALL B*N points (or all B*N*k edges) before any
# normalised value exists, so the fused inference kernels do not apply.  The training formulation is:
#   * per-point layers: raw GEMM -> column statistics -> affine+activation (3 passes over [M,C]);
#   * edge stages: the edge tensor U[(i,t)] = P[nbr] + Q[i] IS materialised ([B*N*k, C] fp32 -- 1.85 GB at
#     B=44, C=128; sized for 288 GB of HBM) so BatchNorm statistics, the DG2 convolution (a plain
#     [E,128]x[128,128] GEMM) and every backward step are row-parallel kernels + lpd_gemm;
#   * backward of the neighbour gather = group-sum (centre term) + atomic row scatter (neighbour term).
# Gradients do not flow through the kNN indices (same as the reference: topk indices are not
# differentiable) nor into the input cloud.
# ------------------------------------------------------------------------------------------------
LEAKY = 0.01

# Storage type of the large saved / materialised training tensors (edge tensors, their gradients, saved activations):
# "f32" (default: what the 1e-4 parity gates are written for) or "bf16" (BASELINE configs[2] as stated: bf16 storage,
# statistics / accumulations / kNN in fp32-fp64; looser stated tolerance, tests/test_train_gpu.py).
TRAIN_STORAGES = ("f32", "bf16")
TRAIN_STORAGE = "f32"


# bf16 storage also keeps the [B N, 1024] conv3 map -- the raw conv3 output, its activated form and both gradients -- in bf16 where the
# trunk hands the raw map to the NetVLAD head (PointNetVlad's train path, lpdnet_features_train(defer_act=True)); LPD_DEBUG=map-bf16=0: fp32 map
MAP_BF16 = _debug.on("map-bf16")
# ... and a bf16 COPY of the point features [x1 | x2 | x3] beside the fp32 ones (written by the activation passes that produce them):
# conv3 takes it as bf16 rows (two products per term), its weight gradient as bf16 rows on both sides (one).  LPD_DEBUG=cat-bf16=0: off
CAT_BF16 = _debug.on("cat-bf16")
PQ3_BF16 = _debug.on("pq3-bf16")      # ... and the gradient of the SN1 projection [B N, 512]


def set_train_storage(kind):
    """-> the previous setting"""
    global TRAIN_STORAGE
    if kind not in TRAIN_STORAGES:
        raise NotImplementedError(f"training storage {kind!r} is not built (available: {TRAIN_STORAGES})")
    prev, TRAIN_STORAGE = TRAIN_STORAGE, kind
    return prev


def _saved(ctx):
    """The forward's saved tensors live in a plain dict on ctx (freed at the end of backward: several GB of edge tensors)."""
    if ctx.saved is None:
        raise RuntimeError("Trying to backward through the LPD-Net HIP graph a second time: the saved edge tensors were freed "
                           "after the first backward (retain_graph=True is not supported on this path; run the forward again)")
    return ctx.saved


def _pick_splits(R, tiles):
    """split-K factor for a reduction over R rows: ~512 blocks, >= 8 k-tiles per block, R % (32*s) == 0."""
    s = 1
    while s * 2 * tiles <= 512 and R // (s * 2) >= 256:
        s *= 2
    return s


def _dweight(dY, X, rows=None):
    """dW [Co, Kin] = dY^T X over the first `rows` rows (reduction on the MFMA GEMM, split-K)."""
    R = dY.shape[0] if rows is None else rows
    Co, Kin = dY.shape[1], X.shape[1]
    if Kin <= 8:
        return ops.dw_smallk(dY[:R], X[:R])
    if ops.gemm_tn_applies(dY, X, R):       # long reductions: the register-transposing split-bf16 kernel
        return ops.gemm_tn(dY, X, rows=R)
    tiles = ((Co + 127) // 128) * ((Kin + 127) // 128)
    return ops.gemm(dY[:R], X[:R], a_kmajor=True, b_kmajor=True, splits=_pick_splits(R, tiles))


class _PointLayer:
    """y = x W^T (raw) ; z = act(BN_train(y)).  Helper used inside the autograd Functions."""

    @staticmethod
    def fwd(x, w2d, bn, act, slope):
        y, st = ops.linear_bn_stats(x, w2d, bn)       # the statistics ride in the GEMM's epilogue where that kernel is built
        z = ops.affine_act(y, st.scale, st.shift, act, slope)
        return y, st, z

    @staticmethod
    def bwd(dz, x, w2d, y, st, act, slope, need_dx=True, inplace=True):
        dy, dgamma, dbeta = ops.bn_act_bwd(dz, y, st, act, slope, out=dz if inplace else None)
        dw = _dweight(dy, x)
        dx = ops.gemm(dy, w2d, b_kmajor=True) if need_dx else None               # dY [R,Co] x W [Co,Kin]
        return dx, dw, dgamma, dbeta


class _Dense:
    """y = act(BN_train(x W^T + bias)) or act(x W^T + bias) -- the general per-row layer (bias and BatchNorm optional;
    T-Net / STN / PointNet stacks: util/lpdnet_model.py:297-305, util/PointNetVlad.py:152-175,213-230)."""

    @staticmethod
    def fwd(x, lin, bn, act, slope=0.0, rows=None):
        from . import engine
        w = engine._w2d(lin)
        raw = ops.linear(x, w, bias=lin.bias)
        if bn is not None:
            st = ops.bn_train_stats(raw, bn, rows=rows)
            out = ops.affine_act(raw, st.scale, st.shift, act, slope, rows=rows)
        else:
            st = None
            out = ops.affine_act(raw, None, None, act, slope, rows=rows) if act != ops.ACT_NONE else raw
        return raw, st, out

    @staticmethod
    def bwd(dout, x, lin, raw, st, act, slope=0.0, need_dx=True, inplace=True, rows=None, dx_accum=None):
        """-> (dx or None, dW, dbias or None, dgamma or None, dbeta or None); dx_accum: add dx into this buffer instead"""
        from . import engine
        w = engine._w2d(lin)
        R = raw.shape[0] if rows is None else rows
        if st is None and act == ops.ACT_NONE:
            draw = dout
            dbias = ops.colsum(draw, rows=R) if lin.bias is not None else None
            dgamma = dbeta = None
        else:
            draw, dgamma, dbeta = ops.bn_act_bwd(dout, raw, st, act, slope, out=dout if inplace else None, rows=R)
            if st is None:
                dbias, dgamma, dbeta = (dbeta if lin.bias is not None else None), None, None   # sum of dpre = bias gradient
            else:
                dbias = ops.colsum(draw, rows=R) if lin.bias is not None else None               # analytically 0 before a BN
        dw = _dweight(draw, x, rows=R)
        dx = None
        if need_dx:
            if w.shape[1] % 4 != 0:   # 3-channel input: the B operand needs a leading dim that is a multiple of 4
                wp = torch.zeros((w.shape[0], (w.shape[1] + 3) // 4 * 4), dtype=w.dtype, device=w.device)
                wp[:, :w.shape[1]] = w
                dx = ops.gemm(draw[:R], wp, b_kmajor=True)[:, :w.shape[1]]
            elif dx_accum is not None:
                dx = ops.gemm(draw[:R], w, b_kmajor=True, out=dx_accum, accumulate=True)
            else:
                dx = ops.gemm(draw[:R], w, b_kmajor=True)
        return dx, dw.reshape(lin.weight.shape), dbias, dgamma, dbeta


class _TNet:
    """TranformNet (lpdnet_model.py:273-313) / STN3d (PointNetVlad.py:126-179) in train mode: rows [B*N, kd] -> [B,kd,kd]."""

    NAMES = ("conv1", "conv2", "conv3", "fc1", "fc2", "fc3")

    @staticmethod
    def param_names(net, use_bn):
        names = []
        for i, lname in enumerate(_TNet.NAMES):
            names += [f"{lname}.weight", f"{lname}.bias"]
            if use_bn and i < 5:
                names += [f"bn{i + 1}.weight", f"bn{i + 1}.bias"]
        return names

    @staticmethod
    def fwd(net, h, B, N, use_bn):
        kd = net.k
        bns = [getattr(net, f"bn{i}") if use_bn else None for i in range(1, 6)]
        S = dict(h=h)
        with ops.exact_gemm():   # BatchNorm over the B rows of the fc stack amplifies GEMM rounding ~100x: f32-input MFMA here
            return _TNet._fwd(net, h, B, N, kd, bns, S)

    @staticmethod
    def _fwd(net, h, B, N, kd, bns, S):
        S["r1"], S["s1"], a1 = _Dense.fwd(h, net.conv1, bns[0], ops.ACT_RELU)
        S["r2"], S["s2"], a2 = _Dense.fwd(a1, net.conv2, bns[1], ops.ACT_RELU)
        S["r3"], S["s3"], a3 = _Dense.fwd(a2, net.conv3, bns[2], ops.ACT_RELU)
        g, S["arg"] = ops.colmax_arg(a3, B, N)
        del a3
        S["r4"], S["s4"], a4 = _Dense.fwd(g, net.fc1, bns[3], ops.ACT_RELU)
        S["r5"], S["s5"], a5 = _Dense.fwd(a4, net.fc2, bns[4], ops.ACT_RELU)
        w3 = net.fc3.weight
        kk = kd * kd
        ld = (kk + 3) // 4 * 4
        t = torch.zeros((B, ld), dtype=torch.float32, device=h.device)        # padded leading dim (k*k = 9 -> 12)
        eye = torch.eye(kd, device=h.device).flatten()
        ops.linear(a5, w3, bias=(net.fc3.bias + eye), out=t[:, :kk])
        S.update(a1=a1, a2=a2, g=g, a4=a4, a5=a5, ld=ld)
        return t[:, :kk].reshape(B, kd, kd).contiguous(), S

    @staticmethod
    def bwd(net, dtrans, S, B, N, use_bn, need_dh, dh_accum=None):
        """dtrans [B,kd,kd] -> (dh [B*N,kd] or None, grads in param_names order); dh_accum: buffer dh is added into"""
        kd = net.k
        kk, ld = kd * kd, S["ld"]
        dt = torch.zeros((B, ld), dtype=torch.float32, device=dtrans.device)
        dt[:, :kk] = dtrans.reshape(B, kk)
        # fc3 (no BN, no activation)
        dw3 = ops.gemm(dt, S["a5"], a_kmajor=True, b_kmajor=True)[:kk]          # [ld,256] -> [kk,256]
        db3 = ops.colsum(dt)[:kk]
        w3p = torch.zeros((ld, net.fc3.weight.shape[1]), dtype=torch.float32, device=dt.device)   # rows padded like dt's columns
        w3p[:kk] = net.fc3.weight
        da5 = ops.gemm(dt, w3p, b_kmajor=True)                                  # [B,256]
        da4, dw2f, db2f, dg5, dbt5 = _Dense.bwd(da5, S["a4"], net.fc2, S["r5"], S["s5"], ops.ACT_RELU)
        dg, dw1f, db1f, dg4, dbt4 = _Dense.bwd(da4, S["g"], net.fc1, S["r4"], S["s4"], ops.ACT_RELU)
        da3 = ops.colmax_bwd(dg, S["arg"], N)
        da2, dwc3, dbc3, dg3, dbt3 = _Dense.bwd(da3, S["a2"], net.conv3, S["r3"], S["s3"], ops.ACT_RELU)
        da1, dwc2, dbc2, dg2, dbt2 = _Dense.bwd(da2, S["a1"], net.conv2, S["r2"], S["s2"], ops.ACT_RELU)
        dh, dwc1, dbc1, dg1, dbt1 = _Dense.bwd(da1, S["h"], net.conv1, S["r1"], S["s1"], ops.ACT_RELU, need_dx=need_dh,
                                                  dx_accum=dh_accum)
        per_layer = [(dwc1, dbc1, dg1, dbt1), (dwc2, dbc2, dg2, dbt2), (dwc3, dbc3, dg3, dbt3), (dw1f, db1f, dg4, dbt4),
                     (dw2f, db2f, dg5, dbt5), (dw3.reshape(net.fc3.weight.shape), db3, None, None)]
        grads = []
        for i, (dw, db, dgm, dbt) in enumerate(per_layer):
            grads += [dw, db]
            if use_bn and i < 5:
                grads += [dgm, dbt]
        return dh, grads


def _transform_bwd(x, trans, dy, B, N):
    """backward of y[m] = x[m] @ trans[cloud(m)]: (dx [M,kd], dtrans [B,kd,kd])."""
    kd = x.shape[1]
    if kd <= 8:
        dx = ops.apply_transform(dy.contiguous(), trans.transpose(1, 2).contiguous(), N)
        dtr = ops.cloud_outer(x, dy, B, N)
    else:
        dyc = dy.contiguous()
        dx = ops.gemm(dyc.view(B, N, kd), trans, b_kmajor=False).view(B * N, kd)            # dy @ trans^T
        dtr = ops.gemm(x.contiguous().view(B, N, kd), dyc.view(B, N, kd), a_kmajor=True, b_kmajor=True)   # x^T dy per cloud
    return dx, dtr


class _Front:
    """The shared front of both LPD-Net trunks: [T-Net3d ->] conv1+bn1 -> conv2+bn2 [-> feature T-Net]
    (util/lpdnet_model.py:85-99 and :226-241).  Bias-free convs; T-Nets optional."""

    @staticmethod
    def tnet_param_names(net):
        names = []
        if net.t3d:      # gradient order of bwd(): t_net3d, then t_net_fea
            names += ["t_net3d." + n for n in _TNet.param_names(net.t_net3d, True)]
        if net.tfea:
            names += ["t_net_fea." + n for n in _TNet.param_names(net.t_net_fea, True)]
        return names

    @staticmethod
    def fwd(net, xyz, w1, bn1, w2, bn2, B, N, act, slope, p_all=None):
        """p_all: the [M, 8] input rows under use_mFea (xyz + 5 features, lpdnet_model.py:215-224); conv1 then sees 8 columns"""
        from . import engine
        S = dict(xyz=xyz, p_in=xyz if p_all is None else p_all)
        with ops.exact_gemm():      # F0 feeds the feature-space kNN: exact fp32 (the backward may use the fast GEMM)
            if net.t3d:
                S["trans3"], S["S3"] = _TNet.fwd(net.t_net3d, xyz, B, N, True)
                S["p_in"] = engine._aligned_input(ops.apply_transform(xyz, S["trans3"], N), p_all, p_all is not None)
            S["y1"], S["st1"], S["f1"] = _PointLayer.fwd(S["p_in"], w1, bn1, act, slope)
            S["y2"], S["st2"], f0 = _PointLayer.fwd(S["f1"], w2, bn2, act, slope)
            if net.tfea:
                S["f_pre"] = f0
                S["transf"], S["Sf"] = _TNet.fwd(net.t_net_fea, f0, B, N, True)
                f0 = ops.apply_transform(f0, S["transf"], N)
        return f0, S

    @staticmethod
    def bwd(net, df0, S, w1, w2, B, N, act, slope):
        """-> ((dW1, dgamma1, dbeta1, dW2, dgamma2, dbeta2), [T-Net grads in tnet_param_names order])"""
        g_f, g_3 = [], []
        if net.tfea:   # f = f_pre @ transf: gradient to f_pre directly and through the feature T-Net
            dfp, dtf = _transform_bwd(S["f_pre"], S["transf"], df0, B, N)
            df0, g_f = _TNet.bwd(net.t_net_fea, dtf, S["Sf"], B, N, True, need_dh=True, dh_accum=dfp)
        df1, dwc2, dg2, db2 = _PointLayer.bwd(df0, S["f1"], w2, S["y2"], S["st2"], act, slope)
        if net.t3d:    # p = xyz @ trans3 feeds conv1: gradient to trans3 only (the cloud itself needs none)
            dy1, dg1, db1 = ops.bn_act_bwd(df1, S["y1"], S["st1"], act, slope, out=df1)
            dwc1 = _dweight(dy1, S["p_in"])
            if w1.shape[1] % 4 == 0:      # use_mFea: 8 input columns, the gradient of the 3 aligned coordinates is the first 3
                dp = ops.gemm(dy1, w1.contiguous(), b_kmajor=True)[:, :3]
            else:
                w1p = torch.zeros((w1.shape[0], 4), dtype=torch.float32, device=w1.device)
                w1p[:, :3] = w1
                dp = ops.gemm(dy1, w1p, b_kmajor=True)[:, :3]
            dt3 = ops.cloud_outer(S["xyz"], dp, B, N)
            _, g_3 = _TNet.bwd(net.t_net3d, dt3, S["S3"], B, N, True, need_dh=False)
        else:
            _, dwc1, dg1, db1 = _PointLayer.bwd(df1, S["p_in"], w1, S["y1"], S["st1"], act, slope, need_dx=False)
        return (dwc1, dg1, db1, dwc2, dg2, db2), list(g_3) + list(g_f)


def _unsplit_cat_nc(dwcat, conv_weight):
    """gradient of the stacked [W_n ; W_c] projection -> gradient of the conv weight [Co, 2Ci, 1, 1]."""
    co = conv_weight.shape[0]
    return torch.cat((dwcat[:co], dwcat[co:]), dim=1).reshape(conv_weight.shape)


class _LPDNetTrainFn(torch.autograd.Function):
    """LPDNet.forward in train mode (util/lpdnet_model.py:211-268, t3d = tfea = False)."""

    PARAMS = ("conv1_lpd.weight", "bn1_lpd.weight", "bn1_lpd.bias", "conv2_lpd.weight", "bn2_lpd.weight", "bn2_lpd.bias",
              "convDG1.0.weight", "convDG1.1.weight", "convDG1.1.bias", "convDG2.0.weight", "convDG2.1.weight",
              "convDG2.1.bias", "convSN1.0.weight", "convSN1.1.weight", "convSN1.1.bias", "conv3_lpd.weight",
              "bn3_lpd.weight", "bn3_lpd.bias")

    @staticmethod
    @_hooked_forward
    def forward(ctx, net, x, *params):
        if TRAIN_STORAGE == "bf16":      # bf16 mode: the dense products behind the kNN take bf16 operands (one MFMA product)
            with ops.bf16_gemm():
                return _LPDNetTrainFn._forward(ctx, net, x, *params)
        with ops.train_forward_gemm(x.shape[0]):
            return _LPDNetTrainFn._forward(ctx, net, x, *params)

    @staticmethod
    def _forward(ctx, net, x, *params):
        from . import engine
        B, N = x.shape[0], x.shape[2]
        M, k = B * N, net.k
        act, slope = (ops.ACT_RELU, 0.0) if net.use_relu else (ops.ACT_LEAKY, LEAKY)
        bf16 = TRAIN_STORAGE == "bf16"
        w2d = engine._w2d
        p_all = None
        if x.shape[3] == 8:       # use_mFea
            xyz, p_all = engine.split_mfea(x)
            x = xyz.view(B, 1, N, 3)
        else:
            xyz = x.view(M, 3)
        f0, front = _Front.fwd(net, xyz, w2d(net.conv1_lpd), net.bn1_lpd, w2d(net.conv2_lpd), net.bn2_lpd, B, N, act, slope, p_all)
        idx_f = engine._knn_rows(f0, B, N, 64, k)
        cat = torch.empty((M, 512), dtype=torch.float32, device=x.device)
        # DG1 -> DG2 (split projection; convDG2 consumes EVERY post-activation edge, so these edge tensors must exist:
        # fp32, or bf16 under set_train_storage("bf16"))
        wcat1 = engine.split_edge_weight(net.convDG1, "cat_nc")
        pq1 = ops.linear(f0, wcat1)                                             # [M,256] = [P | Q]
        w2 = w2d(net.convDG2[0])
        post1 = ops.edge_mlp_train_applies(M, N, k, 128, act, slope) and w2.is_contiguous() and tuple(w2.shape) == (128, 128)
        s1sum = None
        Co3 = net.conv3_lpd.weight.shape[0]
        defer = getattr(_LAST, "defer_act", False) and ops.gemm_act_applies(M, 64, Co3)
        map16 = (defer and bf16 and MAP_BF16 and ops.GEMM_TN and ops.linear_bn_stats_fused_applies(M, Co3, 512) and Co3 % 128 == 0 and Co3 <= ops.STAT_CMAX
                 and (Co3 & (Co3 - 1)) == 0 and N % 128 == 0 and N >= 256 and M >= 16384 and ops.X3T_ROWS and ops.X3W_BATCHED)
        # (the shapes every bf16-map kernel of the head is built for: pooling and weight gradients >= 4096 rows, the batched short
        #  products >= 16384 rows over clouds of whole 128-row tiles; smaller steps keep the fp32 map)
        cat16 = (torch.empty((M, 512), dtype=torch.bfloat16, device=x.device)
                 if (map16 and post1 and CAT_BF16 and Co3 % 256 == 0 and M % 32 == 0) else None)
        c16 = (lambda a, b: cat16[:, a:b]) if cat16 is not None else (lambda a, b: None)
        # with the bf16 copy in place nothing reads the fp32 form of x1 and x3 (conv3 and dW3 take the copy; the SN1 projection and its
        # weight gradient read x2): their fp32 columns of `cat` stay unwritten -- 277 MB of stores per step at 44 clouds
        o16 = cat16 is not None and engine.DEBUG_AUX is None
        post1c = None
        if post1:
            # ONE launch for the stage (lpd_edge_mlp_train): the raw edge tensor U1 is never written.  Its BatchNorm statistics, and
            # x1 = max_k act(BN(U1)) with its arg-max, come from the split-form gather pass (closed-form sums; the activation is monotone,
            # so the maximum is act(BN(.)) of the selected raw value); the fused kernel builds Y1e from the gathered rows, multiplies it
            # by W2 and leaves Y1e, Z, the statistics of Z and its per-point selection (by the sign of gamma2) behind
            s1sum, usel1, arg1, stg1 = ops.edge_split_fwd(pq1[:, :128], pq1[:, 128:], idx_f, N, bn=net.convDG1[1])
            ops.affine_act(usel1, stg1.scale, stg1.shift, act, slope, out=cat[:, 0:128], out16=c16(0, 128), only16=o16)    # x1
            # (fp32 storage: Z as bf16 when the backward is the one that only takes xhat2 from it, lpd_edge_mlp_train_bwd)
            z16 = bf16 or (ops.Z_BF16 and ops.EDGE_MLP_TRAIN_BWD and ops.dg2_bwd_fused_applies(M, k, 128) and M % 32 == 0)
            # ... and (bf16 storage) no Z at all when that backward will run: it forms the xhat2 term as Y1e K (ops.EDGE_NOZ)
            noz = (bf16 and ops.EDGE_NOZ and ops.EDGE_MLP_TRAIN_BWD and ops.dg2_bwd_fused_applies(M, k, 128) and M % 32 == 0 and ops.GEMM_BF16X3
                   and w2.is_contiguous())
            y1e, z, zsel, arg2, stg2 = ops.edge_mlp_train(pq1[:, :128], pq1[:, 128:], idx_f, N, stg1.scale, stg1.shift, w2,
                                                          net.convDG2[1], act, slope, bf16, z_bf16=z16, store_z=not noz)
            ops.affine_act(zsel, stg2.scale, stg2.shift, act, slope, out=cat[:, 128:256], out16=c16(128, 256))             # x2
            u1 = None
            post1c = ops.post_consts(net.convDG1[1], act, slope)      # (beta1, 1 / gamma1, 1 / ns) as of THIS forward, for the backward
            del usel1
        elif bf16:
            u1, stg1 = ops.edge_build_bf16(pq1[:, :128], pq1[:, 128:], idx_f, N, bn=net.convDG1[1])   # [E,128] raw + its BN statistics
            y1e, arg1 = ops.edge_act_max_bf16(u1, k, stg1, act, slope, out=cat[:, 0:128])             # post-activation edges + x1
            z = ops.gemm_bf16s(y1e, w2d(net.convDG2[0]))                        # [E,128] raw, bf16 MFMA
            zsel, arg2, stg2 = ops.group_sel_stats_bf16(z, k, net.convDG2[1])   # statistics of z + the raw selected values
            ops.affine_act(zsel, stg2.scale, stg2.shift, act, slope, out=cat[:, 128:256])             # x2
        else:
            u1, stg1 = ops.edge_build(pq1[:, :128], pq1[:, 128:], idx_f, N, bn=net.convDG1[1])   # [E,128] raw + its BN statistics
            y1e, arg1 = ops.edge_act_max(u1, k, stg1, act, slope, out=cat[:, 0:128])                  # post-activation edges + x1, one pass over u1
            z, stg2 = ops.linear_bn_stats(y1e, w2d(net.convDG2[0]), net.convDG2[1])   # [E,128] raw + its statistics (GEMM epilogue)
            arg2, zsel = ops.group_max(z, k, stg2.scale, stg2.shift, act, slope, cat[:, 128:256], keep_sel=True)   # x2
        # SN1 on the xyz graph, split form: statistics, max and arg-max from one gather pass, no [E,256] tensor
        idx_x = engine._knn_rows(x.view(B * N, 3), B, N, 3, k)
        wcat3 = engine.split_edge_weight(net.convSN1, "cat_nc")
        pq3 = ops.linear(cat[:, 128:256], wcat3)                                # [M,512] = [P | Q]
        s3, usel3, arg3, stg3 = ops.edge_split_fwd(pq3[:, :256], pq3[:, 256:], idx_x, N, bn=net.convSN1[1])
        ops.affine_act(usel3, stg3.scale, stg3.shift, act, slope, out=cat[:, 256:512], out16=c16(256, 512), only16=o16)    # x3
        if defer:
            # PointNetVlad's train path: bn3 + act are applied by the NetVLAD assignment product's operand loader (ops.gemm_act), which
            # also writes the activated map; this Function hands the RAW conv3 output on, with the affine in _LAST.pending
            y3, st3 = ops.linear_bn_stats(cat if cat16 is None else cat16, w2d(net.conv3_lpd), net.bn3_lpd, out_bf16=map16)      # (bf16 storage: a bfloat16 tensor)
            feat = y3
            _LAST.pending = (st3.scale, st3.shift, act, slope)
        else:
            y3, st3, feat = _PointLayer.fwd(cat, w2d(net.conv3_lpd), net.bn3_lpd, act, slope)
            _LAST.pending = None
        ctx.net, ctx.dims, ctx.actslope, ctx.bf16 = net, (B, N, M, k), (act, slope), bf16
        ctx.saved = dict(front=front, f0=f0, idx_f=idx_f, idx_x=idx_x, wcat1=wcat1, post1=post1, post1c=post1c, pq1=pq1 if post1 else None, s1sum=s1sum,
                         u1=u1, stg1=stg1, arg1=arg1, y1e=y1e, z=z, zsel=zsel, stg2=stg2, arg2=arg2, wcat3=wcat3, pq3=pq3, s3=s3, usel3=usel3,
                         stg3=stg3, arg3=arg3, cat=cat, cat16=cat16, y3=y3, st3=st3)
        if engine.DEBUG_AUX is not None:
            engine.DEBUG_AUX.update(F0=f0, idx_feat=idx_f, idx_xyz=idx_x, cat=cat, argsel=dict(x1=arg1, x2=arg2, x3=arg3))
        return feat

    @staticmethod
    @_hooked_backward
    def backward(ctx, dfeat):
        if ctx.bf16:
            with ops.bf16_gemm():
                return _LPDNetTrainFn._backward(ctx, dfeat)
        return _LPDNetTrainFn._backward(ctx, dfeat)

    @staticmethod
    def _backward(ctx, dfeat):
        from . import engine
        net, S = ctx.net, _saved(ctx)
        B, N, M, k = ctx.dims
        act, slope = ctx.actslope
        w2d = engine._w2d
        dfeat = dfeat.contiguous()
        # conv3 + bn3 (the incoming gradient buffer belongs to autograd: not modified in place)
        if S["y3"].dtype == torch.bfloat16:      # bf16 map (forward: map16): bf16 gradient in, bf16 dY3, two-product GEMMs on it
            dy3, dg3, db3 = ops.bn_act_bwd_bf16(dfeat, S["y3"], S["st3"], act, slope)
            dw3 = ops.gemm_tn(dy3, S["cat"] if S["cat16"] is None else S["cat16"])
            dcat = ops.gemm_bf16a(dy3, w2d(net.conv3_lpd), b_kmajor=True)
            del dy3
        else:
            dcat, dw3, dg3, db3 = _PointLayer.bwd(dfeat, S["cat"], w2d(net.conv3_lpd), S["y3"], S["st3"], act, slope,
                                                  inplace=False)
        # SN1: x3 = max_k act(BN(P[nbr] + Q)), split form (closed-form sums over the edges, one pass over the transposed graph)
        # (bf16 storage: the gradient of the SN1 projection as bf16 rows -- its two consumers take them as the hi image)
        pq16 = (ctx.bf16 and PQ3_BF16 and ops.SPLIT_BWD_BF16 and ops.GEMM_TN and ops.GEMM_BF16X3 and M % 32 == 0 and M >= 2048
                and ops._EXACT.depth == 0)
        dpq3 = torch.empty((M, 512), dtype=torch.bfloat16 if pq16 else torch.float32, device=dfeat.device)    # both halves are fully written below
        pq3 = S["pq3"]
        dgs3, dbs3 = ops.edge_split_bwd(dcat[:, 256:512], S["usel3"], S["arg3"], S["s3"], pq3[:, :256], pq3[:, 256:],
                                        ops.GraphT(S["idx_x"], N), S["stg3"], act, slope, k, dP=dpq3[:, :256], dQ=dpq3[:, 256:],
                                        half=ctx.bf16)
        x2 = S["cat"][:, 128:256]
        if pq16:
            dwcat3 = ops.gemm_tn(dpq3, x2)
            ops.gemm_bf16a(dpq3, S["wcat3"], b_kmajor=True, out=dcat[:, 128:256], accumulate=True)
        else:
            dwcat3 = _dweight(dpq3, x2)
            ops.gemm(dpq3, S["wcat3"], b_kmajor=True, out=dcat[:, 128:256], accumulate=True)   # dx2 += dPQ3 Wcat3
        dpq1 = torch.empty((M, 256), dtype=torch.float32, device=dfeat.device)
        w2 = w2d(net.convDG2[0])
        closed = (S["post1"] and ops.EDGE_MLP_TRAIN_BWD and ops.dg2_bwd_fused_applies(M, k, w2.shape[0]) and w2.shape[1] == 128
                  and w2.is_contiguous() and M % 32 == 0 and ops.GEMM_BF16X3
                  and (ops._EXACT.depth == 0 or S["z"] is None or S["z"].dtype != S["y1e"].dtype))
        if S["z"] is None and not closed:
            raise RuntimeError("the forward stored no Z (ops.EDGE_NOZ) but the fused DG2 backward is not available now: "
                               "the switches / weights changed between forward and backward")
        if closed:
            # DG2 + DG1 backward without dZ, dY1e and dU1 tensors: dW2 from one pass over Y1e (arg-max product + Gram matrix); then ONE
            # MFMA launch builds dZ from Z in its operand loader, multiplies by W2 and leaves G = the gradient in front of BatchNorm1 with its
            # reductions (lpd_edge_mlp_train_bwd); then ONE gather pass over the transposed graph gives dP / dQ in closed form
            dt = torch.bfloat16 if ctx.bf16 else torch.float32
            dpre2, red2 = ops.bn_sel_bwd_reduce(dcat[:, 128:256], S["zsel"], S["stg2"], act, slope, dtype=dt)
            dw2 = (ops.edge_dw_sel_bf16 if ctx.bf16 else ops.edge_dw_sel_f32)(S["y1e"], S["arg2"], dpre2, k, w2, S["stg2"], red2)
            redf = red2.float()
            dgs2, dbs2 = redf[1], redf[0]
            G1, gsum1, red1 = ops.edge_mlp_train_bwd(S["z"], S["arg2"], dpre2, w2, S["stg2"], red2, S["y1e"], S["arg1"], dcat[:, 0:128],
                                                     S["post1c"], k, act, slope)
            pq1 = S["pq1"]
            ops.edge_dense_bwd_apply(G1, gsum1, S["s1sum"], pq1[:, :128], pq1[:, 128:], ops.GraphT(S["idx_f"], N), S["stg1"], red1, k,
                                     dP=dpq1[:, :128], dQ=dpq1[:, 128:])
            red1f = red1.float()
            dgs1, dbs1 = red1f[1], red1f[0]
            du1 = dy1e = None
            del G1, gsum1
        elif ctx.bf16:
            # DG2: x2 = groupmax(act(BN(Z))), Z = Y1e W2^T  (bf16 edge tensors, bf16 MFMA products)
            w2 = w2d(net.convDG2[0])
            if ops.dg2_bwd_fused_applies(M, k, w2.shape[0]) and w2.shape[1] == 128:
                # no dZ tensor: dW2 from one pass over Y1e (arg-max product + Gram matrix), dY1e = dZ W2 with dZ built in the loader
                dpre16, red2 = ops.bn_sel_bwd_reduce(dcat[:, 128:256], S["zsel"], S["stg2"], act, slope)
                dw2 = ops.edge_dw_sel_bf16(S["y1e"], S["arg2"], dpre16, k, w2, S["stg2"], red2)
                dy1e = ops.gemm_bf16s_bnbwd(S["z"], S["arg2"], dpre16, k, w2, S["stg2"], red2)
                redf = red2.float()
                dgs2, dbs2 = redf[1], redf[0]
            else:
                dz, dgs2, dbs2 = ops.edge_bn_bwd_bf16(dcat[:, 128:256], S["arg2"], k, S["z"], S["stg2"], act, slope, xsel=S["zsel"])
                dw2 = ops.gemm_tn_bf16(dz, S["y1e"])                            # [Co,Ci] = dZ^T Y1e
                dy1e = ops.gemm_bf16s(dz, w2, b_kmajor=True)                    # [E,128] = dZ W2
                del dz
            # (post1: the forward kept Y1e = act(BN(U1)) instead of U1; the kernels recover the pre-activation from it)
            du1, dgs1, dbs1 = ops.edge_bn_bwd_bf16(dcat[:, 0:128], S["arg1"], k, S["y1e"] if S["post1"] else S["u1"], S["stg1"], act, slope,
                                                   dense=dy1e, dQ=dpq1[:, 128:], post_bn=S["post1c"] if S["post1"] else None)
            ops.gather_sum_rows_bf16(du1, ops.GraphT(S["idx_f"], N), dpq1[:, :128])
        else:
            # DG2: x2 = groupmax(act(BN(Z))), Z = Y1e W2^T
            w2 = w2d(net.convDG2[0])
            if (ops.dg2_bwd_fused_applies(M, k, w2.shape[0]) and w2.shape[1] == 128 and ops.GEMM_BF16X3 and ops._EXACT.depth == 0
                    and S["z"].is_contiguous() and S["y1e"].is_contiguous()):
                # no dZ tensor (as in the bf16 mode, on fp32 tensors with split-bf16 products)
                dpre2, red2 = ops.bn_sel_bwd_reduce(dcat[:, 128:256], S["zsel"], S["stg2"], act, slope, dtype=torch.float32)
                dw2 = ops.edge_dw_sel_f32(S["y1e"], S["arg2"], dpre2, k, w2, S["stg2"], red2)
                dy1e = ops.gemm_f32s_bnbwd(S["z"], S["arg2"], dpre2, k, w2, S["stg2"], red2)
                redf = red2.float()
                dgs2, dbs2 = redf[1], redf[0]
            else:
                dz, dgs2, dbs2 = ops.edge_bn_bwd(dcat[:, 128:256], S["arg2"], k, S["z"], S["stg2"], act, slope, xsel=S["zsel"])
                dw2 = _dweight(dz, S["y1e"])
                dy1e = ops.gemm(dz, w2, b_kmajor=True)                          # [E,128]
                del dz
            # DG1: y1e = act(BN(U1)); consumers: DG2 (dense) and x1 = groupmax (sparse)
            du1, dgs1, dbs1 = ops.edge_bn_bwd(dcat[:, 0:128], S["arg1"], k, S["y1e"] if S["post1"] else S["u1"], S["stg1"], act, slope,
                                              dense=dy1e, dQ=dpq1[:, 128:], post_bn=S["post1c"] if S["post1"] else None)
            ops.gather_sum_rows(du1, ops.GraphT(S["idx_f"], N), dpq1[:, :128])
        del du1, dy1e
        dwcat1 = _dweight(dpq1, S["f0"])
        df0 = ops.gemm(dpq1, S["wcat1"], b_kmajor=True)                         # [M,64]
        if engine.DEBUG_AUX is not None:
            engine.DEBUG_AUX.update(dcat=dcat.clone(), dpq3=dpq3.clone(), dpq1=dpq1.clone(), df0=df0.clone())
        (dwc1, dg1, db1, dwc2, dg2, db2), extra = _Front.bwd(net, df0, S["front"], w2d(net.conv1_lpd), w2d(net.conv2_lpd), B, N,
                                                             act, slope)
        grads = (dwc1.reshape(net.conv1_lpd.weight.shape), dg1, db1, dwc2.reshape(net.conv2_lpd.weight.shape), dg2, db2,
                 _unsplit_cat_nc(dwcat1, net.convDG1[0].weight), dgs1, dbs1, dw2.reshape(net.convDG2[0].weight.shape), dgs2,
                 dbs2, _unsplit_cat_nc(dwcat3, net.convSN1[0].weight), dgs3, dbs3, dw3.reshape(net.conv3_lpd.weight.shape),
                 dg3, db3)
        ctx.saved = None
        return (None, None) + grads + tuple(extra)


def _unsplit_cat_cd(dwcat, conv_weight):
    """DGCNN-style input cat(centre, neighbour - centre): W = [Wa | Wb], P = Wb f_j, Q = (Wa - Wb) f_i
    => dWb = dP-weight - dQ-weight ... in terms of the stacked gradient [dWn ; dWc]: dWa = dWc, dWb = dWn - dWc."""
    co = conv_weight.shape[0]
    dwn, dwc = dwcat[:co], dwcat[co:]
    return torch.cat((dwc, dwn - dwc), dim=1).reshape(conv_weight.shape)


class _EdgeChain:
    """Two chained edge convolutions followed by max over k (LPDNetOrign: convDG1->convDG2->max, lpdnet_model.py:97-100,
    and convSN1->convSN2->max, :105-107), train mode, on materialised edge tensors."""

    @staticmethod
    def fwd(pq, c, has_q, idx, N, k, bn_a, w_b, bn_b, act, slope, out):
        u, st_a = ops.edge_build(pq[:, :c], pq[:, c:] if has_q else None, idx, N, bn=bn_a)   # [E,c] raw + its BN statistics
        ya = ops.affine_act(u, st_a.scale, st_a.shift, act, slope)               # [E,c]
        z = ops.linear(ya, w_b)                                                   # [E,co] raw
        st_b = ops.bn_train_stats(z, bn_b)
        arg, zsel = ops.group_max(z, k, st_b.scale, st_b.shift, act, slope, out, keep_sel=True)
        return dict(u=u, st_a=st_a, ya=ya, z=z, zsel=zsel, st_b=st_b, arg=arg)

    @staticmethod
    def bwd(dout, S, w_b, idx, N, k, c, has_q, act, slope):
        """returns (dPQ [M, c or 2c], dW_b, dgamma_b, dbeta_b, dgamma_a, dbeta_a)"""
        M = dout.shape[0]
        dz, dg_b, db_b = ops.edge_bn_bwd(dout, S["arg"], k, S["z"], S["st_b"], act, slope, xsel=S["zsel"])
        dw_b = _dweight(dz, S["ya"])
        dya = ops.gemm(dz, w_b, b_kmajor=True)
        del dz
        du, dg_a, db_a = ops.bn_act_bwd(dya, S["u"], S["st_a"], act, slope, out=dya)
        dpq = torch.empty((M, 2 * c if has_q else c), dtype=torch.float32, device=dout.device)   # fully written below
        if has_q:
            ops.group_sum(du, k, dpq[:, c:])
        ops.gather_sum_rows(du, ops.GraphT(idx, N), dpq[:, :c])
        return dpq, dw_b, dg_b, db_b, dg_a, db_a


class _LPDNetOrignTrainFn(torch.autograd.Function):
    """LPDNetOrign.forward in train mode (util/lpdnet_model.py:68-114, t3d = tfea = False)."""

    PARAMS = tuple(f"{blk}.{leaf}" for blk in ("conv1_lpd", "conv2_lpd", "convDG1", "convDG2", "convSN1", "convSN2",
                                               "conv3_lpd", "conv4_lpd", "conv5_lpd")
                   for leaf in ("0.weight", "1.weight", "1.bias"))

    @staticmethod
    @_hooked_forward
    def forward(ctx, net, x, *params):
        with ops.train_forward_gemm(x.shape[0]):
            return _LPDNetOrignTrainFn._forward(ctx, net, x, *params)

    @staticmethod
    def _forward(ctx, net, x, *params):
        from . import engine
        B, N = x.shape[0], x.shape[2]
        M, k = B * N, net.k
        act, slope = (ops.ACT_RELU, 0.0) if net.use_relu else (ops.ACT_LEAKY, LEAKY)
        w2d = engine._w2d
        p_all = None
        if x.shape[3] == 8:       # use_mFea
            xyz, p_all = engine.split_mfea(x)
            x = xyz.view(B, 1, N, 3)
        else:
            xyz = x.view(M, 3)
        S = {}
        f0, S["front"] = _Front.fwd(net, xyz, w2d(net.conv1_lpd[0]), net.conv1_lpd[1], w2d(net.conv2_lpd[0]), net.conv2_lpd[1],
                                    B, N, act, slope, p_all)
        S["f0"] = f0
        idx_f = engine._knn_rows(f0, B, N, 64, k)
        wcat1 = engine.split_edge_weight(net.convDG1, "cat_cd")
        pq1 = ops.linear(f0, wcat1)                                               # [M,128] = [P | Q]
        g = torch.empty((M, 64), dtype=torch.float32, device=x.device)
        S["dg"] = _EdgeChain.fwd(pq1, 64, True, idx_f, N, k, net.convDG1[1], w2d(net.convDG2[0]), net.convDG2[1], act, slope, g)
        idx_x = engine._knn_rows(x.view(B * N, 3), B, N, 3, k)
        wsn1 = engine.split_edge_weight(net.convSN1, "nbr")
        pn = ops.linear(g, wsn1)                                                  # [M,64] neighbours only
        h = torch.empty((M, 64), dtype=torch.float32, device=x.device)
        S["sn"] = _EdgeChain.fwd(pn, 64, False, idx_x, N, k, net.convSN1[1], w2d(net.convSN2[0]), net.convSN2[1], act, slope, h)
        S["y3"], S["st3"], h3 = _PointLayer.fwd(h, w2d(net.conv3_lpd[0]), net.conv3_lpd[1], act, slope)
        S["y4"], S["st4"], h4 = _PointLayer.fwd(h3, w2d(net.conv4_lpd[0]), net.conv4_lpd[1], act, slope)
        S["y5"], S["st5"], feat = _PointLayer.fwd(h4, w2d(net.conv5_lpd[0]), net.conv5_lpd[1], act, slope)
        S.update(g=g, h=h, h3=h3, h4=h4, idx_f=idx_f, idx_x=idx_x, wcat1=wcat1, wsn1=wsn1)
        ctx.net, ctx.dims, ctx.actslope, ctx.saved = net, (B, N, M, k), (act, slope), S
        if engine.DEBUG_AUX is not None:
            engine.DEBUG_AUX.update(F0=f0, idx_feat=idx_f, idx_xyz=idx_x, argsel=dict(dg=S["dg"]["arg"], sn=S["sn"]["arg"]))
        return feat

    @staticmethod
    @_hooked_backward
    def backward(ctx, dfeat):
        from . import engine
        net, S = ctx.net, _saved(ctx)
        B, N, M, k = ctx.dims
        act, slope = ctx.actslope
        w2d = engine._w2d
        dfeat = dfeat.contiguous()
        dh4, dw5, dg5, db5 = _PointLayer.bwd(dfeat, S["h4"], w2d(net.conv5_lpd[0]), S["y5"], S["st5"], act, slope, inplace=False)
        dh3, dw4, dg4, db4 = _PointLayer.bwd(dh4, S["h3"], w2d(net.conv4_lpd[0]), S["y4"], S["st4"], act, slope)
        dh, dw3, dg3, db3 = _PointLayer.bwd(dh3, S["h"], w2d(net.conv3_lpd[0]), S["y3"], S["st3"], act, slope)
        # SN1 -> SN2 -> max on the xyz graph (neighbours only)
        dpn, dwsn2, dgsn2, dbsn2, dgsn1, dbsn1 = _EdgeChain.bwd(dh, S["sn"], w2d(net.convSN2[0]), S["idx_x"], N, k, 64, False, act, slope)
        dwsn1 = _dweight(dpn, S["g"])
        dg_ = ops.gemm(dpn, S["wsn1"], b_kmajor=True)                             # [M,64]
        # DG1 -> DG2 -> max on the feature graph
        dpq1, dwdg2, dgdg2, dbdg2, dgdg1, dbdg1 = _EdgeChain.bwd(dg_, S["dg"], w2d(net.convDG2[0]), S["idx_f"], N, k, 64, True, act, slope)
        dwcat1 = _dweight(dpq1, S["f0"])
        df0 = ops.gemm(dpq1, S["wcat1"], b_kmajor=True)
        (dwc1, dg1, db1, dwc2, dg2, db2), extra = _Front.bwd(net, df0, S["front"], w2d(net.conv1_lpd[0]), w2d(net.conv2_lpd[0]),
                                                             B, N, act, slope)

        def shp(dw, seq):
            return dw.reshape(seq[0].weight.shape)
        grads = (shp(dwc1, net.conv1_lpd), dg1, db1, shp(dwc2, net.conv2_lpd), dg2, db2,
                 _unsplit_cat_cd(dwcat1, net.convDG1[0].weight), dgdg1, dbdg1, shp(dwdg2, net.convDG2), dgdg2, dbdg2,
                 shp(dwsn1, net.convSN1), dgsn1, dbsn1, shp(dwsn2, net.convSN2), dgsn2, dbsn2,
                 shp(dw3, net.conv3_lpd), dg3, db3, shp(dw4, net.conv4_lpd), dg4, db4, shp(dw5, net.conv5_lpd), dg5, db5)
        ctx.saved = None
        return (None, None) + grads + tuple(extra)


def lpdnet_origin_features_train(net, x, reorder=True):
    """LPDNetOrign training-mode forward: ([B*N, E] features with autograd, B, N)."""
    from . import engine
    x = engine._check_input(x, 8) if net.use_mFea else engine.reorder_points(engine._check_input(x), reorder)
    params = _named(net, list(_LPDNetOrignTrainFn.PARAMS) + _Front.tnet_param_names(net))
    feat = _LPDNetOrignTrainFn.apply(net, x, *params)
    return feat, x.shape[0], x.shape[2]


def _named(module, names):
    out = []
    for n in names:
        obj = module
        for part in n.split("."):
            obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
        out.append(obj)
    return out


def lpdnet_features_train(net, x, reorder=True, defer_act=False):
    """LPDNet training-mode forward: ([B*N, E] features with autograd, B, N).  defer_act (PointNetVlad's train path only): where the
    fused form is built the result is the RAW conv3 output and a 4th value (scale, shift, act, slope) -- bn3's affine and the
    activation, which netvlad_train(pending=...) applies inside its assignment product; None when nothing is pending."""
    from . import engine
    x = engine._check_input(x, 8) if net.use_mFea else engine.reorder_points(engine._check_input(x), reorder)
    names = list(_LPDNetTrainFn.PARAMS) + _Front.tnet_param_names(net)
    _LAST.defer_act, _LAST.pending = bool(defer_act), None
    try:
        feat = _LPDNetTrainFn.apply(net, x, *_named(net, names))
    finally:
        _LAST.defer_act = False
    pending, _LAST.pending = _LAST.pending, None
    if defer_act:
        return feat, x.shape[0], x.shape[2], pending
    return feat, x.shape[0], x.shape[2]


class _PointNetTrainFn(torch.autograd.Function):
    """PointNetfeat.forward in train mode, max_pool=False (util/PointNetVlad.py:204-233): STN3d(k=3, no BN) alignment,
    conv1..conv5 with bias + BatchNorm2d (ReLU after bn1..bn4, none after bn5), optional 64x64 feature transform."""

    LAYERS = (("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3"), ("conv4", "bn4"), ("conv5", "bn5"))

    @staticmethod
    def param_names(net):
        names = ["stn." + n for n in _TNet.param_names(net.stn, net.stn.use_bn)]
        for c, b in _PointNetTrainFn.LAYERS:
            names += [f"{c}.weight", f"{c}.bias", f"{b}.weight", f"{b}.bias"]
        if net.apply_feature_trans:
            names += ["feature_trans." + n for n in _TNet.param_names(net.feature_trans, net.feature_trans.use_bn)]
        return names

    @staticmethod
    @_hooked_forward
    def forward(ctx, net, x, *params):
        with ops.train_forward_gemm(x.shape[0]):
            return _PointNetTrainFn._forward(ctx, net, x, *params)

    @staticmethod
    def _forward(ctx, net, x, *params):
        B, N = x.shape[0], x.shape[2]
        xyz = x.view(B * N, 3)
        S = dict(xyz=xyz)
        S["trans"], S["S3"] = _TNet.fwd(net.stn, xyz, B, N, net.stn.use_bn)
        S["p"] = ops.apply_transform(xyz, S["trans"], N)
        S["r1"], S["s1"], S["a1"] = _Dense.fwd(S["p"], net.conv1, net.bn1, ops.ACT_RELU)
        S["r2"], S["s2"], S["a2"] = _Dense.fwd(S["a1"], net.conv2, net.bn2, ops.ACT_RELU)
        S["h"] = S["a2"]
        if net.apply_feature_trans:
            S["transf"], S["Sf"] = _TNet.fwd(net.feature_trans, S["a2"], B, N, net.feature_trans.use_bn)
            S["h"] = ops.apply_transform(S["a2"], S["transf"], N)
        S["r3"], S["s3"], S["a3"] = _Dense.fwd(S["h"], net.conv3, net.bn3, ops.ACT_RELU)
        S["r4"], S["s4"], S["a4"] = _Dense.fwd(S["a3"], net.conv4, net.bn4, ops.ACT_RELU)
        S["r5"], S["s5"], out = _Dense.fwd(S["a4"], net.conv5, net.bn5, ops.ACT_NONE)
        ctx.net, ctx.saved, ctx.dims = net, S, (B, N)
        _LAST.trans = S["trans"]
        return out

    @staticmethod
    @_hooked_backward
    def backward(ctx, dfeat):
        net, S = ctx.net, _saved(ctx)
        B, N = ctx.dims
        R = ops.ACT_RELU
        dfeat = dfeat.contiguous()
        da4, dw5, dc5, dg5, db5 = _Dense.bwd(dfeat, S["a4"], net.conv5, S["r5"], S["s5"], ops.ACT_NONE, inplace=False)
        da3, dw4, dc4, dg4, db4 = _Dense.bwd(da4, S["a3"], net.conv4, S["r4"], S["s4"], R)
        dh, dw3, dc3, dg3, db3 = _Dense.bwd(da3, S["h"], net.conv3, S["r3"], S["s3"], R)
        g_f = []
        if net.apply_feature_trans:
            da2, dtf = _transform_bwd(S["a2"], S["transf"], dh, B, N)
            da2, g_f = _TNet.bwd(net.feature_trans, dtf, S["Sf"], B, N, net.feature_trans.use_bn, need_dh=True, dh_accum=da2)
        else:
            da2 = dh
        da1, dw2, dc2, dg2, db2 = _Dense.bwd(da2, S["a1"], net.conv2, S["r2"], S["s2"], R)
        dp, dw1, dc1, dg1, db1 = _Dense.bwd(da1, S["p"], net.conv1, S["r1"], S["s1"], R)
        dt3 = ops.cloud_outer(S["xyz"], dp, B, N)
        _, g_3 = _TNet.bwd(net.stn, dt3, S["S3"], B, N, net.stn.use_bn, need_dh=False)
        grads = list(g_3) + [dw1, dc1, dg1, db1, dw2, dc2, dg2, db2, dw3, dc3, dg3, db3, dw4, dc4, dg4, db4,
                             dw5, dc5, dg5, db5] + list(g_f)
        ctx.saved = None
        return (None, None) + tuple(grads)


def pointnet_features_train(net, x):
    """PointNetfeat (max_pool=False) train-mode trunk: -> (features [B*N, emb_dims] point-major, B, N)."""
    from . import engine
    x = engine._check_input(x)
    B, N = x.shape[0], x.shape[2]
    if N != net.num_points:
        raise ValueError(f"PointNetfeat was built for num_points={net.num_points}, got N={N} (MaxPool2d((num_points,1)))")
    feat = _PointNetTrainFn.apply(net, x, *_named(net, _PointNetTrainFn.param_names(net)))
    return feat, B, N


class _ToPointMajor(torch.autograd.Function):
    """[B, E, N] channel-major -> [B*N, E] point-major rows with a gradient (the standalone NetVLADLoupe.forward in train mode;
    inside PointNetVlad the trunk hands its rows over directly)."""

    @staticmethod
    @_hooked_forward
    def forward(ctx, x3):
        ctx.shape = x3.shape
        return ops.transpose(x3).view(x3.shape[0] * x3.shape[2], x3.shape[1])

    @staticmethod
    @_hooked_backward
    def backward(ctx, d):
        B, E, N = ctx.shape
        return ops.transpose(d.contiguous().view(B, N, E))


class _ToChannelMajor(torch.autograd.Function):
    """[B*N, E] point-major rows -> the reference's [B, E, N, 1] with a gradient (the public trunk forwards in train mode)"""

    @staticmethod
    @_hooked_forward
    def forward(ctx, feat, B, N):
        ctx.dims = (B, N, feat.shape[1])
        return ops.transpose(feat.view(B, N, feat.shape[1])).unsqueeze(-1)

    @staticmethod
    @_hooked_backward
    def backward(ctx, d):
        B, N, E = ctx.dims
        return ops.transpose(d.reshape(B, E, N).contiguous()).view(B * N, E), None, None


def to_channel_major_train(feat, B, N):
    return _ToChannelMajor.apply(feat, B, N)


def to_point_major_train(x4):
    B, E, N = x4.shape[0], x4.shape[1], x4.shape[2]
    return _ToPointMajor.apply(x4.reshape(B, E, N).float().contiguous()), B, N


class _Last(__import__("threading").local):
    trans = None      # the input alignment matrix of the latest PointNet train-mode forward on this thread
    defer_act = False # lpdnet_features_train(defer_act=True) around its Function call
    pending = None    # ... and what that forward left for the head: (scale, shift, act, slope) of bn3 + activation


_LAST = _Last()


class _CloudMaxFn(torch.autograd.Function):
    """per-cloud max over the N points of point-major rows [B*N, C] -> [B, C] (MaxPool2d((num_points,1)), PointNetVlad.py:235)"""

    @staticmethod
    @_hooked_forward
    def forward(ctx, feat, B, N):
        out, arg = ops.colmax_arg(feat, B, N)
        ctx.arg, ctx.N = arg, N
        return out

    @staticmethod
    @_hooked_backward
    def backward(ctx, dout):
        return ops.colmax_bwd(dout.contiguous(), ctx.arg, ctx.N), None, None


def pointnet_global_train(net, x):
    """PointNetfeat(max_pool=True, global_feat=True) in train mode (PointNetVlad.py:204-239): (global feature [B, emb_dims] with
    autograd, trans [B,3,3]).  The alignment matrix is returned as a constant (no gradient flows from a caller's use of it;
    the reference returns it attached -- nothing on the PointNetVLAD path consumes it)."""
    feat, B, N = pointnet_features_train(net, x)
    trans = _LAST.trans.detach()
    _LAST.trans = None
    return _CloudMaxFn.apply(feat, B, N), trans


class _NetVLADTrainFn(torch.autograd.Function):
    """NetVLADLoupe.forward + GatingContext in train mode (util/PointNetVlad.py:45-83,103-115): with BatchNorm + gating (what
    PointNetVlad constructs) and the other constructor variants -- add_batch_norm=False (cluster_biases / gating_biases in place
    of the two BatchNorms, :33-36,55-56,94-96,108-109) and gating=False (:80-81)."""

    @staticmethod
    def param_names(vlad):
        names = ["cluster_weights", "cluster_weights2", "hidden1_weights"]
        names += ["bn1.weight", "bn1.bias"] if vlad.add_batch_norm else ["cluster_biases"]
        names += ["bn2.weight", "bn2.bias"]
        if vlad.gating:
            names += ["context_gating.gating_weights"]
            names += ["context_gating.bn1.weight", "context_gating.bn1.bias"] if vlad.context_gating.add_batch_norm else \
                ["context_gating.gating_biases"]
        return names

    @staticmethod
    @_hooked_forward
    def forward(ctx, vlad, B, N, pending, feat, *params):
        ctx.bf16 = TRAIN_STORAGE == "bf16"
        if ctx.bf16:      # the big per-point products (assignment, pooling) take bf16 operands; the B-row head products stay
            with ops.bf16_gemm():      # on the exact / split path by their shapes (ops.gemm's policy: skinny outputs)
                return _NetVLADTrainFn._forward(ctx, vlad, B, N, pending, feat, *params)
        with ops.train_forward_gemm(B):
            return _NetVLADTrainFn._forward(ctx, vlad, B, N, pending, feat, *params)

    @staticmethod
    def _forward(ctx, vlad, B, N, pending, feat, *params):
        from . import engine
        E, K, O = vlad.feature_size, vlad.cluster_size, vlad.output_dim
        M = B * N
        dev = feat.device
        Bp = (B + 31) // 32 * 32                                    # rows padded so K = Bp weight-gradient GEMMs are legal
        aff = None
        if pending is not None:      # `feat` is the trunk's RAW conv3 output: bn3 affine + activation in the assignment's operand loader
            raw, raw16 = feat.detach(), feat.dtype == torch.bfloat16
            if ops.feat_in_loader_applies(B, N, E, K):
                # ... and in the loaders of the pooling, dA and assignment-weight-gradient products: the activated map is never stored
                _, a0 = ops.gemm_act(raw, vlad.cluster_weights, *pending, out_bf16=raw16, store=False)
                feat, aff = raw, tuple(pending)
                if engine.DEBUG_AUX is not None:      # test hook: the trunk's ACTIVATED output rows
                    act_rows = ops.affine_act(raw.float() if raw16 else raw, *pending)
                    engine.DEBUG_AUX["feat"] = act_rows.to(torch.bfloat16) if raw16 else act_rows
            else:
                feat, a0 = ops.gemm_act(raw, vlad.cluster_weights, *pending, out_bf16=raw16)
                if engine.DEBUG_AUX is not None:
                    engine.DEBUG_AUX["feat"] = feat
        else:
            a0 = ops.gemm(feat, vlad.cluster_weights, b_kmajor=True)    # [M,K] raw
        if vlad.add_batch_norm:
            sta = ops.bn_train_stats(a0, vlad.bn1)
            a = ops.softmax_affine(a0, sta.scale, sta.shift)
        else:
            sta = None
            a = ops.softmax_affine(a0, torch.ones_like(vlad.cluster_biases), vlad.cluster_biases)
        vraw = ops.gemm_tn(feat.view(B, N, E), a.view(B, N, K), a_affine=aff) if aff is not None else ops.pool_tn(feat.view(B, N, E), a.view(B, N, K))
        if vraw is None:
            vraw = ops.gemm(feat.view(B, N, E), a.view(B, N, K), a_kmajor=True, b_kmajor=True, splits=engine._pool_splits(B, N, E))
        aux = {}
        v = torch.zeros((Bp, E * K), dtype=torch.float32, device=dev)
        ops.vlad_finalize(vraw, a.view(B, N, K), vlad.cluster_weights2.view(E, K), out=v, aux=aux)
        h0 = ops.gemm(v, vlad.hidden1_weights, b_kmajor=True, splits=engine._head_splits(E * K))     # [Bp,O] raw
        sth = ops.bn_train_stats(h0, vlad.bn2, rows=B)
        h = torch.zeros((Bp, O), dtype=torch.float32, device=dev)
        ops.affine_act(h0, sth.scale, sth.shift, out=h, rows=B)
        ctx.vlad, ctx.dims = vlad, (B, N, M, E, K, O, Bp)
        S = dict(feat=feat, aff=aff, a0=a0, sta=sta, a=a, aux=aux, v=v, h0=h0, sth=sth, h=h)
        ctx.saved = S
        if not vlad.gating:
            return h[:B].clone()
        gc = vlad.context_gating
        g0 = ops.gemm(h, gc.gating_weights, b_kmajor=True)           # [Bp,O] raw
        if gc.add_batch_norm:
            stg = ops.bn_train_stats(g0, gc.bn1, rows=B)
            gates = ops.affine_act(g0, stg.scale, stg.shift, ops.ACT_SIGMOID, rows=B)
        else:
            stg = None
            g0 = ops.affine_act(g0, torch.ones_like(gc.gating_biases), gc.gating_biases, ops.ACT_NONE, out=g0, rows=B)   # pre-activation
            gates = ops.affine_act(g0, None, None, ops.ACT_SIGMOID, rows=B)
        S.update(g0=g0, stg=stg, gates=gates)
        return ops.mul(h[:B], gates[:B])

    @staticmethod
    @_hooked_backward
    def backward(ctx, dout):
        if ctx.bf16:
            with ops.bf16_gemm():
                return _NetVLADTrainFn._backward(ctx, dout)
        return _NetVLADTrainFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        vlad, S = ctx.vlad, _saved(ctx)
        B, N, M, E, K, O, Bp = ctx.dims
        dev = dout.device
        dout = dout.contiguous()
        h = S["h"]
        dh = torch.zeros((Bp, O), dtype=torch.float32, device=dev)
        g_gate = []
        if vlad.gating:
            gc, gates = vlad.context_gating, S["gates"]
            # out = h * gates
            dgates = torch.zeros((Bp, O), dtype=torch.float32, device=dev)
            dgates[:B] = ops.mul(dout, h[:B])
            dh[:B] = ops.mul(dout, gates[:B])
            # gates = sigmoid(BN(g0)) or sigmoid(g0 + bias), g0 = h Wg
            dg0, dgam_g, dbet_g = ops.bn_act_bwd(dgates, S["g0"], S["stg"], ops.ACT_SIGMOID, out=dgates, rows=B)
            dwg = ops.gemm(h, dg0, a_kmajor=True, b_kmajor=True)                       # h^T dG0  [O,O]  (K = Bp, zero rows pad)
            ops.gemm(dg0, gc.gating_weights, b_kmajor=False, out=dh, accumulate=True)  # dh += dG0 Wg^T
            g_gate = [dwg, dgam_g, dbet_g] if gc.add_batch_norm else [dwg, dbet_g]     # no BN: the sum of dpre is the bias gradient
        else:
            dh[:B] = dout
        # h = BN(h0), h0 = v Wh
        dh0, dgam_h, dbet_h = ops.bn_act_bwd(dh, S["h0"], S["sth"], ops.ACT_NONE, out=dh, rows=B)
        if Bp > B:
            dh0[B:].zero_()
        dwh = ops.gemm(S["v"], dh0, a_kmajor=True, b_kmajor=True)                  # v^T dH0  [E*K, O]
        dv = ops.gemm(dh0, vlad.hidden1_weights, b_kmajor=False)                   # dH0 Wh^T [Bp, E*K]
        cw2 = vlad.cluster_weights2.view(E, K)
        dvraw, dasum, dcw2 = ops.vlad_finalize_bwd(dv, S["v"], S["aux"], cw2, B, E, K)
        del dv
        feat, a, aff = S["feat"], S["a"], S["aff"]      # (aff: `feat` holds the RAW map, the products transform it in their loaders)
        # vraw[b] = feat[b]^T a[b]
        da = ops.gemm(feat.view(B, N, E), dvraw, a_kmajor=False, b_kmajor=True, a_affine=aff)    # [B,N,K]
        ds = ops.softmax_bwd(a, da.view(M, K), dasum, N)
        del da
        # dfeat = a dVraw^T (pooling) + dA0 Wc^T (assignment): ONE batched product with the operands side by side,
        #   [a | dA0] [M, 2K]  x  [dVraw_b | Wc] [E, 2K] per cloud,
        # instead of a K-deep product that writes the 738 MB gradient and a second one that reads it back and accumulates
        # (0.33 + 1.07 ms at B = 44)
        ada = torch.empty((M, 2 * K), dtype=torch.float32, device=dev)
        ops.affine_act(a, None, None, ops.ACT_NONE, out=ada[:, :K])
        if vlad.add_batch_norm:
            da0, dgam_a, dbet_a = ops.bn_act_bwd(ds, S["a0"], S["sta"], ops.ACT_NONE, out=ada[:, K:])
            g_assign = [dgam_a, dbet_a]
        else:
            da0 = ops.affine_act(ds, None, None, ops.ACT_NONE, out=ada[:, K:])         # a = softmax(a0 + cluster_biases)
            g_assign = [ops.colsum(ds)]
        dwc = ops.gemm_tn(feat, da0, a_affine=aff) if aff is not None else _dweight(feat, da0)      # feat^T dA0 [E,K]
        rhs = torch.empty((B, E, 2 * K), dtype=torch.float32, device=dev)
        ops.affine_act(dvraw.view(B * E, K), None, None, ops.ACT_NONE, out=rhs.view(B * E, 2 * K)[:, :K])
        rhs[:, :, K:] = vlad.cluster_weights.detach()                              # parameter-sized broadcast (plumbing)
        dfeat = ops.gemm(ada.view(B, N, 2 * K), rhs, a_kmajor=False, b_kmajor=False, out_bf16=feat.dtype == torch.bfloat16).view(M, E)
        ctx.saved = None
        return (None, None, None, None, dfeat, dwc, dcw2.view(1, E, K), dwh) + tuple(g_assign) + (dgam_h, dbet_h) + tuple(g_gate)


def netvlad_train(vlad, feat, B, N, pending=None):
    """pending: (scale, shift, act, slope) -- `feat` is then the trunk's RAW last-layer output and that BatchNorm affine + activation
    is applied inside the assignment product (lpdnet_features_train(defer_act=True)); the gradient this Function returns for `feat`
    is the gradient w.r.t. the ACTIVATED features either way, which is what the trunk's backward expects."""
    if N != vlad.max_samples:
        raise ValueError(f"NetVLADLoupe was built for max_samples={vlad.max_samples}, got N={N}")
    if pending is not None and (vlad.cluster_size != 64 or not ops.gemm_act_applies(B * N, vlad.cluster_size, vlad.feature_size)):
        raise ValueError("netvlad_train: a pending activation needs the fused assignment product (64 clusters)")
    params = _named(vlad, _NetVLADTrainFn.param_names(vlad))
    return _NetVLADTrainFn.apply(vlad, B, N, pending, feat, *params)


# ------------------------------------------------------------------------------------------------
# standalone train-mode forwards of the sub-modules the trunks normally drive from the inside
# (TranformNet lpdnet_model.py:295-313, STN3d PointNetVlad.py:152-179, GatingContext PointNetVlad.py:103-115)
# ------------------------------------------------------------------------------------------------
class _TNetTrainFn(torch.autograd.Function):
    """rows [B*N, kd] point-major -> [B, kd, kd]; the layers, BatchNorm statistics and backward of _TNet"""

    @staticmethod
    @_hooked_forward
    def forward(ctx, net, use_bn, B, N, rows, *params):
        t, S = _TNet.fwd(net, rows, B, N, use_bn)
        ctx.net, ctx.use_bn, ctx.dims, ctx.saved = net, use_bn, (B, N), S
        ctx.need_dh = rows.requires_grad
        return t

    @staticmethod
    @_hooked_backward
    def backward(ctx, dt):
        B, N = ctx.dims
        dh, grads = _TNet.bwd(ctx.net, dt.contiguous(), _saved(ctx), B, N, ctx.use_bn, need_dh=ctx.need_dh)
        ctx.saved = None
        return (None, None, None, None, dh) + tuple(grads)


def tnet_train(net, rows, B, N, use_bn=True):
    params = _named(net, _TNet.param_names(net, use_bn))
    return _TNetTrainFn.apply(net, use_bn, B, N, rows, *params)


class _GatingTrainFn(torch.autograd.Function):
    """out = x * sigmoid(BN_train(x Wg)) (or sigmoid(x Wg + bias)); rows padded to a multiple of 32 like the head's"""

    @staticmethod
    @_hooked_forward
    def forward(ctx, gc, x, *params):
        B, O = x.shape
        Bp = (B + 31) // 32 * 32
        h = torch.zeros((Bp, O), dtype=torch.float32, device=x.device)
        h[:B] = x
        with ops.exact_gemm():
            g0 = ops.gemm(h, gc.gating_weights, b_kmajor=True)
        if gc.add_batch_norm:
            stg = ops.bn_train_stats(g0, gc.bn1, rows=B)
            gates = ops.affine_act(g0, stg.scale, stg.shift, ops.ACT_SIGMOID, rows=B)
        else:
            stg = None
            g0 = ops.affine_act(g0, torch.ones_like(gc.gating_biases), gc.gating_biases, ops.ACT_NONE, out=g0, rows=B)
            gates = ops.affine_act(g0, None, None, ops.ACT_SIGMOID, rows=B)
        ctx.gc, ctx.dims, ctx.saved = gc, (B, O, Bp), dict(h=h, g0=g0, stg=stg, gates=gates)
        return ops.mul(h[:B], gates[:B])

    @staticmethod
    @_hooked_backward
    def backward(ctx, dout):
        gc, S = ctx.gc, _saved(ctx)
        B, O, Bp = ctx.dims
        dout = dout.contiguous()
        h, gates = S["h"], S["gates"]
        dgates = torch.zeros((Bp, O), dtype=torch.float32, device=dout.device)
        dh = torch.zeros((Bp, O), dtype=torch.float32, device=dout.device)
        dgates[:B] = ops.mul(dout, h[:B])
        dh[:B] = ops.mul(dout, gates[:B])
        dg0, dgam, dbet = ops.bn_act_bwd(dgates, S["g0"], S["stg"], ops.ACT_SIGMOID, out=dgates, rows=B)
        with ops.exact_gemm():
            dwg = ops.gemm(h, dg0, a_kmajor=True, b_kmajor=True)
            ops.gemm(dg0, gc.gating_weights, b_kmajor=False, out=dh, accumulate=True)
        ctx.saved = None
        tail = (dwg, dgam, dbet) if gc.add_batch_norm else (dwg, dbet)
        return (None, dh[:B].clone()) + tail


def gating_train(gc, x):
    names = ["gating_weights"] + (["bn1.weight", "bn1.bias"] if gc.add_batch_norm else ["gating_biases"])
    return _GatingTrainFn.apply(gc, x.float().contiguous(), *_named(gc, names))

