"""Training path: torch.autograd.Function wrappers whose forward AND backward run on the HIP kernels.

torch supplies the tape, the parameter/gradient tensors and the optimizer; all arithmetic on
activations happens in liblpd_hip.so.
"""
import ctypes

import torch

from . import _lib, ops
from .ops import _ptr, _req, _stream


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
    def backward(ctx, g):
        gq, gpos, gneg, gother = ctx.saved_tensors
        return (g * gq, g * gpos, g * gneg, (g * gother) if ctx.quad else None, None, None, None, None, None, None)


def metric_loss(q, pos, neg, other, m1, m2, use_min, lazy, ignore_zero, quad):
    return _MetricLoss.apply(q, pos, neg, other, m1, m2, use_min, lazy, ignore_zero, quad)


def best_pos_distance(query, pos_vecs):
    """min / max squared distance to the positives (no autograd: the losses above carry their own)."""
    q, pos = _rows3(query, "query"), _rows3(pos_vecs, "pos_vecs")
    bq, P, D = pos.shape
    dev = q.device
    scratch = torch.empty((), dtype=torch.float32, device=dev)
    minmax = torch.empty((2, bq), dtype=torch.float32, device=dev)
    gq = torch.empty((bq, 1, D), dtype=torch.float32, device=dev)
    gpos = torch.empty((bq, P, D), dtype=torch.float32, device=dev)
    lib = _lib.load()
    # reuse the fused kernel with the positives standing in as negatives (triplet form)
    _lib.check(lib.lpd_metric_loss(_ptr(q), q.stride(0), _ptr(pos), pos.stride(0), pos.stride(1), _ptr(pos), pos.stride(0),
                                   pos.stride(1), None, 0, bq, P, P, D, 0.0, 0.0, 0, 0, 0, 0, _ptr(scratch), _ptr(minmax),
                                   _ptr(gq), _ptr(gpos), _ptr(torch.empty_like(gpos)), None, _stream()), "lpd_metric_loss")
    return minmax[0], minmax[1]


# ------------------------------------------------------------------------------------------------
# model training forward/backward (filled in below)
# ------------------------------------------------------------------------------------------------
def lpdnet_features_train(net, x):
    raise NotImplementedError("LPDNet training-mode forward is not built yet; call .eval() for inference")


def pointnet_features_train(net, x):
    raise NotImplementedError("PointNetfeat training-mode forward is not built yet; call .eval() for inference")


def netvlad_train(vlad, feat, B, N):
    raise NotImplementedError("NetVLADLoupe training-mode forward is not built yet; call .eval() for inference")
