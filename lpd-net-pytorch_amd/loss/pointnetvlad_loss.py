"""Lazy triplet / quadruplet losses behind the reference's function API
(loss/pointnetvlad_loss.py:6-97 of qiaozhijian/LPD-Net-Pytorch).

Same names, argument order, defaults and return type (0-dim tensor with autograd).  On GPU tensors
the whole loss (squared distances, best positive, hinges, lazy max / sum, mean or hard-count
normalisation) and its gradient are ONE fused HIP kernel (csrc/lpd_loss.hip) instead of ~25 tiny
elementwise launches.
"""
from lpdnet_hip import autograd as _ag


def best_pos_distance(query, pos_vecs):
    """(:6-12) query [bq,1,D], pos_vecs [bq,P,D] -> (min_pos [bq], max_pos [bq]) squared distances."""
    return _ag.best_pos_distance(query, pos_vecs)


def triplet_loss(q_vec, pos_vecs, neg_vecs, margin, use_min=False, lazy=False, ignore_zero_loss=False):
    """(:15-42)"""
    return _ag.metric_loss(q_vec, pos_vecs, neg_vecs, None, margin, 0.0, use_min, lazy, ignore_zero_loss, quad=False)


def triplet_loss_wrapper(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2, use_min=False, lazy=False,
                         ignore_zero_loss=False):
    """(:45-46) ignores other_neg and m2, like the reference."""
    return triplet_loss(q_vec, pos_vecs, neg_vecs, m1, use_min, lazy, ignore_zero_loss)


def quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2, use_min=False, lazy=False, ignore_zero_loss=False):
    """(:49-97)"""
    return _ag.metric_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2, use_min, lazy, ignore_zero_loss, quad=True)
