"""Counterparts of the two callers that sit directly on the hot path.

The reference's train/eval scripts (`train_pointnetvlad.py`, `evaluate.py`) read argparse singletons, pickles and
the Oxford dataset at import time and never travel; these functions reproduce exactly what those scripts do
AROUND the model call, with plain arguments:

  run_model          train_pointnetvlad.py:202-217   cat(q, pos, neg, other) -> [B,1,N,3] -> model -> split
  train_step         train_pointnetvlad.py:121-134   zero_grad / run_model / loss / backward / optimizer.step
  get_latent_vectors evaluate.py:96-159              eval mode, batched forward with a ragged tail, numpy out,
                                                     model.train() afterwards
"""
import numpy as np
import torch

import loss.pointnetvlad_loss as L


def run_model(model, queries, positives, negatives, other_neg, require_grad=True, output_dim=256):
    """queries [bq,1,N,3], positives [bq,P,N,3], negatives [bq,Ng,N,3], other_neg [bq,1,N,3] (any device/dtype)
    -> (q [bq,1,D], pos [bq,P,D], neg [bq,Ng,D], other [bq,1,D]); cloud order inside a tuple = q, pos, neg, other."""
    bq, P, Ng = queries.shape[0], positives.shape[1], negatives.shape[1]
    N = queries.shape[2]
    feed = torch.cat((queries, positives, negatives, other_neg), 1).reshape(-1, 1, N, 3)
    dev = next(model.parameters()).device
    feed = feed.to(dev, dtype=torch.float32, non_blocking=True)
    if require_grad:
        out = model(feed)
    else:
        with torch.no_grad():
            out = model(feed)
    out = out.view(bq, -1, output_dim)
    return torch.split(out, [1, P, Ng, 1], dim=1)


def train_step(model, optimizer, queries, positives, negatives, other_neg, *, margin_1=0.5, margin_2=0.2,
               loss_function="quadruplet", use_min=True, lazy=True, ignore_zero_loss=False):
    """One optimisation step with the reference's defaults (util/initPara.py:32-88: margins 0.5/0.2, quadruplet,
    lazy, best positive).  Returns the loss tensor (0-dim, on the device)."""
    model.train()
    optimizer.zero_grad()
    q, p, n, o = run_model(model, queries, positives, negatives, other_neg)
    fn = L.quadruplet_loss if loss_function == "quadruplet" else L.triplet_loss_wrapper
    loss = fn(q, p, n, o, margin_1, margin_2, use_min=use_min, lazy=lazy, ignore_zero_loss=ignore_zero_loss)
    loss.backward()
    optimizer.step()
    return loss


def get_latent_vectors(model, clouds, batch_size):
    """clouds: [n, N, 3] array-like (float64 like the benchmark's .bin submaps, or float32) -> [n, D] float32 numpy.

    Mirrors evaluate.get_latent_vectors: full batches of `batch_size` clouds, then the ragged remainder in one
    call, eval mode inside, train mode restored afterwards (evaluate.py:97,156)."""
    was_training = model.training
    model.eval()
    dev = next(model.parameters()).device
    clouds = np.asarray(clouds)
    n = clouds.shape[0]
    outs = []
    try:
        with torch.no_grad():
            for s in range(0, n, batch_size):
                chunk = torch.from_numpy(np.ascontiguousarray(clouds[s:s + batch_size])).float().unsqueeze(1).to(dev)
                outs.append(model(chunk).detach().cpu().numpy())
    finally:
        model.train(was_training)
    if not outs:
        return np.zeros((0, 0), dtype=np.float32)
    return np.concatenate(outs, axis=0)
