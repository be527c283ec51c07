"""Oracle (test infrastructure): CPU restatement of the reference's in-batch retrieval scoring + loss.

Pinned by `retrieval_grad_*.npz` (produced by running the reference's `RetrievalGradients` + autograd).

Follows /root/reference/src/vod_models/vod_gradients/retrieval.py:
  * `RetrievalGradients.__call__`   :30-92   (padding mask, log-softmax, targets, n_positives fallback)
  * `_compute_retriever_scores`     :186-203 (einsum "bh,dh->bd" | "bh,bdh->bd", masked_fill -inf)
  * `_cast_data_targets`            :206-215
  * `_compute_loss`                 :153-177 (w = (p - t) / n_pos detached; mean over rows with positives)
  * `_masked_logprobs`/`_compute_kld` :218-243

Forward AND the analytic backward are written out in float64 NumPy (no autograd), so the fused HIP
forward/backward kernel has an independent checker.  Auxiliary losses (guidance, self-supervision,
score decay) have weight 0 in the shipped config (hydra/model/gradients/retrieval.yaml) and are not covered.
"""
from __future__ import annotations

import numpy as np


def _log_softmax(x: np.ndarray) -> np.ndarray:
    with np.errstate(all="ignore"):
        m = np.max(x, axis=-1, keepdims=True)
        z = x - m  # all -inf row -> nan, as torch
        return z - np.log(np.sum(np.exp(z), axis=-1, keepdims=True))


def kld(p_logits: np.ndarray, q_logits: np.ndarray) -> np.ndarray:
    """KL(q || p) per row over entries finite in both (retrieval.py:225-243)."""
    with np.errstate(all="ignore"):
        p_def = np.isfinite(p_logits)
        q_def = np.isfinite(q_logits)
        p_lp = _log_softmax(np.where(p_def, p_logits, -np.inf))
        q_lp = _log_softmax(np.where(q_def, q_logits, -np.inf))
        terms = np.where(p_def & q_def, np.exp(q_lp) * (q_lp - p_lp), 0.0)
        return terms.sum(-1)


def retrieval_gradients(q, s, score, relevance, sparse=None, dense=None):
    """Returns dict(loss, retriever_scores, dq, ds, kl_score, kl_sparse, kl_dense), float64."""
    q = np.asarray(q, dtype=np.float64)
    s = np.asarray(s, dtype=np.float64)
    score = np.asarray(score, dtype=np.float64)
    pad = np.isinf(score) & (score < 0)
    three_d = s.ndim == 3
    scores = np.einsum("bh,bdh->bd", q, s) if three_d else q @ s.T
    scores = np.where(pad, -np.inf, scores)
    with np.errstate(all="ignore"):
        logp = _log_softmax(scores)
        p = np.exp(logp)
    t = ((np.asarray(relevance) > 0) & ~pad).astype(np.float64)
    npos = t.sum(1)
    npos = np.where(npos == 0, (~pad).sum(1).astype(np.float64), npos)
    has_pos = npos > 0
    with np.errstate(all="ignore"):
        w = (p - t) / npos[:, None]
        row = np.where(pad, 0.0, w * logp).sum(-1)
    n_rows = has_pos.sum()
    if n_rows > 0:
        loss = np.where(has_pos, row, 0.0).sum() / n_rows
        g = np.where(pad | ~has_pos[:, None], 0.0, w) / n_rows  # dloss/dlogp
        with np.errstate(all="ignore"):
            d_scores = g - np.where(pad, 0.0, p) * g.sum(-1, keepdims=True)  # log-softmax backward
        d_scores = np.where(pad, 0.0, d_scores)  # masked_fill_ backward
        d_scores = np.nan_to_num(d_scores, nan=0.0) if not np.all(np.isfinite(d_scores)) else d_scores
    else:
        loss = np.nan
        d_scores = np.full_like(scores, np.nan)
    if three_d:
        dq = np.einsum("bd,bdh->bh", d_scores, s)
        ds = d_scores[:, :, None] * q[:, None, :]
    else:
        dq = d_scores @ s
        ds = d_scores.T @ q
    out = {"loss": loss, "retriever_scores": scores, "dq": dq, "ds": ds, "d_scores": d_scores}
    for name, ref in (("kl_score", score), ("kl_sparse", sparse), ("kl_dense", dense)):
        if ref is not None:
            out[name] = kld(logp, np.asarray(ref, dtype=np.float64)).mean()
    return out
