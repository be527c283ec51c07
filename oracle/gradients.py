"""Oracle (test infrastructure): CPU restatement of the reference's in-batch retrieval scoring + loss.

Pinned by `retrieval_grad_*.npz` (produced by running the reference's `RetrievalGradients` + autograd).

Follows /root/reference/src/vod_models/vod_gradients/retrieval.py:
  * `RetrievalGradients.__call__`   :30-92   (padding mask, log-softmax, targets, n_positives fallback)
  * `_compute_retriever_scores`     :186-203 (einsum "bh,dh->bd" | "bh,bdh->bd", masked_fill -inf)
  * `_cast_data_targets`            :206-215
  * `_compute_loss`                 :153-177 (w = (p - t) / n_pos detached; mean over rows with positives)
  * `_masked_logprobs`/`_compute_kld` :218-243

  * `_auxiliary_losses` + `_guidance_loss` / `_self_supervision_loss` / `_score_decay_loss` / `_compute_hubert_loss`
                                    :94-150,180-183 (pinned by `retrieval_aux_*.npz`)

Forward AND the analytic backward are written out in float64 NumPy (no autograd), so the fused HIP
forward/backward kernel has an independent checker.
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


def retrieval_gradients(q, s, score, relevance, sparse=None, dense=None, guidance="zero", guidance_weight=0.0,
                        self_supervision_weight=0.0, score_decay=0.0):
    """Returns dict(loss, retriever_scores, dq, ds, kl_score, kl_sparse, kl_dense [, <guidance>_guidance, self_supervision,
    score_decay]), float64."""
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
    aux = {}
    with np.errstate(all="ignore"):
        if guidance_weight > 0:  # huber(logp - ref), delta 1, mean over entries finite in both (:116-126,180-183)
            ref = np.asarray(sparse, dtype=np.float64) if guidance == "sparse" else np.zeros_like(scores)
            m = np.isfinite(logp) & np.isfinite(ref)
            x = np.where(m, logp - np.where(m, ref, 0.0), 0.0)
            hub = np.where(np.abs(x) < 1, 0.5 * x * x, np.abs(x) - 0.5)
            val = hub[m].mean() if m.any() else np.nan
            g = np.where(m, np.clip(x, -1, 1), 0.0) / max(m.sum(), 1)
            aux[f"{guidance}_guidance"] = (val, guidance_weight, g - np.where(np.isfinite(logp), p, 0.0) * g.sum(-1, keepdims=True))
        if self_supervision_weight > 0:  # cross entropy of the positives' log-probs vs their own arg-max (:129-140)
            lpos = np.where(t > 0, logp, -np.inf)
            rows = npos > 0  # AFTER the n_positives fallback: a row without positives stays in and makes the loss NaN
            idx = np.argmax(lpos, axis=-1)
            lsm = _log_softmax(lpos)
            ce = -lsm[np.arange(len(idx)), idx]
            val = ce[rows].mean() if rows.any() else np.nan
            has = (t > 0).any(-1)
            sm = np.where(t > 0, np.exp(np.where(has[:, None], lsm, 0.0)), 0.0)
            onehot = np.zeros_like(sm)
            onehot[np.arange(len(idx)), idx] = 1.0
            g = np.where((rows & has)[:, None], sm - np.where(t > 0, onehot, 0.0), 0.0) / max(rows.sum(), 1)
            aux["self_supervision"] = (val, self_supervision_weight, g)  # sums to zero per row: log-softmax backward is the identity
        if score_decay > 0:  # mean of the squared finite scores (:143-145)
            fin = np.isfinite(scores)
            val = (scores[fin] ** 2).mean() if fin.any() else np.nan
            aux["score_decay"] = (val, score_decay, np.where(fin, 2.0 * np.where(fin, scores, 0.0), 0.0) / max(fin.sum(), 1))
    for name, (val, weight, g) in aux.items():
        loss = loss + weight * val
        d_scores = d_scores + weight * g
    if three_d:
        dq = np.einsum("bd,bdh->bh", d_scores, s)
        ds = d_scores[:, :, None] * q[:, None, :]
    else:
        dq = d_scores @ s
        ds = d_scores.T @ q
    out = {"loss": loss, "retriever_scores": scores, "dq": dq, "ds": ds, "d_scores": d_scores}
    out.update({name: val for name, (val, _w, _g) in aux.items()})
    for name, ref in (("kl_score", score), ("kl_sparse", sparse), ("kl_dense", dense)):
        if ref is not None:
            out[name] = kld(logp, np.asarray(ref, dtype=np.float64)).mean()
    return out
