"""Oracle-side restatement of the reference's retrieval metrics (test / bench-verification infrastructure, never shipped).

Restates /root/reference/src/vod_models/monitoring/functional.py in NumPy:
  `_mask_rank_inputs` :15-25 (NaN and +inf scores are masked: score -> -inf, relevance -> 0; ranking by score, descending),
  `prepare_for_metric_computation` :166-180 (n_positives is counted BEFORE masking; the ranked lists are cut to `topk`),
  `_compute_mrr` :41-51, `_compute_hitrate` :54-60, `_compute_precision` :63-71, `_compute_recall` :74-81, `_compute_ndcg` :140-161.
Pinned by tests/golden/metrics_recall_ndcg.npz (generated from the imported reference by tests/golden/make_golden.py).
The reference ranks with `torch.argsort(descending=True)` (not stable): entries with EQUAL scores may come in either order - the
restatement uses a stable sort, and the fixture's tied row carries equal relevances on its tied entries.

`recall_of_ids` is how bench.py's `verify.recall_at_k` is defined: the reference's recall@k of the returned list, with the comparator's
top-k ids as the positives.
"""
from __future__ import annotations

import numpy as np


def rank_inputs(relevances: np.ndarray, scores: np.ndarray, topk: int | None = None):
    """-> (ranked_relevances, ranked_scores, n_positives); functional.py:15-25,166-180."""
    relevances = np.asarray(relevances)
    scores = np.asarray(scores, dtype=np.float32)
    n_positives = (relevances > 0).sum(axis=-1)
    mask = np.isnan(scores) | (np.isinf(scores) & (scores > 0))
    scores = np.where(mask, -np.inf, scores).astype(np.float32)
    relevances = np.where(mask, 0, relevances)
    order = np.argsort(-scores, axis=-1, kind="stable")
    rr = np.take_along_axis(relevances, order, axis=-1)
    rs = np.take_along_axis(scores, order, axis=-1)
    if topk:
        rr, rs = rr[..., :topk], rs[..., :topk]
    return rr, rs, n_positives


def recall(relevances, scores, topk=None) -> np.ndarray:
    """functional.py:74-81: retrieved relevant / positives (NaN for a row without positives: 0 / 0, as the reference)."""
    rr, _, n_pos = rank_inputs(relevances, scores, topk)
    with np.errstate(divide="ignore", invalid="ignore"):
        return ((rr > 0).sum(axis=-1).astype(np.float32) / n_pos.astype(np.float32)).astype(np.float32)


def precision(relevances, scores, topk=None) -> np.ndarray:
    """functional.py:63-71: retrieved relevant / finite retrieved scores."""
    rr, rs, _ = rank_inputs(relevances, scores, topk)
    with np.errstate(divide="ignore", invalid="ignore"):
        return ((rr > 0).sum(axis=-1).astype(np.float32) / np.isfinite(rs).sum(axis=-1).astype(np.float32)).astype(np.float32)


def hitrate(relevances, scores, topk=None) -> np.ndarray:
    """functional.py:54-60."""
    rr, _, _ = rank_inputs(relevances, scores, topk)
    return (rr > 0).any(axis=-1)


def mrr(relevances, scores, topk=None) -> np.ndarray:
    """functional.py:8-12,41-51: 1 / (1 + index of the first positive), 0 without one."""
    rr, _, _ = rank_inputs(relevances, scores, topk)
    pos = rr > 0
    first = np.where(pos.any(axis=-1), pos.argmax(axis=-1), 0)
    return np.where(pos.any(axis=-1), (1.0 / (1 + first)).astype(np.float32), np.float32(0)).astype(np.float32)


def ndcg(relevances, scores, topk=None) -> np.ndarray:
    """functional.py:140-161: DCG with graded relevances / log2(rank + 1); the IDEAL ordering is taken over the cut list itself."""
    rr, _, _ = rank_inputs(relevances, scores, topk)
    rr = rr.astype(np.float32)
    log2_ranks = np.log2(np.arange(2, rr.shape[-1] + 2, dtype=np.float32)).astype(np.float32)
    dcg = (rr / log2_ranks).sum(axis=-1, dtype=np.float32)
    ideal = -np.sort(-rr, axis=-1)
    idcg = (ideal / log2_ranks).sum(axis=-1, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(idcg > 0, dcg / idcg, np.float32(0)).astype(np.float32)


def recall_of_ids(got_ids: np.ndarray, got_scores: np.ndarray, ref_ids: np.ndarray) -> float:
    """The reference's recall@k (mean over rows) of a returned top-k list against a comparator's top-k ids: per row the candidate set
    is the union of both lists - the comparator's ids are the positives (relevance 1), the returned ids carry their scores, a
    positive that was NOT returned scores NaN (masked, functional.py:18-20: it still counts in n_positives, which is taken before
    the masking, :172, but can never be retrieved) - cut at k = the list length."""
    got_ids, ref_ids = np.asarray(got_ids), np.asarray(ref_ids)
    got_scores = np.asarray(got_scores, dtype=np.float32)
    k = got_ids.shape[1]
    vals = []
    for gi, gs, ri in zip(got_ids, got_scores, ref_ids):
        ref = [int(v) for v in ri if v >= 0]
        if not ref:
            continue
        got = [(int(i), float(s)) for i, s in zip(gi, gs) if i >= 0]
        have = {i for i, _ in got}
        ids = [i for i, _ in got] + [i for i in ref if i not in have]
        sc = np.array([s for _, s in got] + [np.nan] * (len(ids) - len(got)), dtype=np.float32)
        refset = set(ref)
        rel = np.array([1 if i in refset else 0 for i in ids], dtype=np.int64)
        vals.append(float(recall(rel[None], sc[None], k)[0]))
    return float(np.mean(vals)) if vals else 1.0
