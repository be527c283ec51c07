"""Oracle (test infrastructure): CPU restatement of the reference's hybrid score merge.

Pinned by golden fixtures produced by running the reference's own modules
(tests/golden/make_golden.py -> merge_*.npz, normalize.npz, gather.npz).

Follows (paths relative to /root/reference/src):
  * `_subtract_min_score`            vod_dataloaders/core/normalize.py:17-20
  * `_write_1d_arr`/`_search_1d_arr` vod_dataloaders/core/merge.py:71-105
  * `_nopy_merge_two_search_results` vod_dataloaders/core/merge.py:108-164
  * `_merge_n_search_results`        vod_dataloaders/core/merge.py:31-62
  * `gather_values_by_indices`       vod_dataloaders/core/numpy_ops.py:24-143
  * `_merge_search_results`          vod_dataloaders/core/search.py:79-125
  * `RetrievalBatch.__mul__`         vod_types/retrieval.py:222-233

Pure-Python loops: meant for fixture-sized inputs (<= a few thousand entries).
"""
from __future__ import annotations

import warnings

import numpy as np

LOOKUP = "lookup"  # core/search.py:17


def subtract_min_score(scores: np.ndarray, offset: float = 0.0) -> np.ndarray:
    """Row-wise `s - min_finite(s) + offset`; NaN/inf entries are ignored by the min (normalize.py:17-20)."""
    with np.errstate(all="ignore"):
        non_nan = np.where(np.isinf(scores) | np.isnan(scores), np.inf, scores)
        mn = np.amin(non_nan, axis=-1, keepdims=True)
        return scores - mn + offset


def merge_two(a_s: np.ndarray, a_i: np.ndarray, b_s: np.ndarray, b_i: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Union of ids in first-seen order with score accumulation; output truncated to max_cursor+1 columns."""
    nq = a_s.shape[0]
    width = a_s.shape[1] + b_s.shape[1]
    scores = np.full((nq, width), -np.inf, dtype=a_s.dtype)
    indices = np.full((nq, width), -1, dtype=a_i.dtype)
    cursors = np.zeros(nq, dtype=np.int64)
    with np.errstate(all="ignore"):
        for r in range(nq):
            cur = 0
            pos: dict[int, int] = {}
            for src_s, src_i in ((a_s[r], a_i[r]), (b_s[r], b_i[r])):
                for j in range(src_i.shape[0]):
                    idx = int(src_i[j])
                    if idx < 0:
                        continue
                    f = pos.get(idx, -1)
                    if f < 0:
                        scores[r, cur] = src_s[j]
                        indices[r, cur] = idx
                        pos[idx] = cur
                        cur += 1
                    else:
                        scores[r, f] = src_s[j] + scores[r, f]
            cursors[r] = cur
    m = int(cursors.max()) if nq else 0
    return scores[:, : m + 1], indices[:, : m + 1]


def gather_values(queries: np.ndarray, keys: np.ndarray, values: np.ndarray, fill_value=None) -> np.ndarray:
    """For each query id the value at the FIRST position where `keys == query`, else fill (NaN for floats, -1 else)."""
    if fill_value is None:
        fill_value = np.nan if values.dtype.kind == "f" else -1
    out = np.full(queries.shape, fill_value, dtype=values.dtype)
    if queries.ndim == 1:
        queries2, keys2, values2, out2 = queries[None], keys[None], values[None], out[None]
    else:
        queries2, out2 = queries, out
        if keys.ndim == 1:
            keys2 = np.broadcast_to(keys, (queries.shape[0],) + keys.shape)
            values2 = np.broadcast_to(values, (queries.shape[0],) + values.shape)
        else:
            keys2, values2 = keys, values
    for r in range(queries2.shape[0]):
        first: dict[int, int] = {}
        for j, kx in enumerate(keys2[r]):
            first.setdefault(int(kx), j)
        for c, qx in enumerate(queries2[r]):
            j = first.get(int(qx), -1)
            if j >= 0:
                out2[r, c] = values2[r, j]
    return out


def merge_search_results(results: dict[str, tuple], weights: dict[str, float]):
    """`merge.merge_search_results` for >= 2 engines.

    `results[name] = (scores, indices, labels_or_None)`; returns (scores, indices, labels_or_None, raw_scores dict).
    """
    keys = list(results.keys())
    if len(keys) == 1:  # a single result set is returned as is, weighted: no union, labels as given (merge.py:18-22)
        s0, i0, l0 = results[keys[0]]
        with warnings.catch_warnings(), np.errstate(all="ignore"):
            warnings.simplefilter("ignore")
            return s0 * weights[keys[0]], i0, l0, {keys[0]: s0}
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        s0, i0, _ = results[keys[0]]
        out_s, out_i = s0 * weights[keys[0]], i0
        for name in keys[1:]:
            s, i, _ = results[name]
            out_s, out_i = merge_two(out_s, out_i, s * weights[name], i)
    raw = {name: gather_values(out_i, results[name][1], results[name][0]) for name in keys}
    labels = None
    for name in keys:
        if results[name][2] is not None:
            labels = gather_values(out_i, results[name][1], results[name][2], fill_value=-1)
    return out_s, out_i, labels, raw


def merge_hybrid(lookup: tuple, engines: dict[str, tuple], weights: dict[str, float]):
    """The collate-side merge `core/search.py:79-125`.

    lookup = (indices, scores, labels); engines[name] = (indices, scores).
    Returns (indices, scores, labels, raw_scores{name}) with the reference's layout: first-seen order
    (lookup, then each engine in dict order), one trailing pad column, raw scores min-subtracted.
    """
    l_idx, l_scr, l_lbl = lookup
    res: dict[str, tuple] = {LOOKUP: (np.zeros_like(l_scr), l_idx, l_lbl)}  # lookup scores are discarded (:92)
    for name, (idx, scr) in engines.items():
        res[name] = (scr, idx, None)  # labels dropped for the other engines (:93-96)
    res = {n: (subtract_min_score(s, 0.0) if s.size else s, i, lab) for n, (s, i, lab) in res.items()}  # (:109)
    out_s, out_i, out_l, raw = merge_search_results(res, {LOOKUP: 0.0, **weights})  # (:117-120)
    raw.pop(LOOKUP)
    return out_i, out_s, out_l, raw
