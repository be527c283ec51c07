"""Oracle (test infrastructure): CPU restatement of the reference's labeled priority sampling.

Pinned by `sampling_fixed_noise.npz` / `flatten_inbatch.npz` / `collate_chain.npz` (produced by running the reference).

Follows (paths relative to /root/reference/src/vod_dataloaders/core):
  * `log_softmax_1d_`, `max_1d`, `_logsumexp_1d`   numpy_ops.py:162-216
  * `_priority_sampling_1d`                         sample.py:160-219
  * `_labeled_priority_sampling_1d_`                sample.py:245-320
  * `flatten_samples`                               in_batch_negatives.py:10-52
  * `sample_search_results`                         sample.py:22-84 (with the Exp(1) draw passed in: sample.py:398)

Reference behaviours kept on purpose: Q8 -- support truncation masks the entries
`>= threshold` (it REMOVES the top `max_support_size` entries), and the "not enough positives"
branch assigns to `n_pos_finite` (a no-op); Q7 -- flatten pads with index 1.
"""
from __future__ import annotations

import numpy as np


def _log_softmax_1d(x: np.ndarray) -> np.ndarray:
    x = x.copy()
    dt = x.dtype.type
    x[np.isnan(x)] = dt(-np.inf)
    xm = dt(-np.inf)
    for v in x:  # numpy_ops.py:176-188
        if v > xm:
            xm = v
    if not np.isfinite(xm) and xm < 0:
        xm = dt(0.0)
    x = (x + (-xm)).astype(x.dtype)
    lse = dt(0.0)
    with np.errstate(all="ignore"):
        for v in x:  # sequential accumulation in the array dtype (numpy_ops.py:207-213)
            lse = dt(lse + np.exp(v))
        x = (x + (-np.log(lse))).astype(x.dtype)
    return x


def priority_sampling_1d(scores: np.ndarray, noise: np.ndarray, k: int, temperature: float = 1.0, max_support_size: int = -1, keep_top: bool = False):
    dt = scores.dtype.type
    with np.errstate(all="ignore"):
        t_inv = dt(temperature if temperature > 0 else 1.0)
        log_p = (scores * t_inv).astype(scores.dtype)
        if max_support_size > 0 and len(log_p) > max_support_size:
            thr = np.sort(log_p)[-max_support_size]
            if keep_top:  # the CORRECTED truncation (no reference counterpart: what `max_support_size` is named for) - keep the best entries
                log_p[log_p < thr] = dt(-np.inf)
            else:
                log_p[log_p >= thr] = dt(-np.inf)  # Q8: removes the top entries
        log_p = _log_softmax_1d(log_p)
        log_norm = np.log(np.sum(np.exp(log_p)))
        log_u = np.log(noise)
        log_keys = (log_p - log_u) if temperature > 0 else log_p.copy()
        sorted_ids = np.argsort(-log_keys)[: k + 1]
        if k < log_p.shape[-1]:
            log_tau = log_keys[sorted_ids[-1]]
        else:
            log_tau = dt(-np.inf)
        sorted_ids = sorted_ids[:k]
        log_pi = np.take(log_p, sorted_ids)
        if log_tau > -np.inf:
            log_qz = np.log1p(-np.exp(-np.exp((log_pi + (-log_tau)).astype(scores.dtype))))
            log_w = log_pi - log_qz
        else:
            log_w = log_pi.copy()
    return sorted_ids, log_w.astype(scores.dtype), log_norm


def labeled_priority_sampling_2d(scores, labels, noise, k_positive, k_total, normalized=True, temperature=1.0, max_support_size=-1, keep_top=False):
    nq, n = scores.shape
    out_samples = np.full((nq, k_total), -1, dtype=np.int64)
    out_logw = np.full((nq, k_total), -np.inf, dtype=scores.dtype)
    out_labels = np.zeros((nq, k_total), dtype=np.bool_)
    out_lse = np.zeros((nq, 2), dtype=scores.dtype)
    for r in range(nq):
        idx = np.arange(n)
        lab = labels[r] > 0
        nlab = ~lab
        is_inf = np.isinf(scores[r])
        n_neg_finite = int(np.sum(nlab & ~is_inf))
        kt = n if k_total > n else k_total
        kp = k_positive
        if n_neg_finite < kt - kp:
            kp = kt - n_neg_finite
        ps, pw, plse = priority_sampling_1d(scores[r][lab], noise[r][lab], kp, temperature, max_support_size, keep_top)
        pos = idx[lab][ps]
        if normalized and len(pos) > 0:
            pw = _log_softmax_1d(pw)
        ns, nw, nlse = priority_sampling_1d(scores[r][nlab], noise[r][nlab], kt - len(ps), temperature, max_support_size, keep_top)
        neg = idx[nlab][ns]
        if normalized and len(neg) > 0:
            nw = _log_softmax_1d(nw)
        out_lse[r, 0], out_lse[r, 1] = plse, nlse
        j = 0
        for i in range(len(pos)):
            out_samples[r, j], out_logw[r, j], out_labels[r, j] = pos[i], pw[i], True
            j += 1
        for i in range(len(neg)):
            out_samples[r, j], out_logw[r, j], out_labels[r, j] = neg[i], nw[i], False
            j += 1
    return out_samples, out_logw, out_labels, out_lse


def flatten_samples(indices, scores, labels, log_weights, raw_scores: dict, padding: bool = True):
    from .hybrid import gather_values

    uniq = np.unique(indices)
    if padding:
        n_pad = int(np.prod(indices.shape)) - uniq.shape[0]
        uniq = np.concatenate([uniq, np.ones((n_pad,), dtype=np.int64)])  # Q7
    uq = uniq[None, :].repeat(indices.shape[0], axis=0)
    return {
        "indices": uniq,
        "scores": gather_values(uq, indices, scores),
        "labels": gather_values(uq, indices, labels, fill_value=0),
        "log_weights": gather_values(uq, indices, log_weights),
        "raw": {k: gather_values(uq, indices, v) for k, v in raw_scores.items()},
    }


def sample_search_results(indices, scores, labels, raw_scores: dict, noise, total, max_pos_sections, temperature=1.0, max_support_size=None, keep_top=False):
    """sample.py:22-84 with the noise of sample.py:398 passed in.  Returns a dict of the `PrioritySampledSections` fields."""
    total = total or scores.shape[-1]
    max_pos_sections = max_pos_sections or total
    labels_ref = np.zeros_like(scores, dtype=np.bool_) if labels is None else labels > 0
    max_support_size = max_support_size or -1          # sample.py:123-128
    if max_support_size >= 0:
        max_support_size = max(max_support_size, total)
    with np.errstate(all="ignore"):
        local, logw, lab, lse = labeled_priority_sampling_2d(scores, labels_ref, noise, max_pos_sections, total, True, temperature,
                                                             max_support_size, keep_top)
    take = lambda a: np.take_along_axis(a, local, axis=-1)  # noqa: E731  (-1 pads take the LAST column, as NumPy indexes)
    smp_scores = take(scores)
    min_neg = np.amin(np.where((lab <= 0) & np.isfinite(smp_scores), smp_scores, np.inf), axis=-1, keepdims=True)
    larger = (labels_ref <= 0) & np.isfinite(scores) & (scores >= min_neg)
    return {
        "local": local, "indices": take(indices), "scores": smp_scores, "labels": lab, "log_weights": logw,
        "lse_pos": lse[..., 0], "lse_neg": lse[..., 1], "max_sampling_id": np.sum(larger.astype(np.float32), axis=-1),
        "raw": {k: take(v) for k, v in raw_scores.items()},
    }
