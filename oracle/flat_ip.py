"""Oracle (test infrastructure): exact brute-force inner-product top-k, CPU, NumPy.

PARITY UNPINNED.  This restates what the reference obtains from
`faiss_index.search(query_vec, k)` (src/vod_search/faiss_search/server.py:72,84) on an index
built by `faiss.index_factory(D, "Flat", faiss.METRIC_INNER_PRODUCT)`
(src/vod_search/faiss_search/build.py:60, src/vod_configs/search.py:128-130) after
`index.add(batch.astype(np.float32))` (build.py:67-73).  The arithmetic lives in the
third-party wheel faiss-cpu==1.7.4 (requirements.txt:42), which is neither vendored under
/root/reference nor installable here, and the reference has no test, fixture or golden
vector at that boundary.  What is restated is faiss's published IndexFlatIP contract:

  scores[i, :] = the k largest <q_i, x_j>, sorted descending; ids[i, :] the matching row
  numbers as int64; when fewer than k rows exist the tail is id -1.

Two deliberate, documented choices where faiss leaves the result open:
  * ties are broken by the smaller row id (faiss's heap order among equal scores is
    implementation-defined) -- `north_star` asks for "deterministic tie-break";
  * the pad score is -inf, the repo-wide padding convention
    (src/vod_types/retrieval.py:284-285), not faiss's -FLT_MAX-like sentinel that the
    reference has to scrub afterwards (src/vod_dataloaders/core/search.py:16,149-161).

Accumulation is float64 over the exactly-representable inputs, so for fp16/bf16 inputs the
result is the correctly rounded reference the GPU's fp32 accumulation is compared against
(tolerance 1e-3, stated in the tests); for the small-integer fixtures every partial sum is
exact in fp32, so GPU and oracle must agree bit for bit.
"""
from __future__ import annotations

import numpy as np


_NAN_ID = np.int64(1) << 62


def flat_ip_scores(q: np.ndarray, x: np.ndarray) -> np.ndarray:
    """Full score matrix in float64 (small cases only)."""
    return np.asarray(q, dtype=np.float64) @ np.asarray(x, dtype=np.float64).T


def topk_desc_tiebreak(scores: np.ndarray, k: int, id_base: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """Row-wise top-k of a score matrix: score descending, then id ascending; pad -inf / -1."""
    nq, n = scores.shape
    out_s = np.full((nq, k), -np.inf, dtype=np.float32)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    kk = min(k, n)
    if kk == 0:
        return out_s, out_i
    ids = np.arange(n, dtype=np.int64)
    for i in range(nq):
        row = scores[i]
        valid = ~np.isnan(row)  # a NaN score never enters the result (faiss: comparisons are false)
        cand = ids[valid]
        if cand.size > 4 * kk:
            # cut to the candidates that can matter, keeping every tie of the boundary value
            kth = np.partition(row[cand], cand.size - kk)[cand.size - kk]
            cand = cand[row[cand] >= kth]
        order = np.lexsort((cand, -row[cand]))[:kk]
        sel = cand[order]
        out_s[i, : sel.size] = row[sel].astype(np.float32)
        out_i[i, : sel.size] = sel + id_base
    return out_s, out_i


def flat_ip_topk(q: np.ndarray, x: np.ndarray, k: int, block: int = 65536, id_base: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """Exact MIPS top-k, float64 accumulation, blocked over corpus rows to bound memory."""
    q64 = np.asarray(q, dtype=np.float64)
    nq = q64.shape[0]
    n = x.shape[0]
    best_s = np.full((nq, 0), -np.inf, dtype=np.float64)
    best_i = np.full((nq, 0), -1, dtype=np.int64)
    for lo in range(0, n, block):
        hi = min(n, lo + block)
        s = q64 @ np.asarray(x[lo:hi], dtype=np.float64).T
        ids = np.broadcast_to(np.arange(lo, hi, dtype=np.int64), s.shape)
        cs = np.concatenate([best_s, s], axis=1)
        ci = np.concatenate([best_i, ids], axis=1)
        # a NaN score never enters the result (faiss: comparisons are false): it sinks below every real row, -inf ones
        # included (sentinel id above every row id), and leaves as a pad
        nan = np.isnan(cs)
        ci = np.where(nan, _NAN_ID, ci)
        cs = np.where(nan, -np.inf, cs)
        keep = min(k, cs.shape[1])
        new_s = np.empty((nq, keep), dtype=np.float64)
        new_i = np.empty((nq, keep), dtype=np.int64)
        for r in range(nq):
            row_s, row_i = cs[r], ci[r]
            if row_s.size > 4 * keep:
                # only entries >= the keep-th largest value can be selected (every tie of that value is kept)
                kth = np.partition(row_s, row_s.size - keep)[row_s.size - keep]
                sel = np.nonzero(row_s >= kth)[0]
                row_s, row_i = row_s[sel], row_i[sel]
            order = np.lexsort((row_i, -row_s))[:keep]
            new_s[r] = row_s[order]
            new_i[r] = row_i[order]
        best_s, best_i = new_s, new_i
    out_s = np.full((nq, k), -np.inf, dtype=np.float32)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    kk = best_s.shape[1]
    out_s[:, :kk] = best_s.astype(np.float32)
    out_i[:, :kk] = np.where(best_i == _NAN_ID, -1, best_i + id_base)
    return out_s, out_i


def merge_shard_topk(scores: list[np.ndarray], ids: list[np.ndarray], k: int) -> tuple[np.ndarray, np.ndarray]:
    """Top-k of the union of per-shard top-k lists (ids already global): what the all-gather merge must equal.

    Follows the reference's logical-shard rule `indices += offset` then stack
    (src/vod_search/sharded_search.py:92-106,198-203) but keeps pads at -1.
    """
    cs = np.concatenate(scores, axis=1).astype(np.float64)
    ci = np.concatenate(ids, axis=1)
    cs = np.where(ci < 0, -np.inf, cs)
    nq = cs.shape[0]
    out_s = np.full((nq, k), -np.inf, dtype=np.float32)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    for r in range(nq):
        valid = ci[r] >= 0
        cand = np.nonzero(valid)[0]
        order = cand[np.lexsort((ci[r, cand], -cs[r, cand]))][:k]
        out_s[r, : order.size] = cs[r, order]
        out_i[r, : order.size] = ci[r, order]
    return out_s, out_i


def recall_at_k(ids: np.ndarray, ref_ids: np.ndarray) -> float:
    """Fraction of the reference's (valid) ids recovered per row, averaged."""
    tot, hit = 0, 0
    for a, b in zip(ids, ref_ids):
        bset = set(int(v) for v in b if v >= 0)
        tot += len(bset)
        hit += len(bset.intersection(int(v) for v in a if v >= 0))
    return hit / max(tot, 1)
