"""Generate the golden fixtures under tests/golden/ by RUNNING the reference's leaf modules.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Every fixture is data: seeded inputs plus the outputs the reference produced for them
(`*.npz`), with the parameters in `manifest.json`.  No reference source text is stored.
The reference functions exercised (paths relative to /root/reference/src):

  merge        vod_dataloaders/core/search.py:79-125  (_merge_search_results)
               vod_dataloaders/core/merge.py:8-164    (merge_search_results and helpers)
  normalize    vod_dataloaders/core/normalize.py:17-20 (_subtract_min_score)
  gather       vod_dataloaders/core/numpy_ops.py:126-143 (gather_values_by_indices)
  sampling     vod_dataloaders/core/sample.py:323-352 (_labeled_priority_sampling_2d_)
  flatten      vod_dataloaders/core/in_batch_negatives.py:10-52
  collate      merge -> vod_dataloaders/core/sample.py:22-84 (sample_search_results) -> flatten, chained as realm_collate.py:110-139
  shard        vod_search/sharded_search.py:65-106,176-203
  stack        vod_types/retrieval.py:259-287
  io           vod_search/io.py:17-32
  gradients    vod_models/vod_gradients/retrieval.py:30-243 (incl. the auxiliary losses :94-150)
  metrics      vod_models/monitoring/functional.py:41-161,166-200 (compute_recall / ndcg / mrr / hitrate / precision)
"""
from __future__ import annotations

import json
import pathlib
import sys
import warnings

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import _ref_shim  # noqa: E402

warnings.filterwarnings("ignore")
M = _ref_shim.install()
RB = M["retrieval"].RetrievalBatch
manifest: dict[str, dict] = {}


def _save(name: str, params: dict, **arrays: np.ndarray) -> None:
    np.savez_compressed(HERE / f"{name}.npz", **arrays)
    manifest[name] = {"params": params, "arrays": {k: [str(v.dtype), list(v.shape)] for k, v in arrays.items()}}


def _rb(scores, indices, labels=None):
    return RB(scores=np.array(scores), indices=np.array(indices), labels=None if labels is None else np.array(labels))


# ----------------------------------------------------------------------------------------
# H4: hybrid merge (lookup + dense + sparse), through the collate-side entry point
# ----------------------------------------------------------------------------------------
def _hybrid_case(rng: np.random.Generator, nq: int, kk: int, n_ids: int, pad_frac: float, overlap: float):
    """Random lookup/dense/sparse results shaped like the three engine replies in core/search.py:42-62."""

    def engine(k, with_labels):
        idx = np.full((nq, k), -1, dtype=np.int64)
        scr = np.full((nq, k), -np.inf, dtype=np.float32)
        for i in range(nq):
            n_valid = k if rng.uniform() > pad_frac else int(rng.integers(0, k + 1))
            ids = rng.choice(n_ids, size=n_valid, replace=False)
            idx[i, :n_valid] = ids
            scr[i, :n_valid] = np.sort(rng.normal(size=n_valid).astype(np.float32) * 3.0)[::-1]
        lbl = (scr > -np.inf).astype(np.int64) if with_labels else None
        return idx, scr, lbl

    l_idx, l_scr, l_lbl = engine(max(2, kk // 4), True)
    d_idx, d_scr, _ = engine(kk, False)
    s_idx, s_scr, _ = engine(kk, False)
    # force some dense/sparse overlap and some lookup hits inside dense/sparse
    for i in range(nq):
        for j in range(kk):
            if d_idx[i, j] >= 0 and rng.uniform() < overlap:
                tgt = int(rng.integers(0, kk))
                if s_idx[i, tgt] >= 0 and d_idx[i, j] not in s_idx[i]:
                    s_idx[i, tgt] = d_idx[i, j]
        for j in range(l_idx.shape[1]):
            if l_idx[i, j] >= 0 and rng.uniform() < 0.5:
                tgt = int(rng.integers(0, kk))
                if d_idx[i, tgt] >= 0 and l_idx[i, j] not in d_idx[i]:
                    d_idx[i, tgt] = l_idx[i, j]
    return (l_idx, l_scr, l_lbl), (d_idx, d_scr), (s_idx, s_scr)


def _run_hybrid(lookup, dense, sparse, weights):
    res = {
        "lookup": RB(scores=lookup[1].copy(), indices=lookup[0].copy(), labels=lookup[2].copy()),
        "dense": RB(scores=dense[1].copy(), indices=dense[0].copy()),
        "sparse": RB(scores=sparse[1].copy(), indices=sparse[0].copy()),
    }
    merged, raw = M["search"]._merge_search_results(res, dict(weights))
    return merged, raw


def gen_merge() -> None:
    # the hand-written 3-engine case
    lookup = (np.array([[4, -1]], dtype=np.int64), np.array([[7.0, -np.inf]], dtype=np.float32), np.array([[1, 0]], dtype=np.int64))
    dense = (np.array([[2, 4, 8]], dtype=np.int64), np.array([[1.0, 0.6, 0.2]], dtype=np.float32))
    sparse = (np.array([[4, 6, -1]], dtype=np.int64), np.array([[5.0, 2.0, -np.inf]], dtype=np.float32))
    merged, raw = _run_hybrid(lookup, dense, sparse, {"dense": 2.0, "sparse": 0.5})
    _save(
        "merge_3engine_basic",
        {"weights": {"dense": 2.0, "sparse": 0.5}},
        lookup_idx=lookup[0], lookup_scr=lookup[1], lookup_lbl=lookup[2],
        dense_idx=dense[0], dense_scr=dense[1], sparse_idx=sparse[0], sparse_scr=sparse[1],
        out_idx=merged.indices, out_scr=merged.scores, out_lbl=merged.labels,
        raw_dense=raw["dense"], raw_sparse=raw["sparse"],
    )
    cases = [
        (0, 4, 16, 60, 0.3, 0.3, {"dense": 1.0, "sparse": 0.5}),
        (1, 4, 16, 60, 0.3, 0.3, {"dense": 0.0, "sparse": 1.0}),
        (2, 8, 32, 5000, 0.1, 0.3, {"dense": 1.0, "sparse": 1.0}),
        (3, 8, 32, 80, 0.5, 0.6, {"dense": 1.0, "sparse": 1.0}),
        (4, 16, 128, 100000, 0.1, 0.3, {"dense": 1.0, "sparse": 1.0}),
        (5, 16, 128, 400, 0.2, 0.5, {"dense": 0.7, "sparse": 1.3}),
        (6, 3, 5, 12, 0.9, 0.5, {"dense": 1.0, "sparse": 1.0}),
        (7, 64, 128, 1000000, 0.1, 0.3, {"dense": 1.0, "sparse": 1.0}),
    ]
    for seed, nq, kk, n_ids, pad, ov, w in cases:
        rng = np.random.default_rng(1000 + seed)
        lookup, dense, sparse = _hybrid_case(rng, nq, kk, n_ids, pad, ov)
        merged, raw = _run_hybrid(lookup, dense, sparse, w)
        _save(
            f"merge_random_{seed}",
            {"weights": w, "seed": 1000 + seed, "nq": nq, "k": kk},
            lookup_idx=lookup[0], lookup_scr=lookup[1], lookup_lbl=lookup[2],
            dense_idx=dense[0], dense_scr=dense[1], sparse_idx=sparse[0], sparse_scr=sparse[1],
            out_idx=merged.indices, out_scr=merged.scores, out_lbl=merged.labels,
            raw_dense=raw["dense"], raw_sparse=raw["sparse"],
        )

    # two-engine merges in the style of the reference's own test (float64 scores, bool labels)
    arrays = {}
    params = []
    n = 0
    for seed in range(10):
        for seq_length in (10, 30):
            for n_values in (300, 1000):
                rgn = np.random.default_rng(seed)
                alen = seq_length // 2
                blen = seq_length - alen
                ids = np.arange(n_values)
                a_idx = rgn.choice(ids, size=(alen,), replace=False)
                b_idx = rgn.choice(ids, size=(blen,), replace=False)
                a_scr = rgn.uniform(0.0, 10.0, size=(1, alen))
                b_scr = rgn.uniform(0.0, 10.0, size=(1, blen))
                wa, wb = float(rgn.uniform(0.0, 1.0)), float(rgn.uniform(0.0, 1.0))
                res = {"a": RB(scores=a_scr.copy(), indices=a_idx[None, :].copy()), "b": RB(scores=b_scr.copy(), indices=b_idx[None, :].copy())}
                merged, raw = M["merge"].merge_search_results(res, {"a": wa, "b": wb})
                arrays.update({
                    f"a_idx_{n}": a_idx[None, :], f"a_scr_{n}": a_scr, f"b_idx_{n}": b_idx[None, :], f"b_scr_{n}": b_scr,
                    f"w_{n}": np.array([wa, wb]), f"out_idx_{n}": merged.indices, f"out_scr_{n}": merged.scores,
                    f"raw_a_{n}": raw["a"], f"raw_b_{n}": raw["b"],
                })
                params.append({"seed": seed, "seq_length": seq_length, "n_values": n_values})
                n += 1
    _save("merge_two_engines", {"cases": params}, **arrays)


def gen_merge_corners() -> None:
    """Corner cases a randomised campaign (tests/fuzz/fuzz_collate.py) found the oracle restating differently from the reference:
    ids repeated INSIDE one engine's row (different labels / scores per occurrence) and NaN scores, with 0, 1 and 2 scored
    engines.  With the lookup alone `merge_search_results` returns it untouched (merge.py:18-22): no union, labels as given."""
    rng = np.random.default_rng(4242)
    nq, n_ids = 6, 9

    def eng(k, labels):
        idx = rng.integers(0, n_ids, size=(nq, k)).astype(np.int64)   # many repeats
        scr = (rng.normal(size=(nq, k)) * 3).astype(np.float32)
        pad = rng.uniform(size=(nq, k)) < 0.2
        idx[pad], scr[pad] = -1, -np.inf
        scr[0, 1] = np.nan
        lbl = None
        if labels:
            lbl = rng.integers(1, 4, size=(nq, k)).astype(np.int64)
            lbl[pad] = 0
        return idx, scr, lbl

    lookup = eng(7, True)
    dense = eng(8, False)
    sparse = eng(5, False)
    arrays = dict(lookup_idx=lookup[0], lookup_scr=lookup[1], lookup_lbl=lookup[2], dense_idx=dense[0], dense_scr=dense[1],
                  sparse_idx=sparse[0], sparse_scr=sparse[1])
    cases = {"lookup_only": {}, "one_engine": {"dense": 0.5}, "two_engines": {"dense": 1.0, "sparse": 2.0}}
    for name, w in cases.items():
        res = {"lookup": RB(scores=lookup[1].copy(), indices=lookup[0].copy(), labels=lookup[2].copy())}
        if "dense" in w:
            res["dense"] = RB(scores=dense[1].copy(), indices=dense[0].copy())
        if "sparse" in w:
            res["sparse"] = RB(scores=sparse[1].copy(), indices=sparse[0].copy())
        merged, raw = M["search"]._merge_search_results(res, dict(w))
        arrays[f"{name}_out_idx"], arrays[f"{name}_out_scr"], arrays[f"{name}_out_lbl"] = merged.indices, merged.scores, merged.labels
        for e, r in raw.items():
            arrays[f"{name}_raw_{e}"] = r
    _save("merge_corners", {"cases": cases}, **arrays)


# ----------------------------------------------------------------------------------------
def gen_normalize() -> None:
    arrays = {}
    params = []
    n = 0
    for dtype in ("float32", "float64"):
        for seed in range(4):
            for nan_prob in (0.0, 0.1, 0.3):
                for inf_prob in (0.0, 0.3, 1.0):
                    rgn = np.random.default_rng(seed)
                    scores = rgn.uniform(0.0, 10.0, size=(6, 50))
                    scores = np.where(rgn.uniform(0.0, 1.0, size=scores.shape) < nan_prob, np.nan, scores)
                    scores = np.where(rgn.uniform(0.0, 1.0, size=scores.shape) < inf_prob, -np.inf, scores)
                    scores = scores.astype(dtype)
                    arrays[f"in_{n}"] = scores
                    for off in (0.0, 1.0, -10.0):
                        arrays[f"out_{n}_{off}"] = M["normalize"]._subtract_min_score(scores.copy(), offset=off)
                    params.append({"seed": seed, "nan_prob": nan_prob, "inf_prob": inf_prob, "dtype": dtype})
                    n += 1
    _save("normalize", {"cases": params, "offsets": [0.0, 1.0, -10.0]}, **arrays)


def gen_gather() -> None:
    rng = np.random.default_rng(7)
    npo = M["numpy_ops"]
    keys2 = np.stack([rng.choice(50, size=12, replace=False) for _ in range(5)]).astype(np.int64)
    keys2[1, 9:] = -1
    vals2 = rng.normal(size=keys2.shape).astype(np.float32)
    q2 = rng.integers(-1, 50, size=(5, 20)).astype(np.int64)
    lbl2 = rng.integers(0, 2, size=keys2.shape).astype(np.int64)
    out = {
        "q2": q2, "keys2": keys2, "vals2": vals2, "lbl2": lbl2,
        "out_2d_f32": npo.gather_values_by_indices(q2.copy(), keys2.copy(), vals2.copy()),
        "out_2d_lbl": npo.gather_values_by_indices(q2.copy(), keys2.copy(), lbl2.copy(), fill_value=-1),
        "out_2d_lbl_default": npo.gather_values_by_indices(q2.copy(), keys2.copy(), lbl2.copy()),
        "out_1d": npo.gather_values_by_indices(q2[0].copy(), keys2[0].copy(), vals2[0].copy()),
        "out_2d_from_1d": npo.gather_values_by_indices(q2.copy(), keys2[0].copy(), vals2[0].copy()),
    }
    # duplicate keys: first match wins (numpy_ops.py:31-36)
    kd = np.array([[3, 5, 3, 9]], dtype=np.int64)
    vd = np.array([[1.0, 2.0, 3.0, 4.0]], dtype=np.float32)
    qd = np.array([[3, 9, 4]], dtype=np.int64)
    out.update({"kd": kd, "vd": vd, "qd": qd, "out_dup": npo.gather_values_by_indices(qd.copy(), kd.copy(), vd.copy())})
    _save("gather", {}, **out)


# ----------------------------------------------------------------------------------------
def gen_sampling() -> None:
    smp = M["sample"]
    arrays = {}
    params = []
    n = 0
    for seed, (nq, nc, k_pos, k_tot, support, temp, pos_frac, inf_frac) in enumerate([
        (6, 64, 8, 32, -1, 1.0, 0.2, 0.1),
        (6, 64, 8, 32, 40, 1.0, 0.2, 0.1),
        (6, 64, 8, 32, -1, 0.0, 0.2, 0.1),
        (6, 64, 8, 32, 40, 0.0, 0.2, 0.0),
        (4, 20, 4, 32, -1, 1.0, 0.5, 0.3),   # fewer candidates than k_total
        (4, 48, 8, 16, -1, 1.0, 0.0, 0.2),   # no positives
        (4, 48, 8, 16, -1, 2.0, 1.0, 0.0),   # only positives
        (8, 385, 8, 32, 100, 1.0, 0.02, 0.15),  # shipped train config shape (support 100)
    ]):
        rng = np.random.default_rng(500 + seed)
        scores = (rng.normal(size=(nq, nc)) * 2).astype(np.float32)
        scores[rng.uniform(size=scores.shape) < inf_frac] = -np.inf
        labels = rng.uniform(size=scores.shape) < pos_frac
        noise = rng.exponential(size=scores.shape).astype(np.float32)
        samples = np.full((nq, k_tot), -1, dtype=np.int64)
        logw = np.full((nq, k_tot), -np.inf, dtype=np.float32)
        olab = np.zeros((nq, k_tot), dtype=np.bool_)
        lse = np.zeros((nq, 2), dtype=np.float32)
        with np.errstate(all="ignore"):
            smp._labeled_priority_sampling_2d_(
                scores.copy(), labels.copy(), noise.copy(), k_pos, k_tot,
                out_samples_=samples, out_log_weights_=logw, out_labels_=olab, out_lse_=lse,
                normalized=True, temperature=temp, max_support_size=support,
            )
        arrays.update({
            f"scores_{n}": scores, f"labels_{n}": labels, f"noise_{n}": noise,
            f"out_samples_{n}": samples, f"out_logw_{n}": logw, f"out_labels_{n}": olab, f"out_lse_{n}": lse,
        })
        params.append({"k_positive": k_pos, "k_total": k_tot, "max_support_size": support, "temperature": temp})
        n += 1
    _save("sampling_fixed_noise", {"cases": params}, **arrays)

    # in-batch negative flattening
    rng = np.random.default_rng(77)
    idx = rng.integers(0, 30, size=(4, 6)).astype(np.int64)
    for i in range(4):  # unique per row
        idx[i] = rng.choice(30, size=6, replace=False)
    scr = rng.normal(size=idx.shape).astype(np.float32)
    lbl = rng.uniform(size=idx.shape) < 0.3
    logw = rng.normal(size=idx.shape).astype(np.float32)
    raw = {"dense": rng.normal(size=idx.shape).astype(np.float32), "sparse": rng.normal(size=idx.shape).astype(np.float32)}
    ps = smp.PrioritySampledSections(
        batch=RB(scores=scr, indices=idx, labels=lbl), log_weights=logw, max_sampling_id=np.zeros(4),
        lse_pos=np.zeros(4, dtype=np.float32), lse_neg=np.zeros(4, dtype=np.float32), raw_scores=raw,
    )
    flat = M["in_batch_negatives"].flatten_samples(ps, padding=True)
    _save(
        "flatten_inbatch", {},
        idx=idx, scr=scr, lbl=lbl, logw=logw, raw_dense=raw["dense"], raw_sparse=raw["sparse"],
        out_idx=flat.batch.indices, out_scr=flat.batch.scores, out_lbl=flat.batch.labels, out_logw=flat.log_weights,
        out_raw_dense=flat.raw_scores["dense"], out_raw_sparse=flat.raw_scores["sparse"],
    )


# ----------------------------------------------------------------------------------------
# the collate chain end to end: merge -> sample_search_results -> flatten_samples (realm_collate.py:110-139)
# ----------------------------------------------------------------------------------------
def gen_collate_chain() -> None:
    """The reference's own three functions chained as `RealmCollate.__call__` chains them, with `np.random` seeded so that
    the Exp(1) draw inside `labeled_priority_sampling` (sample.py:398) is reproducible: the fixture stores that draw
    (`noise`, re-drawn with the same seed and shape) next to every intermediate and final array."""
    smp = M["sample"]
    cases = [
        # nq, k per engine, id pool, pad_frac, overlap, total, max_pos, temperature, support, weights
        (6, 16, 60, 0.3, 0.4, 8, 2, 1.0, None, {"dense": 1.0, "sparse": 0.5}),
        (5, 24, 40, 0.2, 0.6, 16, 4, 0.0, None, {"dense": 1.0, "sparse": 1.0}),
        (12, 128, 3000, 0.1, 0.3, 32, 8, 1.0, 100, {"dense": 1.0, "sparse": 1.0}),   # shipped training shape (C5): W ~ 385, support 100
        (4, 12, 30, 0.5, 0.5, 48, 6, 1.0, None, {"dense": 0.0, "sparse": 1.0}),      # fewer candidates than `total`
    ]
    params = []
    arrays = {}
    for c, (nq, kk, n_ids, pad_frac, overlap, total, max_pos, temp, support, weights) in enumerate(cases):
        rng = np.random.default_rng(900 + c)
        lookup, dense, sparse = _hybrid_case(rng, nq, kk, n_ids, pad_frac, overlap)
        with np.errstate(all="ignore"):
            merged, raw = _run_hybrid(lookup, dense, sparse, weights)
            seed = 4000 + c
            np.random.seed(seed)
            noise = np.random.exponential(size=merged.scores.shape).astype(merged.scores.dtype)
            np.random.seed(seed)
            sampled = smp.sample_search_results(search_results=merged, raw_scores=raw, total=total, max_pos_sections=max_pos,
                                                temperature=temp, max_support_size=support)
            flat = M["in_batch_negatives"].flatten_samples(sampled, padding=True)
        arrays.update({
            f"l_idx_{c}": lookup[0], f"l_lbl_{c}": lookup[2], f"d_idx_{c}": dense[0], f"d_scr_{c}": dense[1],
            f"s_idx_{c}": sparse[0], f"s_scr_{c}": sparse[1], f"noise_{c}": noise,
            f"m_idx_{c}": merged.indices, f"m_scr_{c}": merged.scores, f"m_lbl_{c}": merged.labels,
            f"smp_idx_{c}": sampled.batch.indices, f"smp_scr_{c}": sampled.batch.scores, f"smp_lbl_{c}": sampled.batch.labels,
            f"smp_logw_{c}": sampled.log_weights, f"smp_lse_pos_{c}": sampled.lse_pos, f"smp_lse_neg_{c}": sampled.lse_neg,
            f"smp_max_id_{c}": sampled.max_sampling_id, f"smp_dense_{c}": sampled.raw_scores["dense"], f"smp_sparse_{c}": sampled.raw_scores["sparse"],
            f"flat_idx_{c}": flat.batch.indices, f"flat_scr_{c}": flat.batch.scores, f"flat_lbl_{c}": flat.batch.labels,
            f"flat_logw_{c}": flat.log_weights, f"flat_dense_{c}": flat.raw_scores["dense"], f"flat_sparse_{c}": flat.raw_scores["sparse"],
        })
        params.append({"total": total, "max_pos_sections": max_pos, "temperature": temp, "max_support_size": support, "weights": weights,
                       "seed": seed})
    _save("collate_chain", {"cases": params}, **arrays)


# ----------------------------------------------------------------------------------------
def gen_search_plumbing() -> None:
    ss = M["sharded_search"]
    base = M["base"]

    class _Fake(base.SearchClient):
        def __init__(self, table):
            self.table = table

        def ping(self):
            return True

        def search(self, *, text, vector=None, subset_ids=None, ids=None, shard=None, top_k=3):  # noqa: ARG002
            n = len(text)
            return RB(scores=self.table[0][:n, :top_k].copy(), indices=self.table[1][:n, :top_k].copy())

    ta = (np.array([[3.0, 2.0], [1.5, -np.inf], [0.5, 0.25]], dtype=np.float32), np.array([[0, 5], [7, -1], [2, 3]], dtype=np.int64))
    tb = (np.array([[9.0, 8.0, 7.0], [6.0, 5.0, 4.0]], dtype=np.float32), np.array([[1, 2, -1], [4, 0, 6]], dtype=np.int64))
    client = ss.ShardedSearchClient(shards={"a": _Fake(ta), "b": _Fake((tb[0][:, :2], tb[1][:, :2]))}, offsets={"a": 0, "b": 100})
    shard = ["a", "b", "a", "b", "a"]
    out = client.search(text=[""] * 5, vector=np.zeros((5, 4), dtype=np.float32), shard=shard, top_k=2)
    _save(
        "shard_scatter_gather", {"shard": shard, "offsets": {"a": 0, "b": 100}, "top_k": 2},
        a_scr=ta[0], a_idx=ta[1], b_scr=tb[0][:, :2], b_idx=tb[1][:, :2], out_scr=out.scores, out_idx=out.indices,
    )
    # ragged stack
    RS = M["retrieval"].RetrievalSample
    rows = [
        RS(scores=np.array([1.0, 0.5, 0.25], dtype=np.float32), indices=np.array([4, 5, 6], dtype=np.int64)),
        RS(scores=np.array([2.0], dtype=np.float32), indices=np.array([9], dtype=np.int64)),
        RS(scores=np.array([], dtype=np.float32), indices=np.array([], dtype=np.int64)),
    ]
    st = RB.stack_samples(rows)
    _save("stack_samples_ragged", {}, r0_s=rows[0].scores, r0_i=rows[0].indices, r1_s=rows[1].scores, r1_i=rows[1].indices,
          out_scr=st.scores, out_idx=st.indices)

    # wire codec: exact strings
    io = M["io"]
    f = np.arange(6, dtype=np.float32).reshape(2, 3) / 4
    i = np.array([[1, -1, 7], [0, 2**40, 3]], dtype=np.int64)
    h = np.arange(8, dtype=np.float16).reshape(2, 4)
    codec = {"f32": io.serialize_np_array(f), "i64": io.serialize_np_array(i), "f16": io.serialize_np_array(h)}
    (HERE / "io_codec.json").write_text(json.dumps(codec, indent=1))
    _save("io_codec", {"strings": "io_codec.json"}, f32=f, i64=i, f16=h)


# ----------------------------------------------------------------------------------------
def gen_gradients() -> None:
    import torch

    grad = M["gradients"]
    RealmBatch = M["batch"].RealmBatch
    fn = grad.RetrievalGradients()

    def run(name, nq, nd, h, three_d, pad_frac, nopos_row, seed):
        g = torch.Generator().manual_seed(seed)
        q = torch.randn(nq, h, generator=g, dtype=torch.float32, requires_grad=True)
        s = torch.randn(*((nq, nd, h) if three_d else (nd, h)), generator=g, dtype=torch.float32, requires_grad=True)
        score = torch.randn(nq, nd, generator=g)
        pad = torch.rand(nq, nd, generator=g) < pad_frac
        pad[:, 0] = False
        score = score.masked_fill(pad, -float("inf"))
        rel = (torch.rand(nq, nd, generator=g) < 0.25).long()
        rel[:, 0] = 1
        if nopos_row:
            rel[1, :] = 0
        sparse = torch.randn(nq, nd, generator=g).masked_fill(torch.rand(nq, nd, generator=g) < 0.2, float("nan"))
        dense = torch.randn(nq, nd, generator=g).masked_fill(torch.rand(nq, nd, generator=g) < 0.2, float("nan"))
        dummy = torch.zeros(1, dtype=torch.long)
        batch = RealmBatch(
            query__input_ids=dummy, query__attention_mask=dummy, query__id="", query__subset_ids=[], query__section_ids=[],
            section__input_ids=dummy, section__attention_mask=dummy, section__id="",
            section__relevance=rel, section__idx=torch.zeros(nq, nd, dtype=torch.long), section__score=score,
            section__sparse=sparse, section__dense=dense, section__log_weight=torch.zeros(nq, nd),
            section__lse_pos=torch.zeros(nq), section__lse_neg=torch.zeros(nq),
        )
        out = fn(batch=batch, query_encoding=q, section_encoding=s)
        dq, ds = torch.autograd.grad(out.loss, [q, s])
        _save(
            name, {"three_d": three_d, "seed": seed},
            q=q.detach().numpy(), s=s.detach().numpy(), score=score.numpy(), relevance=rel.numpy(),
            sparse=sparse.numpy(), dense=dense.numpy(),
            loss=out.loss.detach().numpy(), retriever_scores=out.retriever_scores.detach().numpy(),
            dq=dq.numpy(), ds=ds.numpy(),
            kl_score=out.diagnostics["kl_score"].numpy(), kl_sparse=out.diagnostics["kl_sparse"].numpy(),
            kl_dense=out.diagnostics["kl_dense"].numpy(),
        )

    run("retrieval_grad_2d", 4, 8, 16, False, 0.0, False, 11)
    run("retrieval_grad_3d", 4, 8, 16, True, 0.0, False, 12)
    run("retrieval_grad_nopos", 5, 12, 32, False, 0.2, True, 13)
    run("retrieval_grad_padded", 6, 10, 64, True, 0.4, False, 14)
    run("retrieval_grad_inbatch", 16, 96, 128, False, 0.1, False, 15)


def gen_gradients_aux() -> None:
    """The auxiliary losses of RetrievalGradients (retrieval.py:94-150), one fixture per term and one with all three."""
    import torch

    grad = M["gradients"]
    RealmBatch = M["batch"].RealmBatch

    def run(name, cfg, nq, nd, h, three_d, pad_frac, nopos_row, seed):
        fn = grad.RetrievalGradients(**cfg)
        g = torch.Generator().manual_seed(seed)
        q = (0.3 * torch.randn(nq, h, generator=g, dtype=torch.float32)).requires_grad_(True)
        s = torch.randn(*((nq, nd, h) if three_d else (nd, h)), generator=g, dtype=torch.float32, requires_grad=True)
        score = torch.randn(nq, nd, generator=g)
        pad = torch.rand(nq, nd, generator=g) < pad_frac
        pad[:, 0] = False
        score = score.masked_fill(pad, -float("inf"))
        rel = (torch.rand(nq, nd, generator=g) < 0.25).long()
        rel[:, 0] = 1
        if nopos_row:
            rel[1, :] = 0
        sparse = (-2.0 + torch.randn(nq, nd, generator=g)).masked_fill(torch.rand(nq, nd, generator=g) < 0.2, float("nan"))
        dense = torch.randn(nq, nd, generator=g).masked_fill(torch.rand(nq, nd, generator=g) < 0.2, float("nan"))
        dummy = torch.zeros(1, dtype=torch.long)
        batch = RealmBatch(
            query__input_ids=dummy, query__attention_mask=dummy, query__id="", query__subset_ids=[], query__section_ids=[],
            section__input_ids=dummy, section__attention_mask=dummy, section__id="",
            section__relevance=rel, section__idx=torch.zeros(nq, nd, dtype=torch.long), section__score=score,
            section__sparse=sparse, section__dense=dense, section__log_weight=torch.zeros(nq, nd),
            section__lse_pos=torch.zeros(nq), section__lse_neg=torch.zeros(nq),
        )
        out = fn(batch=batch, query_encoding=q, section_encoding=s)
        dq, ds = torch.autograd.grad(out.loss, [q, s])
        diag = {k: v.detach().numpy() for k, v in out.diagnostics.items()}
        _save(
            name, {"three_d": three_d, "seed": seed, "config": cfg, "diagnostic_keys": list(out.diagnostics)},
            q=q.detach().numpy(), s=s.detach().numpy(), score=score.numpy(), relevance=rel.numpy(),
            sparse=sparse.numpy(), dense=dense.numpy(),
            loss=out.loss.detach().numpy(), retriever_scores=out.retriever_scores.detach().numpy(),
            dq=dq.numpy(), ds=ds.numpy(), **{f"diag_{k}": v for k, v in diag.items()},
        )

    run("retrieval_aux_guidance_sparse", {"guidance": "sparse", "guidance_weight": 0.5}, 5, 12, 32, False, 0.2, False, 31)
    run("retrieval_aux_guidance_zero", {"guidance": "zero", "guidance_weight": 0.25}, 4, 9, 16, True, 0.3, False, 32)
    run("retrieval_aux_self_supervision", {"self_supervision_weight": 0.7}, 6, 10, 24, False, 0.2, False, 33)
    run("retrieval_aux_score_decay", {"score_decay": 0.01}, 5, 11, 16, True, 0.3, False, 34)
    run("retrieval_aux_all", {"guidance": "sparse", "guidance_weight": 0.3, "self_supervision_weight": 0.4, "score_decay": 0.02},
        16, 96, 64, False, 0.15, False, 35)
    run("retrieval_aux_all_nopos", {"guidance": "zero", "guidance_weight": 0.3, "self_supervision_weight": 0.4, "score_decay": 0.02},
        5, 12, 32, False, 0.2, True, 36)


def gen_metrics() -> None:
    """`metrics_recall_ndcg`: the reference's retrieval metrics on seeded [B, N] scores / graded relevances with NaN, +inf and
    -inf scores, rows without positives and a row whose positives are all masked; topk in {None, 1, 5, 10}."""
    import torch

    F = M["functional"]
    rng = np.random.default_rng(4242)
    B, N = 12, 24
    scores = rng.normal(size=(B, N)).astype(np.float32)
    rel = (rng.random((B, N)) < 0.2).astype(np.int64) * rng.integers(1, 4, size=(B, N))   # graded relevances 0..3
    scores[1, :5] = np.nan
    scores[2, 3] = np.inf        # masked by the reference (+inf), its relevance zeroed
    scores[3, 10:] = -np.inf     # padding: ranked last, NOT masked (relevance 0, as pads have: equal scores are ranked in an
    rel[3, 10:] = 0              # implementation-defined order by the reference's unstable argsort, so tied entries carry equal relevances)
    rel[4] = 0                   # a row without positives
    rel[5] = 0
    rel[5, 2] = 2
    scores[5, 2] = np.nan        # the only positive is masked
    scores[6, :4] = scores[6, 4]  # ties
    rel[6, :5] = rel[6, 4]
    arrays = {"scores": scores, "relevances": rel}
    topks = [0, 1, 5, 10]  # 0 = None
    for name in ("recall", "ndcg", "mrr", "hitrate", "precision"):
        fn = getattr(F, f"compute_{name}")
        for tk in topks:
            out = fn.compute(relevances=torch.from_numpy(rel), scores=torch.from_numpy(scores), topk=tk or None)
            arrays[f"{name}_top{tk}"] = out.to(torch.float64).numpy() if out.dtype != torch.bool else out.numpy()
    _save("metrics_recall_ndcg", {"seed": 4242, "B": B, "N": N, "topk": topks, "fn": "vod_models.monitoring.functional.compute_*"}, **arrays)


def gen_flat_ip() -> None:
    """Build-owned (NOT from the reference: faiss is absent) exact fixtures: small-integer fp16 vectors."""
    sys.path.insert(0, str(HERE.parent.parent))
    from oracle.flat_ip import flat_ip_topk

    for name, n, d, nq, k, seed in [("flat_ip_exact_small", 4096, 64, 16, 10, 21), ("flat_ip_exact_768", 20000, 768, 64, 100, 22)]:
        rng = np.random.default_rng(seed)
        x = rng.integers(-8, 9, size=(n, d)).astype(np.float16)
        q = rng.integers(-8, 9, size=(nq, d)).astype(np.float16)
        scores, ids = flat_ip_topk(q, x, k)
        _save(name, {"seed": seed, "n": n, "d": d, "nq": nq, "k": k, "source": "build-owned fp64 restatement; parity with faiss unpinned"},
              out_scores=scores, out_ids=ids)


if __name__ == "__main__":
    if sys.argv[1:] == ["gradients_aux"]:  # add the round-2 fixtures without touching the others
        manifest.update(json.loads((HERE / "manifest.json").read_text()))
        gen_gradients_aux()
        (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1, sort_keys=True))
        raise SystemExit(0)
    if sys.argv[1:] == ["collate_chain"]:
        manifest.update(json.loads((HERE / "manifest.json").read_text()))
        gen_collate_chain()
        (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1, sort_keys=True))
        raise SystemExit(0)
    if sys.argv[1:] == ["metrics"]:
        manifest.update(json.loads((HERE / "manifest.json").read_text()))
        gen_metrics()
        (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1, sort_keys=True))
        raise SystemExit(0)
    if sys.argv[1:] == ["merge_corners"]:
        manifest.update(json.loads((HERE / "manifest.json").read_text()))
        gen_merge_corners()
        (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1, sort_keys=True))
        raise SystemExit(0)
    gen_merge()
    gen_merge_corners()
    gen_normalize()
    gen_gather()
    gen_sampling()
    gen_collate_chain()
    gen_search_plumbing()
    gen_gradients()
    gen_gradients_aux()
    gen_metrics()
    gen_flat_ip()
    (HERE / "manifest.json").write_text(json.dumps(manifest, indent=1, sort_keys=True))
    total = sum(p.stat().st_size for p in HERE.glob("*.npz"))
    print(f"wrote {len(manifest)} fixtures, {total/1024:.0f} KiB")
