"""GPU parity tests for the hybrid score merge (H4): bit-exact against the golden vectors captured from
the reference's own modules and against the CPU oracle on seeded random cases, through the C-ABI."""
import json

import numpy as np
import pytest

from conftest import GOLDEN

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

MANIFEST = json.loads((GOLDEN / "manifest.json").read_text())


def _eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert a.dtype == b.dtype, (a.dtype, b.dtype)
    assert np.array_equal(a, b, equal_nan=(a.dtype.kind == "f"))


@pytest.mark.parametrize("name", ["merge_3engine_basic"] + [f"merge_random_{i}" for i in range(8)])
def test_merge_matches_reference_golden(name):
    from vod_amd.core.merge import merge_hybrid

    g = np.load(GOLDEN / f"{name}.npz")
    w = MANIFEST[name]["params"]["weights"]
    idx, scr, lbl, raw = merge_hybrid(
        g["lookup_idx"], g["lookup_lbl"],
        {"dense": (g["dense_idx"], g["dense_scr"]), "sparse": (g["sparse_idx"], g["sparse_scr"])},
        {"dense": w["dense"], "sparse": w["sparse"]},
    )
    _eq(idx, g["out_idx"])
    _eq(scr, g["out_scr"])
    _eq(lbl, g["out_lbl"])
    _eq(raw["dense"], g["raw_dense"])
    _eq(raw["sparse"], g["raw_sparse"])


@pytest.mark.parametrize("case", ["lookup_only", "one_engine", "two_engines"])
def test_merge_corner_cases_match_reference_golden(case):
    """Repeated ids inside an engine's row, NaN scores, the lookup alone (returned untouched): reference-generated."""
    from vod_amd.core.merge import merge_hybrid

    g = np.load(GOLDEN / "merge_corners.npz")
    w = MANIFEST["merge_corners"]["params"]["cases"][case]
    engines = {n: (g[f"{n}_idx"], g[f"{n}_scr"]) for n in w}
    idx, scr, lbl, raw = merge_hybrid(g["lookup_idx"], g["lookup_lbl"], engines, dict(w))
    _eq(idx, g[f"{case}_out_idx"])
    _eq(scr, g[f"{case}_out_scr"])
    _eq(lbl, g[f"{case}_out_lbl"])
    for n in w:
        _eq(raw[n], g[f"{case}_raw_{n}"])


def test_collate_entry_point_matches_reference_golden():
    """`merge_search_results` (the `_merge_search_results` mirror) on RetrievalBatch inputs."""
    from vod_amd import types as vt
    from vod_amd.core.search import merge_search_results

    g = np.load(GOLDEN / "merge_random_5.npz")
    w = MANIFEST["merge_random_5"]["params"]["weights"]
    res = {
        "lookup": vt.RetrievalBatch(scores=g["lookup_scr"].copy(), indices=g["lookup_idx"], labels=g["lookup_lbl"], meta={"time": 1.0}),
        "dense": vt.RetrievalBatch(scores=g["dense_scr"].copy(), indices=g["dense_idx"]),
        "sparse": vt.RetrievalBatch(scores=g["sparse_scr"].copy(), indices=g["sparse_idx"]),
    }
    merged, raw = merge_search_results(res, w)
    _eq(merged.indices, g["out_idx"])
    _eq(merged.scores, g["out_scr"])
    _eq(merged.labels, g["out_lbl"])
    _eq(raw["dense"], g["raw_dense"])
    assert set(raw) == {"dense", "sparse"} and merged.meta == {"lookup_time": 1.0}
    with pytest.raises(ValueError):
        merge_search_results({"dense": res["dense"]}, w)


def _random_case(rng, nq, kl, ks, n_ids, pad_frac, dup):
    def eng(k, scored=True):
        idx = np.full((nq, k), -1, dtype=np.int64)
        scr = np.full((nq, k), -np.inf, dtype=np.float32)
        for r in range(nq):
            nv = k if rng.uniform() > pad_frac else int(rng.integers(0, k + 1))
            idx[r, :nv] = rng.choice(n_ids, size=nv, replace=dup)
            scr[r, :nv] = rng.normal(size=nv).astype(np.float32) * 5
        return idx, scr

    l_idx, l_scr = eng(kl)
    l_lbl = (l_scr > -np.inf).astype(np.int64) * rng.integers(1, 3, size=l_scr.shape)
    return l_idx, l_lbl, [eng(k) for k in ks]


@pytest.mark.parametrize(
    "nq,kl,ks,n_ids,pad,dup",
    [
        (64, 128, [128, 128], 1_000_000, 0.1, False),   # C5: B=64, K=128 per engine
        (64, 128, [128, 128], 300, 0.1, False),         # heavy overlap
        (7, 3, [5], 10, 0.5, True),                     # one engine, duplicate ids inside an engine
        (5, 0, [16, 16], 40, 0.2, False),               # empty lookup
        (9, 8, [32, 16, 8, 4], 60, 0.3, False),         # four engines
        (3, 6, [], 20, 0.0, False),                     # lookup only
        (4, 4, [0, 6], 20, 0.0, False),                 # an engine that returned nothing
        (2, 512, [1024, 1024], 5000, 0.05, False),      # wide rows
    ],
)
def test_merge_matches_oracle_random(nq, kl, ks, n_ids, pad, dup):
    from oracle.hybrid import merge_hybrid as oracle_merge
    from vod_amd.core.merge import merge_hybrid

    rng = np.random.default_rng(nq * 1000 + kl + len(ks))
    l_idx, l_lbl, engs = _random_case(rng, nq, kl, ks, n_ids, pad, dup)
    names = [f"e{i}" for i in range(len(engs))]
    weights = {n: float(w) for n, w in zip(names, [1.0, 0.5, 0.0, 1.7])}
    engines = {n: e for n, e in zip(names, engs)}
    idx, scr, lbl, raw = merge_hybrid(l_idx, l_lbl, engines, weights)
    o_idx, o_scr, o_lbl, o_raw = oracle_merge((l_idx, np.zeros(l_idx.shape, np.float32), l_lbl), engines, weights)
    _eq(idx, o_idx)
    _eq(scr, o_scr)
    _eq(lbl, o_lbl)
    for n in names:
        _eq(raw[n], o_raw[n])


def test_all_nonfinite_engine_rows_and_nan_scores():
    from oracle.hybrid import merge_hybrid as oracle_merge
    from vod_amd.core.merge import merge_hybrid

    l_idx = np.array([[1, -1], [2, 3]], dtype=np.int64)
    l_lbl = np.array([[1, 0], [1, 1]], dtype=np.int64)
    d_idx = np.array([[1, 5, 6], [7, 8, -1]], dtype=np.int64)
    d_scr = np.array([[np.nan, -np.inf, -np.inf], [np.nan, 2.0, -np.inf]], dtype=np.float32)
    eng = {"dense": (d_idx, d_scr)}
    for w in (1.0, 0.0):
        idx, scr, lbl, raw = merge_hybrid(l_idx, l_lbl, eng, {"dense": w})
        o = oracle_merge((l_idx, np.zeros(l_idx.shape, np.float32), l_lbl), eng, {"dense": w})
        _eq(idx, o[0]); _eq(scr, o[1]); _eq(lbl, o[2]); _eq(raw["dense"], o[3]["dense"])  # noqa: E702
