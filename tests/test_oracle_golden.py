"""The CPU oracle must reproduce every golden vector captured from the reference's own modules."""
import json

import numpy as np
import pytest

from oracle import gradients as ograd
from oracle import hybrid as ohyb
from oracle import sampling as osmp
from oracle.flat_ip import flat_ip_topk, merge_shard_topk, topk_desc_tiebreak

from conftest import GOLDEN


def _load(name):
    return np.load(GOLDEN / f"{name}.npz")


def _eq(a, b):
    """Exact equality with NaN == NaN."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype.kind == "f":
        assert np.array_equal(a, b, equal_nan=True)
    else:
        assert np.array_equal(a, b)


MANIFEST = json.loads((GOLDEN / "manifest.json").read_text())


@pytest.mark.parametrize("name", ["merge_3engine_basic"] + [f"merge_random_{i}" for i in range(8)])
def test_hybrid_merge_matches_reference(name):
    g = _load(name)
    w = MANIFEST[name]["params"]["weights"]
    idx, scr, lbl, raw = ohyb.merge_hybrid(
        (g["lookup_idx"], g["lookup_scr"], g["lookup_lbl"]),
        {"dense": (g["dense_idx"], g["dense_scr"]), "sparse": (g["sparse_idx"], g["sparse_scr"])},
        {"dense": w["dense"], "sparse": w["sparse"]},
    )
    _eq(idx, g["out_idx"])
    _eq(scr, g["out_scr"])
    _eq(lbl, g["out_lbl"])
    _eq(raw["dense"], g["raw_dense"])
    _eq(raw["sparse"], g["raw_sparse"])


def test_3engine_layout_quirk():
    """first-seen order, trailing pad column with label 0 / raw dense NaN / raw sparse -inf."""
    g = _load("merge_3engine_basic")
    assert g["out_idx"].tolist() == [[4, 2, 8, 6, -1]]
    assert g["out_lbl"].tolist() == [[1, -1, -1, -1, 0]]
    assert np.isneginf(g["out_scr"][0, -1]) and np.isnan(g["raw_dense"][0, -1]) and np.isneginf(g["raw_sparse"][0, -1])


def test_two_engine_merges_match_reference():
    g = _load("merge_two_engines")
    n = len(MANIFEST["merge_two_engines"]["params"]["cases"])
    assert n == 40
    for c in range(n):
        wa, wb = g[f"w_{c}"]
        s, i, lab, raw = ohyb.merge_search_results(
            {"a": (g[f"a_scr_{c}"], g[f"a_idx_{c}"], None), "b": (g[f"b_scr_{c}"], g[f"b_idx_{c}"], None)},
            {"a": float(wa), "b": float(wb)},
        )
        _eq(i, g[f"out_idx_{c}"])
        _eq(s, g[f"out_scr_{c}"])
        _eq(raw["a"], g[f"raw_a_{c}"])
        _eq(raw["b"], g[f"raw_b_{c}"])
        assert lab is None


@pytest.mark.parametrize("case", ["lookup_only", "one_engine", "two_engines"])
def test_hybrid_merge_corner_cases_match_reference(case):
    """Ids repeated inside one engine's row (different labels / scores per occurrence), NaN scores, and the lookup alone -
    which the reference returns untouched (merge.py:18-22).  Found by tests/fuzz/fuzz_collate.py; fixture from the reference."""
    g = _load("merge_corners")
    w = MANIFEST["merge_corners"]["params"]["cases"][case]
    engines = {n: (g[f"{n}_idx"], g[f"{n}_scr"]) for n in w}
    idx, scr, lbl, raw = ohyb.merge_hybrid((g["lookup_idx"], g["lookup_scr"], g["lookup_lbl"]), engines, dict(w))
    _eq(idx, g[f"{case}_out_idx"])
    _eq(scr, g[f"{case}_out_scr"])
    _eq(lbl, g[f"{case}_out_lbl"])
    assert set(raw) == set(w)
    for n in w:
        _eq(raw[n], g[f"{case}_raw_{n}"])


def test_normalize_matches_reference():
    g = _load("normalize")
    p = MANIFEST["normalize"]["params"]
    for c in range(len(p["cases"])):
        for off in p["offsets"]:
            _eq(ohyb.subtract_min_score(g[f"in_{c}"], off), g[f"out_{c}_{off}"])


def test_gather_matches_reference():
    g = _load("gather")
    _eq(ohyb.gather_values(g["q2"], g["keys2"], g["vals2"]), g["out_2d_f32"])
    _eq(ohyb.gather_values(g["q2"], g["keys2"], g["lbl2"], fill_value=-1), g["out_2d_lbl"])
    _eq(ohyb.gather_values(g["q2"], g["keys2"], g["lbl2"]), g["out_2d_lbl_default"])
    _eq(ohyb.gather_values(g["q2"][0], g["keys2"][0], g["vals2"][0]), g["out_1d"])
    _eq(ohyb.gather_values(g["q2"], g["keys2"][0], g["vals2"][0]), g["out_2d_from_1d"])
    _eq(ohyb.gather_values(g["qd"], g["kd"], g["vd"]), g["out_dup"])


def test_sampling_matches_reference():
    g = _load("sampling_fixed_noise")
    cases = MANIFEST["sampling_fixed_noise"]["params"]["cases"]
    for c, p in enumerate(cases):
        smp, logw, lab, lse = osmp.labeled_priority_sampling_2d(
            g[f"scores_{c}"], g[f"labels_{c}"], g[f"noise_{c}"], p["k_positive"], p["k_total"],
            normalized=True, temperature=p["temperature"], max_support_size=p["max_support_size"],
        )
        _eq(smp, g[f"out_samples_{c}"])
        _eq(lab, g[f"out_labels_{c}"])
        np.testing.assert_allclose(logw, g[f"out_logw_{c}"], rtol=2e-6, atol=2e-6, equal_nan=True)
        np.testing.assert_allclose(lse, g[f"out_lse_{c}"], rtol=2e-6, atol=2e-6, equal_nan=True)


def test_sampling_corrected_support_mode_of_the_oracle():
    """`keep_top=True` (SURVEY 9 Q8's corrected truncation; no reference counterpart) differs from the reference mode only in which side
    of the support threshold is masked: with a support that covers a whole class the two agree, otherwise the corrected mode samples
    from the best `support` candidates and the reference mode from the rest."""
    rng = np.random.default_rng(3)
    scores = rng.normal(size=(6, 60)).astype(np.float32)
    labels = rng.uniform(size=scores.shape) < 0.4
    noise = rng.exponential(size=scores.shape).astype(np.float32)
    a = osmp.labeled_priority_sampling_2d(scores, labels, noise, 4, 12, True, 1.0, 60, keep_top=True)
    b = osmp.labeled_priority_sampling_2d(scores, labels, noise, 4, 12, True, 1.0, 60)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)                          # support >= class size: no truncation either way
    smp, logw, lab, _ = osmp.labeled_priority_sampling_2d(scores, labels, noise, 4, 12, True, 1.0, 12, keep_top=True)
    smp_r, _, lab_r, _ = osmp.labeled_priority_sampling_2d(scores, labels, noise, 4, 12, True, 1.0, 12)
    for r in range(len(scores)):
        neg = np.flatnonzero(~labels[r])
        thr = np.sort(scores[r][neg])[-12]
        picked = smp[r][(smp[r] >= 0) & ~lab[r] & np.isfinite(logw[r])]
        assert len(picked) and np.all(scores[r][picked] >= thr)      # corrected: from the 12 best negatives
        picked_r = smp_r[r][(smp_r[r] >= 0) & ~lab_r[r]]
        assert np.all(scores[r][picked_r] < thr)                     # reference (Q8): from everything BUT them


def test_flatten_matches_reference():
    g = _load("flatten_inbatch")
    out = osmp.flatten_samples(g["idx"], g["scr"], g["lbl"], g["logw"], {"dense": g["raw_dense"], "sparse": g["raw_sparse"]})
    _eq(out["indices"], g["out_idx"])
    _eq(out["scores"], g["out_scr"])
    _eq(out["labels"], g["out_lbl"])
    _eq(out["log_weights"], g["out_logw"])
    _eq(out["raw"]["dense"], g["out_raw_dense"])
    _eq(out["raw"]["sparse"], g["out_raw_sparse"])


@pytest.mark.parametrize("name", ["retrieval_grad_2d", "retrieval_grad_3d", "retrieval_grad_nopos", "retrieval_grad_padded", "retrieval_grad_inbatch"])
def test_gradients_match_reference(name):
    g = _load(name)
    out = ograd.retrieval_gradients(g["q"], g["s"], g["score"], g["relevance"], g["sparse"], g["dense"])
    tol = dict(rtol=1e-5, atol=1e-6)  # reference ran in fp32; oracle in fp64
    np.testing.assert_allclose(out["loss"], g["loss"], **tol)
    np.testing.assert_allclose(out["retriever_scores"], g["retriever_scores"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out["dq"], g["dq"], **tol)
    np.testing.assert_allclose(out["ds"], g["ds"], **tol)
    for k in ("kl_score", "kl_sparse", "kl_dense"):
        np.testing.assert_allclose(out[k], g[k], **tol)


AUX_NAMES = ["retrieval_aux_guidance_sparse", "retrieval_aux_guidance_zero", "retrieval_aux_self_supervision",
             "retrieval_aux_score_decay", "retrieval_aux_all", "retrieval_aux_all_nopos"]


@pytest.mark.parametrize("name", AUX_NAMES)
def test_auxiliary_losses_match_reference(name):
    """RetrievalGradients._auxiliary_losses (retrieval.py:94-150): guidance (huber), self-supervision, score decay."""
    g = _load(name)
    cfg = MANIFEST[name]["params"]["config"]
    out = ograd.retrieval_gradients(g["q"], g["s"], g["score"], g["relevance"], g["sparse"], g["dense"], **cfg)
    tol = dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["loss"], g["loss"], **tol)  # NaN == NaN: a row without positives poisons the loss
    np.testing.assert_allclose(out["dq"], g["dq"], **tol)
    np.testing.assert_allclose(out["ds"], g["ds"], **tol)
    for key in MANIFEST[name]["params"]["diagnostic_keys"]:
        np.testing.assert_allclose(out[key], g[f"diag_{key}"], **tol)


@pytest.mark.parametrize("name", ["flat_ip_exact_small", "flat_ip_exact_768"])
def test_flat_ip_fixture_selfconsistent(name):
    """Build-owned fixture (faiss parity unpinned): blocked fp64 top-k == full-matrix lexsort, ties -> smaller id."""
    p = MANIFEST[name]["params"]
    g = _load(name)
    rng = np.random.default_rng(p["seed"])
    x = rng.integers(-8, 9, size=(p["n"], p["d"])).astype(np.float16)
    q = rng.integers(-8, 9, size=(p["nq"], p["d"])).astype(np.float16)
    s, i = flat_ip_topk(q, x, p["k"], block=4096)
    _eq(s, g["out_scores"])
    _eq(i, g["out_ids"])
    s2, i2 = topk_desc_tiebreak(q.astype(np.float64) @ x.astype(np.float64).T, p["k"])
    _eq(s2, s)
    _eq(i2, i)
    # ties must exist (that is what exercises the tie-break) and be ordered by id
    tie = (s[:, 1:] == s[:, :-1])
    assert tie.any()
    assert np.all(i[:, 1:][tie] > i[:, :-1][tie])


def test_shard_merge_equals_global_topk():
    rng = np.random.default_rng(3)
    x = rng.integers(-4, 5, size=(3000, 32)).astype(np.float32)
    q = rng.integers(-4, 5, size=(7, 32)).astype(np.float32)
    k = 20
    ref_s, ref_i = flat_ip_topk(q, x, k)
    parts = [flat_ip_topk(q, x[lo:hi], k, id_base=lo) for lo, hi in ((0, 1000), (1000, 1010), (1010, 3000))]
    s, i = merge_shard_topk([p[0] for p in parts], [p[1] for p in parts], k)
    _eq(s, ref_s)
    _eq(i, ref_i)


def test_flat_ip_pads_when_fewer_rows_than_k():
    x = np.eye(3, 8, dtype=np.float32)
    q = np.ones((2, 8), dtype=np.float32)
    s, i = flat_ip_topk(q, x, 5)
    assert i[:, 3:].tolist() == [[-1, -1], [-1, -1]] and np.all(np.isneginf(s[:, 3:]))
    assert i[:, :3].tolist() == [[0, 1, 2], [0, 1, 2]]


def test_cpu_baseline_port_is_an_exact_topk():
    """bench.py's `cpu_baseline` leg times this port: it must return the same ids as the fp64 oracle."""
    import torch

    from oracle.cpu_baseline import flat_ip_topk_cpu

    rng = np.random.default_rng(0)
    x = rng.normal(size=(30000, 48)).astype(np.float32)
    q = rng.normal(size=(19, 48)).astype(np.float32)
    for k, block in ((20, 4096), (100, 1000), (5, 50000)):
        s, i = flat_ip_topk_cpu(torch.from_numpy(q), torch.from_numpy(x), k, block=block)
        rs, ri = flat_ip_topk(q, x, k)
        assert np.array_equal(i.numpy(), ri)
        np.testing.assert_allclose(s.numpy(), rs, atol=1e-4)
    # ascending scores along the rows: every block floods the candidate buffer (fallback path)
    xs = x[np.argsort(x @ q[0])]
    s, i = flat_ip_topk_cpu(torch.from_numpy(q[:1]), torch.from_numpy(xs), 10, block=2048)
    rs, ri = flat_ip_topk(q[:1], xs, 10)
    assert np.array_equal(i.numpy(), ri)


# ---- the second, independent restatement of IndexFlatIP (plain C, fp32 accumulation + bounded heap: the arithmetic the
# ---- faiss CPU path uses) must agree with the NumPy fp64 oracle and with the committed flat_ip_exact_* outputs ----------


def _c_oracle():
    import ctypes

    from vod_amd.build import build_oracle

    lib = ctypes.CDLL(str(build_oracle()))
    lib.oracle_flat_ip_f32.restype = ctypes.c_int
    lib.oracle_flat_ip_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int64] * 5 + [ctypes.c_void_p, ctypes.c_void_p]

    def run(q, x, k, id_base=0):
        q = np.ascontiguousarray(q, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        s = np.empty((len(q), k), dtype=np.float32)
        i = np.empty((len(q), k), dtype=np.int64)
        rc = lib.oracle_flat_ip_f32(q.ctypes.data, x.ctypes.data, len(q), len(x), q.shape[1], k, id_base, s.ctypes.data, i.ctypes.data)
        assert rc == 0
        return s, i

    return run


@pytest.mark.parametrize("name", ["flat_ip_exact_small", "flat_ip_exact_768"])
def test_c_restatement_reproduces_the_flat_ip_fixtures(name):
    p = MANIFEST[name]["params"]
    g = _load(name)
    rng = np.random.default_rng(p["seed"])
    x = rng.integers(-8, 9, size=(p["n"], p["d"])).astype(np.float16)
    q = rng.integers(-8, 9, size=(p["nq"], p["d"])).astype(np.float16)
    s, i = _c_oracle()(q, x, p["k"])
    _eq(i, g["out_ids"])
    _eq(s, g["out_scores"])


@pytest.mark.parametrize("n,d,nq,k,id_base", [(1, 8, 2, 3, 0), (257, 72, 9, 16, 1000), (5000, 64, 33, 100, 0), (300, 32, 5, 400, 7)])
def test_c_restatement_agrees_with_the_numpy_oracle(n, d, nq, k, id_base):
    """Integer-valued data (exact in fp32 and fp64, tie-heavy): the fp32 + heap restatement and the fp64 + lexsort one are
    two independent routes to the same (score desc, id asc) answer, pads included."""
    rng = np.random.default_rng(n + d)
    x = rng.integers(-4, 5, size=(n, d)).astype(np.float32)
    q = rng.integers(-4, 5, size=(nq, d)).astype(np.float32)
    x[n // 2, 0] = np.nan  # a NaN score never enters either restatement
    s, i = _c_oracle()(q, x, k, id_base)
    rs, ri = flat_ip_topk(q, x, k, id_base=id_base)
    _eq(i, ri)
    _eq(s, rs)


def test_collate_chain_matches_reference():
    """merge -> sample_search_results -> flatten_samples, the chain `RealmCollate.__call__` runs (realm_collate.py:110-139),
    restated by the oracle and compared with what the reference's own functions produced for the same inputs and the same
    Exp(1) draw: ids / labels / -inf / NaN positions exact, float32 weights to 2e-5 (sequential vs NumPy summation order)."""
    from oracle.hybrid import merge_hybrid

    g = _load("collate_chain")
    for c, p in enumerate(MANIFEST["collate_chain"]["params"]["cases"]):
        lookup = (g[f"l_idx_{c}"], np.zeros(g[f"l_idx_{c}"].shape, np.float32), g[f"l_lbl_{c}"])
        m_idx, m_scr, m_lbl, m_raw = merge_hybrid(lookup, {"dense": (g[f"d_idx_{c}"], g[f"d_scr_{c}"]), "sparse": (g[f"s_idx_{c}"], g[f"s_scr_{c}"])},
                                                  p["weights"])
        np.testing.assert_array_equal(m_idx, g[f"m_idx_{c}"])
        np.testing.assert_array_equal(m_scr, g[f"m_scr_{c}"])
        np.testing.assert_array_equal(m_lbl, g[f"m_lbl_{c}"])
        out = osmp.sample_search_results(m_idx, m_scr, m_lbl, m_raw, g[f"noise_{c}"], p["total"], p["max_pos_sections"], p["temperature"],
                                         p["max_support_size"])
        fin = np.isfinite(g[f"smp_logw_{c}"])
        np.testing.assert_array_equal(out["indices"][fin], g[f"smp_idx_{c}"][fin])
        np.testing.assert_array_equal(out["labels"], g[f"smp_lbl_{c}"])
        np.testing.assert_array_equal(out["scores"][fin], g[f"smp_scr_{c}"][fin])
        np.testing.assert_array_equal(np.isfinite(out["log_weights"]), fin)
        np.testing.assert_allclose(out["log_weights"][fin], g[f"smp_logw_{c}"][fin], rtol=2e-5, atol=2e-5)
        for key, ref in (("lse_pos", g[f"smp_lse_pos_{c}"]), ("lse_neg", g[f"smp_lse_neg_{c}"])):
            both = np.isfinite(ref)
            np.testing.assert_array_equal(np.isfinite(out[key]), both)
            np.testing.assert_allclose(out[key][both], ref[both], rtol=2e-5, atol=2e-5)
        np.testing.assert_array_equal(out["max_sampling_id"], g[f"smp_max_id_{c}"])
        np.testing.assert_array_equal(out["raw"]["dense"][fin], g[f"smp_dense_{c}"][fin])
        np.testing.assert_array_equal(out["raw"]["sparse"][fin], g[f"smp_sparse_{c}"][fin])
        # the flattening of the REFERENCE's sampled sections (exact inputs -> exact outputs)
        flat = osmp.flatten_samples(g[f"smp_idx_{c}"], g[f"smp_scr_{c}"], g[f"smp_lbl_{c}"], g[f"smp_logw_{c}"],
                                    {"dense": g[f"smp_dense_{c}"], "sparse": g[f"smp_sparse_{c}"]})
        np.testing.assert_array_equal(flat["indices"], g[f"flat_idx_{c}"])
        np.testing.assert_array_equal(flat["scores"], g[f"flat_scr_{c}"])
        np.testing.assert_array_equal(flat["labels"], g[f"flat_lbl_{c}"])
        np.testing.assert_array_equal(flat["log_weights"], g[f"flat_logw_{c}"])
        np.testing.assert_array_equal(flat["raw"]["dense"], g[f"flat_dense_{c}"])
        np.testing.assert_array_equal(flat["raw"]["sparse"], g[f"flat_sparse_{c}"])


# ---- the C restatement of the reference's numba loops (oracle/collate_ref.c: the CPU figure beside the C5 kernels) must reproduce the
# ---- same reference-generated fixtures as the NumPy restatements ------------------------------------------------------------------


@pytest.mark.parametrize("name", ["merge_3engine_basic"] + [f"merge_random_{i}" for i in range(8)])
def test_c_collate_restatement_merge_matches_reference(name):
    from oracle import collate_ref as cref

    g = _load(name)
    w = MANIFEST[name]["params"]["weights"]
    idx, scr, lbl, raw = cref.merge_hybrid((g["lookup_idx"], g["lookup_scr"], g["lookup_lbl"]),
                                           {"dense": (g["dense_idx"], g["dense_scr"]), "sparse": (g["sparse_idx"], g["sparse_scr"])},
                                           {"dense": w["dense"], "sparse": w["sparse"]})
    _eq(idx, g["out_idx"])
    _eq(scr, g["out_scr"])
    _eq(lbl, g["out_lbl"])
    _eq(raw["dense"], g["raw_dense"])
    _eq(raw["sparse"], g["raw_sparse"])


def test_c_collate_restatement_chain_matches_reference():
    from oracle import collate_ref as cref

    g = _load("collate_chain")
    for c, p in enumerate(MANIFEST["collate_chain"]["params"]["cases"]):
        lookup = (g[f"l_idx_{c}"], None, g[f"l_lbl_{c}"])
        m_idx, m_scr, m_lbl, m_raw = cref.merge_hybrid(lookup, {"dense": (g[f"d_idx_{c}"], g[f"d_scr_{c}"]), "sparse": (g[f"s_idx_{c}"], g[f"s_scr_{c}"])},
                                                       p["weights"])
        np.testing.assert_array_equal(m_idx, g[f"m_idx_{c}"])
        np.testing.assert_array_equal(m_scr, g[f"m_scr_{c}"])
        np.testing.assert_array_equal(m_lbl, g[f"m_lbl_{c}"])
        out = cref.sample_search_results(m_idx, m_scr, m_lbl, m_raw, g[f"noise_{c}"], p["total"], p["max_pos_sections"], p["temperature"],
                                         p["max_support_size"])
        fin = np.isfinite(g[f"smp_logw_{c}"])
        np.testing.assert_array_equal(out["indices"][fin], g[f"smp_idx_{c}"][fin])
        np.testing.assert_array_equal(out["labels"], g[f"smp_lbl_{c}"])
        np.testing.assert_array_equal(out["scores"][fin], g[f"smp_scr_{c}"][fin])
        np.testing.assert_array_equal(np.isfinite(out["log_weights"]), fin)
        np.testing.assert_allclose(out["log_weights"][fin], g[f"smp_logw_{c}"][fin], rtol=2e-5, atol=2e-5)
        for key, ref in (("lse_pos", g[f"smp_lse_pos_{c}"]), ("lse_neg", g[f"smp_lse_neg_{c}"])):
            both = np.isfinite(ref)
            np.testing.assert_array_equal(np.isfinite(out[key]), both)
            np.testing.assert_allclose(out[key][both], ref[both], rtol=2e-5, atol=2e-5)
        np.testing.assert_array_equal(out["max_sampling_id"], g[f"smp_max_id_{c}"])
        np.testing.assert_array_equal(out["raw"]["dense"][fin], g[f"smp_dense_{c}"][fin])
        np.testing.assert_array_equal(out["raw"]["sparse"][fin], g[f"smp_sparse_{c}"][fin])
        flat = cref.flatten_samples(g[f"smp_idx_{c}"], g[f"smp_scr_{c}"], g[f"smp_lbl_{c}"], g[f"smp_logw_{c}"],
                                    {"dense": g[f"smp_dense_{c}"], "sparse": g[f"smp_sparse_{c}"]})
        np.testing.assert_array_equal(flat["indices"], g[f"flat_idx_{c}"])
        np.testing.assert_array_equal(flat["scores"], g[f"flat_scr_{c}"])
        np.testing.assert_array_equal(flat["labels"], g[f"flat_lbl_{c}"])
        np.testing.assert_array_equal(flat["log_weights"], g[f"flat_logw_{c}"])
        np.testing.assert_array_equal(flat["raw"]["dense"], g[f"flat_dense_{c}"])
        np.testing.assert_array_equal(flat["raw"]["sparse"], g[f"flat_sparse_{c}"])


def test_c_collate_restatement_sampling_matches_reference():
    from oracle import collate_ref as cref

    g = _load("sampling_fixed_noise")
    for c, p in enumerate(MANIFEST["sampling_fixed_noise"]["params"]["cases"]):
        scores, labels, noise = g[f"scores_{c}"], g[f"labels_{c}"], g[f"noise_{c}"]
        if scores.ndim != 2:
            continue
        ids = np.tile(np.arange(scores.shape[1], dtype=np.int64), (scores.shape[0], 1))
        out = cref.sample_search_results(ids, scores, labels.astype(np.int64), {}, noise, p["k_total"], p["k_positive"], p["temperature"],
                                         p["max_support_size"] if p["max_support_size"] and p["max_support_size"] > 0 else None)
        ref_w = g[f"out_logw_{c}"]
        fin = np.isfinite(ref_w)
        np.testing.assert_array_equal(out["local"][fin], g[f"out_samples_{c}"][fin])
        np.testing.assert_array_equal(out["labels"], g[f"out_labels_{c}"])
        np.testing.assert_allclose(out["log_weights"][fin], ref_w[fin], rtol=2e-5, atol=2e-5)


def test_metrics_restatement_matches_the_reference_fixture():
    """SURVEY 10's `metrics_recall_ndcg`: recall / ndcg / mrr / hitrate / precision of the reference
    (vod_models/monitoring/functional.py:41-161) on scores with NaN / +inf / -inf, rows without positives, a masked positive, ties."""
    from oracle import metrics

    z = np.load(GOLDEN / "metrics_recall_ndcg.npz")
    scores, rel = z["scores"], z["relevances"]
    for name in ("recall", "ndcg", "mrr", "hitrate", "precision"):
        for tk in (0, 1, 5, 10):
            want = z[f"{name}_top{tk}"]
            got = getattr(metrics, name)(rel, scores, tk or None)
            if want.dtype == bool:
                np.testing.assert_array_equal(got, want, err_msg=f"{name}@{tk}")
            else:
                np.testing.assert_allclose(got.astype(np.float64), want, rtol=2e-6, atol=0, equal_nan=True, err_msg=f"{name}@{tk}")
    # bench.py's verify.recall_at_k = this recall with the comparator's top-k as the positives
    got_ids = np.array([[5, 3, 9, -1], [1, 2, 3, 4]])
    got_s = np.array([[3.0, 2.0, 1.0, -np.inf], [4.0, 3.0, 2.0, 1.0]], dtype=np.float32)
    ref_ids = np.array([[5, 9, 7, -1], [4, 3, 2, 1]])
    assert metrics.recall_of_ids(got_ids, got_s, ref_ids) == pytest.approx((2 / 3 + 1.0) / 2)
