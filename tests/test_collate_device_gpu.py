"""GPU parity tests of the device-resident collate chain: `collate_on_device` = merge -> priority sampling
(+ gathers + rank diagnostic) -> in-batch flattening in three launches with no host synchronisation, against

  * `collate_chain.npz`: what the REFERENCE's own `_merge_search_results` -> `sample_search_results` -> `flatten_samples`
    produced for the same inputs and the same Exp(1) draw (tests/golden/make_golden.py, `np.random` seeded), and
  * the CPU oracle on random cases (wide rows, ids repeated inside an engine, rows without positives, pads).

Ids / labels / pad positions / NaN positions: exact wherever the reference's weight is finite (slots with a -inf key are
padding-like; NumPy's order among equal keys is unspecified).  float32 log-weights and lse: 2e-5 (tree vs sequential sums)."""
import json

import numpy as np
import pytest

from conftest import GOLDEN

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
TOL = dict(rtol=2e-5, atol=2e-5)
MANIFEST = json.loads((GOLDEN / "manifest.json").read_text())


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _pad_noise(noise, stride):
    out = np.ones((noise.shape[0], stride), dtype=np.float32)
    out[:, : noise.shape[1]] = noise
    return out


# random pools hold samples with inclusion probability << 1, where the reference's float32 `log1p(-exp(-exp(x)))` is
# ill-conditioned: its own float32 and float64 evaluations differ by up to 1e-4 there
TOL_RANDOM = dict(rtol=1e-4, atol=1e-4)


def _check_sampled(out, ref, raw_names, tol=None):
    """`out`: DeviceSampledSections (not flattened); `ref`: dict with the reference / oracle arrays."""
    tol = tol or TOL
    fin = np.isfinite(ref["log_weights"])
    got_logw = out.log_weights.cpu().numpy()
    np.testing.assert_array_equal(np.isfinite(got_logw), fin)
    np.testing.assert_array_equal(out.indices.cpu().numpy()[fin], ref["indices"][fin])
    np.testing.assert_array_equal(out.labels.cpu().numpy(), ref["labels"])
    np.testing.assert_array_equal(out.scores.cpu().numpy()[fin], ref["scores"][fin])
    np.testing.assert_allclose(got_logw[fin], ref["log_weights"][fin], **tol)
    for key in ("lse_pos", "lse_neg"):
        both = np.isfinite(ref[key])
        got = getattr(out, key).cpu().numpy()
        np.testing.assert_array_equal(np.isfinite(got), both)
        np.testing.assert_allclose(got[both], ref[key][both], **tol)
    # the rank diagnostic depends on WHICH sections were sampled: comparable on rows whose every sample has a finite key (a slot
    # filled from the tie among -inf keys - e.g. entries masked by the support truncation - is unspecified in the reference)
    settled = (fin | (ref["indices"] < 0) & ~np.isfinite(ref["scores"])).all(axis=1) if "local" not in ref else (fin | (ref["local"] < 0)).all(axis=1)
    np.testing.assert_array_equal(out.max_sampling_id.cpu().numpy()[settled], ref["max_sampling_id"][settled])
    for name in raw_names:
        np.testing.assert_array_equal(out.raw_scores[name].cpu().numpy()[fin], ref["raw"][name][fin])


def test_device_pipeline_matches_the_reference_chain():
    from vod_amd.core.collate import collate_on_device, flatten_on_device, sample_merged_on_device
    from vod_amd.core.merge import merge_hybrid_device

    g = np.load(GOLDEN / "collate_chain.npz")
    for c, p in enumerate(MANIFEST["collate_chain"]["params"]["cases"]):
        engines = {"dense": (_t(g[f"d_idx_{c}"]), _t(g[f"d_scr_{c}"])), "sparse": (_t(g[f"s_idx_{c}"]), _t(g[f"s_scr_{c}"]))}
        l_idx, l_lbl = _t(g[f"l_idx_{c}"]), _t(g[f"l_lbl_{c}"])
        stride = l_idx.shape[1] + sum(v[0].shape[1] for v in engines.values()) + 1
        noise = _t(_pad_noise(g[f"noise_{c}"], stride))
        kw = dict(total=p["total"], max_pos_sections=p["max_pos_sections"], temperature=p["temperature"], max_support_size=p["max_support_size"])
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")  # any host synchronisation inside the chain raises
        try:
            merged = merge_hybrid_device(l_idx, l_lbl, engines, p["weights"])
            out = sample_merged_on_device(merged, noise, **kw)
            flat = collate_on_device(l_idx, l_lbl, engines, p["weights"], noise, in_batch_negatives=True, **kw)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        # the merge, cut on the host with the width read back, is the reference's merged batch bit for bit
        m_idx, m_scr, m_lbl, _ = merged.cut()
        np.testing.assert_array_equal(m_idx.cpu().numpy(), g[f"m_idx_{c}"])
        np.testing.assert_array_equal(m_scr.cpu().numpy(), g[f"m_scr_{c}"])
        np.testing.assert_array_equal(m_lbl.cpu().numpy(), g[f"m_lbl_{c}"])
        ref = {"indices": g[f"smp_idx_{c}"], "scores": g[f"smp_scr_{c}"], "labels": g[f"smp_lbl_{c}"], "log_weights": g[f"smp_logw_{c}"],
               "lse_pos": g[f"smp_lse_pos_{c}"], "lse_neg": g[f"smp_lse_neg_{c}"], "max_sampling_id": g[f"smp_max_id_{c}"],
               "raw": {"dense": g[f"smp_dense_{c}"], "sparse": g[f"smp_sparse_{c}"]}}
        _check_sampled(out, ref, ("dense", "sparse"))
        d = out.to_dict("section__")
        assert set(d) == {"section__idx", "section__score", "section__label", "section__log_weight", "section__lse_pos", "section__lse_neg",
                          "section__dense", "section__sparse"}
        assert d["section__label"].dtype == torch.bool and d["section__idx"].dtype == torch.int64 and all(v.is_cuda for v in d.values())
        # flattening: exact inputs (the reference's sampled sections) -> exact outputs
        from vod_amd.core.collate import DeviceSampledSections

        z = torch.zeros((l_idx.shape[0],), device="cuda")
        ref_in = DeviceSampledSections(indices=_t(g[f"smp_idx_{c}"]), scores=_t(g[f"smp_scr_{c}"]), labels=_t(g[f"smp_lbl_{c}"]),
                                       log_weights=_t(g[f"smp_logw_{c}"]), lse_pos=z, lse_neg=z, max_sampling_id=z,
                                       raw_scores={"dense": _t(g[f"smp_dense_{c}"]), "sparse": _t(g[f"smp_sparse_{c}"])})
        fl = flatten_on_device(ref_in)
        np.testing.assert_array_equal(fl.indices.cpu().numpy(), g[f"flat_idx_{c}"])
        np.testing.assert_array_equal(fl.scores.cpu().numpy(), g[f"flat_scr_{c}"])
        np.testing.assert_array_equal(fl.labels.cpu().numpy(), g[f"flat_lbl_{c}"])
        np.testing.assert_array_equal(fl.log_weights.cpu().numpy(), g[f"flat_logw_{c}"])
        np.testing.assert_array_equal(fl.raw_scores["dense"].cpu().numpy(), g[f"flat_dense_{c}"])
        np.testing.assert_array_equal(fl.raw_scores["sparse"].cpu().numpy(), g[f"flat_sparse_{c}"])
        assert int(fl.n_unique.item()) == len(np.unique(g[f"smp_idx_{c}"]))
        # the fully chained result: the id set of the flattened batch is the one of the device-sampled sections
        assert flat.indices.shape == (l_idx.shape[0] * p["total"],) and flat.scores.shape == (l_idx.shape[0], l_idx.shape[0] * p["total"])
        uq = np.unique(out.indices.cpu().numpy())
        np.testing.assert_array_equal(flat.indices.cpu().numpy()[: len(uq)], uq)


def test_numpy_drop_in_chain_matches_the_reference_chain():
    """The NumPy drop-in functions (what the reference's collate would call) run the same kernels: merge -> sample (seeded
    `np.random`, so the same Exp(1) draw as the reference) -> flatten."""
    from vod_amd import types as vt
    from vod_amd.core.in_batch_negatives import flatten_samples
    from vod_amd.core.sample import sample_search_results
    from vod_amd.core.search import merge_search_results

    g = np.load(GOLDEN / "collate_chain.npz")
    for c, p in enumerate(MANIFEST["collate_chain"]["params"]["cases"]):
        res = {
            "lookup": vt.RetrievalBatch(scores=np.zeros(g[f"l_idx_{c}"].shape, np.float32), indices=g[f"l_idx_{c}"].copy(), labels=g[f"l_lbl_{c}"].copy()),
            "dense": vt.RetrievalBatch(scores=g[f"d_scr_{c}"].copy(), indices=g[f"d_idx_{c}"].copy()),
            "sparse": vt.RetrievalBatch(scores=g[f"s_scr_{c}"].copy(), indices=g[f"s_idx_{c}"].copy()),
        }
        merged, raw = merge_search_results(res, dict(p["weights"]))
        np.testing.assert_array_equal(merged.indices, g[f"m_idx_{c}"])
        np.testing.assert_array_equal(merged.scores, g[f"m_scr_{c}"])
        np.random.seed(p["seed"])
        smp = sample_search_results(search_results=merged, raw_scores=raw, total=p["total"], max_pos_sections=p["max_pos_sections"],
                                    temperature=p["temperature"], max_support_size=p["max_support_size"])
        fin = np.isfinite(g[f"smp_logw_{c}"])
        np.testing.assert_array_equal(smp.batch.indices[fin], g[f"smp_idx_{c}"][fin])
        np.testing.assert_array_equal(smp.batch.labels, g[f"smp_lbl_{c}"])
        np.testing.assert_allclose(smp.log_weights[fin], g[f"smp_logw_{c}"][fin], **TOL)
        np.testing.assert_array_equal(smp.max_sampling_id, g[f"smp_max_id_{c}"])
        np.testing.assert_array_equal(smp.raw_scores["sparse"][fin], g[f"smp_sparse_{c}"][fin])
        assert smp.batch.labels.dtype == np.bool_ and smp.batch.indices.dtype == np.int64
        flat = flatten_samples(smp, padding=True)
        assert flat.batch.indices.shape == g[f"flat_idx_{c}"].shape and flat.batch.scores.shape == g[f"flat_scr_{c}"].shape


def _random_engines(rng, nq, kl, ks, pool, pad_frac, dup_frac):
    def eng(k):
        idx = np.full((nq, k), -1, dtype=np.int64)
        scr = np.full((nq, k), -np.inf, dtype=np.float32)
        for r in range(nq):
            nv = k if rng.random() > pad_frac else int(rng.integers(0, k + 1))
            idx[r, :nv] = rng.choice(pool, size=nv, replace=pool < nv)
            if nv and rng.random() < dup_frac:  # ids repeated INSIDE the engine (legal: merge_corners)
                src = rng.integers(0, nv, size=max(1, nv // 8))
                dst = rng.integers(0, nv, size=len(src))
                idx[r, dst] = idx[r, src]
            scr[r, :nv] = -np.sort(-(rng.normal(size=nv) * 3).astype(np.float32))
            if nv and rng.random() < 0.2:
                scr[r, rng.integers(0, nv)] = np.nan if rng.random() < 0.5 else -np.inf
        return idx, scr

    l_idx, l_scr = eng(kl)
    l_lbl = np.where(l_idx >= 0, rng.integers(0, 3, size=l_idx.shape), 0).astype(np.int64)
    return (l_idx, l_lbl), [eng(k) for k in ks]


@pytest.mark.parametrize("mode", ["reference", "keep_top"])  # Q8's support truncation as the reference has it / corrected
@pytest.mark.parametrize("nq,kl,ks,pool,pad,dup,total,kpos,temp,support", [
    (64, 128, (128, 128), 1_000_000, 0.1, 0.0, 32, 8, 1.0, 100),      # C5: B = 64, K = 128 per engine, 32 sampled sections
    (64, 128, (128, 128), 600, 0.1, 0.0, 32, 8, 1.0, None),            # heavy overlap between the engines
    (9, 40, (60, 50, 70), 150, 0.3, 0.6, 24, 6, 1.0, None),            # three engines, ids repeated inside engines
    (5, 1300, (1300, 1400), 5000, 0.05, 0.2, 64, 16, 0.5, 2000),       # wide rows: W = 4000 (the kernel's limit is 4096)
    (7, 3, (2,), 4, 0.5, 0.5, 16, 4, 1.0, None),                       # tiny lists, fewer candidates than `total`
    (6, 0, (33, 20, 7, 11), 40, 0.2, 0.3, 8, 2, 0.0, None),            # empty lookup list, four engines, deterministic top-k
])
def test_device_pipeline_matches_the_oracle_random(nq, kl, ks, pool, pad, dup, total, kpos, temp, support, mode):
    if mode == "keep_top" and support is None:
        pytest.skip("no support truncation in this case: the two modes are the same code path")
    from oracle import sampling as osmp
    from oracle.hybrid import merge_hybrid
    from vod_amd.core.collate import collate_on_device, sample_merged_on_device
    from vod_amd.core.merge import merge_hybrid_device

    rng = np.random.default_rng(nq * 131 + kl + sum(ks))
    (l_idx, l_lbl), engs = _random_engines(rng, nq, kl, ks, pool, pad, dup)
    names = [f"e{j}" for j in range(len(ks))]
    weights = {n: float(w) for n, w in zip(names, (1.0, 0.5, 0.0, 2.0))}
    with np.errstate(all="ignore"):
        m_idx, m_scr, m_lbl, m_raw = merge_hybrid((l_idx, np.zeros(l_idx.shape, np.float32), l_lbl), dict(zip(names, engs)), weights)
    stride = kl + sum(ks) + 1
    noise = rng.exponential(size=(nq, stride)).astype(np.float32)
    engines = {n: (_t(i), _t(s)) for n, (i, s) in zip(names, engs)}
    merged = merge_hybrid_device(_t(l_idx), _t(l_lbl), engines, weights)
    c_idx, c_scr, c_lbl, c_raw = merged.cut()
    np.testing.assert_array_equal(c_idx.cpu().numpy(), m_idx)
    np.testing.assert_array_equal(c_scr.cpu().numpy(), m_scr)
    np.testing.assert_array_equal(c_lbl.cpu().numpy(), m_lbl)
    for n in names:
        np.testing.assert_array_equal(c_raw[n].cpu().numpy(), m_raw[n])
    # everything beyond the cut is padding, as the merge kernel leaves it
    w = m_idx.shape[1]
    assert bool((merged.indices[:, w:] == -1).all()) and bool(torch.isneginf(merged.scores[:, w:]).all())
    ref = osmp.sample_search_results(m_idx, m_scr, m_lbl, m_raw, noise[:, :w], total, kpos, temp, support, keep_top=mode == "keep_top")
    out = sample_merged_on_device(merged, _t(noise), total=total, max_pos_sections=kpos, temperature=temp, max_support_size=support, support=mode)
    _check_sampled(out, ref, names, TOL_RANDOM)
    if nq * total <= 8192:
        flat = collate_on_device(_t(l_idx), _t(l_lbl), engines, weights, _t(noise), total=total, max_pos_sections=kpos, temperature=temp,
                                 max_support_size=support, in_batch_negatives=True, support=mode)
        o = out
        rf = osmp.flatten_samples(o.indices.cpu().numpy(), o.scores.cpu().numpy(), o.labels.cpu().numpy(), o.log_weights.cpu().numpy(),
                                  {n: v.cpu().numpy() for n, v in o.raw_scores.items()})
        np.testing.assert_array_equal(flat.indices.cpu().numpy(), rf["indices"])
        np.testing.assert_array_equal(flat.scores.cpu().numpy(), rf["scores"])
        np.testing.assert_array_equal(flat.labels.cpu().numpy(), rf["labels"])
        np.testing.assert_array_equal(flat.log_weights.cpu().numpy(), rf["log_weights"])
        for n in names:
            np.testing.assert_array_equal(flat.raw_scores[n].cpu().numpy(), rf["raw"][n])


def test_device_noise_is_drawn_on_the_device_when_not_given():
    from vod_amd.core.collate import collate_on_device

    rng = np.random.default_rng(3)
    (l_idx, l_lbl), engs = _random_engines(rng, 16, 8, (32, 32), 500, 0.1, 0.0)
    engines = {"dense": (_t(engs[0][0]), _t(engs[0][1])), "sparse": (_t(engs[1][0]), _t(engs[1][1]))}
    gen = torch.Generator(device="cuda").manual_seed(7)
    a = collate_on_device(_t(l_idx), _t(l_lbl), engines, {"dense": 1.0, "sparse": 1.0}, total=8, max_pos_sections=2, generator=gen)
    gen.manual_seed(7)
    b = collate_on_device(_t(l_idx), _t(l_lbl), engines, {"dense": 1.0, "sparse": 1.0}, total=8, max_pos_sections=2, generator=gen)
    assert torch.equal(a.indices, b.indices) and torch.equal(a.log_weights, b.log_weights)
    assert a.indices.shape == (16, 8) and bool((a.local_ids < 8 + 64 + 1).all())
