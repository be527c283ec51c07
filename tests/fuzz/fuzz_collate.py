"""Randomised parity campaign for the collate-side kernels (hybrid merge, labeled priority sampling, in-batch flattening,
retrieval loss forward + backward) against the CPU oracle.  Not part of the test suite: run it on a GPU box when those
kernels change.      python3 tests/fuzz/fuzz_collate.py [--trials 300] [--seed 1] [--seconds 600]
"""
import argparse
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))  # the repo root


LAST = {}


def _eq(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    if not np.array_equal(a, b, equal_nan=(a.dtype.kind == "f")):
        bad = np.argwhere(~((a == b) | ((a != a) & (b != b)) if a.dtype.kind == "f" else (a == b)))
        r = int(bad[0][0])
        ga, gb = (a[r], b[r]) if a.ndim > 1 else (a[max(0, r - 20) : r + 20], b[max(0, r - 20) : r + 20])
        raise AssertionError(f"{what}: {len(bad)} entries differ, first at {bad[0].tolist()}\n  got {ga.tolist()[:40]}\n  ref {gb.tolist()[:40]}")


def fuzz_merge(rng):
    from oracle.hybrid import merge_hybrid as oracle_merge
    from vod_amd.core.merge import merge_hybrid

    nq = int(rng.choice([1, 2, 7, 64, 130]))
    kl = int(rng.choice([0, 1, 5, 32, 128, 600]))
    ks = [int(rng.choice([0, 1, 3, 16, 128, 700])) for _ in range(int(rng.integers(0, 5)))]
    if kl + sum(ks) > 4000:
        ks = ks[:1]
    n_ids = int(rng.choice([5, 40, 300, 100_000]))
    pad_frac = float(rng.choice([0.0, 0.2, 0.9]))
    dup = bool(rng.random() < 0.3)

    def eng(k):
        idx = np.full((nq, k), -1, dtype=np.int64)
        scr = np.full((nq, k), -np.inf, dtype=np.float32)
        for r in range(nq):
            nv = k if rng.uniform() > pad_frac else int(rng.integers(0, k + 1))
            nv = min(nv, n_ids) if not dup else nv
            idx[r, :nv] = rng.choice(n_ids, size=nv, replace=dup)
            scr[r, :nv] = rng.normal(size=nv).astype(np.float32) * 5
            if nv and rng.random() < 0.2:
                scr[r, int(rng.integers(0, nv))] = np.nan
        return idx, scr

    LAST.clear(); LAST.update(kind="merge", nq=nq, kl=kl, ks=ks, n_ids=n_ids, pad=pad_frac, dup=dup)
    l_idx, l_scr = eng(kl)
    l_lbl = (l_scr > -np.inf).astype(np.int64) * rng.integers(1, 3, size=l_scr.shape)
    engs = {f"e{i}": eng(k) for i, k in enumerate(ks)}
    weights = {n: float(rng.choice([0.0, 0.5, 1.0, 1.7])) for n in engs}
    idx, scr, lbl, raw = merge_hybrid(l_idx, l_lbl, engs, weights)
    o_idx, o_scr, o_lbl, o_raw = oracle_merge((l_idx, np.zeros(l_idx.shape, np.float32), l_lbl), engs, weights)
    _eq(idx, o_idx, "merge idx")
    _eq(scr, o_scr, "merge scores")
    _eq(lbl, o_lbl, "merge labels")
    for n in engs:
        _eq(raw[n], o_raw[n], f"merge raw {n}")
    return dict(kind="merge", nq=nq, kl=kl, ks=ks, n_ids=n_ids, pad=pad_frac, dup=dup)


def fuzz_sampling(rng):
    from oracle.sampling import labeled_priority_sampling_2d
    from vod_amd.core.sample import labeled_priority_sampling_tensors

    nq = int(rng.choice([1, 3, 64, 200]))
    n = int(rng.choice([1, 2, 9, 40, 385, 1000, 4096]))
    k_tot = int(rng.choice([1, 2, 8, 32, 64, 128]))
    k_pos = int(rng.integers(0, k_tot + 1))
    temp = float(rng.choice([0.0, 0.5, 1.0, 3.0]))
    support = int(rng.choice([-1, -1, 1, 10, 100, 2000]))
    keep_top = bool(rng.random() < 0.4)  # the corrected support truncation beside the reference's
    LAST.clear(); LAST.update(kind="sampling", nq=nq, n=n, k_pos=k_pos, k_tot=k_tot, temp=temp, support=support, keep_top=keep_top)
    scores = (rng.normal(size=(nq, n)) * 3).astype(np.float32)
    scores[rng.uniform(size=scores.shape) < float(rng.choice([0.0, 0.1, 0.8]))] = -np.inf
    labels = rng.uniform(size=scores.shape) < float(rng.choice([0.0, 0.05, 0.5]))
    noise = rng.exponential(size=scores.shape).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    out = labeled_priority_sampling_tensors(t(scores), t(labels), t(noise), k_pos, k_tot, normalized=True, temperature=temp,
                                            max_support_size=support, support="keep_top" if keep_top else "reference")
    smp, logw, lab, lse = [o.cpu().numpy() for o in out]
    sup = max(support, k_tot) if support >= 0 else -1
    r_smp, r_logw, r_lab, r_lse = labeled_priority_sampling_2d(scores, labels, noise, k_pos, k_tot, True, temp, sup, keep_top)
    # The reference orders the samples by an fp32 key through an unstable argsort and computes the weights with
    # log1p(-exp(-exp(log_p - log_tau))) in fp32, which loses ~eps / x relative accuracy for a sample of inclusion
    # probability x << 1: ids are compared as per-row SETS split by label, weights after aligning by id, and the tolerance
    # of a weight follows its conditioning, measured as the distance between the oracle run in fp32 and in fp64.
    r64 = labeled_priority_sampling_2d(scores.astype(np.float64), labels, noise.astype(np.float64), k_pos, k_tot, True, temp, sup, keep_top)
    finite = np.isfinite(r_logw)
    assert np.array_equal(lab, r_lab), "sample labels"
    assert np.array_equal(smp < 0, r_smp < 0), "sample count"
    assert np.array_equal(np.isfinite(logw), finite), "finite log weights"
    for r in range(nq):
        f = finite[r]
        for want in (True, False):
            m = f & (r_lab[r] == want)
            if sorted(smp[r][m].tolist()) != sorted(r_smp[r][m].tolist()):
                raise AssertionError(f"sample ids row {r}:\n  got {smp[r].tolist()}\n  ref {r_smp[r].tolist()}")
        got = dict(zip(smp[r][f].tolist(), logw[r][f].tolist()))
        ref = dict(zip(r_smp[r][f].tolist(), r_logw[r][f].tolist()))
        f64 = np.isfinite(r64[1][r])
        ref64 = dict(zip(r64[0][r][f64].tolist(), r64[1][r][f64].tolist()))
        ids = sorted(ref)
        if ids and sorted(ref64) == ids:  # (else the fp64 run drew a different sample: the row sits on a tie)
            ga, ra, da = (np.array([d[i] for i in ids]) for d in (got, ref, ref64))
            # (+ the GPU's own 1-2 ulp exp in the same ill-conditioned expression when k << n)
            tol = 2e-5 + 2e-5 * np.abs(ra) + 4.0 * np.abs(ra - da).max() + (2e-4 if k_tot * 100 < n else 0.0)
            bad = np.abs(ga - ra) > tol
            assert not bad.any(), f"log weights row {r}: got {ga[bad][:4]} ref {ra[bad][:4]} fp64 {da[bad][:4]} tol {tol}"
        v = smp[r][smp[r] >= 0]
        assert len(set(v.tolist())) == len(v), "duplicate sample"
    both = np.isfinite(r_lse)
    assert np.array_equal(np.isfinite(lse), both), "finite lse"
    np.testing.assert_allclose(lse[both], r_lse[both], rtol=5e-5, atol=5e-5)  # tree vs sequential float32 sums of up to 4096 terms
    return dict(kind="sampling", nq=nq, n=n, k_pos=k_pos, k_tot=k_tot, temp=temp, support=support, keep_top=keep_top)


def fuzz_gradients(rng):
    from oracle.gradients import retrieval_gradients
    from vod_amd.gradients import RetrievalGradients

    B = int(rng.choice([1, 3, 16, 64]))
    D = int(rng.choice([1, 2, 5, 32, 300, 2048]))
    H = int(rng.choice([1, 17, 64, 768, 1024]))
    three_d = bool(rng.random() < 0.5) and D <= 512
    q = (rng.normal(size=(B, H)) / np.sqrt(H) * 3).astype(np.float32)
    s = rng.normal(size=((B, D, H) if three_d else (D, H))).astype(np.float32)
    score = rng.normal(size=(B, D)).astype(np.float32)
    pad = rng.uniform(size=(B, D)) < float(rng.choice([0.0, 0.1, 0.6]))
    pad[:, 0] = False
    score[pad] = -np.inf
    rel = (rng.uniform(size=(B, D)) < float(rng.choice([0.0, 0.05, 0.5]))).astype(np.int64)
    if B > 1:
        rel[B // 2] = 0
    sparse = dense = None
    if rng.random() < 0.7:
        sparse = rng.normal(size=(B, D)).astype(np.float32)
        sparse[rng.uniform(size=(B, D)) < 0.2] = np.nan
    if rng.random() < 0.5:
        dense = rng.normal(size=(B, D)).astype(np.float32)
    up = float(rng.choice([1.0, 2.5]))
    qt = torch.tensor(q, device="cuda", requires_grad=True)
    st = torch.tensor(s, device="cuda", requires_grad=True)
    batch = {"section__score": torch.tensor(score, device="cuda"), "section__relevance": torch.tensor(rel, device="cuda"),
             "section__sparse": None if sparse is None else torch.tensor(sparse, device="cuda"),
             "section__dense": None if dense is None else torch.tensor(dense, device="cuda")}
    out = RetrievalGradients()(batch=batch, query_encoding=qt, section_encoding=st)
    (out.loss * up).backward()
    ref = retrieval_gradients(q, s, score, rel, sparse, dense)
    tol = dict(rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(out.loss.item(), ref["loss"], **tol)
    np.testing.assert_allclose(out.retriever_scores.cpu().numpy(), ref["retriever_scores"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(qt.grad.cpu().numpy(), up * ref["dq"], **tol)
    np.testing.assert_allclose(st.grad.cpu().numpy(), up * ref["ds"], **tol)
    for key in ("kl_score", "kl_sparse", "kl_dense"):
        if key in ref and ref[key] is not None and key in out.diagnostics:
            np.testing.assert_allclose(out.diagnostics[key].item(), ref[key], **tol)
    return dict(kind="gradients", B=B, D=D, H=H, three_d=three_d)


def fuzz_merge_topk(rng):
    """H3: the per-shard lists of a row-sharded search (sorted, padded, global ids) or arbitrary unsorted lists."""
    from oracle.flat_ip import merge_shard_topk
    from vod_amd.index import merge_topk

    S = int(rng.choice([1, 2, 3, 8, 16]))
    nq = int(rng.choice([1, 5, 64, 300]))
    k = int(rng.choice([1, 2, 10, 100, 128, 200, 512, 2048]))
    if S * k > 16384:
        S = max(1, 16384 // k)
    k_out = int(rng.choice([k, k, max(1, k // 2), min(2048, 2 * k)]))
    sorted_in = bool(rng.random() < 0.7)
    LAST.clear(); LAST.update(kind="merge_topk", S=S, nq=nq, k=k, k_out=k_out, sorted=sorted_in)
    scores = np.round(rng.normal(size=(S, nq, k)) * 4).astype(np.float32) / 2  # many ties
    ids = np.full((S, nq, k), -1, dtype=np.int64)
    for s_ in range(S):
        for r in range(nq):
            nv = k if rng.random() < 0.7 else int(rng.integers(0, k + 1))
            ids[s_, r, :nv] = s_ * 1_000_000 + rng.choice(100_000, size=nv, replace=False)
            scores[s_, r, nv:] = -np.inf
            if sorted_in and nv:  # (score desc, id asc) like a search result
                o = np.lexsort((ids[s_, r, :nv], -scores[s_, r, :nv]))
                scores[s_, r, :nv], ids[s_, r, :nv] = scores[s_, r, :nv][o], ids[s_, r, :nv][o]
    gs, gi = merge_topk(torch.from_numpy(scores).cuda(), torch.from_numpy(ids).cuda(), k_out)
    rs, ri = merge_shard_topk(list(scores), list(ids), k_out)
    _eq(gi.cpu().numpy(), ri, "merge_topk ids")
    _eq(gs.cpu().numpy(), rs, "merge_topk scores")
    return dict(LAST)


def fuzz_flatten(rng):
    """H7b: in-batch flattening (unique ids of the batch, every attribute gathered by id, first match wins)."""
    from oracle import sampling as osmp
    from vod_amd import types as vt
    from vod_amd.core.in_batch_negatives import flatten_samples
    from vod_amd.core.sample import PrioritySampledSections

    b = int(rng.choice([1, 2, 7, 64, 130]))
    n = int(rng.choice([1, 3, 32, 130, 512]))
    pool = int(rng.choice([1, 5, 200, 3000, 10**9]))
    LAST.clear(); LAST.update(kind="flatten", b=b, n=n, pool=pool)
    idx = rng.integers(-1, pool, size=(b, n)).astype(np.int64)
    scr = rng.standard_normal((b, n)).astype(np.float32)
    scr[idx < 0] = -np.inf
    lbl = rng.random((b, n)) < float(rng.choice([0.0, 0.2, 1.0]))
    logw = rng.standard_normal((b, n)).astype(np.float32)
    raw = {"dense": rng.standard_normal((b, n)).astype(np.float32),
           "sparse": np.where(rng.random((b, n)) < 0.3, np.nan, 1.0).astype(np.float32)}
    if rng.random() < 0.3:
        raw.pop("sparse")
    ps = PrioritySampledSections(batch=vt.RetrievalBatch(indices=idx, scores=scr, labels=lbl), log_weights=logw,
                                 max_sampling_id=np.zeros(b), lse_pos=np.zeros(b), lse_neg=np.zeros(b), raw_scores=raw)
    for padding in (True, False):
        out = flatten_samples(ps, padding=padding)
        ref = osmp.flatten_samples(idx, scr, lbl, logw, raw, padding=padding)
        _eq(out.batch.indices, ref["indices"], "flatten ids")
        _eq(out.batch.scores, ref["scores"], "flatten scores")
        _eq(out.batch.labels, ref["labels"], "flatten labels")
        _eq(out.log_weights, ref["log_weights"], "flatten log weights")
        for key in raw:
            _eq(out.raw_scores[key], ref["raw"][key], f"flatten raw {key}")
    return dict(LAST)


def fuzz_chain(rng):
    """The device-resident chain (`collate_on_device` = vodhip_collate: merge -> sampling + gathers + rank diagnostic -> in-batch
    flattening, no host sync) against the oracle's restatement of the reference chain, stage by stage."""
    from oracle import sampling as osmp
    from oracle.hybrid import merge_hybrid
    from vod_amd.core.collate import collate_on_device

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    nq = int(rng.choice([1, 3, 16, 64, 200]))
    kl = int(rng.choice([0, 1, 4, 32, 128]))
    ks = [int(rng.choice([1, 3, 16, 128, 400])) for _ in range(int(rng.integers(1, 5)))]
    pool = int(rng.choice([6, 50, 400, 100_000]))
    total = int(rng.choice([1, 4, 32, 100]))
    kpos = min(total, int(rng.choice([1, 2, 8, total])))  # (k_positive > k_total is undefined in the reference: its numba loop overruns the output row)
    temp = float(rng.choice([0.0, 0.5, 1.0, 3.0]))
    support = None if rng.random() < 0.5 else int(rng.choice([1, 10, 100, 1000]))
    keep_top = bool(rng.random() < 0.4)
    mode = "keep_top" if keep_top else "reference"
    dup = float(rng.choice([0.0, 0.0, 0.5]))
    flat = bool(rng.random() < 0.5) and nq * total <= 8192
    LAST.clear(); LAST.update(kind="chain", nq=nq, kl=kl, ks=ks, pool=pool, total=total, kpos=kpos, temp=temp, support=support, keep_top=keep_top, dup=dup, flat=flat)

    def eng(k):
        idx = np.full((nq, k), -1, dtype=np.int64)
        scr = np.full((nq, k), -np.inf, dtype=np.float32)
        for r in range(nq):
            nv = k if rng.random() > 0.2 else int(rng.integers(0, k + 1))
            idx[r, :nv] = rng.choice(pool, size=nv, replace=pool < nv)
            if nv and rng.random() < dup:
                src = rng.integers(0, nv, size=max(1, nv // 8))
                idx[r, rng.integers(0, nv, size=len(src))] = idx[r, src]
            scr[r, :nv] = -np.sort(-(rng.normal(size=nv) * 3).astype(np.float32))
        return idx, scr

    l_idx, _ = eng(kl)
    l_lbl = np.where(l_idx >= 0, rng.integers(0, 3, size=l_idx.shape), 0).astype(np.int64)
    names = [f"e{j}" for j in range(len(ks))]
    engs = [eng(k) for k in ks]
    weights = {n: float(rng.choice([0.0, 0.5, 1.0, 2.0])) for n in names}
    with np.errstate(all="ignore"):
        m_idx, m_scr, m_lbl, m_raw = merge_hybrid((l_idx, np.zeros(l_idx.shape, np.float32), l_lbl), dict(zip(names, engs)), weights)
    w = m_idx.shape[1]
    noise = rng.exponential(size=(nq, kl + sum(ks) + 1)).astype(np.float32)
    ref = osmp.sample_search_results(m_idx, m_scr, m_lbl, m_raw, noise[:, :w], total, kpos, temp, support, keep_top=keep_top)
    engines = {n: (t(i), t(sc)) for n, (i, sc) in zip(names, engs)}
    out = collate_on_device(t(l_idx), t(l_lbl), engines, weights, t(noise), total=total, max_pos_sections=kpos, temperature=temp,
                            max_support_size=support, support=mode)
    fin = np.isfinite(ref["log_weights"])
    got_w = out.log_weights.cpu().numpy()
    _eq(np.isfinite(got_w), fin, "chain finite weights")
    # temperature 0 = deterministic top-k by score: ids that TIE at the cut (ids known to a zero-weight engine only all score 0)
    # are interchangeable - NumPy's order among equal keys is unspecified, this kernel takes the smaller column - so there the
    # selected SCORES are compared, not the ids behind them
    ids_comparable = temp > 0
    got_ids, got_s = out.indices.cpu().numpy(), out.scores.cpu().numpy()
    _eq(out.labels.cpu().numpy(), ref["labels"], "chain labels")
    lab = ref["labels"]
    if ids_comparable:
        # the same SET of sections per row and class; their order may differ where two priority keys agree to the last ulp
        # (`log_p - log(noise)` is evaluated with this device's logf, the reference's with NumPy's)
        for r in range(nq):
            for cls in (True, False):
                m_ = fin[r] & (lab[r] == cls)
                if sorted(got_ids[r][m_].tolist()) != sorted(ref["indices"][r][m_].tolist()):
                    raise AssertionError(f"chain ids row {r} class {cls}:\n  got {got_ids[r][m_].tolist()}\n  ref {ref['indices'][r][m_].tolist()}")
    # scores: the multiset per row and class (temperature 0: scores a few ulp apart can round to the SAME log-probability - equal keys)
    gs, rs = np.where(fin, got_s, -np.inf), np.where(fin, ref["scores"], -np.inf)
    for cls in (True, False):
        _eq(np.sort(np.where(lab == cls, gs, -np.inf), axis=1), np.sort(np.where(lab == cls, rs, -np.inf), axis=1), f"chain scores (class {cls})")
    # log-weights: `log_p - log1p(-exp(-exp(log_p - log_tau)))` is ill-conditioned for inclusion probabilities << 1; the reference's
    # own float32 and float64 evaluations differ there, and that difference sets the tolerance (as in fuzz_sampling)
    with np.errstate(all="ignore"):
        ref64 = osmp.sample_search_results(m_idx, m_scr.astype(np.float64), m_lbl, {}, noise[:, :w].astype(np.float64), total, kpos, temp, support, keep_top=keep_top)
    same = ids_comparable and np.array_equal(np.where(fin, ref64["local"], -1), np.where(fin, ref["local"], -1)) and np.array_equal(got_ids[fin], ref["indices"][fin])
    if same:
        cond = np.abs(np.where(fin, ref["log_weights"].astype(np.float64) - ref64["log_weights"], 0.0)).max(axis=1, keepdims=True)
        tol = 2e-4 + 2e-4 * np.abs(np.where(fin, ref["log_weights"], 0.0)) + 4.0 * cond
        bad = fin & (np.abs(np.where(fin, got_w.astype(np.float64) - ref["log_weights"], 0.0)) > tol)
        assert not bad.any(), f"chain log weights: got {got_w[bad][:4]} ref {ref['log_weights'][bad][:4]} fp64 {ref64['log_weights'][bad][:4]}"
    for key in ("lse_pos", "lse_neg"):
        both = np.isfinite(ref[key])
        got = getattr(out, key).cpu().numpy()
        _eq(np.isfinite(got), both, key)
        np.testing.assert_allclose(got[both], ref[key][both], rtol=2e-4, atol=2e-4)
    settled = (fin | (ref["local"] < 0)).all(axis=1)
    _eq(out.max_sampling_id.cpu().numpy()[settled], ref["max_sampling_id"][settled], "chain rank diagnostic")
    for n in names:
        if same:
            _eq(out.raw_scores[n].cpu().numpy()[fin], ref["raw"][n][fin], f"chain raw {n}")
    if flat:
        fl = collate_on_device(t(l_idx), t(l_lbl), engines, weights, t(noise), total=total, max_pos_sections=kpos, temperature=temp,
                               max_support_size=support, in_batch_negatives=True, support=mode)
        rf = osmp.flatten_samples(out.indices.cpu().numpy(), out.scores.cpu().numpy(), out.labels.cpu().numpy(), out.log_weights.cpu().numpy(),
                                  {n: v.cpu().numpy() for n, v in out.raw_scores.items()})
        _eq(fl.indices.cpu().numpy(), rf["indices"], "chain flat ids")
        _eq(fl.scores.cpu().numpy(), rf["scores"], "chain flat scores")
        _eq(fl.labels.cpu().numpy(), rf["labels"], "chain flat labels")
        _eq(fl.log_weights.cpu().numpy(), rf["log_weights"], "chain flat log weights")
        for n in names:
            _eq(fl.raw_scores[n].cpu().numpy(), rf["raw"][n], f"chain flat raw {n}")
    return dict(LAST)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=600.0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0, fails, counts = time.time(), 0, {}
    for t in range(a.trials):
        if time.time() - t0 > a.seconds:
            break
        fn = [fuzz_merge, fuzz_sampling, fuzz_gradients, fuzz_merge_topk, fuzz_flatten, fuzz_chain][t % 6]
        state = rng.bit_generator.state
        try:
            info = fn(rng)
            counts[info["kind"]] = counts.get(info["kind"], 0) + 1
        except Exception as e:  # noqa: BLE001
            fails += 1
            print(f"FAIL trial {t} ({fn.__name__}) {LAST}: {type(e).__name__}: {str(e)[:1500]}", flush=True)
            if not isinstance(e, AssertionError):
                traceback.print_exc()
    print(f"fuzz_collate: {sum(counts.values())} ok {counts}, {fails} failures, {time.time() - t0:.0f} s, seed {a.seed}")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
