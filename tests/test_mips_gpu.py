"""GPU parity tests for the fused inner-product top-k (H1/H2/H3), through the C-ABI (ctypes -> libvodhip.so).

Bar: ids bit-exact under the (score desc, id asc) tie-break and scores bit-exact on the integer-valued
fixtures (every partial sum is exact in fp32); on Gaussian data scores within 1e-3 of the fp64 oracle and
ids equal except swaps between neighbours whose fp64 scores differ by < 1e-3 * scale (recall must be 1.0
against the oracle's tolerance-widened candidate set).
"""
import json

import numpy as np
import pytest

from conftest import GOLDEN

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-3  # north_star: "scores within 1e-3 fp32"


def _index(x, dtype=None, capacity=None, **params):
    from vod_amd.index import HipFlatIndex

    dtype = dtype or torch.float16
    ix = HipFlatIndex(x.shape[1], capacity or max(len(x), 1), dtype=dtype, device=0)
    if len(x):
        ix.add(x)
    for k, v in params.items():
        ix.set_param(k, v)
    return ix


def _oracle(q, x, k, id_base=0):
    from oracle.flat_ip import flat_ip_topk

    return flat_ip_topk(q, x, k, id_base=id_base)


def _int_data(seed, n, d, nq):
    rng = np.random.default_rng(seed)
    x = rng.integers(-8, 9, size=(n, d)).astype(np.float16)
    q = rng.integers(-8, 9, size=(nq, d)).astype(np.float16)
    return q, x


def _assert_exact(ix, q, x, k, **kw):
    s, i = ix.search(torch.from_numpy(q).cuda(), k, **kw)
    rs, ri = _oracle(q, x, k, id_base=kw.get("id_base", 0))
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(s.cpu().numpy(), rs)


MANIFEST = json.loads((GOLDEN / "manifest.json").read_text())
# production filter-kernel variants: 1 = 128x128, 42 / 46 = small-batch rings, 8 = persistent 256x256 with
# both waves of a SIMD in lockstep (default for 129..256 queries), 9 = the same with waves 4..7 staggered by one k-step, 14 = the 8-phase
# K loop (kernels_mips_8phase.hip: default above 256 queries)
TILES = [1, 8, 9, 14, 42, 46]


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("name", ["flat_ip_exact_small", "flat_ip_exact_768"])
def test_golden_exact_fixtures(name, tile):
    p = MANIFEST[name]["params"]
    g = np.load(GOLDEN / f"{name}.npz")
    q, x = _int_data(p["seed"], p["n"], p["d"], p["nq"])
    with _index(x, tile=tile) as ix:
        s, i = ix.search(torch.from_numpy(q).cuda(), p["k"])
        np.testing.assert_array_equal(i.cpu().numpy(), g["out_ids"])
        np.testing.assert_array_equal(s.cpu().numpy(), g["out_scores"])
        assert ix.get_stat("last_overflow") == 0


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize(
    "n,d,nq,k",
    [
        (1, 64, 1, 1),
        (5, 64, 3, 10),        # fewer rows than k -> pads
        (255, 128, 7, 32),
        (2048, 64, 130, 100),  # exactly the dense limit
        (2049, 64, 130, 100),  # one row more: too few rows for a bootstrap sample -> dense chunks
        (4200, 64, 130, 5),    # smallest bootstrap (GMAX) schedule
        (8192, 64, 70, 100),
        (10000, 100, 33, 17),  # dim not a multiple of 64
        (70000, 64, 257, 64),  # several geometric chunks, nq not a tile multiple
        (30000, 384, 32, 10),  # config-1 shape (dim 384, batch 32, top-10)
    ],
)
def test_exact_integer_shapes(n, d, nq, k, tile):
    q, x = _int_data(n * 7 + d, n, d, nq)
    with _index(x, tile=tile) as ix:
        _assert_exact(ix, q, x, k)


def test_empty_index_and_empty_query():
    x = np.zeros((0, 64), dtype=np.float16)
    with _index(x, capacity=16) as ix:
        s, i = ix.search(torch.zeros((3, 64), dtype=torch.float16, device="cuda"), 4)
        assert torch.all(i == -1) and torch.all(torch.isneginf(s))
        s, i = ix.search(torch.zeros((0, 64), dtype=torch.float16, device="cuda"), 4)
        assert s.shape == (0, 4) and i.shape == (0, 4)


@pytest.mark.parametrize("n,nq", [(9000, 5), (40000, 300), (40000, 130), (300_000, 130), (300_000, 20)])
def test_k_max_and_large_k(n, nq):
    """k = 2048 / 1000: few rows -> dense chunks; 300 k rows -> a bootstrap with 2k..4k groups and short stages."""
    q, x = _int_data(5, n, 64, nq)
    with _index(x) as ix:
        _assert_exact(ix, q, x, 2048)
        _assert_exact(ix, q, x, 1000)


def test_id_base_offsets_valid_ids_only():
    q, x = _int_data(6, 50, 64, 4)
    with _index(x) as ix:
        s, i = ix.search(torch.from_numpy(q).cuda(), 64, id_base=1000)
        i = i.cpu().numpy()
        assert np.all(i[:, :50] >= 1000) and np.all(i[:, 50:] == -1)
        _assert_exact(ix, q, x, 64, id_base=1000)


def test_incremental_add_and_reset():
    q, x = _int_data(8, 5000, 64, 9)
    with _index(x[:0], capacity=5000) as ix:
        for lo in range(0, 5000, 1234):
            ix.add(x[lo : lo + 1234])
        assert ix.ntotal == 5000
        _assert_exact(ix, q, x, 20)
        ix.reset()
        ix.add(x[:100])
        _assert_exact(ix, q, x[:100], 20)


def test_add_from_device_and_f32_rounding():
    rng = np.random.default_rng(9)
    x32 = rng.normal(size=(3000, 96)).astype(np.float32)
    q32 = rng.normal(size=(11, 96)).astype(np.float32)
    with _index(x32[:0], capacity=3000) as ix:
        ix.add(torch.from_numpy(x32[:1500]).cuda())   # device f32 source
        ix.add(x32[1500:])                            # host f32 source
        stored = ix.stored_rows().cpu().numpy()
        np.testing.assert_array_equal(stored, x32.astype(np.float16))  # round-to-nearest-even, both paths
        s, i = ix.search(torch.from_numpy(q32).cuda(), 10)
        rs, ri = _oracle(q32.astype(np.float16), x32.astype(np.float16), 10)
        np.testing.assert_array_equal(i.cpu().numpy(), ri)
        np.testing.assert_allclose(s.cpu().numpy(), rs, atol=SCORE_TOL, rtol=0)


@pytest.mark.parametrize("nq", [4, 100, 200, 300])
def test_overflow_recovery_stays_exact(nq):
    """A candidate capacity far too small for the stage sizes overflows; the recovery passes (thresholds seeded from the
    incomplete result, then the exhaustive schedule) must stay exact
    (nq selects the kernel family: small-batch rings, persistent with the per-wave survivor lists)."""
    n, d = 60000, 64
    x = np.zeros((n, d), dtype=np.float16)
    x[:, 0] = (np.arange(n) // 32).astype(np.float16)  # non-decreasing, exactly representable (< 2048)
    x[:, 1] = 1
    q = np.zeros((nq, d), dtype=np.float16)
    q[:, 0] = 1
    q[1::4, 0] = -1
    q[2::4, 1] = 3
    # (stages in ROW order: on this score-sorted store every row beats the threshold its stage starts with - the overflow this test is
    # about; the default low-discrepancy stage order calibrates every stage on the whole store and does not overflow here)
    with _index(x, cand_cap=512, dense_rows=256, tile_order=1) as ix:
        _assert_exact(ix, q, x, 40)
        assert ix.get_stat("last_overflow") == 1 and ix.get_stat("last_safe_reruns") >= 1
    with _index(x, cand_cap=512, dense_rows=256) as ix:
        _assert_exact(ix, q, x, 40)
    with _index(x) as ix:  # default capacity: same answer, and the strided bootstrap sample needs no recovery
        _assert_exact(ix, q, x, 40)
        assert ix.get_stat("last_overflow") == 0


@pytest.mark.parametrize("nq", [50, 300])
def test_force_safe_equals_default_schedule(nq):
    q, x = _int_data(12, 40000, 128, nq)
    with _index(x) as ix:
        a = [t.cpu().numpy() for t in ix.search(torch.from_numpy(q).cuda(), 100)]
        ix.set_param("force_safe", 1)
        b = [t.cpu().numpy() for t in ix.search(torch.from_numpy(q).cuda(), 100)]
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.parametrize("nq", [3, 400])
def test_all_equal_scores_tie_break_by_id(nq):
    x = np.ones((20000, 64), dtype=np.float16)
    q = np.ones((nq, 64), dtype=np.float16)
    with _index(x) as ix:
        s, i = ix.search(torch.from_numpy(q).cuda(), 100)
        np.testing.assert_array_equal(i.cpu().numpy(), np.tile(np.arange(100), (nq, 1)))
        assert torch.all(s == 64.0)


@pytest.mark.parametrize("n,nq", [(5000, 6), (150000, 700), (150000, 200), (150000, 100)])
def test_nan_and_inf_rows_never_win(n, nq):
    """All kernel families (small-batch ring, 3+2-slot persistent, persistent with the LDS survivor list) and both the
    dense and the filtered chunks see NaN / inf rows."""
    q, x = _int_data(13, n, 64, nq)
    x = x.copy()
    x[17, 3] = np.nan
    x[4000, 5] = np.inf   # +inf * 0 -> nan for queries with a zero there, +-inf otherwise
    x[n - 3, 7] = np.nan
    x[n // 2, 9] = -np.inf
    with _index(x) as ix:
        s, i = ix.search(torch.from_numpy(q).cuda(), 10)
        s, i = s.cpu().numpy(), i.cpu().numpy()
        assert not np.isnan(s).any()
        assert not (i == 17).any() and not (i == n - 3).any()
        with np.errstate(all="ignore"):
            full = q.astype(np.float64) @ x.astype(np.float64).T
        from oracle.flat_ip import topk_desc_tiebreak

        rs, ri = topk_desc_tiebreak(full, 10)
        np.testing.assert_array_equal(i, ri)


def _check_gaussian(q, x, s, i, k):
    """ids must be a valid exact top-k up to score perturbations below SCORE_TOL."""
    q64, x64 = q.astype(np.float64), x.astype(np.float64)
    rs, ri = _oracle(q, x, k)
    np.testing.assert_allclose(s, rs, atol=SCORE_TOL, rtol=0)
    # every returned id's true score must be within tolerance of the oracle's score at that rank
    true = np.einsum("qkd,qd->qk", x64[i], q64)
    np.testing.assert_allclose(true, rs.astype(np.float64), atol=SCORE_TOL, rtol=0)
    same = (i == ri)
    for r, c in zip(*np.nonzero(~same)):
        # a differing position may only be a swap among near-ties
        assert abs(true[r, c] - rs[r, c]) < SCORE_TOL
    from oracle.flat_ip import recall_at_k

    return recall_at_k(i, ri), float(same.mean())


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
@pytest.mark.parametrize("tile", TILES)
def test_gaussian_matches_fp64_oracle(dtype, tile):
    rng = np.random.default_rng(31)
    n, d, nq, k = 50000, 768, 64, 100
    x = (rng.normal(size=(n, d)) / np.sqrt(d)).astype(np.float32)
    q = (rng.normal(size=(nq, d))).astype(np.float32)
    tdt = getattr(torch, dtype)
    xr = torch.from_numpy(x).to(tdt)
    qr = torch.from_numpy(q).to(tdt)
    with _index(x[:0], dtype=tdt, capacity=n, tile=tile) as ix:
        ix.add(xr.cuda())
        s, i = ix.search(qr.cuda(), k)
    rec, same = _check_gaussian(qr.float().numpy(), xr.float().numpy(), s.cpu().numpy(), i.cpu().numpy(), k)
    assert rec >= 0.999, rec  # swaps among near-ties can move an id across the k-th boundary
    assert same > 0.98


def test_multi_pass_query_batches():
    """nq above the per-pass workspace size (2048) is processed in passes."""
    q, x = _int_data(14, 4096, 64, 2500)
    with _index(x) as ix:
        _assert_exact(ix, q, x, 5)


def test_shard_merge_equals_whole_index():
    """H3: top-k of the union == merge of per-shard top-k (the multi-GPU exchange step), at 1M rows."""
    from vod_amd.index import merge_topk

    g = torch.Generator(device="cuda").manual_seed(1234)
    n, d, nq, k = 1_000_000, 128, 96, 100
    x = torch.randint(-8, 9, (n, d), generator=g, device="cuda").half()
    q = torch.randint(-8, 9, (nq, d), generator=g, device="cuda").half()
    with _index(np.zeros((0, d), np.float16), capacity=n) as whole:
        whole.add(x)
        ws, wi = whole.search(q, k)
    bounds = [0, 300_000, 300_100, 1_000_000]
    parts_s, parts_i = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        with _index(np.zeros((0, d), np.float16), capacity=hi - lo) as sh:
            sh.add(x[lo:hi])
            s, i = sh.search(q, k, id_base=lo)
            parts_s.append(s)
            parts_i.append(i)
    ms, mi = merge_topk(torch.stack(parts_s), torch.stack(parts_i))
    assert torch.equal(mi, wi) and torch.equal(ms, ws)
    # size-independent properties: sorted, ids unique and valid, scores reproduce from the stored rows
    assert torch.all(ws[:, 1:] <= ws[:, :-1])
    assert all(len(set(r.tolist())) == k for r in wi.cpu())
    redo = torch.einsum("qkd,qd->qk", x[wi].float(), q.float())
    assert torch.equal(redo, ws)
    # exactness against the full score matrix (fits: 96 x 1M fp32)
    full = q.float() @ x.float().T
    ts, ti = torch.sort(full, dim=1, descending=True, stable=True)  # stable: ties keep ascending id order
    assert torch.equal(ts[:, :k], ws) and torch.equal(ti[:, :k], wi)


def test_merge_topk_handles_pads_and_ties():
    from oracle.flat_ip import merge_shard_topk
    from vod_amd.index import merge_topk

    rng = np.random.default_rng(2)
    n_sh, nq, k = 8, 33, 100
    s = rng.integers(-3, 4, size=(n_sh, nq, k)).astype(np.float32)
    s = -np.sort(-s, axis=-1)
    ids = np.stack([np.sort(rng.choice(5000, size=(nq, k)), axis=-1) + 10000 * sh for sh in range(n_sh)]).astype(np.int64)
    # per shard: sort by (score desc, id asc) to look like real shard output, then pad some tails
    for sh in range(n_sh):
        for r in range(nq):
            o = np.lexsort((ids[sh, r], -s[sh, r]))
            s[sh, r], ids[sh, r] = s[sh, r][o], ids[sh, r][o]
            cut = rng.integers(0, k + 1)
            if rng.uniform() < 0.3:
                s[sh, r, cut:] = -np.inf
                ids[sh, r, cut:] = -1
    ms, mi = merge_topk(torch.from_numpy(s).cuda(), torch.from_numpy(ids).cuda())
    rs, ri = merge_shard_topk(list(s), list(ids), k)
    np.testing.assert_array_equal(mi.cpu().numpy(), ri)
    np.testing.assert_array_equal(ms.cpu().numpy(), rs)


@pytest.mark.parametrize("n_sh,nq,k,k_out", [(8, 1024, 100, 100), (8, 40, 200, 200), (3, 17, 100, 30), (2, 5, 7, 64), (1, 9, 100, 100), (8, 6, 1000, 1000),
                                             (5, 3, 1638, 2000), (4, 3, 2048, 2048), (7, 11, 33, 33)])
@pytest.mark.parametrize("ordered", [True, False])
def test_merge_topk_ranked_path_and_unsorted_fallback(n_sh, nq, k, k_out, ordered):
    """Sorted shard lists (what the search writes) take the rank-by-binary-search path; lists in any other order must fall
    back to the sorting network and give the same answer.  Duplicated (score, id) pairs across shards, pads, k_out != k."""
    from oracle.flat_ip import merge_shard_topk
    from vod_amd.index import merge_topk

    rng = np.random.default_rng(n_sh * 1000 + k)
    s = rng.integers(-4, 5, size=(n_sh, nq, k)).astype(np.float32)
    ids = np.stack([rng.permutation(max(50_000, nq * k))[: nq * k].reshape(nq, k) + 1_000_000 * sh for sh in range(n_sh)]).astype(np.int64)
    if n_sh > 1:  # the same (score, id) pair reported by two shards (overlapping shards): both copies rank, in shard order
        ids[1, :, 0] = ids[0, :, 0]
        s[1, :, 0] = s[0, :, 0]
    for sh in range(n_sh):
        for r in range(nq):
            if rng.uniform() < 0.3:
                cut = rng.integers(0, k + 1)
                s[sh, r, cut:] = -np.inf
                ids[sh, r, cut:] = -1
            valid = ids[sh, r] >= 0
            o = np.lexsort((ids[sh, r], -s[sh, r], ~valid)) if ordered else rng.permutation(k)
            s[sh, r], ids[sh, r] = s[sh, r][o], ids[sh, r][o]
    ms, mi = merge_topk(torch.from_numpy(s).cuda(), torch.from_numpy(ids).cuda(), k_out=k_out)
    rs, ri = merge_shard_topk(list(s), list(ids), k_out)
    np.testing.assert_array_equal(mi.cpu().numpy(), ri)
    np.testing.assert_array_equal(ms.cpu().numpy(), rs)


def test_packed_record_merge_matches_plain_merge():
    """The one-all-gather exchange format: [scores | ids] records laid end to end, merged in place."""
    from vod_amd.index import PackedTopk, merge_topk

    rng = np.random.default_rng(4)
    world, nq, k = 8, 37, 100
    recs, plain_s, plain_i = [], [], []
    for r in range(world):
        p = PackedTopk(nq, k, torch.device("cuda", 0))
        s = -np.sort(-rng.integers(-5, 6, size=(nq, k)).astype(np.float32), axis=1)
        i = np.sort(rng.choice(100000, size=(nq, k)), axis=1).astype(np.int64) + r * 100000
        for row in range(nq):
            o = np.lexsort((i[row], -s[row]))
            s[row], i[row] = s[row][o], i[row][o]
        p.scores.copy_(torch.from_numpy(s))
        p.ids.copy_(torch.from_numpy(i))
        recs.append(p)
        plain_s.append(p.scores.clone())
        plain_i.append(p.ids.clone())
    gathered = torch.cat([p.buffer for p in recs])
    ms, mi = recs[0].merge_gathered(gathered, world)
    rs, ri = merge_topk(torch.stack(plain_s), torch.stack(plain_i))
    assert torch.equal(ms, rs) and torch.equal(mi, ri)


@pytest.mark.parametrize("nq", [700, 1300, 1600, 1800, 2048])
@pytest.mark.parametrize("tile", [9, 8])
def test_persistent_kernel_odd_qtile_counts(nq, tile):
    """n_qtiles = 3, 6, 7, 8: the persistent grid is rounded to a multiple of 8 * n_qtiles."""
    q, x = _int_data(nq, 24000, 64, nq)
    with _index(x, tile=tile) as ix:
        _assert_exact(ix, q, x, 10)


@pytest.mark.parametrize("tile", [0, 1, 8, 9, 42, 46])
def test_subset_filtered_search(tile):
    """per-row labels + per-query allowed labels; exact top-k over the eligible rows only."""
    from oracle.flat_ip import topk_desc_tiebreak

    rng = np.random.default_rng(77)
    n, d, nq, k = 40000, 64, {1: 100, 42: 50, 46: 100}.get(tile, 300), 50
    q, x = _int_data(78, n, d, nq)
    labels = rng.integers(0, 12, size=n).astype(np.int32)
    subset = np.full((nq, 3), -1, dtype=np.int32)
    for r in range(nq):
        m = rng.integers(0, 4)                       # 0 -> unrestricted query
        subset[r, :m] = rng.choice(12, size=m, replace=False)
    subset[5] = [99, -1, -1]                         # a label nobody carries -> no hits at all
    with _index(x, tile=tile) as ix:
        ix.set_row_labels(labels)
        s, i = ix.search(torch.from_numpy(q).cuda(), k, subset=subset)
        s2, i2 = ix.search(torch.from_numpy(q).cuda(), k)           # filter cleared again
    full = q.astype(np.float64) @ x.astype(np.float64).T
    masked = full.copy()
    for r in range(nq):
        allowed = subset[r][subset[r] >= 0]
        if allowed.size:
            masked[r, ~np.isin(labels, allowed)] = np.nan            # NaN scores never enter the oracle's result
    rs, ri = topk_desc_tiebreak(masked, k)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(s.cpu().numpy(), rs)
    assert np.all(ri[5] == -1)
    us, ui = topk_desc_tiebreak(full, k)
    np.testing.assert_array_equal(i2.cpu().numpy(), ui)


def test_error_behaviour_is_loud():
    from vod_amd._native import NativeLibraryError

    q, x = _int_data(3, 100, 64, 4)
    with _index(x, capacity=100) as ix:
        with pytest.raises(NativeLibraryError, match="index full"):
            ix.add(x[:1])
        with pytest.raises(ValueError):
            ix.add(np.zeros((2, 63), dtype=np.float16))            # wrong dimension
        with pytest.raises(ValueError):
            ix.search(torch.zeros((2, 65), device="cuda"), 3)      # wrong query dimension
        with pytest.raises(NativeLibraryError, match="out of range"):
            ix.search(torch.from_numpy(q).cuda(), 0)
        with pytest.raises(NativeLibraryError, match="out of range"):
            ix.search(torch.from_numpy(q).cuda(), 4096)
        with pytest.raises(NativeLibraryError, match="row labels first"):
            ix.search(torch.from_numpy(q).cuda(), 3, subset=np.zeros((4, 1), dtype=np.int32))
        with pytest.raises(NativeLibraryError, match="unknown parameter"):
            ix.set_param("nope", 1)
        _assert_exact(ix, q, x, 5)                                 # the handle is still usable after the errors


def test_pipelined_searches_finish_in_order_and_stay_exact():
    """Up to 4 searches in flight on one index: `finish` completes the oldest; a 5th enqueue is refused; an
    overflowing search in the middle of the pipeline is recovered without disturbing the others."""
    n, d, k = 60000, 64, 20
    q, x = _int_data(11, n, d, 300)
    with _index(x, capacity=n + 100) as ix:
        outs = []
        for j in range(4):
            outs.append(ix.search_async(torch.from_numpy(q[j * 70 : j * 70 + 70]).cuda(), k))
        with pytest.raises(RuntimeError, match="in flight"):
            ix.search_async(torch.from_numpy(q[:8]).cuda(), k)
        with pytest.raises(RuntimeError, match="in flight"):
            ix.add(x[:10])  # the store may not change under enqueued searches
        for j in range(4):
            ix.finish()
        from oracle.flat_ip import flat_ip_topk

        for j, (s, i) in enumerate(outs):
            rs, ri = flat_ip_topk(q[j * 70 : j * 70 + 70].astype(np.float32), x.astype(np.float32), k)
            np.testing.assert_array_equal(i.cpu().numpy(), ri)
            np.testing.assert_array_equal(s.cpu().numpy(), rs)
        with pytest.raises(RuntimeError, match="no search is pending"):
            ix.finish()
    # a tiny candidate capacity overflows
    xr = np.sort(np.random.default_rng(0).integers(-8, 9, size=(n, 1)), axis=0).astype(np.float16) * np.ones((1, d), np.float16)
    qr = np.ones((40, d), dtype=np.float16)
    with _index(xr) as ix:
        ix.set_param("cand_cap", 256)
        a = ix.search_async(torch.from_numpy(qr[:16]).cuda(), k)
        b = ix.search_async(torch.from_numpy(qr[16:]).cuda(), k)
        ix.finish()
        assert ix.get_stat("last_overflow") == 1
        ix.finish()
        assert ix.get_stat("last_overflow") == 1
        rs, ri = flat_ip_topk(qr.astype(np.float32), xr.astype(np.float32), k)
        np.testing.assert_array_equal(torch.cat([a[1], b[1]]).cpu().numpy(), ri)
        np.testing.assert_array_equal(torch.cat([a[0], b[0]]).cpu().numpy(), rs)


def test_random_shapes_stay_exact():
    """Seeded sweep over shapes the fixed cases do not name: every (n, d, nq, k) goes through the default kernel
    selection and chunk schedule and must reproduce the oracle bit for bit (integer data, tie-heavy)."""
    rng = np.random.default_rng(2026)
    for trial in range(16):
        n = int(rng.choice([1, 17, 300, 1023, 1025, 5000, 33000, 90001, 250000]))
        d = int(rng.choice([8, 64, 72, 128, 200, 384, 768]))
        nq = int(rng.choice([1, 2, 31, 64, 129, 256, 257, 513, 1024, 1500]))
        k = int(rng.choice([1, 3, 10, 64, 100, 128, 130, 500]))
        if n * d > 40_000_000:
            n = 40_000_000 // d
        dtype = torch.float16 if trial % 3 else torch.bfloat16
        x = rng.integers(-8, 9, size=(n, d)).astype(np.float32)
        q = rng.integers(-8, 9, size=(nq, d)).astype(np.float32)
        with _index(x[:0].astype(np.float16), dtype=dtype, capacity=n) as ix:
            ix.add(x)
            s, i = ix.search(torch.from_numpy(q).cuda(), k)
            rs, ri = _oracle(q, x, k)
            np.testing.assert_array_equal(i.cpu().numpy(), ri, err_msg=f"trial {trial}: n={n} d={d} nq={nq} k={k} {dtype}")
            np.testing.assert_array_equal(s.cpu().numpy(), rs, err_msg=f"trial {trial}: n={n} d={d} nq={nq} k={k} {dtype}")


# ---- round 2: C4-shaped workload, bootstrap robustness, recovery corner cases ------------------------------------------


@pytest.mark.parametrize("tile", [0, 8, 9])
def test_c4_shape_exact_bf16_dim1024_k200(tile):
    """BASELINE configs[3] at an oracle-sized N: bf16 store, dim 1024, batch 512, top-200
    (shape source: /root/reference/src/vod_exps/hydra/datasets/msmarco.yaml:9-12 with e5-large dims)."""
    n, d, nq, k = 60000, 1024, 512, 200
    rng = np.random.default_rng(404)
    x = rng.integers(-4, 5, size=(n, d)).astype(np.float32)   # exact in bf16, partial sums exact in fp32
    q = rng.integers(-4, 5, size=(nq, d)).astype(np.float32)
    with _index(x[:0].astype(np.float16), dtype=torch.bfloat16, capacity=n, tile=tile) as ix:
        ix.add(torch.from_numpy(x).cuda())
        s, i = ix.search(torch.from_numpy(q).cuda().to(torch.bfloat16), k)
        assert ix.get_stat("last_overflow") == 0
    rs, ri = _oracle(q, x, k)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(s.cpu().numpy(), rs)


def _clustered(n, d, n_clusters, seed, nq, late=True):
    """Rows sorted by cluster (documents ingested in topic order, /root/reference/src/vod_search/faiss_search/build.py:65-73),
    queries drawn from the LAST clusters: every neighbour of a query sits at the end of the store."""
    rng = np.random.default_rng(seed)
    centers = rng.integers(-3, 4, size=(n_clusters, d)).astype(np.float32)
    assign = np.sort(rng.integers(0, n_clusters, size=n))
    x = centers[assign] + rng.integers(-1, 2, size=(n, d)).astype(np.float32)
    pick = n_clusters - 1 - rng.integers(0, max(1, n_clusters // 10), size=nq) if late else rng.integers(0, n_clusters, size=nq)
    q = centers[pick] + rng.integers(-1, 2, size=(nq, d)).astype(np.float32)
    return q.astype(np.float16), x.astype(np.float16)


@pytest.mark.parametrize("nq,k", [(300, 100), (64, 50), (1024, 100)])
def test_clustered_row_order_needs_no_recovery(nq, k):
    """The bootstrap sample is strided over the WHOLE store, so a topic-sorted corpus does not overflow the candidate
    lists (round 1 re-ran the whole batch in ~N/4096 dense launches here)."""
    q, x = _clustered(200_000, 128, 400, 5, nq)
    with _index(x) as ix:
        _assert_exact(ix, q, x, k)
        assert ix.get_stat("last_overflow") == 0 and ix.get_stat("last_safe_reruns") == 0


@pytest.mark.parametrize("nq", [300, 2300])
def test_only_the_overflowing_queries_are_searched_again(nq):
    """A few queries whose neighbours form one huge block overflow their candidate lists; the recovery pass re-searches
    THEM (as a small batch, results written back through the query map), not the whole batch.  nq = 2300: two workspace
    passes, flagged queries in both."""
    rng = np.random.default_rng(17)
    n, d, k = 120_000, 64, 50
    x = rng.integers(-2, 3, size=(n, d)).astype(np.float16)
    x[70_000:100_000] = x[70_000]        # 30k IDENTICAL rows (duplicated sections): for the queries below every one of
    x[70_000:100_000, :8] = 8            # them ties at the top score, so no threshold can keep them out of the lists
    q = rng.integers(-2, 3, size=(nq, d)).astype(np.float16)
    hot = [3, 77, 150, nq - 1]
    q[hot] = x[70_000]
    with _index(x, cand_cap=4096) as ix:
        _assert_exact(ix, q, x, k)
        assert ix.get_stat("last_overflow") == 1
        # the four hot queries, plus every random query that happens to point towards the duplicated row: a fraction of the batch
        assert len(hot) <= ix.get_stat("last_recovered_queries") < nq // 4
    sub = np.full((nq, 1), -1, dtype=np.int32)  # the same through the subset path: labels follow the query map
    sub[hot] = 1
    labels = (np.arange(n) % 2).astype(np.int32)
    from oracle.flat_ip import topk_desc_tiebreak

    with _index(x, cand_cap=4096) as ix:
        ix.set_row_labels(labels)
        s, i = ix.search(torch.from_numpy(q).cuda(), k, subset=sub)
    full = q.astype(np.float64) @ x.astype(np.float64).T
    for r in hot:
        full[r, labels != 1] = np.nan
    rs, ri = topk_desc_tiebreak(full, k)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(s.cpu().numpy(), rs)


def test_stale_rows_beyond_ntotal_do_not_leak_into_the_bootstrap():
    """reset() keeps the old bytes in the store: the sampled rows of the bootstrap must all lie below ntotal."""
    big = np.full((50000, 64), 8, dtype=np.float16)
    q, x = _int_data(21, 30000, 64, 300)
    with _index(big, capacity=50000) as ix:
        ix.reset()
        ix.add(x)
        _assert_exact(ix, q, x, 100)
        assert ix.get_stat("last_overflow") == 0


def test_pipelined_subset_search_recovers_with_its_own_labels():
    """A subset search whose candidate lists overflow is recovered with ITS labels even though a younger, unrestricted
    search was enqueued behind it (round-1 advisor finding)."""
    from oracle.flat_ip import topk_desc_tiebreak

    rng = np.random.default_rng(5)
    n, d, nq, k = 60000, 64, 200, 20
    x = np.zeros((n, d), dtype=np.float16)
    x[:, 0] = (np.arange(n) // 32).astype(np.float16)
    x[:, 1] = 1
    q = np.zeros((nq, d), dtype=np.float16)
    q[:, 0] = 1
    labels = rng.integers(0, 4, size=n).astype(np.int32)
    subset = np.full((nq, 1), 2, dtype=np.int32)
    with _index(x, cand_cap=256) as ix:
        ix.set_row_labels(labels)
        a = ix.search_async(torch.from_numpy(q).cuda(), k, subset=subset)
        b = ix.search_async(torch.from_numpy(q).cuda(), k)
        ix.finish()
        assert ix.get_stat("last_overflow") == 1
        ix.finish()
    full = q.astype(np.float64) @ x.astype(np.float64).T
    masked = full.copy()
    masked[:, labels != 2] = np.nan
    rs, ri = topk_desc_tiebreak(masked, k)
    np.testing.assert_array_equal(a[1].cpu().numpy(), ri)
    np.testing.assert_array_equal(a[0].cpu().numpy(), rs)
    us, ui = topk_desc_tiebreak(full, k)
    np.testing.assert_array_equal(b[1].cpu().numpy(), ui)


def _full_size_properties(n, d, nq, k, dtype, chunk=500_000, n_check=64, normalized=False):
    """Size-independent properties at a BASELINE config's full size: sorted, unique valid ids, scores reproduce from the
    stored rows, shard-merge == whole, and exactness of a query sample against a chunked fp32 matmul of the stored rows."""
    from vod_amd.index import merge_topk

    dev = torch.device("cuda", 0)
    tdt = getattr(torch, dtype)
    with _index(np.zeros((0, d), np.float16), dtype=tdt, capacity=n) as whole:
        for c, lo in enumerate(range(0, n, chunk)):
            g = torch.Generator(device=dev).manual_seed(1234 + c)
            rows = torch.randn((min(chunk, n - lo), d), generator=g, device=dev, dtype=torch.float32)
            whole.add((10.0 * torch.nn.functional.normalize(rows, dim=1) if normalized else rows).to(tdt))
        gq = torch.Generator(device=dev).manual_seed(4321)
        q = torch.randn((nq, d), generator=gq, device=dev, dtype=torch.float32)
        q = (10.0 * torch.nn.functional.normalize(q, dim=1) if normalized else q).to(tdt)
        ws, wi = whole.search(q, k)
        assert whole.get_stat("last_overflow") == 0
        assert torch.all(ws[:, 1:] <= ws[:, :-1])
        assert torch.all((wi >= 0) & (wi < n))
        srt = torch.sort(wi, dim=1).values
        assert torch.all(srt[:, 1:] != srt[:, :-1])
        # scores reproduce from the stored rows (fp32 accumulation in another order: 1e-3, the north-star tolerance)
        n_redo = 8  # rows are read back one by one through the C-ABI
        gathered = torch.stack([whole.stored_rows(int(r), 1)[0] for r in wi[:n_redo].flatten().tolist()]).view(n_redo, k, d)
        redo = torch.einsum("qkd,qd->qk", gathered.float(), q[:n_redo].float())
        assert torch.allclose(redo, ws[:n_redo], atol=SCORE_TOL, rtol=0)
        # exactness of the sample: chunked brute force over the stored rows
        best_s = torch.full((n_check, k), float("-inf"), device=dev)
        best_i = torch.full((n_check, k), -1, dtype=torch.int64, device=dev)
        qs = q[:n_check].float()
        for lo in range(0, n, 1_000_000):
            blk = whole.stored_rows(lo, min(1_000_000, n - lo)).float()
            sc = qs @ blk.T
            ts, ti = torch.topk(sc, k, dim=1)
            cs = torch.cat([best_s, ts], dim=1)
            ci = torch.cat([best_i, ti + lo], dim=1)
            o = torch.topk(cs, k, dim=1)
            best_s, best_i = o.values, torch.gather(ci, 1, o.indices)
            del blk, sc
        hits = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(wi[:n_check].cpu(), best_i.cpu()))
        assert hits / float(n_check * k) >= 0.999
        assert float((ws[:n_check] - best_s).abs().max()) < SCORE_TOL
        # shard-merge == whole: two shards searched in place of the whole store (rows copied shard by shard)
        bounds = [0, (n * 3 // 8) // 256 * 256 + 17, n]
        parts_s, parts_i = [], []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            with _index(np.zeros((0, d), np.float16), dtype=tdt, capacity=hi - lo) as sh:
                for b0 in range(lo, hi, 2_000_000):
                    sh.add(whole.stored_rows(b0, min(2_000_000, hi - b0)))
                s, i = sh.search(q, k, id_base=lo)
                parts_s.append(s)
                parts_i.append(i)
        ms, mi = merge_topk(torch.stack(parts_s), torch.stack(parts_i))
        assert torch.equal(mi, wi) and torch.equal(ms, ws)


def test_full_size_c3_properties():
    """BASELINE configs[2]: 10 M x 768 fp16, batch 1024, top-100 (15.4 GB store + 15.4 GB of shard copies)."""
    _full_size_properties(10_000_000, 768, 1024, 100, "float16")


def test_full_size_c4_properties():
    """BASELINE configs[3]: 40 M x 1024 bf16, batch 512, top-200 on ONE device (82 GB store + 82 GB of shard copies)."""
    _full_size_properties(40_000_000, 1024, 512, 200, "bfloat16")


# ---- round 3: BASELINE configs[1] (C2) at its full size ------------------------------------------------------------------


def test_full_size_c2_properties():
    """BASELINE configs[1]: 1 M x 768 fp16, batch 256, top-100 on one GPU (the single-q-tile / `nt` cache-policy path of the
    persistent kernel, different code from C3's four q-tiles)."""
    _full_size_properties(1_000_000, 768, 256, 100, "float16", n_check=256)


def test_full_size_c2_scaled_cosine_embeddings():
    """SURVEY 8(d)'s second C2 input: rows and queries L2-normalised x 10 (the encoder's `mpool-scaled-cosine` pooler, scaler 100 =
    sqrt(100) per side): a score distribution 8x narrower than N(0, 1) rows give, i.e. many more near-ties around the k-th score."""
    _full_size_properties(1_000_000, 768, 256, 100, "float16", n_check=256, normalized=True)


@pytest.fixture(scope="module")
def c2_integer_case():
    """C2-sized integer-valued store (every partial sum exact in fp32, many ties) + the blocked fp64 oracle's answer."""
    rng = np.random.default_rng(2002)
    n, d, nq = 1_000_000, 768, 256
    x = np.empty((n, d), dtype=np.float16)
    for lo in range(0, n, 100_000):  # int8 draws, chunked: an int64 draw of the whole store would take 6 GB
        x[lo : lo + 100_000] = rng.integers(-8, 9, size=(100_000, d), dtype=np.int8)
    q = rng.integers(-8, 9, size=(nq, d), dtype=np.int8).astype(np.float16)
    rs, ri = _oracle(q, x, 100)
    return q, x, rs, ri


@pytest.mark.parametrize("tile", [0, 8, 9])
def test_c2_shape_exact_integer(c2_integer_case, tile):
    """C2 (1 M x 768 fp16, nq 256, k 100) bit for bit against the oracle: ids and scores, production tiles."""
    q, x, rs, ri = c2_integer_case
    with _index(x, tile=tile) as ix:
        s, i = ix.search(torch.from_numpy(q).cuda(), 100)
        assert ix.get_stat("last_chunks") >= 3  # bootstrap + filter stages, not the dense schedule
        np.testing.assert_array_equal(i.cpu().numpy(), ri)
        np.testing.assert_array_equal(s.cpu().numpy(), rs)
        # the same rows as two shards + merge, and with an id base
        s2, i2 = ix.search(torch.from_numpy(q).cuda(), 100, id_base=7_000_000)
        np.testing.assert_array_equal(i2.cpu().numpy(), ri + 7_000_000)
        np.testing.assert_array_equal(s2.cpu().numpy(), rs)


@pytest.mark.parametrize("tile,nq", [(8, 1024), (9, 1024), (8, 200), (46, 100), (42, 33)])
def test_repeated_searches_are_bit_identical(tile, nq):
    """Candidates reach the lists in a different order on every run (atomics, wave timing); the answer may not depend on it.
    60 searches of the same Gaussian batch on 1 M rows must return the same bits, and the first must be exact on a sample."""
    dev = torch.device("cuda", 0)
    n, d, k = 1_000_000, 256, 100
    g = torch.Generator(device=dev).manual_seed(77)
    with _index(np.zeros((0, d), np.float16), capacity=n, tile=tile) as ix:
        for _ in range(4):
            ix.add(torch.randn((n // 4, d), generator=g, device=dev).half())
        q = torch.randn((nq, d), generator=g, device=dev).half()
        s0, i0 = ix.search(q, k)
        s0, i0 = s0.clone(), i0.clone()
        for it in range(60):
            s, i = ix.search(q, k)
            assert torch.equal(i, i0) and torch.equal(s, s0), f"run {it} differs"
        full = q[:16].float() @ ix.stored_rows().float().T
        ts, ti = torch.topk(full, k, dim=1)
        assert torch.equal(torch.sort(ti, dim=1).values, torch.sort(i0[:16], dim=1).values) or float((ts - s0[:16]).abs().max()) < SCORE_TOL
        assert float((ts - s0[:16]).abs().max()) < SCORE_TOL


def test_merge_of_more_than_8192_entries_runs_in_levels():
    """8 shards x top-2048 (what an 8-GPU index returns for k = 2048) and 16 x 1024: more than one launch can hold in LDS, so
    the C-ABI merges groups of shards first; the result must equal the one-shot oracle, ties and pads included."""
    from oracle.flat_ip import merge_shard_topk
    from vod_amd.index import merge_topk

    rng = np.random.default_rng(88)
    for S, nq, k, k_out in [(8, 40, 2048, 2048), (16, 33, 1024, 700), (5, 7, 4096, 4096), (9, 12, 1000, 2000)]:
        scores = (np.round(rng.normal(size=(S, nq, k)) * 4) / 2).astype(np.float32)
        ids = np.full((S, nq, k), -1, dtype=np.int64)
        for s_ in range(S):
            for r in range(nq):
                nv = k if rng.random() < 0.6 else int(rng.integers(0, k + 1))
                ids[s_, r, :nv] = s_ * 1_000_000 + rng.choice(200_000, size=nv, replace=False)
                scores[s_, r, nv:] = -np.inf
                o = np.lexsort((ids[s_, r, :nv], -scores[s_, r, :nv]))
                scores[s_, r, :nv], ids[s_, r, :nv] = scores[s_, r, :nv][o], ids[s_, r, :nv][o]
        gs, gi = merge_topk(torch.from_numpy(scores).cuda(), torch.from_numpy(ids).cuda(), k_out)
        rs, ri = merge_shard_topk(list(scores), list(ids), k_out)
        np.testing.assert_array_equal(gi.cpu().numpy(), ri)
        np.testing.assert_array_equal(gs.cpu().numpy(), rs)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("n", [70_001, 131_072, 2_303])
def test_stage_tile_order_does_not_change_the_result(n, tile):
    """FILTER stages walk the store's 256-row tiles in a low-discrepancy order by default (`tile_order` 0) or in row order (1): ids and
    scores are identical, on a score-sorted store (where the ORDER decides how many survivors a stage emits), with a partly filled
    last tile, an id base, a subset filter and rows added in pieces - and equal to the oracle."""
    rng = np.random.default_rng(n + tile)
    d, nq, k = 96, (60 if tile in (1, 42, 46) else 290), 33
    x = rng.integers(-8, 9, size=(n, d)).astype(np.float16)
    w = rng.integers(-8, 9, size=(d,)).astype(np.float32)
    x = x[np.argsort(x.astype(np.float32) @ w, kind="stable")]  # rows sorted by their score against one direction
    q = np.clip(w[None, :] + rng.integers(-2, 3, size=(nq, d)), -8, 8).astype(np.float16)  # ... which the queries look along
    labels = rng.integers(0, 4, size=n).astype(np.int32)
    subset = np.full((nq, 1), -1, dtype=np.int32)
    subset[::3, 0] = 2
    res = []
    for order in (0, 1):
        with _index(x[:1000], capacity=n, tile=tile, tile_order=order) as ix:
            ix.add(x[1000:])
            ix.set_row_labels(labels)
            res.append([t.cpu().numpy() for t in ix.search(torch.from_numpy(q).cuda(), k, id_base=7)]
                       + [t.cpu().numpy() for t in ix.search(torch.from_numpy(q).cuda(), k, subset=subset)])
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, b)
    rs, ri = _oracle(q, x, k, id_base=7)
    np.testing.assert_array_equal(res[0][1], ri)
    np.testing.assert_array_equal(res[0][0], rs)


@pytest.mark.parametrize("lanes", [0, 1, 2])
def test_pipelined_searches_on_two_lanes_equal_synchronous_ones(lanes):
    """Up to four searches in flight alternate between the index's two workspaces / streams (`lanes`: 0 = auto for batches of one query
    tile, 2 = always); batches of different sizes, k and kernel families, one with candidate-list overflow (recovery on the lane's own
    stream), enqueued from a NON-default stream with the queries produced on that stream right before the call: every result must equal
    the one a synchronous search returns."""
    rng = np.random.default_rng(lanes)
    n, d = 150_000, 128
    x = rng.integers(-8, 9, size=(n, d)).astype(np.float16)
    x[100_000:103_000] = x[7]  # 3,000 tied copies: the 256-slot candidate lists below overflow for queries aligned with row 7
    batches = [(200, 50), (64, 10), (300, 100), (17, 7), (256, 100), (1, 1), (130, 33), (700, 20)]
    qs = [rng.integers(-8, 9, size=(nq, d)).astype(np.float16) for nq, _ in batches]
    qs[2][:5] = x[7]
    with _index(x) as ref:
        want = [[t.clone() for t in ref.search(torch.from_numpy(q).cuda(), k)] for q, (_, k) in zip(qs, batches)]
    side = torch.cuda.Stream()
    with _index(x, lanes=lanes, cand_cap=256) as ix:
        for _round in range(3):
            outs = []
            with torch.cuda.stream(side):
                for j, (q, (nq, k)) in enumerate(zip(qs, batches)):
                    tq = (torch.from_numpy(q).cuda(non_blocking=True).float() * 1.0).half()  # produced on `side` just before the enqueue
                    outs.append(ix.search_async(tq, k))
                    if len(outs) - sum(o is None for o in outs) > 3:
                        pass
                    if j % 4 == 3:  # four in flight: finish them oldest first
                        for _ in range(4):
                            ix.finish()
            side.synchronize()
            for (s, i), (ws, wi) in zip(outs, want):
                assert torch.equal(i, wi) and torch.equal(s, ws)
