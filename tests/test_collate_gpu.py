"""GPU test of the collate-side flow around the dense search (config 5: hybrid search fan-out -> merge -> sampling):
`async_hybrid_search` with a real HIP index behind an in-process dense client and table-driven lookup/sparse
clients, checked against the CPU oracle stage by stage."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_async_hybrid_search_end_to_end():
    from oracle.flat_ip import flat_ip_topk
    from oracle.hybrid import merge_hybrid as oracle_merge
    from vod_amd import types as vt
    from vod_amd.core.search import async_hybrid_search
    from vod_amd.index import HipFlatIndex
    from vod_amd.search.base import SearchClient

    rng = np.random.default_rng(11)
    n, d, B, K = 50_000, 128, 16, 32
    x = rng.integers(-6, 7, size=(n, d)).astype(np.float16)
    q = rng.integers(-6, 7, size=(B, d)).astype(np.float16)

    class Dense(SearchClient):  # in-process stand-in for HipMipsClient: same search() contract, real HIP index
        def __init__(self):
            self.ix = HipFlatIndex(d, n)
            self.ix.add(x)

        def ping(self):
            return True

        def search(self, *, text, vector=None, subset_ids=None, ids=None, shard=None, top_k=3):  # noqa: ARG002
            s, i = self.ix.search(vector, top_k)
            return vt.RetrievalBatch(scores=s.cpu().numpy(), indices=i.cpu().numpy())

    sp_idx = np.stack([rng.choice(n, size=K, replace=False) for _ in range(B)]).astype(np.int64)
    sp_scr = -np.sort(-rng.gamma(2.0, 4.0, size=(B, K)).astype(np.float32), axis=1)
    gold = np.stack([rng.choice(n, size=3, replace=False) for _ in range(B)]).astype(np.int64)

    class Sparse(SearchClient):  # BM25 stand-in: returns the gold sections when `ids` is given (the lookup request)
        requires_vectors = False

        def ping(self):
            return True

        def search(self, *, text, vector=None, subset_ids=None, ids=None, shard=None, top_k=3):  # noqa: ARG002
            if ids is not None:
                idx = np.full((B, top_k), -1, dtype=np.int64)
                scr = np.full((B, top_k), -np.inf, dtype=np.float32)
                idx[:, :3], scr[:, :3] = gold, 1.0
                return vt.RetrievalBatch(scores=scr, indices=idx, labels=(scr > -np.inf).astype(np.int64))
            return vt.RetrievalBatch(scores=sp_scr[:, :top_k].copy(), indices=sp_idx[:, :top_k].copy())

    dense = Dense()
    merged, raw = async_hybrid_search(
        text=["q"] * B, shards=["s"] * B, vector=q, section_ids=[[str(g) for g in row] for row in gold], top_k=K,
        clients={"dense": dense, "sparse": Sparse()}, weights={"dense": 1.0, "sparse": 0.5}, lookup_engine_name="sparse",
    )
    ds, di = flat_ip_topk(q, x, K)
    l_idx = np.full((B, K), -1, dtype=np.int64)
    l_idx[:, :3] = gold
    l_lbl = (l_idx >= 0).astype(np.int64)
    o_idx, o_scr, o_lbl, o_raw = oracle_merge((l_idx, np.zeros((B, K), np.float32), l_lbl), {"dense": (di, ds), "sparse": (sp_idx, sp_scr)},
                                              {"dense": 1.0, "sparse": 0.5})
    np.testing.assert_array_equal(merged.indices, o_idx)
    np.testing.assert_array_equal(merged.scores, o_scr)
    np.testing.assert_array_equal(merged.labels, o_lbl)
    np.testing.assert_array_equal(raw["dense"], o_raw["dense"])
    np.testing.assert_array_equal(raw["sparse"], o_raw["sparse"])
    assert "search_time" in merged.meta and "dense_search_time" in merged.meta
    # gold sections come first (first-seen order) and carry label 1
    assert np.all(merged.indices[:, :3] == gold) and np.all(merged.labels[:, :3] == 1)
    dense.ix.close()
