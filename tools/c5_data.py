"""Config 5 inputs: B = 64 queries, K = 128 hits per engine x 3 engines (lookup, dense, sparse), ids drawn without
replacement from 1e6 with ~30 % dense-sparse overlap, dense scores N(0, 1) sorted, sparse (bm25) scores Gamma(2, 4), 10 % of the
rows padded (-1 / -inf tails), the lookup engine returning each question's few gold sections (label 1) and pads."""
import numpy as np
import torch

B, K, H, NS = 64, 128, 768, 32


def make(dev, seed: int = 0):
    rng = np.random.default_rng(seed)
    ids = lambda: np.stack([rng.choice(1_000_000, size=K, replace=False) for _ in range(B)])  # noqa: E731
    d_idx, s_idx = ids(), ids()
    s_idx[:, :40] = d_idx[:, rng.permutation(K)[:40]]  # ~30 % of the sparse hits are dense hits too
    d_scr = -np.sort(-rng.normal(size=(B, K)).astype(np.float32), axis=1)
    s_scr = -np.sort(-rng.gamma(2.0, 4.0, size=(B, K)).astype(np.float32), axis=1)
    for r in rng.choice(B, size=B // 10, replace=False):  # padded rows: an engine returned fewer than K hits
        cut = int(rng.integers(K // 4, K))
        d_idx[r, cut:], d_scr[r, cut:] = -1, -np.inf
        cut = int(rng.integers(K // 4, K))
        s_idx[r, cut:], s_scr[r, cut:] = -1, -np.inf
    l_idx = np.full((B, K), -1, dtype=np.int64)
    l_lbl = np.zeros((B, K), dtype=np.int64)
    for r in range(B):  # 1-4 gold sections per question; half of them are also dense hits
        g = int(rng.integers(1, 5))
        gold = rng.choice(1_000_000, size=g, replace=False)
        gold[: g // 2] = d_idx[r, rng.choice(32, size=g // 2, replace=False)]
        l_idx[r, :g], l_lbl[r, :g] = gold, 1
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    return t(l_idx), t(l_lbl), {"dense": (t(d_idx), t(d_scr)), "sparse": (t(s_idx), t(s_scr))}, {"dense": 1.0, "sparse": 1.0}
