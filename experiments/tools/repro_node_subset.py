#!/usr/bin/env python3
"""Isolate fuzz_search trial 1051 (seed 404): node index, 1 shard, subset filter, nq 300, k 128, n 9000."""
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle.flat_ip import topk_desc_tiebreak  # noqa: E402
from vod_amd.index import HipFlatIndex, HipNodeIndex  # noqa: E402

rng = np.random.default_rng(1520506472)
n, d, nq, k = 9000, 65, 300, 128
base = rng.integers(-8, 9, size=(max(1, n // 50), d)).astype(np.float32)
x = np.clip(base[rng.integers(0, len(base), size=n)], -8, 8)
q = rng.integers(-8, 9, size=(nq, d)).astype(np.float32)
cuts = sorted(set([0, n] + [int(v) for v in rng.integers(0, n + 1, size=3)]))
labels = rng.integers(0, 6, size=n).astype(np.int32)
subset = np.full((nq, 2), -1, dtype=np.int32)
for r in range(nq):
    m = int(rng.integers(0, 3))
    subset[r, :m] = rng.choice(7, size=m, replace=False)
full = q.astype(np.float64) @ x.astype(np.float64).T
unf_s, unf_i = topk_desc_tiebreak(full.copy(), k)
for r in range(nq):
    allowed = subset[r][subset[r] >= 0]
    if allowed.size:
        full[r, ~np.isin(labels, allowed)] = np.nan
rs, ri = topk_desc_tiebreak(full, k)


def report(name, gs, gi):
    bad = np.argwhere((gi != ri).any(axis=1)).ravel()
    print(f"{name:>44}: {len(bad)} bad rows of {nq}; first {bad[:6].tolist()}; equals UNFILTERED answer on the bad rows: "
          f"{bool(len(bad)) and bool((gi[bad] == unf_i[bad]).all())}; restricted rows among bad: {int((subset[bad] >= 0).any(axis=1).sum()) if len(bad) else 0}")
    if len(bad):
        r = int(bad[0])
        cols = np.argwhere(gi[r] != ri[r]).ravel()
        print(f"{'':>44}  row {r}: subset {subset[r].tolist()}, first differing cols {cols[:5].tolist()}, got {gi[r][cols[:3]].tolist()} want {ri[r][cols[:3]].tolist()}")


for params in ({}, {"cand_cap": 512, "sample_div": 2}, {"sample_div": 2}, {"cand_cap": 512}):
    for chunks in (True, False):
        for kind in ("node1", "plain"):
            if kind == "node1":
                ix = HipNodeIndex(d, n, [0], dtype=torch.float16)
            else:
                ix = HipFlatIndex(d, n, dtype=torch.float16, device=0)
            for key, v in params.items():
                ix.set_param(key, v)
            if chunks:
                for lo, hi in zip(cuts[:-1], cuts[1:]):
                    if hi > lo:
                        ix.add(x[lo:hi])
            else:
                ix.add(x)
            ix.set_row_labels(labels)
            if kind == "node1":
                gs, gi = ix.search(q, k, subset=subset)
            else:
                ts, ti = ix.search(torch.from_numpy(q).cuda(), k, subset=subset)
                gs, gi = ts.cpu().numpy(), ti.cpu().numpy()
            report(f"{kind} chunks={chunks} {params}", gs, gi)
            ix.close()
print("cuts", cuts)
