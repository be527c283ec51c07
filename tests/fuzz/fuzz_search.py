"""Randomised parity campaign for the search path (not part of the test suite: run it on a GPU box when the kernels or the
stage planner change).  Every trial draws a store, a query batch, k, a kernel variant, planner knobs (small candidate lists
force overflow recovery, small sample divisors / growth change the stage list) and optionally a subset filter, an id base,
an incremental build or a reset-and-refill, and compares ids AND scores bit for bit with the fp64 oracle (integer-valued
rows: every partial sum is exact in fp32, ties everywhere).

    python3 tests/fuzz/fuzz_search.py [--trials 400] [--seed 1] [--seconds 600]
"""
import argparse
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))  # the repo root
from oracle.flat_ip import topk_desc_tiebreak  # noqa: E402
from vod_amd.index import HipFlatIndex, HipNodeIndex  # noqa: E402


def draw(rng):
    c = {}
    c["n"] = int(rng.choice([1, 2, 63, 255, 256, 257, 1000, 2047, 2049, 4097, 9000, 33000, 70000, 131072, 200001, 400000]))
    c["d"] = int(rng.choice([1, 7, 8, 33, 64, 65, 96, 128, 200, 256, 384, 768, 1024]))
    if c["n"] * c["d"] > 60_000_000:
        c["n"] = 60_000_000 // c["d"]
    c["nq"] = int(rng.choice([1, 2, 5, 31, 32, 33, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513, 700, 1024, 1025, 2300]))
    if c["nq"] * c["n"] > 250_000_000:  # the oracle's score matrix (fp64)
        c["nq"] = max(1, 250_000_000 // c["n"])
    c["k"] = int(rng.choice([1, 2, 7, 10, 63, 64, 65, 100, 128, 129, 200, 256, 500, 1000, 2048]))
    c["dtype"] = str(rng.choice(["f16", "bf16"]))
    c["tile"] = int(rng.choice([0, 0, 0, 1, 8, 9, 14, 42, 46]))
    if c["tile"] == 42 and c["nq"] > 2300:
        c["tile"] = 0
    c["data"] = str(rng.choice(["uniform", "uniform", "few_values", "sorted", "duplicates", "constant"]))
    c["cand_cap"] = int(rng.choice([0, 0, 0, 256, 512, 1024, 4096]))
    c["dense_rows"] = int(rng.choice([0, 0, 256, 1024, 8192]))
    c["sample_div"] = int(rng.choice([0, 0, 2, 8, 32, 500]))
    c["growth"] = int(rng.choice([0, 0, 125, 200, 400, 1600, 25600]))
    c["small_chunk_tiles"] = int(rng.choice([-1, -1, 0, 64, 100000]))
    c["subset"] = bool(rng.random() < 0.2)
    c["id_base"] = int(rng.choice([0, 0, 12345, 1 << 33]))
    c["build"] = str(rng.choice(["once", "once", "chunks", "reset_refill"]))
    c["rng_seed"] = int(rng.integers(0, 2**31))
    # one trial in six goes through the one-process node index (`vodhip_node_index_*`): 1-5 shards, all on device 0
    c["node_shards"] = int(rng.choice([0, 0, 0, 0, 0, 1, 2, 3, 5])) if rng.random() < 0.5 else 0
    # one trial in three runs a VODHIP_EXACT_F32 store; two thirds of those on inputs the scan dtype CANNOT represent ("lossy": rows or
    # queries with more significant bits than fp16 / bf16 keep, sized so that every float32 partial sum is still exact) - the result must
    # equal the float64 oracle on the UNROUNDED values bit for bit; "exact_expand" 1 = the scan lists only k rows (band pass every time)
    c["exact"] = bool(rng.random() < 0.34)
    # round 6: "outliers" = a few rows (1 .. 80: more than the 64 the outlier list holds) scaled by 2^6 .. 2^14 - norms far above the rest of
    # the store's, and in an fp16 store values BEYOND the scan dtype's range (8 x 2^14 > 65504: the scan copy saturates); every float32
    # partial sum stays an exact integer
    c["lossy"] = str(rng.choice(["none", "rows", "queries", "outliers"])) if c["exact"] else "none"
    c["exact_expand"] = int(rng.choice([0, 0, 1, 100, 300])) if c["exact"] else 0
    c["exact_adapt"] = int(rng.choice([1, 1, 0])) if c["exact"] else 1
    if c["lossy"] == "outliers":
        c["d"] = min(c["d"], 32 if c["dtype"] == "f16" else 128)  # 2^17 x 4 x 32 = 2^24
    elif c["lossy"] != "none" and c["dtype"] == "f16":
        c["d"] = min(c["d"], 128)  # wide values x 3-bit values x d terms must stay below 2^24
    return c


def lossy_inputs(rng, c, x, q):
    """Integer inputs whose products and partial sums are exact in float32 but which the scan dtype rounds: the wide side has 13 (fp16
    store, d <= 128) or 10 (bf16) significant bits, the other side 3."""
    wide = 4096 if c["dtype"] == "f16" else 512
    if c["lossy"] == "rows":
        scale = rng.integers(1, wide // 8 + 1, size=(x.shape[0], 1)).astype(np.float32)  # keeps the trial's row structure (ties, duplicates, order)
        x = np.clip(x, -8, 8) * scale + rng.integers(-3, 4, size=x.shape).astype(np.float32)
        x = np.clip(x, -wide, wide)
        q = np.clip(q, -4, 4)
    elif c["lossy"] == "queries":
        q = rng.integers(-wide, wide + 1, size=q.shape).astype(np.float32)
        x = np.clip(x, -4, 4)
    elif c["lossy"] == "outliers":
        q = np.clip(q, -4, 4)
        x = np.clip(x, -8, 8).copy()
        m = min(x.shape[0], int(rng.choice([1, 2, 5, 40, 80])))
        rows = rng.choice(x.shape[0], size=m, replace=False)
        top = 14 if c["dtype"] == "f16" else 10
        x[rows] *= np.exp2(rng.integers(6, top + 1, size=(m, 1))).astype(np.float32)
    return x, q


def make_rows(rng, kind, n, d):
    if kind == "uniform":
        return rng.integers(-8, 9, size=(n, d)).astype(np.float32)
    if kind == "few_values":
        return rng.integers(-1, 2, size=(n, d)).astype(np.float32)
    if kind == "constant":
        return np.full((n, d), 2.0, dtype=np.float32)
    if kind == "duplicates":
        base = rng.integers(-8, 9, size=(max(1, n // 50), d)).astype(np.float32)
        return base[rng.integers(0, len(base), size=n)]
    x = rng.integers(-8, 9, size=(n, d)).astype(np.float32)  # sorted: rows ordered by their score against a fixed direction
    w = rng.integers(-8, 9, size=(d,)).astype(np.float32)
    return x[np.argsort(x @ w, kind="stable")]


def run_node_trial(c):
    """The same draw behind `HipNodeIndex` (host buffers in and out, row shards on device 0): id_base 0, host rows only."""
    rng = np.random.default_rng(c["rng_seed"])
    n, d, nq, k = c["n"], c["d"], c["nq"], c["k"]
    tdt = torch.float16 if c["dtype"] == "f16" else torch.bfloat16
    lim = 4 if c["dtype"] == "bf16" else 8
    x = np.clip(make_rows(rng, c["data"], n, d), -lim, lim)
    q = rng.integers(-lim, lim + 1, size=(nq, d)).astype(np.float32)
    x, q = lossy_inputs(rng, c, x, q)
    labels = subset = None
    with HipNodeIndex(d, n, [0] * c["node_shards"], dtype=tdt, exact_f32=c["exact"]) as nx:
        if c["cand_cap"] and c["cand_cap"] >= k:
            nx.set_param("cand_cap", c["cand_cap"])
        if c["exact_expand"]:
            nx.set_param("exact_expand", c["exact_expand"])
        if c.get("exact_adapt", 1) == 0:
            nx.set_param("exact_adapt", 0)
        for key in ("dense_rows", "sample_div", "growth"):
            if c[key]:
                nx.set_param(key, c[key])
        if c["tile"]:
            nx.set_param("tile", c["tile"])
        if c["build"] == "reset_refill":
            nx.add(rng.integers(-lim, lim + 1, size=(n, d)).astype(np.float32))
            nx.reset()
        if c["build"] == "chunks":
            cuts = sorted(set([0, n] + [int(v) for v in rng.integers(0, n + 1, size=3)]))
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                if hi > lo:
                    nx.add(x[lo:hi])
        else:
            nx.add(x)
        assert nx.ntotal == n
        if c["subset"]:
            labels = rng.integers(0, 6, size=n).astype(np.int32)
            subset = np.full((nq, 2), -1, dtype=np.int32)
            for r in range(nq):
                m = int(rng.integers(0, 3))
                subset[r, :m] = rng.choice(7, size=m, replace=False)
            nx.set_row_labels(labels)
        if subset is None and rng.random() < 0.5:  # device-side entry: queries and results on devices[0]
            ts, ti = nx.search(torch.from_numpy(q).cuda(), k)
            gs, gi = ts.cpu().numpy(), ti.cpu().numpy()
        else:
            gs, gi = nx.search(q, k, subset=subset)
        gs2, gi2 = nx.search(q, k, subset=subset)  # determinism / stale buffers
    full = q.astype(np.float64) @ x.astype(np.float64).T
    if subset is not None:
        for r in range(nq):
            allowed = subset[r][subset[r] >= 0]
            if allowed.size:
                full[r, ~np.isin(labels, allowed)] = np.nan
    rs, ri = topk_desc_tiebreak(full, k)
    assert np.array_equal(gi, ri), f"node index ({c['node_shards']} shards): ids differ: first bad row {np.argwhere((gi != ri).any(axis=1))[:3].ravel().tolist()}"
    assert np.array_equal(gs, rs), "node index: scores differ"
    assert np.array_equal(gi2, gi) and np.array_equal(gs2, gs), "node index: second search differs"
    return {"last_chunks": 0, "last_safe_reruns": 0, "last_recovered_queries": 0, "last_exact_band_queries": 0}


def run_trial(c):
    if c.get("node_shards"):
        return run_node_trial(c)
    rng = np.random.default_rng(c["rng_seed"])
    n, d, nq, k = c["n"], c["d"], c["nq"], c["k"]
    tdt = torch.float16 if c["dtype"] == "f16" else torch.bfloat16
    lim = 4 if c["dtype"] == "bf16" else 8  # bf16 keeps 8 significant bits: stay exact
    x = np.clip(make_rows(rng, c["data"], n, d), -lim, lim)
    q = rng.integers(-lim, lim + 1, size=(nq, d)).astype(np.float32)
    x, q = lossy_inputs(rng, c, x, q)
    labels = subset = None
    with HipFlatIndex(d, n, dtype=tdt, device=0, exact_f32=c["exact"]) as ix:
        if c["cand_cap"] and c["cand_cap"] >= k:
            ix.set_param("cand_cap", c["cand_cap"])
        if c["exact_expand"]:
            ix.set_param("exact_expand", c["exact_expand"])
        if c.get("exact_adapt", 1) == 0:
            ix.set_param("exact_adapt", 0)
        for key in ("dense_rows", "sample_div", "growth"):
            if c[key]:
                ix.set_param(key, c[key])
        if c["small_chunk_tiles"] >= 0:
            ix.set_param("small_chunk_tiles", c["small_chunk_tiles"])
        if c["tile"]:
            ix.set_param("tile", c["tile"])
        if c["build"] == "reset_refill":
            ix.add(rng.integers(-lim, lim + 1, size=(n, d)).astype(np.float32))  # stale rows that must never be seen again
            ix.reset()
        if c["build"] == "chunks":
            cuts = sorted(set([0, n] + [int(v) for v in rng.integers(0, n + 1, size=3)]))
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                if hi > lo:
                    ix.add(torch.from_numpy(x[lo:hi]).cuda() if rng.random() < 0.5 else x[lo:hi])
        else:
            ix.add(x)
        assert ix.ntotal == n
        if c["subset"]:
            labels = rng.integers(0, 6, size=n).astype(np.int32)
            subset = np.full((nq, 2), -1, dtype=np.int32)
            for r in range(nq):
                m = int(rng.integers(0, 3))
                subset[r, :m] = rng.choice(7, size=m, replace=False)  # label 6: nobody carries it
            ix.set_row_labels(labels)
        tq = torch.from_numpy(q).cuda()
        s, i = ix.search(tq, k, id_base=c["id_base"], subset=subset)
        s2, i2 = ix.search(tq, k, id_base=c["id_base"], subset=subset)  # determinism / stale workspace state
        stats = {key: ix.get_stat(key) for key in ("last_chunks", "last_safe_reruns", "last_recovered_queries", "last_exact_band_queries")}
    full = q.astype(np.float64) @ x.astype(np.float64).T
    if subset is not None:
        for r in range(nq):
            allowed = subset[r][subset[r] >= 0]
            if allowed.size:
                full[r, ~np.isin(labels, allowed)] = np.nan
    rs, ri = topk_desc_tiebreak(full, k)
    ri = np.where(ri >= 0, ri + c["id_base"], ri)
    gi, gs = i.cpu().numpy(), s.cpu().numpy()
    assert np.array_equal(gi, ri), f"ids differ: first bad row {np.argwhere((gi != ri).any(axis=1))[:3].ravel().tolist()}"
    assert np.array_equal(gs, rs), "scores differ"
    assert torch.equal(i, i2) and torch.equal(s, s2), "second search differs"
    return stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=600.0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    fails = 0
    done = 0
    recovered = 0
    n_exact = n_band = 0
    for t in range(a.trials):
        if time.time() - t0 > a.seconds:
            break
        c = draw(rng)
        try:
            st = run_trial(c)
            recovered += 1 if st["last_safe_reruns"] else 0
            n_exact += 1 if c["exact"] else 0
            n_band += 1 if st["last_exact_band_queries"] else 0
        except Exception as e:  # noqa: BLE001
            fails += 1
            print(f"FAIL trial {t}: {c}\n  {type(e).__name__}: {e}", flush=True)
            if not isinstance(e, AssertionError):
                traceback.print_exc()
        done += 1
    print(f"fuzz: {done} trials, {fails} failures, {recovered} trials needed a recovery pass, {n_exact} on exact-f32 stores "
          f"({n_band} of them through a band pass), {time.time() - t0:.0f} s, seed {a.seed}")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
