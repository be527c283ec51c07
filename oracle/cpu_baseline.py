"""Oracle-side CPU baseline ("port"): the faiss-CPU IndexFlatIP path RESTATED, timed on the host cores.

Test/bench infrastructure only (bench.py's `cpu_baseline` leg).  It restates what the reference runs per
batch on the CPU -- `faiss_index.search(query_vec, k)` on a Flat / inner-product index holding float32
rows (/root/reference/src/vod_search/faiss_search/server.py:84, build.py:60-73): for batches of >= 20
queries upstream faiss computes blocked `Q.X^T` with BLAS sgemm and keeps a per-query heap of the k best.
Here: float32 corpus in RAM, row blocks -> `torch.mm` (MKL sgemm, all host threads) -> `torch.topk` per
block -> merge with the running top-k.  faiss itself is not installable in this image: every number
produced by this file is labelled "faiss-CPU restated", never "faiss".
"""
from __future__ import annotations

import time

import torch


def flat_ip_topk_cpu(q: torch.Tensor, x: torch.Tensor, k: int, block: int = 16384) -> tuple[torch.Tensor, torch.Tensor]:
    """q [nq,d] f32, x [n,d] f32 (CPU).  Returns (scores, ids) sorted descending.  Tie order = torch.topk's."""
    nq = q.shape[0]
    best_s = torch.full((nq, 0), float("-inf"))
    best_i = torch.full((nq, 0), -1, dtype=torch.int64)
    for lo in range(0, x.shape[0], block):
        xb = x[lo : lo + block]
        s = q @ xb.T
        kk = min(k, s.shape[1])
        ts, ti = torch.topk(s, kk, dim=1)
        cs = torch.cat([best_s, ts], dim=1)
        ci = torch.cat([best_i, ti + lo], dim=1)
        kk = min(k, cs.shape[1])
        ms, mo = torch.topk(cs, kk, dim=1)
        best_s, best_i = ms, torch.gather(ci, 1, mo)
    return best_s, best_i


def time_cpu_baseline(dim: int, nq: int, k: int, n_full: int, target_seconds: float = 15.0, max_rows: int = 2_000_000,
                      seed: int = 1234) -> dict:
    """Time the port on a bounded sample of the workload; extrapolate linearly in N to `n_full` rows."""
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(nq, dim, generator=g)
    probe_rows = 65536
    xp = torch.randn(probe_rows, dim, generator=g)
    flat_ip_topk_cpu(q, xp[:16384], k)  # warm-up (thread pool, MKL)
    t0 = time.perf_counter()
    flat_ip_topk_cpu(q, xp, k)
    t_probe = time.perf_counter() - t0
    rows = int(min(max_rows, max(probe_rows, probe_rows * target_seconds / max(t_probe, 1e-6))))
    rows = min(rows, n_full)
    if rows > probe_rows:
        x = torch.randn(rows, dim, generator=g)
        t0 = time.perf_counter()
        flat_ip_topk_cpu(q, x, k)
        t = time.perf_counter() - t0
    else:
        rows, t = probe_rows, t_probe
    t_full = t * (n_full / rows)
    return {
        "value": nq / t_full,
        "unit": "queries/s",
        "cores": threads,
        "kind": "port",
        "sample": f"faiss-CPU restated (MKL sgemm + top-k merge, fp32): {nq} queries x {rows} of {n_full} rows x {dim}, "
                  f"top-{k}, {t:.2f} s measured, scaled linearly in N",
    }
