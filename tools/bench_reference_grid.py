#!/usr/bin/env python3
"""The reference's ONE published performance artefact, re-run on its own axes (BASELINE.md section 1:
/root/reference/assets/faiss_server_profile.png - batch 10, top-100, N in {1e2, 1e3, 1e4, 1e5}, D in {128, 512, 1024}):

  API_base     POST /search       JSON lists in and out            (faiss_search/server.py:68-73, client.py:47-62)
  API_fast     POST /fast-search  base64 .npy payloads             (server.py:76-91, client.py:64-100)
  main_thread  in-process search of the index                      (examples/search/faiss.py:48-58)

here through `HipMipsMaster` (the server in its own process, default settings) and `HipFlatIndex.search` (host float32 in, host
arrays out, like `faiss_index.search`).  Milliseconds per batch, median of `--repeats` after warm-up.  The chart's hardware is not
stated: its values (read off the plot, +-1 ms) are context beside ours, not a same-node comparison.

    python tools/bench_reference_grid.py [--out profiles/r04_reference_grid.json]
"""
import argparse
import json
import pathlib
import statistics
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

# read off assets/faiss_server_profile.png; None = not legible on the chart
REFERENCE_MS = {
    (128, 100): (24, 6, 0.0), (128, 1000): (20, 6, 0.0), (128, 10000): (21, 6.5, 0.3), (128, 100000): (28, 13.5, 3.7),
    (512, 100000): (47, 27.5, 18), (1024, 100000): (75.5, 53, 53),
}


def median_ms(fn, repeats: int, warm: int = 5) -> float:
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=10)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--rows", type=int, nargs="+", default=[100, 1000, 10_000, 100_000])
    ap.add_argument("--dims", type=int, nargs="+", default=[128, 512, 1024])
    ap.add_argument("--repeats", type=int, default=50)
    ap.add_argument("--http", default="native", choices=["native", "uvicorn"])
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch

    from vod_amd import store
    from vod_amd.index import HipFlatIndex
    from vod_amd.search.client import HipMipsMaster

    tmp = pathlib.Path(tempfile.mkdtemp())
    rng = np.random.default_rng(0)
    out = {"batch": a.batch, "top_k": a.k, "http": a.http, "repeats": a.repeats,
           "series": {"API_base": "POST /search (JSON lists)", "API_fast": "POST /fast-search (base64 .npy)", "main_thread": "in-process HipFlatIndex.search, host arrays in and out"},
           "reference_source": "assets/faiss_server_profile.png, values read off the chart (+-1 ms), hardware not stated", "cells": []}
    for d in a.dims:
        for n in a.rows:
            x = rng.standard_normal((n, d), dtype=np.float32)
            q = rng.standard_normal((a.batch, d), dtype=np.float32)
            k = a.k
            # main_thread: the index in THIS process (the call the reference times around faiss_index.search)
            ix = HipFlatIndex(d, n, dtype=torch.float16, device=0)
            ix.add(x)

            def in_process():
                s, i = ix.search(q, k)
                return s.cpu().numpy(), i.cpu().numpy()

            t_main = median_ms(in_process, a.repeats)
            ref_s, ref_i = in_process()
            ix.close()
            path = tmp / f"v_{n}_{d}.npy"
            store.save_vectors(path, x, dtype=np.float16)
            with HipMipsMaster(path, port=-1, logging_level="warning", http=a.http) as master:
                client = master.get_client()
                fast = client.search(vector=q, top_k=k)
                assert np.array_equal(fast.indices, ref_i) and np.array_equal(fast.scores, ref_s), "served result differs from the in-process one"
                t_fast = median_ms(lambda: client.search(vector=q, top_k=k), a.repeats)
                base = client.search_py(q, top_k=k)
                assert np.array_equal(base.indices, ref_i)
                t_base = median_ms(lambda: client.search_py(q, top_k=k), max(10, a.repeats // 2))
            ref = REFERENCE_MS.get((d, n))
            cell = {"dim": d, "rows": n, "API_base_ms": round(t_base, 3), "API_fast_ms": round(t_fast, 3), "main_thread_ms": round(t_main, 3),
                    "reference_chart_ms": None if ref is None else {"API_base": ref[0], "API_fast": ref[1], "main_thread": ref[2]}}
            out["cells"].append(cell)
            print(json.dumps(cell), flush=True)
    floor = [c["API_fast_ms"] for c in out["cells"] if c["rows"] <= 1000]
    out["fast_search_floor_ms"] = {"ours_min": min(floor), "ours_max": max(floor), "reference_chart": "6 - 7.5"}
    text = json.dumps(out, indent=1)
    if a.out:
        (ROOT / a.out).write_text(text)
    print(json.dumps({"done": True, "fast_search_floor_ms": out["fast_search_floor_ms"]}))


if __name__ == "__main__":
    main()
