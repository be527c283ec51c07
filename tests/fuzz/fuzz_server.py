"""Randomised campaign THROUGH the drop-in boundary: one spawned server (optionally with micro-batching, optionally the
multi-worker group), several client threads firing random requests - `/search` (JSON lists), `/fast-search` (the reference's
base64-npy codec), `/raw-search`, with and without subset ids, random batch sizes and k - every answer compared bit for bit
with the fp64 oracle.      python3 tests/fuzz/fuzz_server.py [--requests 400] [--threads 8] [--wait-ms 0] [--group | --node] [--http native]
(round 4: the server's defaults = libvodhip's native HTTP front + batch-while-busy request fusion; `--wait-ms` adds the optional window)
"""
import argparse
import concurrent.futures
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))  # the repo root
from oracle.flat_ip import topk_desc_tiebreak  # noqa: E402
from vod_amd import store  # noqa: E402
from vod_amd.search.client import HipMipsClient, HipMipsMaster  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=400)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--wait-ms", type=float, default=0.0)
    ap.add_argument("--group", action="store_true", help="serve through the worker group (devices=[0, 0], gloo)")
    ap.add_argument("--node", action="store_true", help="serve through the one-process node index (devices=[0, 0], group_backend=node)")
    ap.add_argument("--http", default="native", choices=["native", "uvicorn"])
    ap.add_argument("--churn", action="store_true", help="a NEW client object (new connections) every 25 requests per route: connection churn for soak runs")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--exact-f32", action="store_true",
                    help="serve an exact-f32 store of rows the fp16 scan cannot represent (13-bit integers; queries 3-bit: float32 sums stay exact): "
                         "every answer must still equal the float64 oracle on the unrounded values bit for bit")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    n, d = 60_000, 96
    x = rng.integers(-6, 7, size=(n, d)).astype(np.float32)
    q_lim = 6
    if a.exact_f32:
        x = np.clip(x * rng.integers(1, 513, size=(n, 1)).astype(np.float32) + rng.integers(-3, 4, size=(n, d)).astype(np.float32), -4096, 4096)
        q_lim = 4
    names = np.array([f"doc{v}" for v in rng.integers(0, 7, size=n)])
    tmp = tempfile.mkdtemp()
    store.save_vectors(f"{tmp}/v.npy", x, dtype=np.float32 if a.exact_f32 else np.float16)
    np.save(f"{tmp}/subsets.npy", names)
    x64 = x.astype(np.float64)

    class Master(HipMipsMaster):
        def _make_cmd(self):
            return super()._make_cmd() + ["--subset-ids-path", f"{tmp}/subsets.npy"]

    kw = dict(devices=[0, 0], group_backend="gloo") if a.group else (dict(devices=[0, 0], group_backend="node") if a.node else {})
    kw.update(http=a.http, micro_batch_wait_ms=a.wait_ms, exact_f32=a.exact_f32)
    jobs = []
    for j in range(a.requests):
        nq = int(rng.choice([1, 1, 2, 5, 17, 64, 130, 300]))
        k = int(rng.choice([1, 3, 10, 100, 257]))
        route = str(rng.choice(["json", "fast", "raw"]))
        subset = None
        if route == "fast" and rng.random() < 0.4:
            subset = [[f"doc{int(v)}" for v in rng.choice(8, size=int(rng.integers(0, 3)), replace=False)] for _ in range(nq)]
        q = rng.integers(-q_lim, q_lim + 1, size=(nq, d)).astype(np.float32)
        jobs.append((j, route, q, k, subset))

    def expected(q, k, subset):
        full = q.astype(np.float64) @ x64.T
        if subset is not None:
            for r, allowed in enumerate(subset):
                if allowed:
                    full[r, ~np.isin(names, allowed)] = np.nan
        return topk_desc_tiebreak(full, k)

    t0 = time.time()
    with Master(f"{tmp}/v.npy", port=-1, logging_level="warning", **kw) as m:
        clients = {
            "fast": HipMipsClient(host=m.host, port=m.port, forward_subset_ids=True),
            "raw": HipMipsClient(host=m.host, port=m.port, binary=True),
            "json": HipMipsClient(host=m.host, port=m.port),
        }

        def server_stats():
            try:
                import requests

                return requests.get(f"{m.host}:{m.port}/stats", timeout=10).json()
            except Exception:  # noqa: BLE001
                return {}

        stats0 = server_stats()

        def run(job):
            j, route, q, k, subset = job
            if a.churn and j % 1000 == 999:
                st = server_stats()
                print(f"  after ~{j + 1} requests: rss {st.get('rss_kb')} kB, connections {st.get('connections')}, open {st.get('open_connections')}", flush=True)
            if a.churn and j % 25 == 24:  # drop this route's client: its threads' connections close, new ones open
                clients[route] = HipMipsClient(host=m.host, port=m.port, forward_subset_ids=(route == "fast"), binary=(route == "raw"))
            try:
                if route == "json":
                    res = clients["json"].search_py(q, top_k=k)
                else:
                    res = clients[route].search(vector=q, subset_ids=subset, top_k=k)
                rs, ri = expected(q, k, subset)
                if not (np.array_equal(np.asarray(res.indices), ri) and np.array_equal(np.asarray(res.scores, dtype=np.float32), rs)):
                    return f"request {j} ({route}, nq={len(q)}, k={k}, subset={'yes' if subset else 'no'}): result differs from the oracle"
            except Exception as e:  # noqa: BLE001
                return f"request {j} ({route}, nq={len(q)}, k={k}): {type(e).__name__}: {str(e)[:300]}"
            return None

        with concurrent.futures.ThreadPoolExecutor(a.threads) as ex:
            errs = [e for e in ex.map(run, jobs) if e]
        stats1 = server_stats()
        if stats0 and stats1:
            print(f"server: rss {stats0.get('rss_kb')} -> {stats1.get('rss_kb')} kB, connections {stats1.get('connections')} (open now {stats1.get('open_connections')}), "
                  f"native {stats1.get('requests_native')} / fallback {stats1.get('requests_fallback')} requests, batches {stats1.get('batches')}")
    for e in errs[:20]:
        print("FAIL", e)
    print(f"fuzz_server: {len(jobs)} requests on {a.threads} threads ({'worker group x2' if a.group else ('node index x2' if a.node else 'single process')}, "
          f"http {a.http}, extra wait window {a.wait_ms} ms, seed {a.seed}), {len(errs)} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if errs else 0)


if __name__ == "__main__":
    main()
