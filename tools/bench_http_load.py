#!/usr/bin/env python3
"""Throughput THROUGH the drop-in boundary under the load shape the trainer produces: many DataLoader workers, each sending
its own small batch (/root/reference/src/vod_dataloaders/realm_dataloader.py:92-118, core/search.py:128-146) to ONE server.

P concurrent client PROCESSES (pickled `HipMipsClient`s, like DataLoader workers) x batch size nq, on the reference's wire format
(`/fast-search`, base64-in-JSON) and on the binary route (`/raw-search`), with the server's micro-batching off and on.
Per cell: aggregate queries/s, request latency p50 / p99, and the ratio to the device-resident rate of ONE fused batch of
P * nq queries (what a perfect boundary in front of the same kernels would deliver).

Round 4: the server's DEFAULT settings are what is measured (`--http native`: libvodhip's front + batch-while-busy request fusion, no
wait window); `--http uvicorn` / `--micro-batch-ms` reproduce round 3's shells for an A/B.  Per cell the server's own counters
(GET /stats) give the mean fused batch and the share of the cell the engine sat idle.

    python tools/bench_http_load.py [--rows 10000000] [--dim 768] [--k 100] [--seconds 2.5] [--out profiles/r04_http_load.json]
"""
import argparse
import json
import multiprocessing as mp
import os
import pathlib
import statistics
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _worker(client, nq, dim, k, seconds, barrier, out_q, seed, think_ms=0.0):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((nq, dim), dtype=np.float32)
    for _ in range(3):
        client.search(vector=q, top_k=k)
    barrier.wait()
    lat = []
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        res = client.search(vector=q, top_k=k)
        lat.append(time.perf_counter() - t0)
        if think_ms > 0:  # an open-ish loop: the worker "tokenises its next batch" for an Exp(think_ms) time before it searches again
            time.sleep(float(rng.exponential(think_ms)) / 1e3)
    assert res.indices.shape == (nq, k)
    out_q.put(lat)


def run_cell(client, P, nq, dim, k, seconds, stats=lambda: {}, think_ms=0.0):
    ctx = mp.get_context("spawn")
    barrier, out_q = ctx.Barrier(P + 1), ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(client, nq, dim, k, seconds, barrier, out_q, 100 + i, think_ms)) for i in range(P)]
    for p in procs:
        p.start()
    barrier.wait()
    s0 = stats()  # the server's counters over the timed window only (not the seconds the client processes take to start)
    t0 = time.perf_counter()
    lats = [out_q.get(timeout=seconds + 120) for _ in range(P)]
    wall = time.perf_counter() - t0
    s1 = stats()
    for p in procs:
        p.join(timeout=60)
    flat = sorted(x for lat in lats for x in lat)
    n_req = len(flat)
    cell = {"clients": P, "nq": nq, "requests": n_req, "qps": n_req * nq / wall, "p50_ms": flat[n_req // 2] * 1e3,
            "p99_ms": flat[min(n_req - 1, int(n_req * 0.99))] * 1e3}
    if s0 and s1 and s1.get("batches", 0) > s0.get("batches", 0):
        nb = s1["batches"] - s0["batches"]
        busy, idle = s1["busy_ns"] - s0["busy_ns"], s1["idle_ns"] - s0["idle_ns"]
        cell["server"] = {"batches": nb, "mean_queries_per_batch": (s1["queries"] - s0["queries"]) / nb,
                          "mean_requests_per_batch": (s1["requests"] - s0["requests"]) / nb, "engine_idle_fraction": idle / max(1, busy + idle),
                          "grace_waits": s1["grace_waits"] - s0["grace_waits"], "grace_expired": s1["grace_expired"] - s0["grace_expired"],
                          "scan_ms_by_tiles": {t: round(s1.get(f"tiles_ns_{t}", 0) / 1e6, 3) for t in (1, 2, 4, 8)}}
    return cell


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--clients", type=int, nargs="+", default=[1, 8, 32])
    ap.add_argument("--nq", type=int, nargs="+", default=[32, 64, 256])
    ap.add_argument("--micro-batch-ms", type=float, nargs="+", default=[0.0])
    ap.add_argument("--http", default="native", choices=["native", "uvicorn"])
    ap.add_argument("--routes", nargs="+", default=["fast", "raw"], choices=["fast", "raw"])
    ap.add_argument("--batcher-param", action="append", default=[], metavar="KEY=VALUE")
    ap.add_argument("--devices", type=int, nargs="+", default=None, help="serve the store row-sharded over these GPUs (a device may repeat)")
    ap.add_argument("--group-backend", default="node", choices=["node", "nccl", "gloo"])
    ap.add_argument("--exact-f32", action="store_true", help="serve an exact-f32 store (float32 rows kept, float32 brute-force results): what a float32 corpus gets by default")
    ap.add_argument("--think-ms", type=float, nargs="+", default=[0.0],
                    help="mean of an exponential pause between a worker's requests (0 = closed loop, the default cells); > 0 shows what the "
                         "fusion policy costs / gives when arrivals are not synchronised by the server itself")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch

    from vod_amd.index import HipFlatIndex
    from vod_amd.search.client import HipMipsClient, HipMipsMaster
    from vod_amd.search.server import synthetic_rows

    os.chdir(tempfile.mkdtemp())
    spec = f"synthetic:{a.rows}x{a.dim}:7"
    from vod_amd.hostcpu import usable_cpus

    out = {"store": f"{a.rows} x {a.dim} fp16 (synthetic N(0,1), generated on the device)", "k": a.k, "http": a.http, "seconds_per_cell": a.seconds,
           "host_cpus_visible": len(os.sched_getaffinity(0)), "host_cpus_usable": usable_cpus(), "batcher_params": a.batcher_param,
           "note": "client processes and the server share the usable host CPUs", "device_resident": {}, "cells": [],
           "server_shape": "single process, one device" if a.devices is None else f"devices={a.devices}, group_backend={a.group_backend}"}
    # device-resident reference: the same store in this process, one fused batch of B queries resident in HBM
    dev = torch.device("cuda", 0)
    ix = HipFlatIndex(a.dim, a.rows, dtype=torch.float16, device=0)
    for rows in synthetic_rows(torch, dev, 0, a.rows, a.dim, 7):
        ix.add(rows.half())
    for B in sorted({min(2048, P * nq) for P in a.clients for nq in a.nq} | set(a.nq)):
        q = torch.randn((B, a.dim), device=dev).half()
        for _ in range(3):
            ix.search(q, a.k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            ix.search_async(q, a.k)
            if ix._keep and len(ix._keep) > 1:
                ix.finish()
        while ix._keep:
            ix.finish()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        out["device_resident"][str(B)] = {"ms_per_batch": ms, "qps": B / ms * 1e3}
    ix.close()
    del ix
    torch.cuda.empty_cache()
    for mb in a.micro_batch_ms:
        bparams = {kv.partition("=")[0]: int(kv.partition("=")[2]) for kv in a.batcher_param}
        multi = {} if a.devices is None else dict(devices=a.devices, group_backend=a.group_backend)
        with HipMipsMaster(spec, port=-1, logging_level="warning", micro_batch_wait_ms=mb, http=a.http, batcher_params=bparams, exact_f32=a.exact_f32, **multi) as master:
            import requests

            def stats():
                try:
                    return requests.get(f"{master.host}:{master.port}/stats", timeout=10).json()
                except Exception:  # noqa: BLE001
                    return {}

            for binary in [r == "raw" for r in a.routes]:
                client = HipMipsClient(host=master.host, port=master.port, binary=binary)
                for P, nq, think in [(P, nq, th) for th in a.think_ms for P in a.clients for nq in a.nq]:
                    if True:
                        cell = run_cell(client, P, nq, a.dim, a.k, a.seconds, stats, think)
                        cell["think_ms"] = think
                        fused = str(min(2048, P * nq))
                        cell.update(route="/raw-search" if binary else "/fast-search", micro_batch_wait_ms=mb,
                                    device_resident_qps_at_fused_batch=out["device_resident"][fused]["qps"],
                                    device_resident_qps_at_request_batch=out["device_resident"][str(nq)]["qps"])
                        cell["fraction_of_device_rate_at_fused_batch"] = cell["qps"] / cell["device_resident_qps_at_fused_batch"]
                        out["cells"].append(cell)
                        print(json.dumps(cell), flush=True)
    text = json.dumps(out, indent=1)
    if a.out:
        pathlib.Path(ROOT / a.out).write_text(text)
    print(json.dumps({"done": True, "cells": len(out["cells"])}))


if __name__ == "__main__":
    main()
