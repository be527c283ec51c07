"""PCIe-inclusive rate of the headline search: host float32 queries in (pageable and pinned), host arrays out, against the
device-resident figure bench.py reports.      python3 tools/bench_pcie.py [--rows 10000000]
"""
import argparse
import json
import time

import torch

from vod_amd.index import HipFlatIndex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--k", type=int, default=100)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    ix = HipFlatIndex(a.dim, a.rows, dtype=torch.float16, device=0)
    for lo in range(0, a.rows, 250_000):
        ix.add(torch.randn((min(250_000, a.rows - lo), a.dim), generator=g, device=dev).half())
    q_host = torch.randn((a.nq, a.dim))
    q_pin = q_host.pin_memory()
    q_dev = q_host.to(dev)
    out = {}
    for name, q in (("device_resident", q_dev), ("host_pageable_in_host_out", q_host), ("host_pinned_in_host_out", q_pin)):
        def step():
            s, i = ix.search(q.to(dev, non_blocking=True) if q.device.type == "cpu" else q, a.k)
            if q.device.type == "cpu":
                s, i = s.cpu(), i.cpu()
            return s, i
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        out[name] = {"ms_per_batch": round(ms, 3), "queries_per_s": round(a.nq / ms * 1e3)}
    print(json.dumps({"rows": a.rows, "dim": a.dim, "nq": a.nq, "k": a.k, **out}))


if __name__ == "__main__":
    main()
