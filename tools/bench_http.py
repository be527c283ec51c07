#!/usr/bin/env python3
"""End-to-end latency THROUGH the drop-in boundary: HipMipsMaster spawns the server process (owner of the GPU index),
the client sends host float32 queries over HTTP - base64-in-JSON `/fast-search` (the reference's wire format,
src/vod_search/faiss_search/server.py:76-91) and the raw-bytes `/raw-search` - and gets host arrays back.
Prints median milliseconds per request and the resulting queries/s, next to the in-process device-resident time.
usage: python tools/bench_http.py [rows] [dim] [single|uds|group|gloo2|node2]"""
import json
import os
import statistics
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
from vod_amd import factory  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
rng = np.random.default_rng(0)
tmp = tempfile.mkdtemp()
os.chdir(tmp)
x = rng.standard_normal((rows, dim), dtype=np.float32).astype(np.float16)
mode = sys.argv[3] if len(sys.argv) > 3 else "single"
# group = worker group on RCCL, devices=[0]; gloo2 / node2 = TWO shards on the box's one GPU behind a gloo worker group / behind the
# one-process node index (what the request broadcast + gather of a process group costs next to peer copies inside one process)
devices, backend = {"single": (None, "nccl"), "uds": (None, "nccl"), "group": ([0], "nccl"), "gloo2": ([0, 0], "gloo"), "node2": ([0, 0], "node")}[mode]
master = factory.build_hip_mips_index(x, config={"port": -1, "logging_level": "warning", "group_backend": backend, "uds": mode == "uds"}, cache_dir=tmp,
                                      devices=devices)
out = {"rows": rows, "dim": dim, "server": {"single": "single process", "uds": "single process, clients on its Unix-domain socket", "group": "worker group, devices=[0]", "gloo2": "worker group on gloo, devices=[0, 0]",
                                             "node2": "one process, node index, devices=[0, 0]"}[mode], "requests": []}
with master:
    json_client = master.get_client()
    raw_client = type(json_client)(host=json_client.host, port=json_client.port, binary=True, uds=json_client.uds)
    for nq, k in [(32, 10), (64, 100), (256, 100), (1024, 100)]:
        q = rng.standard_normal((nq, dim), dtype=np.float32)
        rec = {"nq": nq, "k": k}
        for name, cl in [("fast_search_json_b64", json_client), ("raw_search_bytes", raw_client)]:
            for _ in range(3):
                cl.search(vector=q, top_k=k)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter()
                cl.search(vector=q, top_k=k)
                ts.append(time.perf_counter() - t0)
            ms = statistics.median(ts) * 1e3
            rec[name + "_ms"] = round(ms, 3)
            rec[name + "_qps"] = round(nq / ms * 1e3)
        out["requests"].append(rec)
print(json.dumps(out))
