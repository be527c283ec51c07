"""Node index: synchronous searches vs one batch ahead (search_async / finish), N shards on the box's one GPU (host-staged exchange).
The device work is the same; what the pipelined form hides is the host's part of every batch: N x 7 launches, N finishes, copies, merge."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
from vod_amd.index import HipNodeIndex  # noqa: E402

n_shards = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rows, d, nq, k, steps = 1_000_000, 768, int(sys.argv[2]) if len(sys.argv) > 2 else 256, 100, 60
g = torch.Generator(device="cuda").manual_seed(3)
nx = HipNodeIndex(d, rows, [0] * n_shards)
nx.set_param("host_staging", 1)
for c in range(4):
    nx.add(torch.randn((rows // 4, d), generator=g, device="cuda").half().cpu().numpy())
q = torch.randn((nq, d), generator=g, device="cuda").half()
for _ in range(5):
    ref = nx.search(q, k)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = nx.search(q, k)
    torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nx.search_async(q, k)
    for _ in range(steps - 1):
        nx.search_async(q, k)
        r = nx.finish()
    r = nx.finish()
    torch.cuda.synchronize()
    t_pipe = (time.perf_counter() - t0) / steps * 1e3
    assert torch.equal(r[1], ref[1]) and torch.equal(r[0], ref[0])
    print(f"{n_shards} shards x {rows // n_shards} rows, nq {nq}: synchronous {t_sync:.4f} ms/batch, one ahead {t_pipe:.4f} ms/batch ({(t_pipe / t_sync - 1) * 100:+.1f} %)")
