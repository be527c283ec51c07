#!/usr/bin/env python3
"""Host-side cost of the device-resident collate call (no synchronisation inside the loop): where the CPU time of a launch goes."""
import json
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import c5_data  # noqa: E402
from vod_amd import _native  # noqa: E402
from vod_amd.core.collate import collate_on_device  # noqa: E402

dev = torch.device("cuda", 0)
l_idx, l_lbl, engines, wts = c5_data.make(dev)
noise = torch.empty((c5_data.B, 3 * c5_data.K + 1), device=dev).exponential_()
kw = dict(total=32, max_pos_sections=8, temperature=1.0, max_support_size=100)


def host(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return {"host_us_per_call": (t1 - t0) / n * 1e6, "total_us_per_call": (t2 - t0) / n * 1e6}


out = {
    "collate_merge_sample": host(lambda: collate_on_device(l_idx, l_lbl, engines, wts, noise, **kw)),
    "collate_merge_sample_flatten": host(lambda: collate_on_device(l_idx, l_lbl, engines, wts, noise, in_batch_negatives=True, **kw)),
    "torch_empty": host(lambda: torch.empty((64, 385), dtype=torch.float32, device=dev)),
    "stream_ptr": host(lambda: _native.current_stream_ptr(dev)),
    "struct": host(lambda: _native.CollateArgs()),
    "unbind": host(lambda: noise.unbind(0)),
}
print(json.dumps(out))
