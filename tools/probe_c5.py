#!/usr/bin/env python3
"""Phase stamps of the collate-side kernels on the C5 shape (diagnostic build: VODHIP_LIB=vod_amd/csrc/libvodhip_ablation.so)."""
import ctypes
import json
import pathlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vod_amd import _native  # noqa: E402
from vod_amd.core.collate import flatten_on_device, sample_merged_on_device  # noqa: E402
from vod_amd.core.merge import merge_hybrid_device  # noqa: E402

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import c5_data  # noqa: E402

dev = torch.device("cuda", 0)
B, K, NS = c5_data.B, c5_data.K, c5_data.NS
l_idx, l_lbl, engines, wts = c5_data.make(dev)
noise = torch.empty((B, 3 * K + 1), device=dev).exponential_()
lib = _native.load_library()
out = {}
busy = "--busy" in sys.argv
big = torch.randn((8192, 8192), device=dev, dtype=torch.float16) if busy else None
for rep in range(5):
    if busy:  # keep the shader clock up, as a training step would
        for _ in range(20):
            big @ big
    m = merge_hybrid_device(l_idx, l_lbl, engines, wts)
    smp = sample_merged_on_device(m, noise, total=NS, max_pos_sections=8, temperature=1.0, max_support_size=100)
    fl = flatten_on_device(smp)
    torch.cuda.synchronize()
for which, name in ((0, "merge"), (1, "sample"), (2, "flatten")):
    buf = (ctypes.c_int64 * 256)()
    _native.check(lib.vodhip_debug_read_probe(which, buf, 256))
    v = np.array(buf[:128]).reshape(64, 2)
    n = int((v[:32, 1] > 0).sum())
    t = (v[:n, 1] - v[0, 1]) * 0.01  # us
    cyc = v[:n, 0] - v[0, 0]
    n = min(n, 32)
    t, cyc = t[:n], cyc[:n]
    out[name] = {"us_since_start": [round(float(x), 2) for x in t], "mhz": round(float(cyc[-1] / max(t[-1], 1e-9)), 1) if n > 1 else None}
    wg = np.array(buf[128:256]).reshape(64, 2)
    if wg[:, 0].min() > 0:
        t0 = wg[:, 0].min()
        out[name]["wg_begin_us"] = [round(float(x - t0) * 0.01, 2) for x in wg[:, 0]]
        out[name]["wg_end_us"] = [round(float(x - t0) * 0.01, 2) for x in wg[:, 1]]
print(json.dumps(out))
