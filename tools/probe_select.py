#!/usr/bin/env python3
"""Phase stamps of `mips_select_kernel` (diagnostic build: VODHIP_LIB=vod_amd/csrc/libvodhip_ablation.so).

    python tools/probe_select.py [rows] [nq]      default: the C2 shape, 1,000,000 x 768 fp16, 256 queries, k = 100
Prints, for the threshold-only launch (after the bootstrap), a middle launch and the final one, the microseconds of workgroup 0
from kernel entry to: candidates in LDS | radix select done | compaction done | final sort done | outputs written."""
import ctypes
import json
import pathlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vod_amd import _native  # noqa: E402
from vod_amd.index import HipFlatIndex  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
ix = HipFlatIndex(768, rows, device=0)
for lo in range(0, rows, 1 << 20):
    ix.add(torch.randn((min(1 << 20, rows - lo), 768), device=dev, generator=g, dtype=torch.float16))
q = torch.randn((nq, 768), device=dev, generator=g, dtype=torch.float16)
lib = _native.load_library()
for _ in range(5):
    ix.search(q, 100)
torch.cuda.synchronize()
buf = (ctypes.c_int64 * 256)()
_native.check(lib.vodhip_debug_read_probe(3, buf, 256))
v = np.array(buf[:128]).reshape(64, 2)
out = {}
for name, pb in (("threshold_only", 8), ("middle", 16), ("final", 0)):
    t = (v[pb : pb + 6, 1] - v[pb, 1]) * 0.01
    cyc = v[pb : pb + 6, 0] - v[pb, 0]
    out[name] = {"candidates_of_query_0": int(v[pb + 6, 0]), "us_since_entry": [round(float(x), 2) for x in t],
                 "phases": ["entry", "keys in LDS", "radix select", "compaction", "final sort", "outputs"],
                 "mhz": round(float(cyc[5] / max(t[5], 1e-9)), 1)}
    r = v[24 + 10 * (pb >> 3) : 24 + 10 * (pb >> 3) + 9]
    r = r[r[:, 1] >= v[pb, 1]]  # stamps of THIS launch (older launches leave stale words behind the last pass)
    out[name]["radix_us_since_entry"] = [round(float(x - v[pb, 1]) * 0.01, 2) for x in r[:, 1]]
print(json.dumps(out))
