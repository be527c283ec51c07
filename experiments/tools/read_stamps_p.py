#!/usr/bin/env python3
"""Diagnostic (VODHIP_ABLATION build, tile 29): per-tile K-loop vs epilogue cycles of the persistent kernel."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vod_amd.index import HipFlatIndex
from vod_amd import _native
n, d, nq, k = 10_000_000, 768, 1024, 100
ix = HipFlatIndex(d, n)
for c in range(n // 250_000):
    g = torch.Generator(device="cuda").manual_seed(1234 + c)
    ix.add(torch.randn((250_000, d), generator=g, device="cuda").half())
q = torch.randn((nq, d), device="cuda").half()
ix.set_param("tile", 29)
if len(sys.argv) > 1:
    ix.set_param("krot", int(sys.argv[1]))
for _ in range(2):
    ix.search(q, k)
lib = _native.load_library()
N = 64 * 8 * 16 * 6
buf = (ctypes.c_ulonglong * N)()
lib.vodhip_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
assert lib.vodhip_debug_read_stamps(buf, N) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8, 16 * 6).astype(np.int64)[:, :, :7]
ok = a[:, 0, 3] > 0
a = a[ok]
print("workgroups:", len(a), "tiles per workgroup:", a[0, 0, 3])
per_tile_k = a[:, :, 0] / a[:, :, 3]
per_tile_e = a[:, :, 1] / a[:, :, 3]
per_tile_t = a[:, :, 2] / a[:, :, 3]
print("per tile: K loop %.0f  epilogue %.0f  total %.0f cycles (mean over waves)" % (per_tile_k.mean(), per_tile_e.mean(), per_tile_t.mean()))
print("per tile: own filter %.0f  barrier wait %.0f  flush %.0f  hit blocks per tile per wave %.3f" % (
    (a[:, :, 4] / a[:, :, 3]).mean(), (a[:, :, 5] / a[:, :, 3]).mean(), ((a[:, :, 1] - a[:, :, 4] - a[:, :, 5]) / a[:, :, 3]).mean(), (a[:, :, 6] / a[:, :, 3]).mean()))
print("per wave K:", per_tile_k.mean(axis=0).round().tolist())
print("per wave E:", per_tile_e.mean(axis=0).round().tolist())
