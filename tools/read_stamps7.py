#!/usr/bin/env python3
"""Diagnostic: stamps of the specialised kernel (tile 18).  slots 0..6 = MFMA waves, slot 7 = loader wave 0."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from vod_amd.index import HipFlatIndex
from vod_amd import _native
n, d, nq, k = 4_000_000, 768, 1024, 100
ix = HipFlatIndex(d, n)
for c in range(n // 250_000):
    g = torch.Generator(device="cuda").manual_seed(1234 + c)
    ix.add(torch.randn((250_000, d), generator=g, device="cuda").half())
q = torch.randn((nq, d), device="cuda").half()
ix.set_param("tile", int(sys.argv[1]) if len(sys.argv) > 1 else 18)
for _ in range(3):
    ix.search(q, k)
lib = _native.load_library()
N = 64 * 8 * 16 * 6
buf = (ctypes.c_ulonglong * N)()
lib.vodhip_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
assert lib.vodhip_debug_read_stamps(buf, N) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8, 16, 6).astype(np.int64)
a = a[a[:, 0, 0, 0] > 0]
print("sampled workgroups:", len(a))
T = 15
c = a[:, :7, :T, :]
print("consumer: barrier wait  :", (c[..., 2] - c[..., 0]).mean(axis=(0, 1)).round().tolist())
print("consumer: ds+mfma       :", (c[..., 3] - c[..., 2]).mean(axis=(0, 1)).round().tolist())
print("consumer: slice period  :", (c[:, :, 1:, 0] - c[:, :, :-1, 0]).mean(axis=(0, 1)).round().tolist())
l = a[:, 7, :T, :]
print("loader: vmcnt wait      :", (l[..., 1] - l[..., 0]).mean(axis=0).round().tolist())
print("loader: barrier wait    :", (l[..., 2] - l[..., 1]).mean(axis=0).round().tolist())
print("loader: issue 8 glds    :", (l[..., 3] - l[..., 2]).mean(axis=0).round().tolist())
for w in range(7):
    print("wave", w, "work", (c[:, w, :, 3] - c[:, w, :, 2]).mean().round(), "barrier", (c[:, w, :, 2] - c[:, w, :, 0]).mean().round())
