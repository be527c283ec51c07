#!/usr/bin/env python3
"""Diagnostic (VODHIP_ABLATION build): run the stamped filter kernel on the bench workload and print where a
K-slice spends its cycles.  Never used by the product or the tests."""
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
ix.set_param("tile", 17)
for _ in range(3):
    ix.search(q, k)
lib = _native.load_library()
N = 64 * 8 * 16 * 6
buf = (ctypes.c_ulonglong * N)()
lib.vodhip_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
assert lib.vodhip_debug_read_stamps(buf, N) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8, 16, 6).astype(np.int64)
ok = a[:, 0, 0, 0] > 0
a = a[ok]
print("sampled workgroups:", len(a))
sl = a[:, :, :11, :]                       # 11 instrumented slices (the last slice of 12 runs un-instrumented)
wait_vm = (sl[..., 1] - sl[..., 0]).mean(axis=(0, 1))
barrier = (sl[..., 2] - sl[..., 1]).mean(axis=(0, 1))
work = (sl[..., 3] - sl[..., 2]).mean(axis=(0, 1))
tot = (sl[:, :, 1:, 0] - sl[:, :, :-1, 0]).mean(axis=(0, 1))
print("per slice   wait_vmcnt:", wait_vm.round().tolist())
print("per slice   barrier   :", barrier.round().tolist())
print("per slice   ds+mfma   :", work.round().tolist())
print("slice period          :", tot.round().tolist())
ent, kend, end = a[:, :, 15, 0], a[:, :, 15, 1], a[:, :, 15, 2]
first = a[:, :, 0, 0]
print("prologue (entry -> first slice wait):", (first - ent).mean().round(), " k-loop:", (kend - first).mean().round(),
      " epilogue:", (end - kend).mean().round(), " total:", (end - ent).mean().round())
for w in range(8):
    print("wave", w, "slice work mean", (sl[:, w, :, 3] - sl[:, w, :, 2]).mean().round(), "wait_vm", (sl[:, w, :, 1] - sl[:, w, :, 0]).mean().round(),
          "barrier", (sl[:, w, :, 2] - sl[:, w, :, 1]).mean().round())
