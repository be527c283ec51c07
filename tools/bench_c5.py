#!/usr/bin/env python3
"""Config 5 latency (hybrid merge + sampling + in-batch retrieval loss): B=64 queries, K=128 per engine x 3 engines,
32 sampled sections, H=768.  Prints microseconds per call (median over repeats, device-tensor APIs, inputs resident)."""
import json
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from vod_amd.core.merge import merge_hybrid_tensors
from vod_amd.core.sample import labeled_priority_sampling_tensors
from vod_amd.gradients import RetrievalGradients

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
B, K, H, NS = 64, 128, 768, 32


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(ts)


def ids():
    return torch.from_numpy(np.stack([rng.choice(1_000_000, size=K, replace=False) for _ in range(B)])).to(dev)


l_idx, d_idx, s_idx = ids(), ids(), ids()
s_idx[:, :40] = d_idx[:, :40]  # ~30 % dense/sparse overlap
l_lbl = torch.ones((B, K), dtype=torch.int64, device=dev)
d_scr = torch.randn((B, K), device=dev).sort(dim=1, descending=True).values
s_scr = torch.from_numpy(rng.gamma(2.0, 4.0, size=(B, K)).astype(np.float32)).to(dev)
out = {}
res = merge_hybrid_tensors(l_idx, l_lbl, {"dense": (d_idx, d_scr), "sparse": (s_idx, s_scr)}, {"dense": 1.0, "sparse": 1.0})
out["merge_hybrid_us"] = timeit(lambda: merge_hybrid_tensors(l_idx, l_lbl, {"dense": (d_idx, d_scr), "sparse": (s_idx, s_scr)}, {"dense": 1.0, "sparse": 1.0}))
m_idx, m_scr, m_lbl, _ = res
noise = torch.from_numpy(rng.exponential(size=tuple(m_scr.shape)).astype(np.float32)).to(dev)
out["merged_width"] = int(m_scr.shape[1])
out["priority_sample_us"] = timeit(lambda: labeled_priority_sampling_tensors(m_scr, m_lbl > 0, noise, 8, NS, True, 1.0, 100))
grad = RetrievalGradients()
for name, D, three_d in (("retrieval_loss_3d_64x32", NS, True), ("retrieval_loss_inbatch_64x2048", B * NS, False)):
    q = torch.randn((B, H), device=dev, requires_grad=True)
    s = torch.randn(((B, D, H) if three_d else (D, H)), device=dev, requires_grad=True)
    batch = {"section__score": torch.randn((B, D), device=dev), "section__relevance": (torch.rand((B, D), device=dev) < 0.05).long(),
             "section__sparse": torch.randn((B, D), device=dev), "section__dense": torch.randn((B, D), device=dev)}
    batch["section__relevance"][:, 0] = 1

    def fwd_bwd():
        o = grad(batch=batch, query_encoding=q, section_encoding=s)
        o.loss.backward()

    out[name + "_fwd_bwd_us"] = timeit(fwd_bwd, n=100)
    out[name + "_fwd_us"] = timeit(lambda: grad(batch=batch, query_encoding=q, section_encoding=s), n=100)
print(json.dumps(out))
