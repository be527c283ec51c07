#!/usr/bin/env python3
"""Config 5 latency (hybrid merge + sampling + in-batch retrieval loss): B=64 queries, K=128 per engine x 3 engines,
32 sampled sections, H=768.  Prints microseconds per call (median over repeats, device-tensor APIs, inputs resident)."""
import json
import statistics
import sys
import time

import numpy as np
import torch

import pathlib  # noqa: E402

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vod_amd.core.merge import merge_hybrid_tensors
from vod_amd.core.sample import labeled_priority_sampling_tensors
from vod_amd.gradients import RetrievalGradients

import c5_data  # noqa: E402  (tools/ is on sys.path: this file lives there)

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
B, K, H, NS = c5_data.B, c5_data.K, c5_data.H, c5_data.NS


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


l_idx, l_lbl, engines, wts = c5_data.make(dev)
(d_idx, d_scr), (s_idx, s_scr) = engines["dense"], engines["sparse"]
out = {}
res = merge_hybrid_tensors(l_idx, l_lbl, {"dense": (d_idx, d_scr), "sparse": (s_idx, s_scr)}, {"dense": 1.0, "sparse": 1.0})
out["merge_hybrid_us"] = timeit(lambda: merge_hybrid_tensors(l_idx, l_lbl, {"dense": (d_idx, d_scr), "sparse": (s_idx, s_scr)}, {"dense": 1.0, "sparse": 1.0}))
m_idx, m_scr, m_lbl, _ = res
noise = torch.from_numpy(rng.exponential(size=tuple(m_scr.shape)).astype(np.float32)).to(dev)
out["merged_width"] = int(m_scr.shape[1])
out["priority_sample_us"] = timeit(lambda: labeled_priority_sampling_tensors(m_scr, m_lbl > 0, noise, 8, NS, True, 1.0, 100))
# ---- round 3: the device-resident chain (merge -> sample (+ gathers, rank diagnostic) -> in-batch flattening): launches only ----
from vod_amd.core.collate import collate_on_device, flatten_on_device, sample_merged_on_device  # noqa: E402
from vod_amd.core.merge import merge_hybrid_device  # noqa: E402

noise_full = torch.empty((B, 3 * K + 1), device=dev).exponential_()
merged = merge_hybrid_device(l_idx, l_lbl, engines, wts)
sampled = sample_merged_on_device(merged, noise_full, total=NS, max_pos_sections=8, temperature=1.0, max_support_size=100)
out["device_merge_us"] = timeit(lambda: merge_hybrid_device(l_idx, l_lbl, engines, wts))
out["device_sample_gather_us"] = timeit(lambda: sample_merged_on_device(merged, noise_full, total=NS, max_pos_sections=8, temperature=1.0, max_support_size=100))
out["device_flatten_us"] = timeit(lambda: flatten_on_device(sampled))
out["device_merge_sample_us"] = timeit(lambda: collate_on_device(l_idx, l_lbl, engines, wts, noise_full, total=NS, max_pos_sections=8, temperature=1.0,
                                                                 max_support_size=100))
out["device_merge_sample_flatten_us"] = timeit(lambda: collate_on_device(l_idx, l_lbl, engines, wts, noise_full, total=NS, max_pos_sections=8,
                                                                         temperature=1.0, max_support_size=100, in_batch_negatives=True))
out["device_chain_with_device_noise_us"] = timeit(lambda: collate_on_device(l_idx, l_lbl, engines, wts, None, total=NS, max_pos_sections=8,
                                                                            temperature=1.0, max_support_size=100, in_batch_negatives=True))


def chain_events(n=200):
    """Device time of the chained launches (HIP events on the stream; excludes the host's launch cost)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        collate_on_device(l_idx, l_lbl, engines, wts, noise_full, total=NS, max_pos_sections=8, temperature=1.0, max_support_size=100, in_batch_negatives=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        collate_on_device(l_idx, l_lbl, engines, wts, noise_full, total=NS, max_pos_sections=8, temperature=1.0, max_support_size=100, in_batch_negatives=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


out["device_chain_back_to_back_us"] = chain_events()
if "--collate-only" in sys.argv:
    print(json.dumps(out))
    raise SystemExit(0)
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
