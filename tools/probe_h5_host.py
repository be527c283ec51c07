#!/usr/bin/env python3
"""Where the HOST time of one retrieval-loss step goes (C5 shapes): the calls are issued without synchronisation, so the per-call
figures are CPU time; `wall` figures include a device synchronisation.  usage: python tools/probe_h5_host.py"""
import json
import pathlib
import statistics
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vod_amd import gradients as G  # noqa: E402

dev = torch.device("cuda", 0)
B, H = 64, 768
acc = {}


def timed(name, fn):
    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e6)
    return wrap


G._RetrievalLoss.forward = staticmethod(timed("Function.forward body", G._RetrievalLoss.forward))
G._RetrievalLoss.backward = staticmethod(timed("Function.backward body", G._RetrievalLoss.backward))
out = {}
for name, D, three_d in (("3d_64x32", 32, True), ("inbatch_64x2048", 2048, False)):
    q = torch.randn((B, H), device=dev, requires_grad=True)
    s = torch.randn(((B, D, H) if three_d else (D, H)), device=dev, requires_grad=True)
    batch = {"section__score": torch.randn((B, D), device=dev), "section__relevance": (torch.rand((B, D), device=dev) < 0.05).long(),
             "section__sparse": torch.randn((B, D), device=dev), "section__dense": torch.randn((B, D), device=dev)}
    grad = G.RetrievalGradients()
    rec = {}
    for mode in ("host", "wall"):
        acc.clear()
        f, b, tot = [], [], []
        for it in range(300):
            q.grad = s.grad = None
            if mode == "wall":
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            o = grad(batch=batch, query_encoding=q, section_encoding=s)
            t1 = time.perf_counter()
            o.loss.backward()
            t2 = time.perf_counter()
            if mode == "wall":
                torch.cuda.synchronize()
            t3 = time.perf_counter()
            if it >= 50:
                f.append((t1 - t0) * 1e6)
                b.append((t2 - t1) * 1e6)
                tot.append((t3 - t0) * 1e6)
        torch.cuda.synchronize()
        rec[mode] = {"forward_call_us": statistics.median(f), "backward_call_us": statistics.median(b), "total_us": statistics.median(tot),
                     **{k: statistics.median(v[50:]) for k, v in acc.items()}}
    out[name] = rec
print(json.dumps(out, indent=1))
