#!/usr/bin/env python3
"""Is the fused retrieval loss (forward + backward through autograd) hipGraph-capturable, and what does a replay cost?  C5 shapes.
`torch.cuda.graph` captures every launch of the step - our kernels are enqueued on torch's current (capturing) stream through the C-ABI -
and a replay is ONE host call.  Prints eager vs replay wall microseconds (median, incl. a device sync) and checks the replayed results
against the eager ones on fresh inputs copied into the static buffers.      usage: python tools/probe_h5_graph.py"""
import json
import pathlib
import statistics
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vod_amd.gradients import RetrievalGradients  # noqa: E402

dev = torch.device("cuda", 0)
B, H = 64, 768
out = {}


def wall(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(ts)


for name, D, three_d in (("3d_64x32", 32, True), ("inbatch_64x2048", 2048, False)):
    q = torch.randn((B, H), device=dev, requires_grad=True)
    s = torch.randn(((B, D, H) if three_d else (D, H)), device=dev, requires_grad=True)
    batch = {"section__score": torch.randn((B, D), device=dev), "section__relevance": (torch.rand((B, D), device=dev) < 0.05).long(),
             "section__sparse": torch.randn((B, D), device=dev), "section__dense": torch.randn((B, D), device=dev)}
    batch["section__relevance"][:, 0] = 1
    grad = RetrievalGradients()

    def eager():
        q.grad = s.grad = None
        o = grad(batch=batch, query_encoding=q, section_encoding=s)
        o.loss.backward()
        return o

    rec = {"eager_wall_us": wall(eager)}
    try:
        # static gradient buffers: backward accumulates into them inside the graph, so they are zeroed inside it too
        q.grad = torch.zeros_like(q)
        s.grad = torch.zeros_like(s)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                q.grad.zero_()
                s.grad.zero_()
                grad(batch=batch, query_encoding=q, section_encoding=s).loss.backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            q.grad.zero_()
            s.grad.zero_()
            o_static = grad(batch=batch, query_encoding=q, section_encoding=s)
            o_static.loss.backward()
        rec["replay_wall_us"] = wall(g.replay)
        # fresh inputs through the static buffers: the replay must give what eager gives
        with torch.no_grad():
            q.copy_(torch.randn_like(q))
            s.copy_(torch.randn_like(s))
            batch["section__score"].copy_(torch.randn((B, D), device=dev))
        g.replay()
        torch.cuda.synchronize()
        loss_g, dq_g, ds_g = o_static.loss.clone(), q.grad.clone(), s.grad.clone()
        q.grad = s.grad = None
        o = eager()
        rec["replay_equals_eager"] = bool(torch.equal(loss_g, o.loss) and torch.equal(dq_g, q.grad) and torch.equal(ds_g, s.grad))
    except Exception as exc:  # noqa: BLE001
        rec["capture_error"] = f"{type(exc).__name__}: {exc}"[:400]
    out[name] = rec
print(json.dumps(out, indent=1))
