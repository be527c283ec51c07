#!/usr/bin/env python3
"""BASELINE configs[4] (C5) as one `side` entry of the default bench line: the collate-side chain (hybrid merge -> labeled priority
sampling -> in-batch flattening) and the in-batch retrieval loss at B = 64 queries, 128 hits x 3 engines, 32 sampled sections, H = 768
(SURVEY 8d inputs: tools/c5_data.py), timed on the device-resident API, with

  * `verify`: the same entry points run on the committed REFERENCE-generated fixtures (tests/golden/collate_chain.npz - the reference's
    own merge -> sample -> flatten chain - and the five retrieval_grad_* fixtures) and compared the way the parity tests compare;
  * bench.py's `cpu_baseline` leg adds (oracle/c5_baseline.py - this file never imports the oracle): `cpu_baseline` = the reference's
    numba loops restated in plain C, timed on the host cores, and `reference_op_sequence` = the reference's H5 op sequence restated in
    eager torch, timed on the same GPU.  Both are labelled "restated": the reference itself cannot travel to the GPU box.

    python tools/side_c5.py            # prints the entry as JSON
"""
from __future__ import annotations

import json
import pathlib
import statistics
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = ROOT / "tests" / "golden"


def _timeit(torch, fn, n=100, warm=10):
    """Median wall microseconds of one call INCLUDING a device synchronisation (what a caller that needs the result pays)."""
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


def _device_us(torch, fn, n=100, warm=10):
    """Device microseconds per call with the calls enqueued back to back (HIP events on the current stream)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def _host_syncs(torch, fn) -> int:
    """0 when the call completes under `set_sync_debug_mode('error')` (any host synchronisation inside raises)."""
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        fn()
        return 0
    except RuntimeError:
        return 1
    finally:
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()


def verify_fixtures(torch, dev) -> dict:
    """The device chain and the loss on the reference-generated fixtures, compared as tests/test_collate_device_gpu.py and
    tests/test_gradients_gpu.py compare (ids / labels exact where the reference's weight is finite, float32 sums to 2e-5 / 2e-4)."""
    from vod_amd.core.collate import collate_on_device, sample_merged_on_device
    from vod_amd.core.merge import merge_hybrid_device
    from vod_amd.gradients import RetrievalGradients

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    manifest = json.loads((GOLDEN / "manifest.json").read_text())
    g = np.load(GOLDEN / "collate_chain.npz")
    cases = manifest["collate_chain"]["params"]["cases"]
    max_logw = 0.0
    for c, p in enumerate(cases):
        engines = {"dense": (t(g[f"d_idx_{c}"]), t(g[f"d_scr_{c}"])), "sparse": (t(g[f"s_idx_{c}"]), t(g[f"s_scr_{c}"]))}
        l_idx, l_lbl = t(g[f"l_idx_{c}"]), t(g[f"l_lbl_{c}"])
        stride = l_idx.shape[1] + sum(v[0].shape[1] for v in engines.values()) + 1
        noise = np.ones((l_idx.shape[0], stride), dtype=np.float32)
        noise[:, : g[f"noise_{c}"].shape[1]] = g[f"noise_{c}"]
        kw = dict(total=p["total"], max_pos_sections=p["max_pos_sections"], temperature=p["temperature"], max_support_size=p["max_support_size"])
        merged = merge_hybrid_device(l_idx, l_lbl, engines, p["weights"])
        out = sample_merged_on_device(merged, t(noise), **kw)
        flat = collate_on_device(l_idx, l_lbl, engines, p["weights"], t(noise), in_batch_negatives=True, **kw)
        m_idx, m_scr, m_lbl, _ = merged.cut()
        assert np.array_equal(m_idx.cpu().numpy(), g[f"m_idx_{c}"]) and np.array_equal(m_scr.cpu().numpy(), g[f"m_scr_{c}"]), f"merge differs (case {c})"
        assert np.array_equal(m_lbl.cpu().numpy(), g[f"m_lbl_{c}"]), f"merged labels differ (case {c})"
        fin = np.isfinite(g[f"smp_logw_{c}"])
        logw = out.log_weights.cpu().numpy()
        assert np.array_equal(np.isfinite(logw), fin), f"sampled pads differ (case {c})"
        assert np.array_equal(out.indices.cpu().numpy()[fin], g[f"smp_idx_{c}"][fin]), f"sampled ids differ (case {c})"
        assert np.array_equal(out.labels.cpu().numpy(), g[f"smp_lbl_{c}"]), f"sampled labels differ (case {c})"
        d = float(np.abs(logw[fin] - g[f"smp_logw_{c}"][fin]).max()) if fin.any() else 0.0
        assert d <= 2e-5 + 2e-5 * float(np.abs(g[f"smp_logw_{c}"][fin]).max()), f"log-weights differ by {d} (case {c})"
        max_logw = max(max_logw, d)
        uq = np.unique(out.indices.cpu().numpy())
        assert np.array_equal(flat.indices.cpu().numpy()[: len(uq)], uq), f"flattened id set differs (case {c})"
    max_loss = 0.0
    names = ["retrieval_grad_2d", "retrieval_grad_3d", "retrieval_grad_nopos", "retrieval_grad_padded", "retrieval_grad_inbatch"]
    for name in names:
        f = np.load(GOLDEN / f"{name}.npz")
        qt = torch.tensor(f["q"], device=dev, requires_grad=True)
        st = torch.tensor(f["s"], device=dev, requires_grad=True)
        batch = {"section__score": torch.tensor(f["score"], device=dev), "section__relevance": torch.tensor(f["relevance"], device=dev),
                 "section__sparse": torch.tensor(f["sparse"], device=dev), "section__dense": torch.tensor(f["dense"], device=dev)}
        o = RetrievalGradients()(batch=batch, query_encoding=qt, section_encoding=st)
        o.loss.backward()
        np.testing.assert_allclose(o.loss.item(), f["loss"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(qt.grad.cpu().numpy(), f["dq"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(st.grad.cpu().numpy(), f["ds"], rtol=2e-4, atol=2e-5)
        max_loss = max(max_loss, abs(o.loss.item() - float(f["loss"])))
    return {"fixtures": "tests/golden/collate_chain.npz (the reference's own merge -> sample -> flatten chain) + retrieval_grad_{2d,3d,nopos,padded,inbatch}.npz",
            "collate_cases": len(cases), "gradient_cases": len(names), "ok": True, "max_abs_log_weight_diff": max_logw, "max_abs_loss_diff": max_loss,
            "tolerance": "ids / labels / merged scores bit-exact; float32 log-weights 2e-5; loss and gradients rtol 2e-4"}


def measure(dev) -> tuple[dict, dict]:
    """(the side entry, the context - inputs and sampling parameters - the baseline leg re-uses)."""
    import torch

    import c5_data
    from vod_amd.core.collate import collate_on_device
    from vod_amd.gradients import RetrievalGradients

    B, K, H, NS = c5_data.B, c5_data.K, c5_data.H, c5_data.NS
    entry: dict = {"name": "C5", "workload": f"hybrid merge + priority sampling + in-batch flattening + retrieval loss: batch {B} queries, {K} hits x 3 engines "
                                              f"(lookup, dense, sparse), {NS} sampled sections, H = {H}", "unit": "us per batch"}
    try:
        entry["verify"] = verify_fixtures(torch, dev)
    except AssertionError as exc:
        entry["verify"] = {"ok": False, "error": str(exc)[:300]}
    l_idx, l_lbl, engines, wts = c5_data.make(dev)
    noise = torch.empty((B, 3 * K + 1), device=dev).exponential_()
    kw = dict(total=NS, max_pos_sections=8, temperature=1.0, max_support_size=100)
    chain2 = lambda: collate_on_device(l_idx, l_lbl, engines, wts, noise, **kw)  # noqa: E731
    chain3 = lambda: collate_on_device(l_idx, l_lbl, engines, wts, noise, in_batch_negatives=True, **kw)  # noqa: E731
    entry["collate_merge_sample"] = {"wall_us": _timeit(torch, chain2), "device_us": _device_us(torch, chain2), "launches": 2,
                                     "host_syncs": _host_syncs(torch, chain2)}
    entry["collate_merge_sample_flatten"] = {"wall_us": _timeit(torch, chain3), "device_us": _device_us(torch, chain3), "launches": 3,
                                             "host_syncs": _host_syncs(torch, chain3)}
    entry["launches_note"] = "launch counts are the library's by construction (vodhip_collate); profiles/r04_c5_kernel_stats.csv lists them"
    grad = RetrievalGradients()
    loss_inputs = {}
    for name, D, three_d in (("retrieval_loss_3d_64x32", NS, True), ("retrieval_loss_inbatch_64x2048", B * NS, False)):
        g = torch.Generator(device=dev).manual_seed(5)
        q = torch.randn((B, H), device=dev, generator=g).requires_grad_()
        s = torch.randn(((B, D, H) if three_d else (D, H)), device=dev, generator=g).requires_grad_()
        batch = {"section__score": torch.randn((B, D), device=dev, generator=g), "section__relevance": (torch.rand((B, D), device=dev, generator=g) < 0.05).long(),
                 "section__sparse": torch.randn((B, D), device=dev, generator=g), "section__dense": torch.randn((B, D), device=dev, generator=g)}
        batch["section__relevance"][:, 0] = 1
        loss_inputs[name] = (q, s, batch)

        def fwd_bwd(q=q, s=s, batch=batch):
            q.grad = s.grad = None
            grad(batch=batch, query_encoding=q, section_encoding=s).loss.backward()

        fwd = lambda q=q, s=s, batch=batch: grad(batch=batch, query_encoding=q, section_encoding=s)  # noqa: E731
        entry[name] = {"fwd_wall_us": _timeit(torch, fwd), "fwd_bwd_wall_us": _timeit(torch, fwd_bwd), "fwd_bwd_device_us": _device_us(torch, fwd_bwd)}
        # the same step captured as ONE hipGraph (vod_amd.gradients.GraphedRetrievalStep): one host call per step, bit-identical results
        try:
            from vod_amd.gradients import GraphedRetrievalStep  # noqa: PLC0415

            step = GraphedRetrievalStep(grad, batch_size=B, n_sections=D, hidden=H, sections_3d=three_d, device=dev)
            step.load(batch=batch, query_encoding=q.detach(), section_encoding=s.detach())
            out_g = step.replay()
            q.grad = s.grad = None
            out_e = grad(batch=batch, query_encoding=q, section_encoding=s)
            out_e.loss.backward()
            entry[name]["graphed_fwd_bwd_wall_us"] = _timeit(torch, step.replay)
            entry[name]["graphed_equals_eager"] = bool(torch.equal(out_g.loss, out_e.loss) and torch.equal(step.dq, q.grad) and torch.equal(step.ds, s.grad))
        except Exception as exc:  # noqa: BLE001
            entry[name]["graphed_error"] = f"{type(exc).__name__}: {exc}"[:300]
    return entry, {"c5_data": c5_data, "loss_inputs": loss_inputs, "kw": kw}


if __name__ == "__main__":
    import torch

    print(json.dumps(measure(torch.device("cuda", 0))[0]))
