"""Oracle (test infrastructure; imported only by bench.py's `cpu_baseline` leg): the two "restated" baselines reported beside the C5
kernels (bench.py `side` entry C5; never on the product path):

  cpu_baseline            the reference's numba loops of the collate-side chain (merge.py:71-164, numpy_ops.py:24-143,
                          sample.py:160-352, in_batch_negatives.py:10-52) restated in plain C (oracle/collate_ref.c, gcc -O3 -fopenmp;
                          pinned by the reference-generated fixtures in tests/test_oracle_golden.py), timed on this box's host cores
                          on the SAME inputs the device chain ran on - and, as the checker, compared with the device chain's output;
  reference_op_sequence   the reference's H5 forward as the sequence of eager torch ops it runs (einsum, masked_fill_, log_softmax,
                          target casting, the weighted loss, three KL diagnostics: retrieval.py:30-92,153-243) + autograd backward,
                          timed on the same GPU.  A build-owned restatement of the op sequence: the reference module cannot travel.
"""
from __future__ import annotations

import math
import pathlib
import statistics
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def _median_us(fn, n=30, warm=3) -> float:
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(ts)


def eager_reference_loss(torch, q, s, batch):
    """The op sequence of RetrievalGradients.__call__ with the shipped configuration (auxiliary weights 0)."""
    score = batch["section__score"]
    is_padding = score.isinf() & (score < 0)
    scores = torch.einsum("bh, dh -> bd", q, s) if s.dim() == 2 else torch.einsum("bh, bdh -> bd", q, s)
    scores.masked_fill_(is_padding, -math.inf)
    logp = scores.log_softmax(dim=-1)
    targets = (batch["section__relevance"] > 0).float()
    targets.masked_fill_(is_padding, 0.0)
    n_pos = targets.sum(dim=1)
    n_pos = torch.where(n_pos == 0, (~is_padding).float().sum(dim=1), n_pos)
    w = 1 / n_pos[:, None] * (logp.exp().detach() - targets)
    loss = torch.sum(torch.where(is_padding, 0, w.detach() * logp), dim=-1)
    has_pos = n_pos > 0
    loss = torch.where(has_pos, loss, torch.zeros_like(loss))
    loss = loss.sum() / has_pos.float().sum()
    diag = {}
    for key in ("section__score", "section__sparse", "section__dense"):
        ref = batch.get(key)
        if ref is None:
            continue
        p_def, q_def = logp.isfinite(), ref.isfinite()
        p_lp = logp.masked_fill(~p_def, -math.inf).log_softmax(dim=-1)
        q_lp = ref.masked_fill(~q_def, -math.inf).log_softmax(dim=-1)
        diag[key] = torch.where(p_def & q_def, q_lp.exp() * (q_lp - p_lp), 0.0).sum(dim=-1).mean().detach()
    return loss, scores, diag


def measure(torch, dev, c5_data, loss_inputs: dict, kw: dict) -> dict:
    from oracle import collate_ref as cref  # the reported CPU baseline + checker; never the product path
    from vod_amd.core.collate import collate_on_device
    from vod_amd.hostcpu import usable_cpus

    out: dict = {}
    # ---- collate-side chain on the host cores ----
    l_idx, l_lbl, engines, wts = c5_data.make(dev)
    B, K = c5_data.B, c5_data.K
    noise_dev = torch.empty((B, 3 * K + 1), device=dev).exponential_()
    host = {n: (i.cpu().numpy(), s.cpu().numpy()) for n, (i, s) in engines.items()}
    lookup = (l_idx.cpu().numpy(), None, l_lbl.cpu().numpy())
    noise = noise_dev.cpu().numpy()
    threads = usable_cpus()
    cref._load().vodref_set_threads(int(threads))
    state = {}

    def chain():
        m_idx, m_scr, m_lbl, m_raw = cref.merge_hybrid(lookup, host, wts)
        smp = cref.sample_search_results(m_idx, m_scr, m_lbl, m_raw, noise[:, : m_idx.shape[1]], kw["total"], kw["max_pos_sections"], kw["temperature"],
                                         kw["max_support_size"])
        flat = cref.flatten_samples(smp["indices"], smp["scores"], smp["labels"], smp["log_weights"], smp["raw"])
        state["smp"], state["flat"] = smp, flat

    def merge_sample():
        m_idx, m_scr, m_lbl, m_raw = cref.merge_hybrid(lookup, host, wts)
        cref.sample_search_results(m_idx, m_scr, m_lbl, m_raw, noise[:, : m_idx.shape[1]], kw["total"], kw["max_pos_sections"], kw["temperature"],
                                   kw["max_support_size"])

    us_chain, us_ms = _median_us(chain), _median_us(merge_sample)
    dev_out = collate_on_device(l_idx, l_lbl, engines, wts, noise_dev, **kw)
    fin = np.isfinite(state["smp"]["log_weights"])
    same = bool(np.array_equal(dev_out.indices.cpu().numpy()[fin], state["smp"]["indices"][fin])
                and np.array_equal(dev_out.labels.cpu().numpy().astype(bool), state["smp"]["labels"])
                and np.allclose(dev_out.log_weights.cpu().numpy()[fin], state["smp"]["log_weights"][fin], rtol=1e-4, atol=1e-4))
    out["cpu_baseline"] = {
        "value": us_chain, "unit": "us per batch (merge + sample + flatten)", "merge_sample_us": us_ms, "cores": threads, "threads": threads, "kind": "port",
        "sample": f"the reference's numba loops restated in C (oracle/collate_ref.c, gcc -O3 -fopenmp, {threads} threads), the same {B}-query batch, "
                  "median of 30 runs incl. the NumPy glue between the three calls",
        "device_chain_equals_cpu_restatement": same,
    }
    # ---- the reference's H5 op sequence in eager torch on this GPU ----
    ref = {}
    for name, (q, s, batch) in loss_inputs.items():
        def fwd_bwd(q=q, s=s, batch=batch):
            q.grad = s.grad = None
            loss, _, _ = eager_reference_loss(torch, q, s, batch)
            loss.backward()

        def fwd(q=q, s=s, batch=batch):
            with torch.no_grad():
                eager_reference_loss(torch, q, s, batch)

        def wall(fn, n=100):
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

        ref[name] = {"fwd_wall_us": wall(fwd), "fwd_bwd_wall_us": wall(fwd_bwd)}
        # checker: the fused kernels against this op sequence on the same inputs
        from vod_amd.gradients import RetrievalGradients

        q.grad = s.grad = None
        l_ref, sc_ref, _ = eager_reference_loss(torch, q, s, batch)
        l_ref.backward()
        dq_ref = q.grad.clone()
        q.grad = s.grad = None
        o = RetrievalGradients()(batch=batch, query_encoding=q, section_encoding=s)
        o.loss.backward()
        ref[name]["fused_equals_op_sequence"] = bool(torch.allclose(o.loss, l_ref, rtol=2e-4, atol=2e-5) and torch.allclose(q.grad, dq_ref, rtol=2e-3, atol=2e-5))
    out["reference_op_sequence"] = {"kind": "restated", "what": "eager torch ops of retrieval.py:30-92,153-243 + autograd backward, same GPU, wall us incl. sync", **ref}
    return out
