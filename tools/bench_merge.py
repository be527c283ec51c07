#!/usr/bin/env python3
"""Latency of the exchange step's merge kernel (vodhip_merge_topk) for n_shards per-rank lists: [n_shards, nq, k] -> [nq, k]."""
import sys
import time

import torch

import pathlib

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from vod_amd.index import merge_topk

dev = torch.device("cuda", 0)
for n_shards, nq, k in [(1, 1024, 100), (2, 1024, 100), (4, 1024, 100), (8, 1024, 100), (8, 512, 200), (8, 256, 100)]:
    s = torch.sort(torch.randn((n_shards, nq, k), device=dev), dim=2, descending=True).values
    i = torch.stack([torch.randperm(1_000_000, device=dev)[: nq * k].view(nq, k) + sh * 1_250_000 for sh in range(n_shards)])
    for _ in range(5):
        merge_topk(s, i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        merge_topk(s, i)
    e1.record()
    torch.cuda.synchronize()
    t_sorted = e0.elapsed_time(e1) / 50 * 1e3
    su = s[:, : nq - 1, torch.randperm(k, device=dev)].contiguous()  # lists in arbitrary order: the sorting-network fallback
    iu = i[:, : nq - 1, torch.randperm(k, device=dev)].contiguous()  # (one query fewer: a different grid size in kernel traces)
    merge_topk(su, iu)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        merge_topk(su, iu)
    e1.record()
    torch.cuda.synchronize()
    print(f"n_shards {n_shards} nq {nq} k {k}: sorted lists {t_sorted:.1f} us, unsorted lists {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per merge (incl. output allocation)")
