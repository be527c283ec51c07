"""Search fan-out + merge used by the retrieval collate (mirror of
/root/reference/src/vod_dataloaders/core/search.py:20-161), with the merge on the GPU.

`async_hybrid_search` keeps the reference's contract: a "lookup" request is prepended to fetch the
gold sections from `lookup_engine_name`, every engine is queried concurrently (each call retried with
exponential back-off, search is pure so retries are idempotent), results are merged with
`{lookup: 0, **weights}`.
"""
from __future__ import annotations

import asyncio
import copy
import random
import time
import typing as typ
import warnings

import numpy as np

from vod_amd import types as vt
from vod_amd.core import merge as _merge

FLOAT_INF_THRES = 3e12  # scores above this are treated as bogus +inf (search.py:16); our engine never emits them
LOOKUP_CLIENT_NAME = "lookup"
MAX_ATTEMPTS = 10


def async_hybrid_search(
    *,
    text: list[str],
    shards: list[str],
    vector: None | np.ndarray = None,
    subset_ids: None | list[list[str]] = None,
    section_ids: list[list[str]],
    top_k: int,
    clients: dict[str, typ.Any],
    weights: dict[str, float],
    lookup_engine_name: str = "sparse",
    device: int = 0,
) -> tuple[vt.RetrievalBatch, dict[str, np.ndarray]]:
    if lookup_engine_name not in clients:
        raise ValueError(f"The `{lookup_engine_name}` client must be specified to lookup the golden/positive sections.")
    lookup_payload = {
        "client": clients[lookup_engine_name], "vector": vector, "text": [""] * len(text),
        "subset_ids": subset_ids, "ids": section_ids, "shard": shards, "top_k": top_k,
    }
    names = list(clients)
    payloads = [
        {"client": clients[n], "vector": vector, "text": text, "subset_ids": subset_ids, "shard": shards, "top_k": top_k}
        for n in names
    ]
    t0 = time.perf_counter()
    results = asyncio.run(_execute_search([lookup_payload] + payloads))
    meta = {"search_time": time.perf_counter() - t0}
    merged, raw = merge_search_results(dict(zip([LOOKUP_CLIENT_NAME] + names, results)), weights, device=device)
    merged.meta.update(meta)
    return merged, raw


def merge_search_results(
    search_results: dict[str, vt.RetrievalBatch], weights: dict[str, float], device: int = 0
) -> tuple[vt.RetrievalBatch, dict[str, np.ndarray]]:
    """`_merge_search_results` (search.py:79-125): lookup scores discarded, other engines' labels discarded,
    per-row min-subtraction, weighted union -- executed by the HIP kernel."""
    if LOOKUP_CLIENT_NAME not in search_results:
        raise ValueError(f"The `{LOOKUP_CLIENT_NAME}` client must be specified to lookup the golden/positive sections.")
    meta = {}
    for name, result in search_results.items():
        for key, value in result.meta.items():
            meta[f"{name}_{key}"] = value
    lookup = search_results[LOOKUP_CLIENT_NAME]
    engines = {}
    nq = len(lookup.indices)
    for name, r in search_results.items():
        if name == LOOKUP_CLIENT_NAME:
            continue
        if len(r.scores) != nq:
            raise ValueError(f"All scores must have the same length. Found: {{{nq}, {len(r.scores)}}}")
        scores = r.scores
        if name == "dense":  # scrub faiss-style sentinels (search.py:149-161); inert for the HIP engine
            bogus = scores >= FLOAT_INF_THRES
            if bogus.any():
                warnings.warn(f"Found {int(bogus.sum())} inf scores in the search results.", stacklevel=2)
                scores = np.where(bogus, np.nan, scores)
        engines[name] = (r.indices, scores)
    idx, scr, lbl, raw = _merge.merge_hybrid(lookup.indices, lookup.labels, engines, weights, device=device)
    out = vt.RetrievalBatch(scores=scr, indices=idx, labels=lbl, meta=meta)
    return out, raw


async def _execute_search(payloads: list[dict[str, typ.Any]]) -> list[vt.RetrievalBatch]:
    async def one(args: dict[str, typ.Any]) -> vt.RetrievalBatch:
        delay = 1.0
        for attempt in range(MAX_ATTEMPTS):
            a = copy.copy(args)  # shallow copy: retries must see the original dict
            client = a.pop("client")
            t0 = time.perf_counter()
            try:
                result = await client.async_search(**a)
            except Exception:
                if attempt == MAX_ATTEMPTS - 1:
                    raise
                await asyncio.sleep(random.uniform(0, delay))  # noqa: S311
                delay = min(60.0, delay * 2)
                continue
            result.meta["search_time"] = time.perf_counter() - t0
            return result
        raise RuntimeError("unreachable")

    return await asyncio.gather(*[one(p) for p in payloads])
