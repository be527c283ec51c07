"""Logical corpus shards: route each query row to the engine of its shard, add the shard's row offset,
restore the request order (mirror of /root/reference/src/vod_search/sharded_search.py:28-260).

These are the reference's *logical* shards (one engine per corpus).  Physical row-sharding of one corpus
over several GPUs lives below this layer (`vod_amd.distributed`).  One deliberate difference (SURVEY
quirk Q1): the offset is added to valid ids only, so a pad stays -1 instead of becoming `offset - 1`;
the strict-superset validation bug (Q2) is not reproduced either.
"""
from __future__ import annotations

import asyncio
import collections
import typing as typ

import numpy as np

from vod_amd import types as vt
from vod_amd.search.base import SearchClient, SearchMaster, ShardName


class ShardedSearchClient(SearchClient):
    def __init__(self, shards: dict[ShardName, SearchClient], offsets: dict[ShardName, int]):
        if shards.keys() != offsets.keys():
            raise ValueError(f"Keys of `shards` and `offsets` must be the same. Found {shards.keys()} and {offsets.keys()}")
        self._shards = shards
        self._offsets = offsets

    def __repr__(self) -> str:
        return f"{type(self).__name__}(shards={self._shards})"

    @property
    def shards(self) -> dict[ShardName, SearchClient]:
        return self._shards.copy()

    @property
    def offsets(self) -> dict[ShardName, int]:
        return self._offsets.copy()

    @property
    def requires_vectors(self) -> bool:  # type: ignore[override]
        return any(s.requires_vectors for s in self._shards.values())

    def ping(self) -> bool:
        return all(s.ping() for s in self._shards.values())

    def _validate(self, shard) -> None:
        if shard is None:
            raise ValueError("Must specify `shard`")
        unknown = set(shard) - set(self._shards)
        if unknown:
            raise ValueError(f"Invalid shard names {sorted(unknown)}. Valid names are {list(self._shards)}")

    def _search_one(self, name: ShardName, query: dict, with_vector: bool, top_k: int) -> vt.RetrievalBatch:
        result = self._shards[name].search(
            text=query["text"],
            ids=query.get("ids"),
            subset_ids=query.get("subset_ids"),
            vector=np.stack(query["vector"]) if with_vector else None,
            top_k=top_k,
        )
        off = self._offsets[name]
        result.indices = np.where(result.indices >= 0, result.indices + off, result.indices)
        return result

    def search(self, *, text, vector=None, subset_ids=None, ids=None, shard=None, top_k: int = 3) -> vt.RetrievalBatch:
        self._validate(shard)
        groups, lookup = _scatter_queries(text=text, shard=shard, vector=vector, subset_ids=subset_ids, ids=ids)
        results = {name: self._search_one(name, q, vector is not None, top_k) for name, q in groups.items()}
        return _gather_results(lookup, results)

    async def async_search(self, *, text, shard=None, vector=None, subset_ids=None, ids=None, top_k: int = 3) -> vt.RetrievalBatch:
        self._validate(shard)
        groups, lookup = _scatter_queries(text=text, shard=shard, vector=vector, subset_ids=subset_ids, ids=ids)
        loop = asyncio.get_event_loop()
        names = list(groups)
        futures = [loop.run_in_executor(None, self._search_one, n, groups[n], vector is not None, top_k) for n in names]
        results = dict(zip(names, await asyncio.gather(*futures)))
        return _gather_results(lookup, results)


def _scatter_queries(text, shard, vector=None, subset_ids=None, ids=None):
    """Group the batch rows by shard name; `lookup[i] = (shard, position inside the shard's group)`."""
    groups: dict[ShardName, dict[str, list]] = collections.defaultdict(lambda: collections.defaultdict(list))
    lookup: list[tuple[ShardName, int]] = []
    for i, name in enumerate(shard):
        g = groups[name]
        g["text"].append(text[i])
        lookup.append((name, len(g["text"]) - 1))
        if subset_ids is not None:
            g["subset_ids"].append(subset_ids[i])
        if ids is not None:
            g["ids"].append(ids[i])
        if vector is not None:
            g["vector"].append(vector[i])
    return {k: dict(v) for k, v in groups.items()}, lookup


def _gather_results(lookup, results: dict[ShardName, vt.RetrievalBatch]) -> vt.RetrievalBatch:
    return vt.RetrievalBatch.stack_samples([results[name][j] for name, j in lookup])


class ShardedSearchMaster(SearchMaster[ShardedSearchClient]):
    """Enter / exit every shard's master together (sharded_search.py:206-260)."""

    def __init__(self, shards: dict[ShardName, SearchMaster], offsets: dict[ShardName, int], skip_setup: bool = False,
                 free_resources: bool = False):
        super().__init__(skip_setup=skip_setup, free_resources=free_resources)
        if shards.keys() != offsets.keys():
            raise ValueError("Keys of `shards` and `offsets` must be the same.")
        self.shards = shards
        self.offsets = offsets

    def __enter__(self):
        for m in self.shards.values():
            m.__enter__()
        return self

    def __exit__(self, *exc) -> None:
        for m in self.shards.values():
            m.__exit__(*exc)

    def get_client(self) -> ShardedSearchClient:
        return ShardedSearchClient(shards={k: m.get_client() for k, m in self.shards.items()}, offsets=dict(self.offsets))

    def _make_cmd(self) -> list[str]:
        raise NotImplementedError(f"{type(self).__name__} does not implement `_make_cmd`: it only manages its shards")
