"""Fan-out to several engines ("dense", "sparse", ...) -- mirror of
/root/reference/src/vod_search/hybrid_search.py:20-200.  The dense engine is `HipMipsClient` (usually
wrapped in a `ShardedSearchClient`); sparse engines (BM25 / Elasticsearch) stay the reference's own and
reach the merge only as (indices, scores, labels) arrays.
"""
from __future__ import annotations

import asyncio
import copy

from vod_amd import types as vt
from vod_amd.search.base import SearchClient, SearchMaster, ShardName


class HybridSearchClient(SearchClient):
    def __init__(self, clients: dict[str, SearchClient], shard_list: None | list[ShardName] = None, sections=None) -> None:
        self.clients = clients
        self._shard_list = shard_list
        self._sections = sections

    @property
    def shard_list(self) -> list[ShardName]:
        if self._shard_list is None:
            raise ValueError("The shard list has not been set.")
        return copy.copy(self._shard_list)

    @property
    def sections(self):
        if self._sections is None:
            raise ValueError("The sections have not been set.")
        return self._sections

    def __repr__(self) -> str:
        return f"{type(self).__name__}(clients={self.clients})"

    @property
    def requires_vectors(self) -> bool:  # type: ignore[override]
        return any(c.requires_vectors for c in self.clients.values())

    def ping(self) -> bool:
        return all(c.ping() for c in self.clients.values())

    def search(self, *, text, vector=None, subset_ids=None, ids=None, shard=None, top_k: int = 3) -> dict[str, vt.RetrievalBatch]:  # type: ignore[override]
        return {
            name: c.search(vector=vector, text=text, subset_ids=subset_ids, ids=ids, shard=shard, top_k=top_k)
            for name, c in self.clients.items()
        }

    async def async_search(self, *, text, vector=None, subset_ids=None, ids=None, shard=None, top_k: int = 3):  # type: ignore[override]
        loop = asyncio.get_event_loop()
        names = list(self.clients)

        def run(name):
            return self.clients[name].search(vector=vector, text=text, ids=ids, subset_ids=subset_ids, shard=shard, top_k=top_k)

        results = await asyncio.gather(*[loop.run_in_executor(None, run, n) for n in names])
        return dict(zip(names, results))


class HybridSearchMaster(SearchMaster[HybridSearchClient]):
    """Enter / exit all engines' masters together (the reference spells it `HyrbidSearchMaster`)."""

    def __init__(self, servers: dict[str, SearchMaster], skip_setup: bool = False, free_resources: bool = False,
                 shard_list: None | list[ShardName] = None, sections=None):
        super().__init__(skip_setup=skip_setup, free_resources=free_resources)
        self.servers = servers
        self._shard_list = shard_list
        self._sections = sections

    def __enter__(self):
        for m in self.servers.values():
            m.__enter__()
        return self

    def __exit__(self, *exc) -> None:
        for m in self.servers.values():
            m.__exit__(*exc)

    def get_client(self) -> HybridSearchClient:
        return HybridSearchClient(clients={k: m.get_client() for k, m in self.servers.items()}, shard_list=self._shard_list,
                                  sections=self._sections)

    def _make_cmd(self) -> list[str]:
        raise NotImplementedError(f"{type(self).__name__} does not implement `_make_cmd`: it only manages its engines")


HyrbidSearchMaster = HybridSearchMaster  # the reference's spelling, kept importable
