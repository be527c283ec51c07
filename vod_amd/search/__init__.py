"""Host-side mirror of the reference's `vod_search` interface for the dense (MIPS) path.

`HipMipsClient` / `HipMipsMaster` stand where `FaissClient` / `FaissMaster`
(/root/reference/src/vod_search/faiss_search/client.py) stand; the wire protocol, the keyword-only
`search()` contract and the `RetrievalBatch` return type are unchanged.
"""
from vod_amd.search.base import DoNotPickleError, SearchClient, SearchMaster  # noqa: F401
from vod_amd.search.client import HipMipsClient, HipMipsMaster  # noqa: F401
from vod_amd.search.hybrid import HybridSearchClient, HybridSearchMaster  # noqa: F401
from vod_amd.search.sharded import ShardedSearchClient, ShardedSearchMaster  # noqa: F401
